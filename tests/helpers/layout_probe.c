/* tests/helpers/layout_probe.c -- prints sizeof / offsetof of the public
 * structs; compiled once against the reference's headers and once against the
 * drop-in's (tests/test_abi.py), the two outputs must be identical. */
#include <stddef.h>
#include <stdio.h>
#include <h264/h264.h>
#define S(t) printf("sizeof " #t " %zu\n", sizeof(struct t))
#define O(t, m) printf("offsetof " #t "." #m " %zu\n", offsetof(struct t, m))
int main(void)
{
	S(h264_bitstream); O(h264_bitstream, cdata); O(h264_bitstream, len);
	O(h264_bitstream, off); O(h264_bitstream, cache);
	O(h264_bitstream, cachebits); O(h264_bitstream, emulation_prevention);
	O(h264_bitstream, dynamic); O(h264_bitstream, priv);
	S(h264_nalu_header); O(h264_nalu_header, nal_ref_idc);
	O(h264_nalu_header, nal_unit_type);
	S(h264_scaling_matrix); O(h264_scaling_matrix, scaling_list_4x4);
	O(h264_scaling_matrix, scaling_list_8x8);
	O(h264_scaling_matrix, use_default_4x4);
	O(h264_scaling_matrix, _optimized_8x8);
	S(h264_hrd); O(h264_hrd, cpb);
	O(h264_hrd, initial_cpb_removal_delay_length_minus1);
	O(h264_hrd, time_offset_length);
	S(h264_vui); O(h264_vui, sar_width); O(h264_vui, matrix_coefficients);
	O(h264_vui, num_units_in_tick); O(h264_vui, nal_hrd);
	O(h264_vui, vcl_hrd); O(h264_vui, pic_struct_present_flag);
	O(h264_vui, max_dec_frame_buffering);
	S(h264_sps); O(h264_sps, constraint_set5_flag); O(h264_sps, level_idc);
	O(h264_sps, seq_parameter_set_id); O(h264_sps, seq_scaling_matrix);
	O(h264_sps, log2_max_frame_num_minus4); O(h264_sps, offset_for_ref_frame);
	O(h264_sps, max_num_ref_frames); O(h264_sps, pic_width_in_mbs_minus1);
	O(h264_sps, frame_mbs_only_flag); O(h264_sps, frame_crop_bottom_offset);
	O(h264_sps, vui_parameters_present_flag); O(h264_sps, vui);
	S(h264_pps); O(h264_pps, entropy_coding_mode_flag);
	O(h264_pps, run_length_minus1); O(h264_pps, slice_group_id);
	O(h264_pps, num_ref_idx_l0_default_active_minus1);
	O(h264_pps, pic_init_qp_minus26); O(h264_pps, _more_rbsp_data_present);
	O(h264_pps, transform_8x8_mode_flag); O(h264_pps, pic_scaling_matrix);
	O(h264_pps, second_chroma_qp_index_offset);
	S(h264_aud);
	S(h264_slice_header); O(h264_slice_header, slice_type);
	O(h264_slice_header, frame_num); O(h264_slice_header, idr_pic_id);
	O(h264_slice_header, delta_pic_order_cnt);
	O(h264_slice_header, num_ref_idx_l1_active_minus1);
	O(h264_slice_header, rplm); O(h264_slice_header, pwt);
	O(h264_slice_header, drpm); O(h264_slice_header, cabac_init_idc);
	O(h264_slice_header, slice_qp_delta);
	O(h264_slice_header, disable_deblocking_filter_idc);
	O(h264_slice_header, slice_group_change_cycle);
	S(h264_sei); O(h264_sei, buffering_period); O(h264_sei, pic_timing);
	O(h264_sei, user_data_unregistered); O(h264_sei, raw);
	S(h264_sei_pic_timing); O(h264_sei_pic_timing, clk_ts);
	S(h264_sei_buffering_period); O(h264_sei_buffering_period, vcl_hrd_cpb);
	S(h264_sei_pan_scan_rect);
	O(h264_sei_pan_scan_rect, pan_scan_rect_repetition_period);
	S(h264_sei_user_data_registered); S(h264_sei_user_data_unregistered);
	S(h264_sei_recovery_point); S(h264_sei_filler_payload);
	S(h264_sps_derived); O(h264_sps_derived, RawMbBits);
	O(h264_sps_derived, PicWidthInMbs); O(h264_sps_derived, Height);
	S(h264_info); O(h264_info, crop_left); O(h264_info, sar_width);
	O(h264_info, framerate); O(h264_info, nal_hrd_bitrate);
	O(h264_info, vcl_hrd_cpb_size);
	S(h264_ctx_cbs); O(h264_ctx_cbs, nalu_begin); O(h264_ctx_cbs, slice);
	O(h264_ctx_cbs, slice_data_mb); O(h264_ctx_cbs, sps);
	O(h264_ctx_cbs, sei); O(h264_ctx_cbs, sei_recovery_point);
	printf("enum %d %d %d %d %d %d %d\n", H264_NALU_TYPE_SLICE_IDR,
	       H264_NALU_TYPE_FILLER, H264_SLICE_TYPE_SI, H264_MB_TYPE_I_PCM,
	       H264_MB_TYPE_B_SKIP, H264_SEI_TYPE_RECOVERY_POINT,
	       H264_READER_FLAGS_SLICE_DATA);
	return 0;
}

"""ctypes mirrors of the public ABI structs (include/h264/h264_types.h,
include/h264_amd.h) used by the tests and by bench.py.  Sizes are pinned in
tests/test_abi.py against the values measured on the reference
(SURVEY.md Appendix B)."""
import ctypes as C

u32, i32, cint, u8, u16, i16, u64 = (C.c_uint32, C.c_int32, C.c_int, C.c_uint8,
                                      C.c_uint16, C.c_int16, C.c_uint64)


class S(C.Structure):
    def asdict(self):
        out = {}
        for name, _t in self._fields_:
            v = getattr(self, name)
            if isinstance(v, S):
                v = v.asdict()
            elif hasattr(v, "_length_"):
                v = list(v) if not isinstance(v[0], (S, C.Array)) else None
            out[name] = v
        return out

    def raw(self):
        return bytes(self)


class NaluHeader(S):
    _fields_ = [("forbidden_zero_bit", u32), ("nal_ref_idc", u32),
                ("nal_unit_type", u32)]


class ScalingMatrix(S):
    _fields_ = [("scaling_list_present_flag", cint * 12),
                ("scaling_list_4x4", (i32 * 16) * 6),
                ("scaling_list_8x8", (i32 * 64) * 6),
                ("use_default_4x4", cint * 6), ("use_default_8x8", cint * 6),
                ("_optimized_4x4", cint * 6), ("_optimized_8x8", cint * 6)]


class HrdCpb(S):
    _fields_ = [("bit_rate_value_minus1", u32), ("cpb_size_value_minus1", u32),
                ("cbr_flag", cint)]


class Hrd(S):
    _fields_ = [("cpb_cnt_minus1", u32), ("bit_rate_scale", u32),
                ("cpb_size_scale", u32), ("cpb", HrdCpb * 32),
                ("initial_cpb_removal_delay_length_minus1", u32),
                ("cpb_removal_delay_length_minus1", u32),
                ("dpb_output_delay_length_minus1", u32),
                ("time_offset_length", u32)]


class Vui(S):
    _fields_ = [("aspect_ratio_info_present_flag", cint),
                ("aspect_ratio_idc", u32), ("sar_width", u32),
                ("sar_height", u32), ("overscan_info_present_flag", cint),
                ("overscan_appropriate_flag", cint),
                ("video_signal_type_present_flag", cint),
                ("video_format", u32), ("video_full_range_flag", cint),
                ("colour_description_present_flag", cint),
                ("colour_primaries", u32), ("transfer_characteristics", u32),
                ("matrix_coefficients", u32),
                ("chroma_loc_info_present_flag", cint),
                ("chroma_sample_loc_type_top_field", u32),
                ("chroma_sample_loc_type_bottom_field", u32),
                ("timing_info_present_flag", cint),
                ("num_units_in_tick", u32), ("time_scale", u32),
                ("fixed_frame_rate_flag", cint),
                ("nal_hrd_parameters_present_flag", cint), ("nal_hrd", Hrd),
                ("vcl_hrd_parameters_present_flag", cint), ("vcl_hrd", Hrd),
                ("low_delay_hrd_flag", cint),
                ("pic_struct_present_flag", cint),
                ("bitstream_restriction_flag", cint),
                ("motion_vectors_over_pic_boundaries_flag", cint),
                ("max_bytes_per_pic_denom", u32),
                ("max_bits_per_mb_denom", u32),
                ("log2_max_mv_length_horizontal", u32),
                ("log2_max_mv_length_vertical", u32),
                ("max_num_reorder_frames", u32),
                ("max_dec_frame_buffering", u32)]


class Sps(S):
    _fields_ = [("profile_idc", u32), ("constraint_set0_flag", cint),
                ("constraint_set1_flag", cint), ("constraint_set2_flag", cint),
                ("constraint_set3_flag", cint), ("constraint_set4_flag", cint),
                ("constraint_set5_flag", cint), ("reserved_zero_2bits", u32),
                ("level_idc", u32), ("seq_parameter_set_id", u32),
                ("chroma_format_idc", u32),
                ("separate_colour_plane_flag", cint),
                ("bit_depth_luma_minus8", u32),
                ("bit_depth_chroma_minus8", u32),
                ("qpprime_y_zero_transform_bypass_flag", cint),
                ("seq_scaling_matrix_present_flag", cint),
                ("seq_scaling_matrix", ScalingMatrix),
                ("log2_max_frame_num_minus4", u32),
                ("pic_order_cnt_type", u32),
                ("log2_max_pic_order_cnt_lsb_minus4", u32),
                ("delta_pic_order_always_zero_flag", cint),
                ("offset_for_non_ref_pic", i32),
                ("offset_for_top_to_bottom_field", i32),
                ("num_ref_frames_in_pic_order_cnt_cycle", u32),
                ("offset_for_ref_frame", i32 * 256),
                ("max_num_ref_frames", u32),
                ("gaps_in_frame_num_value_allowed_flag", cint),
                ("pic_width_in_mbs_minus1", u32),
                ("pic_height_in_map_units_minus1", u32),
                ("frame_mbs_only_flag", cint),
                ("mb_adaptive_frame_field_flag", cint),
                ("direct_8x8_inference_flag", cint),
                ("frame_cropping_flag", cint),
                ("frame_crop_left_offset", u32),
                ("frame_crop_right_offset", u32),
                ("frame_crop_top_offset", u32),
                ("frame_crop_bottom_offset", u32),
                ("vui_parameters_present_flag", cint), ("vui", Vui)]


class Pps(S):
    _fields_ = [("pic_parameter_set_id", u32), ("seq_parameter_set_id", u32),
                ("entropy_coding_mode_flag", cint),
                ("bottom_field_pic_order_in_frame_present_flag", cint),
                ("num_slice_groups_minus1", u32),
                ("slice_group_map_type", u32), ("run_length_minus1", u32 * 8),
                ("top_left", u32 * 8), ("bottom_right", u32 * 8),
                ("slice_group_change_direction_flag", cint),
                ("slice_group_change_rate_minus1", u32),
                ("pic_size_in_map_units_minus1", u32),
                ("slice_group_id", u32 * 256),
                ("num_ref_idx_l0_default_active_minus1", u32),
                ("num_ref_idx_l1_default_active_minus1", u32),
                ("weighted_pred_flag", cint), ("weighted_bipred_idc", u32),
                ("pic_init_qp_minus26", i32), ("pic_init_qs_minus26", i32),
                ("chroma_qp_index_offset", i32),
                ("deblocking_filter_control_present_flag", cint),
                ("constrained_intra_pred_flag", cint),
                ("redundant_pic_cnt_present_flag", cint),
                ("_more_rbsp_data_present", cint),
                ("transform_8x8_mode_flag", cint),
                ("pic_scaling_matrix_present_flag", cint),
                ("pic_scaling_matrix", ScalingMatrix),
                ("second_chroma_qp_index_offset", i32)]


class RplmItem(S):
    _fields_ = [("modification_of_pic_nums_idc", u32), ("value", u32)]


class Rplm(S):
    _fields_ = [("ref_pic_list_modification_flag_l0", cint),
                ("pic_num_l0", RplmItem * 32),
                ("ref_pic_list_modification_flag_l1", cint),
                ("pic_num_l1", RplmItem * 32)]


class PwtItem(S):
    _fields_ = [("luma_weight_flag", cint), ("luma_weight", i32),
                ("luma_offset", i32), ("chroma_weight_flag", cint),
                ("chroma_weight", i32 * 2), ("chroma_offset", i32 * 2)]


class Pwt(S):
    _fields_ = [("luma_log2_weight_denom", u32),
                ("chroma_log2_weight_denom", u32), ("l0", PwtItem * 32),
                ("l1", PwtItem * 32)]


class DrpmItem(S):
    _fields_ = [("memory_management_control_operation", u32),
                ("difference_of_pic_nums_minus1", u32),
                ("long_term_pic_num", u32), ("long_term_frame_idx", u32),
                ("max_long_term_frame_idx_plus1", u32)]


class Drpm(S):
    _fields_ = [("no_output_of_prior_pics_flag", cint),
                ("long_term_reference_flag", cint),
                ("adaptive_ref_pic_marking_mode_flag", cint),
                ("mm", DrpmItem * 64)]


class SliceHeader(S):
    _fields_ = [("first_mb_in_slice", u32), ("slice_type", u32),
                ("pic_parameter_set_id", u32), ("colour_plane_id", u32),
                ("frame_num", u32), ("field_pic_flag", cint),
                ("bottom_field_flag", cint), ("idr_pic_id", u32),
                ("pic_order_cnt_lsb", u32),
                ("delta_pic_order_cnt_bottom", i32),
                ("delta_pic_order_cnt", i32 * 2), ("redundant_pic_cnt", u32),
                ("direct_spatial_mv_pred_flag", cint),
                ("num_ref_idx_active_override_flag", cint),
                ("num_ref_idx_l0_active_minus1", u32),
                ("num_ref_idx_l1_active_minus1", u32), ("rplm", Rplm),
                ("pwt", Pwt), ("drpm", Drpm), ("cabac_init_idc", u32),
                ("slice_qp_delta", i32), ("sp_for_switch_flag", cint),
                ("slice_qs_delta", i32),
                ("disable_deblocking_filter_idc", u32),
                ("slice_alpha_c0_offset_div2", i32),
                ("slice_beta_offset_div2", i32),
                ("slice_group_change_cycle", u32)]


class SpsDerived(S):
    _fields_ = [(n, u32) for n in (
        "ChromaArrayType SubWidthC SubHeightC MbWidthC MbHeightC BitDepthLuma "
        "QpBdOffsetLuma BitDepthChroma QpBdOffsetChroma RawMbBits MaxFrameNum "
        "MaxPicOrderCntLsb PicWidthInMbs PicWidthInSamplesLuma "
        "PicWidthInSamplesChroma PicHeightInMapUnits PicSizeInMapUnits "
        "FrameHeightInMbs CropUnitX CropUnitY Width Height").split()]


class Info(S):
    _fields_ = [("width", u32), ("height", u32), ("bit_depth_luma", u8),
                ("sar_width", u32), ("sar_height", u32), ("crop_left", u32),
                ("crop_top", u32), ("crop_width", u32), ("crop_height", u32),
                ("full_range", cint), ("colour_description_present", cint),
                ("colour_primaries", u32), ("transfer_characteristics", u32),
                ("matrix_coefficients", u32), ("num_units_in_tick", u32),
                ("time_scale", u32), ("framerate", C.c_float),
                ("framerate_num", u32), ("framerate_den", u32),
                ("nal_hrd_bitrate", u32), ("nal_hrd_cpb_size", u32),
                ("vcl_hrd_bitrate", u32), ("vcl_hrd_cpb_size", u32)]


class Bitstream(S):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("off", C.c_size_t),
                ("cache", u8), ("cachebits", u8),
                ("emulation_prevention", cint), ("dynamic", cint),
                ("priv", C.c_void_p)]


# ---- engine records (include/h264_amd.h) ----------------------------------

class Nal(S):
    _fields_ = [("start", u64), ("end", u64)]


class SliceDesc(S):
    _fields_ = [("offset", u64), ("size", u32), ("epoch", u32)]


class MbSummary(S):
    _fields_ = [("mb_type", u8), ("cbp", u8), ("flags", u8),
                ("raw_mb_type", u8), ("mb_qp_delta", i16),
                ("intra_chroma_pred_mode", u8), ("intra16x16_pred_mode", u8),
                ("pred", u8 * 8), ("nz_coeff", u8 * 48)]


class SliceResult(S):
    _fields_ = [("status", i32), ("flags", u32), ("nal_ref_idc", u32),
                ("nal_unit_type", u32), ("slice_type", i32),
                ("first_mb_in_slice", u32), ("mb_count", u32),
                ("slice_qp", i32), ("hdr_len_bits", u32),
                ("rawdata_off", u32), ("rawdata_len", u32), ("partial", u8),
                ("partialbits", u8), ("chroma_array_type", u16),
                ("end_bit", u32), ("pic_width_in_mbs", u32),
                ("checksum", u64)]


class Ev(S):
    _fields_ = [("kind", u32), ("a", u32), ("b", u64), ("c", u64)]


class Rewrite(S):
    _fields_ = [("off", u64), ("len", u32), ("res", i32), ("nal_index", u32),
                ("match", i32)]


EV_NAMES = {1: "au_end", 2: "nalu_begin", 3: "nalu_end", 4: "slice",
            5: "slice_data_begin", 6: "slice_data_end", 7: "slice_data_mb",
            8: "sps", 9: "pps", 10: "aud", 11: "sei", 12: "sei_typed",
            13: "parse_result"}

# sizes measured on the reference
REFERENCE_SIZEOF = {
    "h264_bitstream": 48, "h264_nalu_header": 12, "h264_sps": 4184,
    "h264_pps": 3276, "h264_slice_header": 3968, "h264_sei": 544,
    "h264_sps_derived": 88, "h264_info": 92, "h264_ctx_cbs": 144,
}


def t2_stride(cat):
    return 256 + 2 * {0: 256, 1: 384, 2: 512}.get(cat, 768)


# ---- engine argument blocks and generator structs --------------------------

class DecodeArgs(S):
    _fields_ = [("d_buf", C.c_void_p), ("d_descs", C.c_void_p), ("n", u32),
                ("reader_flags", u32), ("tier", u32), ("mb_capacity", u32),
                ("d_results", C.c_void_p), ("d_headers", C.c_void_p),
                ("d_mb_out", C.c_void_p), ("mb_stride", u32),
                ("reserved", u32), ("d_buf_len", u64)]


class WriteArgs(S):
    _fields_ = [("d_src", C.c_void_p), ("d_descs", C.c_void_p),
                ("d_results", C.c_void_p), ("d_headers", C.c_void_p),
                ("n", u32), ("wflags", u32), ("d_out", C.c_void_p),
                ("d_out_offsets", C.c_void_p), ("d_out_sizes", C.c_void_p),
                ("out_capacity", u32), ("reserved2", u32),
                ("d_out_hdr_bits", C.c_void_p), ("d_src_len", u64)]


class SynthDesc(S):
    _fields_ = [("kind", u32), ("epoch", u32), ("mb_count", u32),
                ("nal_ref_idc", u32), ("nal_unit_type", u32),
                ("reserved", u32)]


class SynthArgs(S):
    _fields_ = [("d_descs", C.c_void_p), ("d_headers", C.c_void_p),
                ("n", u32), ("out_capacity", u32), ("d_out", C.c_void_p),
                ("d_out_offsets", C.c_void_p), ("d_out_sizes", C.c_void_p),
                ("d_out_hdr_bits", C.c_void_p), ("wflags", u32),
                ("reserved", u32)]


class GenCfg(S):
    _fields_ = ([(n, u32) for n in (
        "width_mbs height_mbs profile_idc level_idc chroma_format_idc "
        "bit_depth_luma_minus8 bit_depth_chroma_minus8 "
        "log2_max_frame_num_minus4 pic_order_cnt_type "
        "log2_max_poc_lsb_minus4 max_num_ref_frames "
        "direct_8x8_inference_flag crop_bottom sps_id pps_id "
        "entropy_coding_mode_flag transform_8x8_mode_flag "
        "deblocking_filter_control_present_flag "
        "num_ref_idx_l0_default_minus1 "
        "num_ref_idx_l1_default_minus1").split()] +
        [("pic_init_qp_minus26", i32)] +
        [(n, u32) for n in (
            "weighted_pred_flag weighted_bipred_idc "
            "redundant_pic_cnt_present_flag "
            "bottom_field_pic_order_in_frame_present_flag "
            "constrained_intra_pred_flag vui field_coding "
            "delta_pic_order_always_zero_flag "
            "num_slice_groups_minus1 slice_group_map_type").split()] +
        [("run_length_minus1", u32 * 8), ("top_left", u32 * 8),
         ("bottom_right", u32 * 8),
         ("slice_group_change_direction_flag", u32),
         ("slice_group_change_rate_minus1", u32),
         ("slice_group_id", C.c_uint8 * 256)])


class GenSlice(S):
    _fields_ = ([(n, u32) for n in (
        "slice_type idr nal_ref_idc first_mb num_mbs frame_num idr_pic_id "
        "poc_lsb").split()] +
        [("slice_qp_delta", i32), ("disable_deblocking_filter_idc", u32),
         ("alpha_c0_offset_div2", i32), ("beta_offset_div2", i32)] +
        [(n, u32) for n in (
            "num_ref_idx_l0_minus1 num_ref_idx_l1_minus1 cabac_init_idc "
            "hdr_extras pm_i16x16 pm_pcm pm_skip pm_intra_in_inter pm_t8x8 "
            "pm_escape pm_zero_pcm density pm_cbp_zero end_with_skip "
            "cabac_payload field_pic slice_group_change_cycle "
            "cabac_syntax").split()] +
        [("seed", u64)])


TIER_NONE, TIER_T0, TIER_T1, TIER_T2 = 0, 1, 2, 3
READER_FLAGS_SLICE_DATA = 1
READER_FLAGS_CABAC = 0x100
READER_FLAGS_CABAC_X264_PRE151 = 0x200  # engine extension, see h264_amd.h
READER_FLAGS_CABAC_STD_QP = 0x400  # engine extension, see h264_amd.h
READER_FLAGS_CABAC_SUBSET = READER_FLAGS_CABAC  # round-2 name
SR_HEADER_OK, SR_DATA_PARSED, SR_CABAC = 1, 2, 4

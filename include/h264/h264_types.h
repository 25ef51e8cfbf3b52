/**
 * libh264_amd -- drop-in public types for the libh264 C API.
 *
 * These declarations ARE the output ABI of the engine: every struct below has
 * the same member order, member types and therefore the same size/offsets as
 * the reference's include/h264/h264_types.h (cited per struct), so that a
 * caller compiled against the reference headers can be re-linked against
 * libh264_amd.so unchanged.  tests/test_abi.py pins sizeof() against the
 * values measured on the reference.
 *
 * The device kernels include this header too: the slice-header record written
 * by the GPU is a byte image of struct h264_slice_header.
 */
#ifndef _H264_TYPES_H_
#define _H264_TYPES_H_

/* ---- enums (ref: include/h264/h264_types.h:34-212) ---------------------- */

/* nal_unit_type, H.264 Table 7-1 (ref :34-48) */
enum h264_nalu_type {
	H264_NALU_TYPE_UNKNOWN = 0,
	H264_NALU_TYPE_SLICE = 1,
	H264_NALU_TYPE_SLICE_DPA = 2,
	H264_NALU_TYPE_SLICE_DPB = 3,
	H264_NALU_TYPE_SLICE_DPC = 4,
	H264_NALU_TYPE_SLICE_IDR = 5,
	H264_NALU_TYPE_SEI = 6,
	H264_NALU_TYPE_SPS = 7,
	H264_NALU_TYPE_PPS = 8,
	H264_NALU_TYPE_AUD = 9,
	H264_NALU_TYPE_END_OF_SEQ = 10,
	H264_NALU_TYPE_END_OF_STREAM = 11,
	H264_NALU_TYPE_FILLER = 12,
};

/* slice_type modulo 5, H.264 Table 7-6 (ref :54-64) */
enum h264_slice_type {
	H264_SLICE_TYPE_UNKNOWN = -1,
	H264_SLICE_TYPE_P = 0,
	H264_SLICE_TYPE_B = 1,
	H264_SLICE_TYPE_I = 2,
	H264_SLICE_TYPE_SP = 3,
	H264_SLICE_TYPE_SI = 4,
};
#define H264_SLICE_TYPE(_val) ((_val) % 5)

/* Macroblock classes; numbering is the reference's, not the spec's (ref
 * :72-91).  The values travel in the slice_data_mb callback and in the GPU
 * macroblock records. */
enum h264_mb_type {
	H264_MB_TYPE_UNKNOWN = 0,
	H264_MB_TYPE_I_NxN,
	H264_MB_TYPE_I_16x16,
	H264_MB_TYPE_I_PCM,
	H264_MB_TYPE_SI,
	H264_MB_TYPE_P_16x16,
	H264_MB_TYPE_P_16x8,
	H264_MB_TYPE_P_8x16,
	H264_MB_TYPE_P_8x8,
	H264_MB_TYPE_P_8x8ref0,
	H264_MB_TYPE_P_SKIP,
	H264_MB_TYPE_B_Direct_16x16,
	H264_MB_TYPE_B_16x16,
	H264_MB_TYPE_B_16x8,
	H264_MB_TYPE_B_8x16,
	H264_MB_TYPE_B_8x8,
	H264_MB_TYPE_B_SKIP,
};

/* profile_idc (ref :97-106) */
enum h264_profile {
	H264_PROFILE_CAVLC_444 = 44,
	H264_PROFILE_BASELINE = 66,
	H264_PROFILE_MAIN = 77,
	H264_PROFILE_EXTENDED = 88,
	H264_PROFILE_HIGH = 100,
	H264_PROFILE_HIGH_10 = 110,
	H264_PROFILE_HIGH_422 = 122,
	H264_PROFILE_HIGH_444 = 244,
};

/* chroma_format_idc (ref :112-124) */
enum h264_color_format {
	H264_COLOR_FORMAT_MONO = 0,
	H264_COLOR_FORMAT_YUV420 = 1,
	H264_COLOR_FORMAT_YUV422 = 2,
	H264_COLOR_FORMAT_YUV444 = 3,
};

/* aspect_ratio_idc, H.264 Table E-1 (ref :130-150) */
enum h264_aspect_ratio {
	H264_ASPECT_RATIO_UNSPECIFIED = 0,
	H264_ASPECT_RATIO_1_1 = 1,
	H264_ASPECT_RATIO_12_11 = 2,
	H264_ASPECT_RATIO_10_11 = 3,
	H264_ASPECT_RATIO_16_11 = 4,
	H264_ASPECT_RATIO_40_33 = 5,
	H264_ASPECT_RATIO_24_11 = 6,
	H264_ASPECT_RATIO_20_11 = 7,
	H264_ASPECT_RATIO_32_11 = 8,
	H264_ASPECT_RATIO_80_33 = 9,
	H264_ASPECT_RATIO_18_11 = 10,
	H264_ASPECT_RATIO_15_11 = 11,
	H264_ASPECT_RATIO_64_33 = 12,
	H264_ASPECT_RATIO_160_99 = 13,
	H264_ASPECT_RATIO_4_3 = 14,
	H264_ASPECT_RATIO_3_2 = 15,
	H264_ASPECT_RATIO_2_1 = 16,
	H264_ASPECT_RATIO_EXTENDED_SAR = 255,
};

/* SEI payloadType, H.264 D.1 (ref :156-212); enumerators 0..54 in order */
enum h264_sei_type {
	H264_SEI_TYPE_BUFFERING_PERIOD = 0,
	H264_SEI_TYPE_PIC_TIMING,
	H264_SEI_TYPE_PAN_SCAN_RECT,
	H264_SEI_TYPE_FILLER_PAYLOAD,
	H264_SEI_TYPE_USER_DATA_REGISTERED,
	H264_SEI_TYPE_USER_DATA_UNREGISTERED,
	H264_SEI_TYPE_RECOVERY_POINT,
	H264_SEI_TYPE_DEC_REF_PIC_MARKING_REPETITION,
	H264_SEI_TYPE_SPARE_PIC,
	H264_SEI_TYPE_SCENE_INFO,
	H264_SEI_TYPE_SUB_SEQ_INFO, /* 10 */
	H264_SEI_TYPE_SUB_SEQ_LAYER_CHARACTERISTICS,
	H264_SEI_TYPE_SUB_SEQ_CHARACTERISTICS,
	H264_SEI_TYPE_FULL_FRAME_FREEZE,
	H264_SEI_TYPE_FULL_FRAME_FREEZE_RELEASE,
	H264_SEI_TYPE_FULL_FRAME_SNAPSHOT,
	H264_SEI_TYPE_PROGRESSIVE_REFINEMENT_SEGMENT_START,
	H264_SEI_TYPE_PROGRESSIVE_REFINEMENT_SEGMENT_END,
	H264_SEI_TYPE_MOTION_CONSTRAINED_SLICE_GROUP_SET,
	H264_SEI_TYPE_FILM_GRAIN_CHARACTERISTICS,
	H264_SEI_TYPE_DEBLOCKING_FILTER_DISPLAY_PREFERENCE, /* 20 */
	H264_SEI_TYPE_STEREO_VIDEO_INFO,
	H264_SEI_TYPE_POST_FILTER_HINT,
	H264_SEI_TYPE_TONE_MAPPING_INFO,
	H264_SEI_TYPE_SCALABILITY_INFO,
	H264_SEI_TYPE_SUB_PIC_SCALABLE_LAYER,
	H264_SEI_TYPE_NON_REQUIRED_LAYER_REP,
	H264_SEI_TYPE_PRIORITY_LAYER_INFO,
	H264_SEI_TYPE_LAYERS_NOT_PRESENT,
	H264_SEI_TYPE_LAYER_DEPENDENCY_CHANGE,
	H264_SEI_TYPE_SCALABLE_NESTING, /* 30 */
	H264_SEI_TYPE_BASE_LAYER_TEMPORAL_HRD,
	H264_SEI_TYPE_QUALITY_LAYER_INTEGRITY_CHECK,
	H264_SEI_TYPE_REDUNDANT_PIC_PROPERTY,
	H264_SEI_TYPE_TL0_DEP_REP_INDEX,
	H264_SEI_TYPE_TL_SWITCHING_POINT,
	H264_SEI_TYPE_PARALLEL_DECODING_INFO,
	H264_SEI_TYPE_MVC_SCALABLE_NESTING,
	H264_SEI_TYPE_VIEW_SCALABILITY_INFO,
	H264_SEI_TYPE_MULTIVIEW_SCENE_INFO,
	H264_SEI_TYPE_MULTIVIEW_ACQUISITION_INFO, /* 40 */
	H264_SEI_TYPE_NON_REQUIRED_VIEW_COMPONENT,
	H264_SEI_TYPE_VIEW_DEPENDENCY_CHANGE,
	H264_SEI_TYPE_OPERATION_POINTS_NOT_PRESENT,
	H264_SEI_TYPE_BASE_VIEW_TEMPORAL_HRD,
	H264_SEI_TYPE_FRAME_PACKING_ARRANGEMENT,
	H264_SEI_TYPE_MULTIVIEW_VIEW_POSITION,
	H264_SEI_TYPE_DISPLAY_ORIENTATION,
	H264_SEI_TYPE_MVCD_SCALABLE_NESTING,
	H264_SEI_TYPE_MVCD_VIEW_SCALABILITY_INFO,
	H264_SEI_TYPE_DEPTH_REPRESENTATION_INFO, /* 50 */
	H264_SEI_TYPE_THREE_DIMENSIONAL_REFERENCE_DISPLAYS_INFO,
	H264_SEI_TYPE_DEPTH_TIMING,
	H264_SEI_TYPE_DEPTH_SAMPLING_INFO,
	H264_SEI_TYPE_CONSTRAINED_DEPTH_PARAMETER_SET_IDENTIFIER,
};

/* ---- syntax structures --------------------------------------------------- */

/* 7.3.1 NAL unit header, 12 bytes (ref :218-222) */
struct h264_nalu_header {
	uint32_t forbidden_zero_bit;
	uint32_t nal_ref_idc;
	uint32_t nal_unit_type;
};

/* Scaling lists of SPS / PPS (ref :229-238).  The two _optimized_* arrays
 * record how the list was terminated in the original bitstream so that the
 * writer can reproduce the exact same bits. */
struct h264_scaling_matrix {
	int scaling_list_present_flag[12];
	int32_t scaling_list_4x4[6][16];
	int32_t scaling_list_8x8[6][64];
	int use_default_4x4[6];
	int use_default_8x8[6];
	int _optimized_4x4[6];
	int _optimized_8x8[6];
};

/* E.1.2 HRD parameters (ref :244-260) */
struct h264_hrd {
	uint32_t cpb_cnt_minus1;
	uint32_t bit_rate_scale;
	uint32_t cpb_size_scale;
	struct {
		uint32_t bit_rate_value_minus1;
		uint32_t cpb_size_value_minus1;
		int cbr_flag;
	} cpb[32];
	uint32_t initial_cpb_removal_delay_length_minus1;
	uint32_t cpb_removal_delay_length_minus1;
	uint32_t dpb_output_delay_length_minus1;
	uint32_t time_offset_length;
};

/* E.1.1 VUI parameters (ref :266-301) */
struct h264_vui {
	int aspect_ratio_info_present_flag;
	uint32_t aspect_ratio_idc;
	uint32_t sar_width;
	uint32_t sar_height;
	int overscan_info_present_flag;
	int overscan_appropriate_flag;
	int video_signal_type_present_flag;
	uint32_t video_format;
	int video_full_range_flag;
	int colour_description_present_flag;
	uint32_t colour_primaries;
	uint32_t transfer_characteristics;
	uint32_t matrix_coefficients;
	int chroma_loc_info_present_flag;
	uint32_t chroma_sample_loc_type_top_field;
	uint32_t chroma_sample_loc_type_bottom_field;
	int timing_info_present_flag;
	uint32_t num_units_in_tick;
	uint32_t time_scale;
	int fixed_frame_rate_flag;
	int nal_hrd_parameters_present_flag;
	struct h264_hrd nal_hrd;
	int vcl_hrd_parameters_present_flag;
	struct h264_hrd vcl_hrd;
	int low_delay_hrd_flag;
	int pic_struct_present_flag;
	int bitstream_restriction_flag;
	int motion_vectors_over_pic_boundaries_flag;
	uint32_t max_bytes_per_pic_denom;
	uint32_t max_bits_per_mb_denom;
	uint32_t log2_max_mv_length_horizontal;
	uint32_t log2_max_mv_length_vertical;
	uint32_t max_num_reorder_frames;
	uint32_t max_dec_frame_buffering;
};

/* 7.3.2.1 Sequence parameter set, 4184 bytes (ref :307-352) */
struct h264_sps {
	uint32_t profile_idc;
	int constraint_set0_flag;
	int constraint_set1_flag;
	int constraint_set2_flag;
	int constraint_set3_flag;
	int constraint_set4_flag;
	int constraint_set5_flag;
	uint32_t reserved_zero_2bits;
	uint32_t level_idc;
	uint32_t seq_parameter_set_id;
	uint32_t chroma_format_idc;
	int separate_colour_plane_flag;
	uint32_t bit_depth_luma_minus8;
	uint32_t bit_depth_chroma_minus8;
	int qpprime_y_zero_transform_bypass_flag;
	int seq_scaling_matrix_present_flag;
	struct h264_scaling_matrix seq_scaling_matrix;
	uint32_t log2_max_frame_num_minus4;
	uint32_t pic_order_cnt_type;
	uint32_t log2_max_pic_order_cnt_lsb_minus4;
	int delta_pic_order_always_zero_flag;
	int32_t offset_for_non_ref_pic;
	int32_t offset_for_top_to_bottom_field;
	uint32_t num_ref_frames_in_pic_order_cnt_cycle;
	int32_t offset_for_ref_frame[256];
	uint32_t max_num_ref_frames;
	int gaps_in_frame_num_value_allowed_flag;
	uint32_t pic_width_in_mbs_minus1;
	uint32_t pic_height_in_map_units_minus1;
	int frame_mbs_only_flag;
	int mb_adaptive_frame_field_flag;
	int direct_8x8_inference_flag;
	int frame_cropping_flag;
	uint32_t frame_crop_left_offset;
	uint32_t frame_crop_right_offset;
	uint32_t frame_crop_top_offset;
	uint32_t frame_crop_bottom_offset;
	int vui_parameters_present_flag;
	struct h264_vui vui;
};

/* 7.3.2.2 Picture parameter set, 3276 bytes (ref :358-398) */
struct h264_pps {
	uint32_t pic_parameter_set_id;
	uint32_t seq_parameter_set_id;
	int entropy_coding_mode_flag;
	int bottom_field_pic_order_in_frame_present_flag;
	uint32_t num_slice_groups_minus1;
	uint32_t slice_group_map_type;
	uint32_t run_length_minus1[8];
	uint32_t top_left[8];
	uint32_t bottom_right[8];
	int slice_group_change_direction_flag;
	uint32_t slice_group_change_rate_minus1;
	uint32_t pic_size_in_map_units_minus1;
	uint32_t slice_group_id[256];
	uint32_t num_ref_idx_l0_default_active_minus1;
	uint32_t num_ref_idx_l1_default_active_minus1;
	int weighted_pred_flag;
	uint32_t weighted_bipred_idc;
	int32_t pic_init_qp_minus26;
	int32_t pic_init_qs_minus26;
	int32_t chroma_qp_index_offset;
	int deblocking_filter_control_present_flag;
	int constrained_intra_pred_flag;
	int redundant_pic_cnt_present_flag;
	/* set by the reader when the optional PPS tail is present, consumed by
	 * the writer */
	int _more_rbsp_data_present;
	int transform_8x8_mode_flag;
	int pic_scaling_matrix_present_flag;
	struct h264_scaling_matrix pic_scaling_matrix;
	int32_t second_chroma_qp_index_offset;
};

/* 7.3.2.4 Access unit delimiter (ref :404-406) */
struct h264_aud {
	uint32_t primary_pic_type;
};

/* 7.3.3.1 ref_pic_list_modification (ref :413-436) */
struct h264_rplm_item {
	uint32_t modification_of_pic_nums_idc;
	union {
		uint32_t abs_diff_pic_num_minus1;
		uint32_t long_term_pic_num;
		uint32_t abs_diff_view_idx_minus1;
	};
};

struct h264_rplm {
	int ref_pic_list_modification_flag_l0;
	struct h264_rplm_item pic_num_l0[32];
	int ref_pic_list_modification_flag_l1;
	struct h264_rplm_item pic_num_l1[32];
};

/* 7.3.3.2 pred_weight_table (ref :442-464) */
struct h264_pwt_item {
	int luma_weight_flag;
	int32_t luma_weight;
	int32_t luma_offset;
	int chroma_weight_flag;
	int32_t chroma_weight[2];
	int32_t chroma_offset[2];
};

struct h264_pwt {
	uint32_t luma_log2_weight_denom;
	uint32_t chroma_log2_weight_denom;
	struct h264_pwt_item l0[32];
	struct h264_pwt_item l1[32];
};

/* 7.3.3.3 dec_ref_pic_marking (ref :470-490) */
struct h264_drpm_item {
	uint32_t memory_management_control_operation;
	uint32_t difference_of_pic_nums_minus1;
	uint32_t long_term_pic_num;
	uint32_t long_term_frame_idx;
	uint32_t max_long_term_frame_idx_plus1;
};

struct h264_drpm {
	int no_output_of_prior_pics_flag;
	int long_term_reference_flag;
	int adaptive_ref_pic_marking_mode_flag;
	struct h264_drpm_item mm[64];
};

/* 7.3.3 Slice header, 3968 bytes (ref :496-528) */
struct h264_slice_header {
	uint32_t first_mb_in_slice;
	uint32_t slice_type;
	uint32_t pic_parameter_set_id;
	uint32_t colour_plane_id;
	uint32_t frame_num;
	int field_pic_flag;
	int bottom_field_flag;
	uint32_t idr_pic_id;
	uint32_t pic_order_cnt_lsb;
	int32_t delta_pic_order_cnt_bottom;
	int32_t delta_pic_order_cnt[2];
	uint32_t redundant_pic_cnt;
	int direct_spatial_mv_pred_flag;
	int num_ref_idx_active_override_flag;
	uint32_t num_ref_idx_l0_active_minus1;
	uint32_t num_ref_idx_l1_active_minus1;
	struct h264_rplm rplm;
	struct h264_pwt pwt;
	struct h264_drpm drpm;
	uint32_t cabac_init_idc;
	int32_t slice_qp_delta;
	int sp_for_switch_flag;
	int32_t slice_qs_delta;
	uint32_t disable_deblocking_filter_idc;
	int32_t slice_alpha_c0_offset_div2;
	int32_t slice_beta_offset_div2;
	uint32_t slice_group_change_cycle;
};

/* ---- SEI payloads (ref :534-654) ----------------------------------------- */

struct h264_sei_buffering_period {
	uint32_t seq_parameter_set_id;
	struct {
		uint32_t initial_cpb_removal_delay;
		uint32_t initial_cpb_removal_delay_offset;
	} nal_hrd_cpb[32];
	struct {
		uint32_t initial_cpb_removal_delay;
		uint32_t initial_cpb_removal_delay_offset;
	} vcl_hrd_cpb[32];
};

struct h264_sei_pic_timing {
	uint32_t cpb_removal_delay;
	uint32_t dpb_output_delay;
	uint32_t pic_struct;
	struct {
		int clock_timestamp_flag;
		uint32_t ct_type;
		int nuit_field_based_flag;
		uint32_t counting_type;
		int full_timestamp_flag;
		int discontinuity_flag;
		int cnt_dropped_flag;
		uint32_t n_frames;
		uint32_t seconds_value;
		uint32_t minutes_value;
		uint32_t hours_value;
		int seconds_flag;
		int minutes_flag;
		int hours_flag;
		int32_t time_offset;
	} clk_ts[3];
};

struct h264_sei_pan_scan_rect {
	uint32_t pan_scan_rect_id;
	int pan_scan_rect_cancel_flag;
	uint32_t pan_scan_cnt_minus1;
	struct {
		int32_t left_offset;
		int32_t right_offset;
		int32_t top_offset;
		int32_t bottom_offset;
	} pan_scan_rect[4];
	uint32_t pan_scan_rect_repetition_period;
};

struct h264_sei_filler_payload {
	const uint8_t *buf;
	size_t len;
};

struct h264_sei_user_data_registered {
	uint32_t country_code;
	uint32_t country_code_extension_byte;
	const uint8_t *buf;
	size_t len;
};

struct h264_sei_user_data_unregistered {
	uint8_t uuid[16];
	const uint8_t *buf;
	size_t len;
};

struct h264_sei_recovery_point {
	uint32_t recovery_frame_cnt;
	int exact_match_flag;
	int broken_link_flag;
	uint32_t changing_slice_group_idc;
};

/* One SEI message, 544 bytes (ref :637-654).  raw is owned by the context. */
struct h264_sei {
	enum h264_sei_type type;
	union {
		struct h264_sei_buffering_period buffering_period;
		struct h264_sei_pic_timing pic_timing;
		struct h264_sei_pan_scan_rect pan_scan_rect;
		struct h264_sei_filler_payload filler_payload;
		struct h264_sei_user_data_registered user_data_registered;
		struct h264_sei_user_data_unregistered user_data_unregistered;
		struct h264_sei_recovery_point recovery_point;
	};
	struct {
		uint8_t *buf;
		size_t len;
	} raw;
};

/* ---- derived values ------------------------------------------------------- */

/* Variables derived from an SPS, 22 x uint32 = 88 bytes (ref :663-687).
 * The slice kernels take these as constants. */
struct h264_sps_derived {
	uint32_t ChromaArrayType;
	uint32_t SubWidthC;
	uint32_t SubHeightC;
	uint32_t MbWidthC;
	uint32_t MbHeightC;
	uint32_t BitDepthLuma;
	uint32_t QpBdOffsetLuma;
	uint32_t BitDepthChroma;
	uint32_t QpBdOffsetChroma;
	uint32_t RawMbBits;
	uint32_t MaxFrameNum;
	uint32_t MaxPicOrderCntLsb;
	uint32_t PicWidthInMbs;
	uint32_t PicWidthInSamplesLuma;
	uint32_t PicWidthInSamplesChroma;
	uint32_t PicHeightInMapUnits;
	uint32_t PicSizeInMapUnits;
	uint32_t FrameHeightInMbs;
	uint32_t CropUnitX;
	uint32_t CropUnitY;
	uint32_t Width;
	uint32_t Height;
};

/* Stream summary from one SPS + PPS, 92 bytes (ref :691-764) */
struct h264_info {
	uint32_t width;
	uint32_t height;
	uint8_t bit_depth_luma;
	uint32_t sar_width;
	uint32_t sar_height;
	uint32_t crop_left;
	uint32_t crop_top;
	uint32_t crop_width;
	uint32_t crop_height;
	int full_range;
	int colour_description_present;
	uint32_t colour_primaries;
	uint32_t transfer_characteristics;
	uint32_t matrix_coefficients;
	uint32_t num_units_in_tick;
	uint32_t time_scale;
	float framerate;
	uint32_t framerate_num;
	uint32_t framerate_den;
	uint32_t nal_hrd_bitrate;
	uint32_t nal_hrd_cpb_size;
	uint32_t vcl_hrd_bitrate;
	uint32_t vcl_hrd_cpb_size;
};

/* ---- string / predicate helpers (ref :767-802) --------------------------- */

H264_API const char *h264_nalu_type_str(enum h264_nalu_type val);
H264_API const char *h264_slice_type_str(enum h264_slice_type val);
H264_API const char *h264_mb_type_str(enum h264_mb_type val);
H264_API int h264_mb_type_is_intra(enum h264_mb_type val);
H264_API int h264_mb_type_is_inter(enum h264_mb_type val);
H264_API const char *h264_profile_str(enum h264_profile val);
H264_API const char *h264_color_format_str(enum h264_color_format val);
/* returns a malloc()ed string, caller frees */
H264_API char *h264_aspect_ratio_str_alloc(enum h264_aspect_ratio val,
					   uint32_t sar_width,
					   uint32_t sar_height);
H264_API const char *h264_sei_type_str(enum h264_sei_type val);

#endif /* !_H264_TYPES_H_ */

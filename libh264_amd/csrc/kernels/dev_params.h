/**
 * dev_params.h -- device-side view of parameter sets (host + device).
 *
 * A slice header can only be parsed with the SPS/PPS that were active at the
 * slice's position in the stream (ref src/h264_syntax.h:1256-1258: the header
 * reads pic_parameter_set_id and calls h264_ctx_set_active_pps, which then
 * activates the SPS the PPS names, src/h264_ctx.c:366-386).  The host keeps
 * every parameter-set VERSION it has seen in compact form and, per epoch, the
 * version each of the 256 PPS ids / 32 SPS ids pointed to; the kernels do the
 * two look-ups themselves.
 */
#ifndef H264_AMD_DEV_PARAMS_H
#define H264_AMD_DEV_PARAMS_H

#include <stdint.h>

#define DEV_NO_VERSION 0xffffu

/* fields of struct h264_sps the slice kernels read (32 bytes).  The reference
 * range-checks almost nothing in an SPS/PPS (src/h264_syntax.h:278-345), so a
 * damaged parameter set can carry any 32-bit value: byte fields that are only
 * compared with small constants SATURATE at 255 (DEV_SAT8), field widths keep
 * all 32 bits (a u(n) with n > 32 is legal for the reference's reader,
 * include/h264/h264_bitstream.h:192-218). */
struct dev_sps {
	uint8_t chroma_format_idc; /* saturating */
	uint8_t separate_colour_plane_flag;
	uint8_t bit_depth_luma; /* 8..14, checked by the SPS parser */
	uint8_t bit_depth_chroma;
	uint8_t pic_order_cnt_type; /* saturating */
	uint8_t delta_pic_order_always_zero_flag;
	uint8_t frame_mbs_only_flag;
	uint8_t mb_adaptive_frame_field_flag;
	uint8_t direct_8x8_inference_flag;
	uint8_t pad0[3];
	uint32_t pic_width_in_mbs;
	uint32_t pic_height_in_map_units;
	uint32_t log2_max_frame_num; /* 4 + minus4, modulo 2^32 like the reference */
	uint32_t log2_max_pic_order_cnt_lsb;
	uint32_t pad1;
};
#define DEV_SAT8(x) ((uint8_t)((x) < 255u ? (x) : 255u))

/* fields of struct h264_pps the slice kernels read (32 bytes) */
struct dev_pps {
	uint8_t seq_parameter_set_id;
	uint8_t entropy_coding_mode_flag;
	uint8_t bottom_field_pic_order_in_frame_present_flag;
	uint8_t num_slice_groups_minus1; /* saturating */
	uint8_t slice_group_map_type; /* saturating */
	uint8_t weighted_pred_flag;
	uint8_t weighted_bipred_idc;
	uint8_t deblocking_filter_control_present_flag;
	uint8_t redundant_pic_cnt_present_flag;
	uint8_t transform_8x8_mode_flag;
	uint8_t constrained_intra_pred_flag;
	uint8_t pad0;
	int32_t pic_init_qp_minus26;
	int32_t pic_init_qs_minus26;
	uint32_t num_ref_idx_l0_default_active_minus1;
	uint32_t num_ref_idx_l1_default_active_minus1;
	uint32_t slice_group_change_rate_minus1;
};

/* slice-group fields of a PPS (7.3.2.2), one record per PPS version next to
 * its dev_pps; only read when num_slice_groups_minus1 > 0.  Values are kept
 * in full: the map builders bound every use (kernels/slice_decode.hip,
 * sd_fmo_build). */
struct dev_fmo {
	uint32_t slice_group_change_direction_flag;
	uint32_t pic_size_in_map_units_minus1;
	uint32_t run_length_minus1[8];
	uint32_t top_left[8];
	uint32_t bottom_right[8];
	uint8_t slice_group_id[256];
};
/* largest picture (macroblocks of a frame) the per-slice map scratch is sized
 * for: level 6.2 allows 139264; beyond it slices of a picture with slice
 * groups answer H264_AMD_E_UNSUPPORTED */
#define DEV_FMO_MAX_MBS H264_AMD_FMO_MAX_PIC_MBS

/* which version every id names during one epoch */
struct dev_epoch {
	uint16_t pps_ver[256];
	uint16_t sps_ver[32];
};

/* CAVLC prefix-code tables as the slice kernel keeps them in LDS (layout and
 * use: vlc_lanes.hip.h).  One code word per entry: mask covers the code's bits
 * (left aligned), cw = the code left aligned | payload in the low 16 bits. */
struct dev_vlc_entry {
	uint32_t mask, cw;
};
#define DEV_VLC_BAD 0x8000u /* payload bit of a coeff_token catch-all entry */
#define DEV_VLC_CT_CLASSES 6 /* nC<2, <4, <8, >=8 (FLC), chroma DC 420, 422 */
#define DEV_VLC_TZ_TABLES 25 /* 4x4 tzVlcIndex 1..15, 2x2 1..3, 2x4 1..7 */
struct dev_vlc_tables {
	struct dev_vlc_entry ct[DEV_VLC_CT_CLASSES][64];
	struct dev_vlc_entry tz[DEV_VLC_TZ_TABLES][16];
	/* coded_block_pattern mapping, entry = codeNum: bytes cat 1/2 intra,
	 * inter, cat 0/3 intra, inter (Table 9-4) */
	uint32_t cbp[64];
};
#define DEV_VLC_LDS_BYTES                                                      \
	((DEV_VLC_CT_CLASSES * 64 + DEV_VLC_TZ_TABLES * 16) * 8)

#endif /* H264_AMD_DEV_PARAMS_H */

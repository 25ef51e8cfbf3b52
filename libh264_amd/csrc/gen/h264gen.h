/**
 * h264gen -- seeded synthetic H.264 stream generator (host, plain C).
 *
 * The container has no encoder (no ffmpeg/x264) and the reference ships no
 * streams, so the workloads of BASELINE.json are synthesised: parameter sets,
 * slice headers and random-but-valid CAVLC slice_data (I/P/B macroblocks,
 * I_PCM, escapes, skip runs), written through an emulation-preventing bit
 * writer.  The generator is an ENCODER written from the syntax tables; it
 * shares no code with the parsers it feeds, and it can emit, next to the
 * bytes, the macroblock records (T2 layout of <h264_amd.h>) it encoded --
 * an independent expectation for the oracle and for the kernels.
 *
 * Workload recipe: SURVEY.md 8(d) (C1..C5) and Appendix E.
 */
#ifndef H264GEN_H
#define H264GEN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct h264gen_cfg {
	uint32_t width_mbs, height_mbs;
	uint32_t profile_idc, level_idc;
	uint32_t chroma_format_idc; /* != 1 needs a High-family profile_idc */
	uint32_t bit_depth_luma_minus8, bit_depth_chroma_minus8;
	uint32_t log2_max_frame_num_minus4;
	uint32_t pic_order_cnt_type; /* 0, 1 or 2 */
	uint32_t log2_max_poc_lsb_minus4;
	uint32_t max_num_ref_frames;
	uint32_t direct_8x8_inference_flag;
	uint32_t crop_bottom; /* frame_crop_bottom_offset, 0 = no cropping */
	uint32_t sps_id, pps_id;
	uint32_t entropy_coding_mode_flag;
	uint32_t transform_8x8_mode_flag;
	uint32_t deblocking_filter_control_present_flag;
	uint32_t num_ref_idx_l0_default_minus1, num_ref_idx_l1_default_minus1;
	int32_t pic_init_qp_minus26;
	uint32_t weighted_pred_flag, weighted_bipred_idc;
	uint32_t redundant_pic_cnt_present_flag;
	uint32_t bottom_field_pic_order_in_frame_present_flag;
	uint32_t constrained_intra_pred_flag;
	uint32_t vui; /* 1: emit a VUI with timing + HRD-less restriction info */
	/* 0: frame_mbs_only_flag = 1; 1: interlaced material without MBAFF
	 * (height_mbs = map units of one FIELD); 2: same with
	 * mb_adaptive_frame_field_flag = 1 */
	uint32_t field_coding;
	uint32_t delta_pic_order_always_zero_flag; /* pic_order_cnt_type 1 */
	/* slice groups (FMO, 7.3.2.2 / 8.2.2); num_slice_groups_minus1 = 0: none.
	 * A slice then codes macroblocks of ONE slice group: first_mb picks the
	 * group, num_mbs counts that group's macroblocks from there on. */
	uint32_t num_slice_groups_minus1; /* <= 7 */
	uint32_t slice_group_map_type; /* 0..6 */
	uint32_t run_length_minus1[8]; /* type 0 */
	uint32_t top_left[8], bottom_right[8]; /* type 2 */
	uint32_t slice_group_change_direction_flag; /* types 3..5 */
	uint32_t slice_group_change_rate_minus1;
	uint8_t slice_group_id[256]; /* type 6: pictures of <= 256 map units */
};

#define H264GEN_HDR_RPLM 0x1 /* ref_pic_list_modification entries */
#define H264GEN_HDR_MMCO 0x2 /* adaptive ref pic marking (non-IDR, ref) */
#define H264GEN_HDR_OVERRIDE 0x4 /* num_ref_idx_active_override_flag */

struct h264gen_slice {
	uint32_t slice_type; /* 0 P, 1 B, 2 I (5..7 allowed too) */
	uint32_t idr; /* nal_unit_type 5, else 1 */
	uint32_t nal_ref_idc;
	uint32_t first_mb, num_mbs;
	uint32_t frame_num, idr_pic_id, poc_lsb;
	int32_t slice_qp_delta;
	uint32_t disable_deblocking_filter_idc;
	int32_t alpha_c0_offset_div2, beta_offset_div2;
	uint32_t num_ref_idx_l0_minus1, num_ref_idx_l1_minus1; /* w/ OVERRIDE */
	uint32_t cabac_init_idc;
	uint32_t hdr_extras; /* H264GEN_HDR_* */
	/* content knobs, per mille */
	uint32_t pm_i16x16, pm_pcm, pm_skip, pm_intra_in_inter, pm_t8x8;
	uint32_t pm_escape; /* large levels (level_prefix >= 15) */
	uint32_t pm_zero_pcm; /* I_PCM filled with zeros: escape-byte trigger */
	uint32_t density; /* 0..16: how full coded 4x4 blocks are */
	uint32_t pm_cbp_zero; /* coded MBs without residual */
	uint32_t end_with_skip; /* finish a P/B slice on an mb_skip_run */
	uint32_t cabac_payload; /* CABAC PPS: bytes of opaque slice data */
	uint32_t field_pic; /* field_coding != 0: 0 frame picture, 1 top, 2 bottom field */
	uint32_t slice_group_change_cycle; /* slice group map types 3..5 */
	/* CABAC PPS: 1 = real macroblock syntax through the generator's own
	 * arithmetic encoder (h264gen_cabac.inc; frame and field pictures, no
	 * MBAFF), 2 = the same with the context variables of a slice at
	 * SliceQPY 0 initialised as the standard says (9.3.1.1: QP clipped
	 * to 0..51) instead of the reference's way (1..51, src/h264_bac.c:
	 * 216-230); 0 = cabac_payload bytes of opaque slice data */
	uint32_t cabac_syntax;
	uint64_t seed;
};

/* NAL units without start code; return size or -errno */
int h264gen_sps(const struct h264gen_cfg *cfg, uint8_t *out, size_t cap);
int h264gen_pps(const struct h264gen_cfg *cfg, uint8_t *out, size_t cap);
int h264gen_aud(uint32_t primary_pic_type, uint8_t *out, size_t cap);
int h264gen_filler(uint32_t n_ff, uint8_t *out, size_t cap);
/* user_data_unregistered SEI with n payload bytes after the uuid */
int h264gen_sei_user_data(uint32_t n, uint64_t seed, uint8_t *out, size_t cap);

/* One slice NAL.  t2 (optional): num_mbs * t2_stride bytes receive the
 * expected macroblock records; *mbs receives the number of macroblocks coded
 * (== num_mbs).  Returns size or -errno. */
int h264gen_slice_nal(const struct h264gen_cfg *cfg,
		      const struct h264gen_slice *sl, uint8_t *out, size_t cap,
		      uint8_t *t2, uint32_t t2_stride);

/* n slices generated with `threads` workers.  slices[i].seed distinguishes
 * them.  out receives the NAL units back to back, each starting at a 16-byte
 * aligned offset; offs[i] / sizes[i] describe them.  Returns total bytes used
 * (last NAL end) or -errno (-ENOBUFS when cap is too small). */
long h264gen_batch(const struct h264gen_cfg *cfg,
		   const struct h264gen_slice *slices, uint32_t n, int threads,
		   uint8_t *out, size_t cap, uint64_t *offs, uint32_t *sizes);

/* The macroblock-to-slice-group map of a picture (8.2.2.1 - 8.2.2.8, written
 * from the standard), one byte per macroblock address of a picture coded as
 * `field_pic` (see struct h264gen_slice); returns the number of macroblocks
 * or -errno.  For building slices: the addresses of group g from `first` on
 * are those with map[a] == g, a >= first. */
long h264gen_mb_to_slice_group_map(const struct h264gen_cfg *cfg,
				   uint32_t slice_group_change_cycle,
				   uint32_t field_pic, uint8_t *map, size_t cap);

/* Bins (context coded, bypass and terminating) the generator's arithmetic
 * encoder has coded in all CABAC slices so far -- the unit a CABAC decoder's rate
 * is quoted in; reset != 0 starts the count again. */
uint64_t h264gen_stat_cabac_bins(int reset);

/* Worst-case size of one slice NAL for a config (for sizing `cap`). */
size_t h264gen_slice_bound(const struct h264gen_cfg *cfg,
			   const struct h264gen_slice *sl);

#ifdef __cplusplus
}
#endif
#endif /* H264GEN_H */

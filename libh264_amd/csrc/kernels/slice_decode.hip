/**
 * slice_decode.hip -- one slice NAL unit per wavefront: slice header +
 * CAVLC slice_data, hand written for gfx950 (MI355X).
 *
 * Replaces, for nal_unit_type 1 and 5, what the reference does per NAL in
 *   _h264_read_nalu / slice_layer / slice_header   src/h264_syntax.h:1239-1604
 *   _h264_read_slice_data(_internal)               src/h264_syntax_slice_data.h:701-811
 *   macroblock_layer / mb_pred / sub_mb_pred       src/h264_syntax_slice_data.h:422-696
 *   residual / residual_luma / residual_block      src/h264_syntax_slice_data.h:103-419
 *   mb_type, sub_mb_type, cbp, coeff_token & co    src/h264_slice_data.c:839-1416
 *   neighbour derivation (non-MBAFF)               src/h264_macroblock.c:84-104,306-433
 * with the reference's observable results (slice header struct, rawdata extent,
 * per-macroblock syntax, nz_coeff bookkeeping incl. its quirks, error points).
 *
 * Execution model: one slice per wave, SD_WAVES_PER_BLOCK waves per workgroup
 * (they share the CAVLC tables in LDS and one barrier after staging them, and
 * are independent afterwards); grid = slices / SD_WAVES_PER_BLOCK.  The parse
 * state is wave-uniform and lives in SGPRs (see wave_bits.hip.h); the lanes
 * cooperate for
 *   - ingesting/de-escaping the bitstream (coalesced loads, ballot compaction),
 *   - matching prefix codes (one candidate code word per lane, vlc_lanes.hip.h),
 *   - the 16 intra prediction modes of a macroblock at once (pointer doubling,
 *     sd_intra_pred_modes), I_PCM samples 64 at a time,
 *   - runs of empty 4x4 blocks (prefix sum of code lengths over the blocks still
 *     to come against the string of one bits at the cursor),
 *   - scattering run/level pairs of a 4x4 block into the record (lane i owns
 *     coefficient i; position = zerosLeft_i + TotalCoeff - 1 - i),
 *   - the nz_coeff neighbour context kept as running sums (lane = comp*16 +
 *     blkIdx, struct MbCtx),
 *   - staging the macroblock record in LDS and writing it out with 16-byte
 *     stores, plus the order-independent checksum over it.
 * No MFMA: this is integer/bit work.  LDS: 6.3 KiB of tables per workgroup; per
 * wave 1 KiB RBSP ring + 4 KiB record staging + 16 B per macroblock column of
 * top-row nz context.  Cost model and measurements:
 * profiles/r01_decode_cost_model.md.
 *
 * Slice groups (FMO): the map of 8.2.2 is built per slice by sd_fmo_build and
 * walked by sd_fmo_next (ref src/h264_fmo.c).
 * MBAFF frames: macroblock pairs, see PairCtx (a third instantiation of the
 * slice-data walk).
 */
#include <hip/hip_runtime.h>
#include <errno.h>
#include <stddef.h>
#include <stdint.h>

#include <h264_amd.h>

#include "dev_params.h"
#include "vbits.hip.h"
#include "vlc_lanes.hip.h"
#include "wave_bits.hip.h"

/* LDS: the workgroup's copy of the CAVLC tables, then one private block per
 * wave (SD_WAVES_PER_BLOCK slices per workgroup share the tables) */
#define SD_WAVES_PER_BLOCK 4u
#define SD_TABS_BYTES ((DEV_VLC_LDS_BYTES + 255u) & ~255u)
/* offsets inside a wave's block */
#define SD_RING_OFF 0u
#define SD_STATE_OFF 1024u
#define SD_DUMP_OFF (1024u + WB_STATE_BYTES)
#define SD_STAGE_OFF (SD_DUMP_OFF + 1024u) /* dump area right before stage */
#define SD_STAGE_BYTES 4096u /* the slice header image (3968 B) is built here */
/* the macroblock record staged during slice data needs 1792 bytes at most, and
 * the header image has left for HBM by then: the row context starts inside the
 * staging area's second half (20 instead of 16 waves per CU fit at 1080p) */
#define SD_REC_BYTES 2048u
#define SD_ROW_OFF (SD_STAGE_OFF + SD_REC_BYTES)

/* private enums of the reference (src/h264_macroblock.h:36-60) */
enum { PM_I4x4, PM_I8x8, PM_I16x16, PM_L0, PM_L1, PM_BI, PM_DIRECT };

struct SliceState {
	/* parameter sets */
	uint32_t cat, W, bdl, bdc, pcm_chroma;
	uint32_t t8x8_mode, direct8x8, entropy, unsupported;
	/* slice groups */
	uint32_t mbaff; /* MbaffFrameFlag */
	uint32_t fmo; /* num_slice_groups_minus1 (0: none) */
	uint32_t fmo_type, pps_version, frame_mbs_only, map_units_h;
	uint32_t prio_slot; /* age rank of the wave on its SIMD, modulo 4 */
	uint32_t prio_on; /* 0: the launch fits one dispatch layer, no rotation */
	uint32_t change_cycle, change_rate;
	/* header */
	uint32_t slice_type, first_mb, nref0, nref1, field_pic;
	int32_t slice_qp;
	uint32_t nal_ref_idc, nal_unit_type;
	uint32_t hdr_len_bits, rawdata_off, rawdata_len, partial, partialbits;
};

/* NO DIVERGENT BRANCHES in this kernel.  The whole design rests on the compiler
 * proving the parse state wave-uniform; an `if (lane == 0) store` is a
 * divergent branch, and once the optimiser merges its join block with the
 * surrounding (uniform) control flow the cursor ends up in a phi "below a
 * divergent branch", is declared divergent and drags the entire syntax walk
 * onto the vector unit with exec-mask branches (measured: 2.6 k VALU + 3.1 k
 * SALU instructions per macroblock).  So per-lane predication is done with
 * ADDRESS selection instead: lanes that must not store write to a per-lane
 * dump slot in LDS (or repeat a neighbour's identical store). */
#define SD_DUMP_BYTES 1024u /* 64 lanes x 16 B */

/* a condition that is rarely true: its code goes out of line and the usual way
 * is the branch not taken -- 19 cycles of a wave alone on its SIMD against
 * 31-34 for one taken (profiles/r02_decode_cost_model.md) */
#define SD_RARE(c) __builtin_expect(!!(c), 0)
#define SD_USUAL(c) __builtin_expect(!!(c), 1)

__device__ __forceinline__ uint8_t *lds_sel(uint8_t *real, uint8_t *dump,
					    bool take, uint32_t lane)
{
	return take ? real : dump + 16u * lane;
}

__device__ __forceinline__ void lds_w32(uint8_t *p, uint32_t off, uint32_t v,
					uint32_t lane)
{
	uint8_t *dump = p - SD_DUMP_BYTES; /* the dump area precedes `stage` */
	*reinterpret_cast<uint32_t *>(lds_sel(p + off, dump, lane == 0, lane)) = v;
}

__device__ __forceinline__ void lds_w16(uint8_t *p, uint32_t off, uint32_t v,
					uint32_t lane)
{
	uint8_t *dump = p - SD_DUMP_BYTES;
	*reinterpret_cast<uint16_t *>(lds_sel(p + off, dump, lane == 0, lane)) =
		(uint16_t)v;
}

__device__ __forceinline__ void lds_w8(uint8_t *p, uint32_t off, uint32_t v,
				       uint32_t lane)
{
	uint8_t *dump = p - SD_DUMP_BYTES;
	*lds_sel(p + off, dump, lane == 0, lane) = (uint8_t)v;
}

#define HOFF(f) ((uint32_t)offsetof(struct h264_slice_header, f))
#define HP(f, v) lds_w32(stage, HOFF(f), (uint32_t)(v), lane)

#define RD_UE(var)                                                             \
	do {                                                                   \
		var = wb_ue(wb, lane, ok);                                     \
		if (!ok)                                                       \
			return -EIO;                                           \
	} while (0)
#define RD_SE(var)                                                             \
	do {                                                                   \
		var = wb_se(wb, lane, ok);                                     \
		if (!ok)                                                       \
			return -EIO;                                           \
	} while (0)

/* Load a small POD from global memory so that every field is wave-uniform
 * (SGPR): the compiler cannot prove a vector load from a uniform address
 * uniform by itself, and everything derived from the parameter sets steers
 * scalar control flow. */
template <typename T>
__device__ __forceinline__ void sd_load_uniform(T *dst, const T *src)
{
	static_assert(sizeof(T) % 4 == 0, "dword multiple");
	const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
	uint32_t *d = reinterpret_cast<uint32_t *>(dst);
#pragma unroll
	for (uint32_t i = 0; i < sizeof(T) / 4; i++)
		d[i] = wb_uniform(s[i]);
}

__device__ __forceinline__ uint32_t sd_t2_stride(uint32_t cat)
{
	return 256u + 2u * (cat == 0 ? 256u : cat == 1 ? 384u : cat == 2 ? 512u : 768u);
}

/* Calculate log2 rounded up, 0 -> 0 (ref src/h264_priv.h:185-193) */
__device__ __forceinline__ uint32_t sd_intlog2(uint32_t x)
{
	uint32_t r = x ? 32u - (uint32_t)__builtin_clz(x) : 0u;
	if (r > 0 && x == (1u << (r - 1u)))
		r--;
	return r;
}

/* ---- slice header (ref src/h264_syntax.h:1014-1387) --------------------- */

__device__ __forceinline__ int sd_rplm_items(WaveBits &wb, uint32_t lane,
					     uint8_t *stage, uint32_t off)
{
	bool ok = true;
	uint32_t i = 0, val;
	do {
		if (SD_RARE(i >= 32))
			return -EIO;
		wb_sync(wb, lane);
		RD_UE(val);
		lds_w32(stage, off + 8 * i, val, lane);
		if (val == 0 || val == 1 || val == 2 || val == 4 || val == 5) {
			uint32_t v;
			RD_UE(v);
			lds_w32(stage, off + 8 * i + 4, v, lane);
		}
		i++;
	} while (val != 3);
	return 0;
}

__device__ __forceinline__ int sd_pwt_list(WaveBits &wb, uint32_t lane,
					   uint8_t *stage, uint32_t off,
					   uint32_t count, uint32_t cat)
{
	bool ok = true;
	for (uint32_t i = 0; i < count; i++) {
		uint32_t o = off + 32 * i; /* sizeof(struct h264_pwt_item) */
		int32_t v;
		wb_sync(wb, lane);
		uint32_t f = wb_u1(wb, lane);
		lds_w32(stage, o, f, lane);
		if (f) {
			RD_SE(v);
			lds_w32(stage, o + 4, (uint32_t)v, lane);
			RD_SE(v);
			lds_w32(stage, o + 8, (uint32_t)v, lane);
		}
		if (cat != 0) {
			f = wb_u1(wb, lane);
			lds_w32(stage, o + 12, f, lane);
			if (f) {
				/* weight[0], offset[0], weight[1], offset[1] */
				RD_SE(v);
				lds_w32(stage, o + 16, (uint32_t)v, lane);
				RD_SE(v);
				lds_w32(stage, o + 24, (uint32_t)v, lane);
				RD_SE(v);
				lds_w32(stage, o + 20, (uint32_t)v, lane);
				RD_SE(v);
				lds_w32(stage, o + 28, (uint32_t)v, lane);
			}
		}
	}
	return 0;
}

__device__ __forceinline__ int
sd_slice_header(WaveBits &wb, uint32_t lane, uint8_t *stage, SliceState &S,
		const dev_sps *spsv, const dev_pps *ppsv, const dev_epoch *ep,
		uint32_t n_sps, uint32_t n_pps)
{
	bool ok = true;
	uint32_t v, type, raw_type, pps_id;
	int32_t sv;

	RD_UE(v);
	HP(first_mb_in_slice, v);
	S.first_mb = v;
	RD_UE(raw_type);
	HP(slice_type, raw_type);
	type = raw_type % 5u;
	RD_UE(pps_id);
	HP(pic_parameter_set_id, pps_id);

	/* a ue that ran past the end of the NAL is the reference's -EIO on the
	 * read itself, before it looks at the value (found by the drop-in fuzz:
	 * the zero padding made pps_id large and -EINVAL won) */
	if (SD_RARE(wb_overrun(wb)))
		return -EIO;
	/* h264_ctx_set_active_pps -> set_active_sps (src/h264_ctx.c:366-386) */
	if (SD_RARE(pps_id >= 256))
		return -EINVAL;
	uint32_t pv = wb_uniform(ep->pps_ver[pps_id]);
	if (SD_RARE(pv == DEV_NO_VERSION || pv >= n_pps))
		return -EINVAL;
	dev_pps pps;
	sd_load_uniform(&pps, ppsv + pv);
	if (SD_RARE(pps.seq_parameter_set_id >= 32))
		return -EINVAL;
	uint32_t sver = wb_uniform(ep->sps_ver[pps.seq_parameter_set_id]);
	if (SD_RARE(sver == DEV_NO_VERSION || sver >= n_sps))
		return -EINVAL;
	dev_sps sps;
	sd_load_uniform(&sps, spsv + sver);

	S.cat = sps.separate_colour_plane_flag ? 0u : sps.chroma_format_idc;
	S.W = sps.pic_width_in_mbs;
	S.bdl = sps.bit_depth_luma;
	S.bdc = sps.bit_depth_chroma;
	/* MbWidthC * MbHeightC; 0 for a chroma_format_idc above 3, which the
	 * reference does not refuse (src/h264.c:46-71) */
	S.pcm_chroma = S.cat == 1 ? 64u : S.cat == 2 ? 128u : S.cat == 3 ? 256u : 0u;
	S.t8x8_mode = pps.transform_8x8_mode_flag;
	S.direct8x8 = sps.direct_8x8_inference_flag;
	S.entropy = pps.entropy_coding_mode_flag;
	S.slice_type = type;

	uint32_t nref0 = pps.num_ref_idx_l0_default_active_minus1;
	uint32_t nref1 = pps.num_ref_idx_l1_default_active_minus1;
	uint32_t field_pic = 0;

	if (sps.separate_colour_plane_flag) {
		v = wb_u(wb, lane, 2);
		HP(colour_plane_id, v);
	}
	v = wb_u_any(wb, lane, sps.log2_max_frame_num, ok);
	if (SD_RARE(!ok))
		return -EIO;
	HP(frame_num, v);
	if (!sps.frame_mbs_only_flag) {
		field_pic = wb_u1(wb, lane);
		HP(field_pic_flag, field_pic);
		if (field_pic) {
			v = wb_u1(wb, lane);
			HP(bottom_field_flag, v);
		}
	}
	if (S.nal_unit_type == 5) {
		RD_UE(v);
		HP(idr_pic_id, v);
	}
	wb_sync(wb, lane);
	if (sps.pic_order_cnt_type == 0) {
		v = wb_u_any(wb, lane, sps.log2_max_pic_order_cnt_lsb, ok);
		if (SD_RARE(!ok))
			return -EIO;
		HP(pic_order_cnt_lsb, v);
		if (pps.bottom_field_pic_order_in_frame_present_flag && !field_pic) {
			RD_SE(sv);
			HP(delta_pic_order_cnt_bottom, sv);
		}
	}
	if (sps.pic_order_cnt_type == 1 && !sps.delta_pic_order_always_zero_flag) {
		RD_SE(sv);
		HP(delta_pic_order_cnt[0], sv);
		if (pps.bottom_field_pic_order_in_frame_present_flag && !field_pic) {
			RD_SE(sv);
			HP(delta_pic_order_cnt[1], sv);
		}
	}
	if (pps.redundant_pic_cnt_present_flag) {
		RD_UE(v);
		HP(redundant_pic_cnt, v);
	}
	if (type == H264_SLICE_TYPE_B) {
		v = wb_u1(wb, lane);
		HP(direct_spatial_mv_pred_flag, v);
	}
	wb_sync(wb, lane);
	if (type == H264_SLICE_TYPE_P || type == H264_SLICE_TYPE_SP ||
	    type == H264_SLICE_TYPE_B) {
		v = wb_u1(wb, lane);
		HP(num_ref_idx_active_override_flag, v);
		if (v) {
			RD_UE(nref0);
			if (type == H264_SLICE_TYPE_B)
				RD_UE(nref1);
		}
	}
	HP(num_ref_idx_l0_active_minus1, nref0);
	HP(num_ref_idx_l1_active_minus1, nref1);

	/* ref_pic_list_modification */
	if (type != H264_SLICE_TYPE_I && type != H264_SLICE_TYPE_SI) {
		v = wb_u1(wb, lane);
		HP(rplm.ref_pic_list_modification_flag_l0, v);
		if (v) {
			int r = sd_rplm_items(wb, lane, stage, HOFF(rplm.pic_num_l0));
			if (r)
				return r;
		}
	}
	if (type == H264_SLICE_TYPE_B) {
		v = wb_u1(wb, lane);
		HP(rplm.ref_pic_list_modification_flag_l1, v);
		if (v) {
			int r = sd_rplm_items(wb, lane, stage, HOFF(rplm.pic_num_l1));
			if (r)
				return r;
		}
	}
	wb_sync(wb, lane);
	/* pred_weight_table */
	if ((pps.weighted_pred_flag &&
	     (type == H264_SLICE_TYPE_P || type == H264_SLICE_TYPE_SP)) ||
	    (pps.weighted_bipred_idc == 1 && type == H264_SLICE_TYPE_B)) {
		uint32_t pcat = sps.separate_colour_plane_flag
					? 0u
					: sps.chroma_format_idc;
		RD_UE(v);
		HP(pwt.luma_log2_weight_denom, v);
		if (pcat != 0) {
			RD_UE(v);
			HP(pwt.chroma_log2_weight_denom, v);
		}
		if (SD_RARE(nref0 > 32))
			return -EIO;
		int r = sd_pwt_list(wb, lane, stage, HOFF(pwt.l0), nref0 + 1, pcat);
		if (r)
			return r;
		if (type == H264_SLICE_TYPE_B) {
			if (SD_RARE(nref1 > 32))
				return -EIO;
			r = sd_pwt_list(wb, lane, stage, HOFF(pwt.l1), nref1 + 1,
					pcat);
			if (r)
				return r;
		}
	}
	wb_sync(wb, lane);
	/* dec_ref_pic_marking */
	if (S.nal_ref_idc != 0) {
		if (S.nal_unit_type == 5) {
			v = wb_u1(wb, lane);
			HP(drpm.no_output_of_prior_pics_flag, v);
			v = wb_u1(wb, lane);
			HP(drpm.long_term_reference_flag, v);
		} else {
			v = wb_u1(wb, lane);
			HP(drpm.adaptive_ref_pic_marking_mode_flag, v);
			if (v) {
				uint32_t i = 0, op;
				do {
					uint32_t o = HOFF(drpm.mm) + 20 * i;
					if (SD_RARE(i >= 64))
						return -EIO;
					wb_sync(wb, lane);
					RD_UE(op);
					lds_w32(stage, o, op, lane);
					if (op == 1 || op == 3) {
						RD_UE(v);
						lds_w32(stage, o + 4, v, lane);
					}
					if (op == 2) {
						RD_UE(v);
						lds_w32(stage, o + 8, v, lane);
					}
					if (op == 3 || op == 6) {
						RD_UE(v);
						lds_w32(stage, o + 12, v, lane);
					}
					if (op == 4) {
						RD_UE(v);
						lds_w32(stage, o + 16, v, lane);
					}
					i++;
				} while (op != 0);
			}
		}
	}
	wb_sync(wb, lane);
	if (pps.entropy_coding_mode_flag && type != H264_SLICE_TYPE_I &&
	    type != H264_SLICE_TYPE_SI) {
		RD_UE(v);
		HP(cabac_init_idc, v);
	}
	RD_SE(sv);
	HP(slice_qp_delta, sv);
	S.slice_qp = pps.pic_init_qp_minus26 + 26 + sv;
	if (type == H264_SLICE_TYPE_SP || type == H264_SLICE_TYPE_SI) {
		if (type == H264_SLICE_TYPE_SP) {
			v = wb_u1(wb, lane);
			HP(sp_for_switch_flag, v);
		}
		RD_SE(sv);
		HP(slice_qs_delta, sv);
	}
	if (pps.deblocking_filter_control_present_flag) {
		RD_UE(v);
		HP(disable_deblocking_filter_idc, v);
		if (v != 1) {
			RD_SE(sv);
			HP(slice_alpha_c0_offset_div2, sv);
			RD_SE(sv);
			HP(slice_beta_offset_div2, sv);
		}
	}
	if (pps.num_slice_groups_minus1 > 0 && pps.slice_group_map_type >= 3 &&
	    pps.slice_group_map_type <= 5) {
		uint32_t units = sps.pic_width_in_mbs * sps.pic_height_in_map_units;
		uint32_t n = sd_intlog2(
			units / (pps.slice_group_change_rate_minus1 + 1u) + 1u);
		v = n ? wb_u(wb, lane, n) : 0u;
		HP(slice_group_change_cycle, v);
		S.change_cycle = v;
	}
	S.nref0 = nref0;
	S.nref1 = nref1;
	S.field_pic = field_pic;
	S.fmo = pps.num_slice_groups_minus1;
	S.fmo_type = pps.slice_group_map_type;
	S.pps_version = pv;
	S.frame_mbs_only = sps.frame_mbs_only_flag;
	S.map_units_h = sps.pic_height_in_map_units;
	S.change_rate = pps.slice_group_change_rate_minus1 + 1u;
	S.mbaff = (sps.mb_adaptive_frame_field_flag && !field_pic) ? 1u : 0u;
	S.unsupported = 0;
	return 0;
}

/* ---- slice groups (ref src/h264_fmo.c:33-319) ------------------------------
 * The macroblock-to-slice-group map of the picture, one byte per macroblock
 * ADDRESS, written to this slice's piece of the map scratch in HBM behind the
 * slice header (types 3..5 depend on slice_group_change_cycle).  The map units
 * of 8.2.2.1 - 8.2.2.7 are laid out over the lanes; 8.2.2.8 (map unit ->
 * macroblock) is folded in.  Box-out (type 3) is a walk with memory: it runs
 * on a one-bit-per-unit image in LDS (the header staging area, free by now),
 * every lane following the same steps.  Where the reference indexes without
 * bounds the choices are the oracle's (oracle/orc_fmo.c): cells outside the
 * map are skipped, ids past the PPS's 256 are group 0.
 * Returns 0 or H264_AMD_E_UNSUPPORTED (picture too large for the scratch /
 * the LDS image). */
#define SD_FMO_LDS_UNITS (SD_STAGE_BYTES * 8u)
static_assert(SD_FMO_LDS_UNITS == H264_AMD_FMO_MAX_BOX_OUT_UNITS,
	      "box-out bitmap and the documented limit differ");

/* slice group of one map unit, per lane, without a divergent branch (see the
 * note on divergent branches above: loops and switches here are uniform, what
 * differs between lanes is selected) */
__device__ __forceinline__ uint32_t sd_fmo_unit_group(const dev_fmo *f,
						      const SliceState &S,
						      uint32_t unit, uint32_t n,
						      uint32_t W, uint32_t units0)
{
	const uint32_t G = S.fmo + 1u;
	const uint32_t dir = f->slice_group_change_direction_flag;
	const uint32_t upper = dir ? n - units0 : units0;
	uint32_t res = 0;
	if (S.fmo_type == 0) {
		/* interleaved: runs repeat with period sum(run_length + 1) */
		uint64_t period = 0, acc = 0, pos;
		for (uint32_t g = 0; g < 8u; g++)
			period += g < G ? (uint64_t)f->run_length_minus1[g] + 1u : 0u;
		pos = unit % period;
		for (uint32_t g = 0; g < 8u; g++) {
			uint64_t run = g < G ? (uint64_t)f->run_length_minus1[g] + 1u
					     : 0u;
			res = (pos >= acc && pos < acc + run) ? g : res;
			acc += run;
		}
	} else if (S.fmo_type == 1) {
		res = ((unit % W) + (((unit / W) * G) / 2u)) % G;
	} else if (S.fmo_type == 2) {
		uint32_t x = unit % W, y = unit / W;
		res = G - 1u;
		for (uint32_t k = 0; k < 7u; k++) {
			/* groups G - 2 down to 0: a lower one overrides */
			uint32_t g = 6u - k;
			uint32_t yt = f->top_left[g] / W, xl = f->top_left[g] % W;
			uint32_t yb = f->bottom_right[g] / W;
			uint32_t xr = f->bottom_right[g] % W;
			bool in = g + 1u < G && y >= yt && y <= yb && x >= xl &&
				  x <= xr;
			res = in ? g : res;
		}
	} else if (S.fmo_type == 4) {
		res = unit < upper ? dir : 1u - dir;
	} else if (S.fmo_type == 5) {
		uint32_t H = n / W;
		uint32_t k = (unit % W) * H + unit / W;
		res = k < upper ? dir : 1u - dir;
	} else if (S.fmo_type == 6) {
		uint32_t have = f->pic_size_in_map_units_minus1 < 255u
					? f->pic_size_in_map_units_minus1 + 1u
					: 256u;
		uint32_t id = f->slice_group_id[unit & 255u];
		res = unit < have ? id : 0u;
	}
	return res;
}

__device__ __forceinline__ int sd_fmo_build(const dev_fmo *f, const SliceState &S,
					    uint8_t *stage, uint8_t *map,
					    uint32_t map_cap, uint32_t lane,
					    uint32_t &pic_mbs)
{
	const uint32_t W = S.W;
	const uint64_t n64 = (uint64_t)W * S.map_units_h;
	const uint64_t mbs64 = (S.frame_mbs_only || S.field_pic) ? n64 : 2u * n64;
	if (mbs64 > map_cap || n64 == 0)
		return H264_AMD_E_UNSUPPORTED;
	const uint32_t n = (uint32_t)n64, H = S.map_units_h;
	pic_mbs = (uint32_t)mbs64;
	/* MapUnitsInSliceGroup0 (ref src/h264_ctx.c:119-122) */
	uint64_t u0 = (uint64_t)S.change_cycle * S.change_rate;
	const uint32_t units0 = u0 < n ? (uint32_t)u0 : n;
	uint32_t *bits = reinterpret_cast<uint32_t *>(stage);

	if (S.fmo_type == 3) {
		if (n > SD_FMO_LDS_UNITS)
			return H264_AMD_E_UNSUPPORTED;
		/* bit set = unit in group 0 */
		for (uint32_t i = lane; i < SD_STAGE_BYTES / 4u; i += 64u)
			bits[i] = 0;
		const int32_t dir = (int32_t)f->slice_group_change_direction_flag;
		int32_t x = (int32_t)((W - (uint32_t)dir) / 2u);
		int32_t y = (int32_t)((H - (uint32_t)dir) / 2u);
		int32_t left = x, top = y, right = x, bottom = y;
		int32_t xd = dir - 1, yd = dir;
		/* every vacant step fills a unit; steps over filled units come
		 * with a box that the picture's edge clips, and are bounded
		 * (H264_AMD_FMO_BOX_OUT_STEPS, h264_amd.h: the oracle stops at the
		 * same step) */
		uint32_t k = 0, guard = (uint32_t)H264_AMD_FMO_BOX_OUT_STEPS(n);
		while (k < units0 && guard-- > 0) {
			uint32_t cell = (uint32_t)y * W + (uint32_t)x;
			uint32_t word = wb_uniform(bits[cell >> 5]);
			uint32_t bit = 1u << (cell & 31u);
			if (!(word & bit)) {
				bits[cell >> 5] = word | bit; /* all lanes, same */
				k++;
			}
			if (xd == -1 && x == left) {
				left = left > 0 ? left - 1 : 0;
				x = left;
				xd = 0;
				yd = 2 * dir - 1;
			} else if (xd == 1 && x == right) {
				right = right + 1 < (int32_t)W - 1 ? right + 1
								   : (int32_t)W - 1;
				x = right;
				xd = 0;
				yd = 1 - 2 * dir;
			} else if (yd == -1 && y == top) {
				top = top > 0 ? top - 1 : 0;
				y = top;
				xd = 1 - 2 * dir;
				yd = 0;
			} else if (yd == 1 && y == bottom) {
				bottom = bottom + 1 < (int32_t)H - 1 ? bottom + 1
								     : (int32_t)H - 1;
				y = bottom;
				xd = 2 * dir - 1;
				yd = 0;
			} else {
				x += xd;
				y += yd;
			}
		}
	}
	/* one byte per macroblock address (8.2.2.8: map units are macroblocks
	 * for frame_mbs_only and for field pictures; a frame picture of
	 * interlaced material has two macroblock rows per unit row) */
	for (uint32_t a0 = 0; a0 < pic_mbs; a0 += 64u) {
		/* lanes past the end repeat the last macroblock's store */
		uint32_t a = a0 + lane < pic_mbs ? a0 + lane : pic_mbs - 1u;
		uint32_t unit = (S.frame_mbs_only || S.field_pic) ? a
				: S.mbaff		      ? a >> 1
					: (a / (2u * W)) * W + (a % W);
		uint32_t g3 = (bits[(unit >> 5) & (SD_STAGE_BYTES / 4u - 1u)] >>
			       (unit & 31u)) & 1u ? 0u : 1u;
		uint32_t g = S.fmo_type == 3
				     ? g3
				     : sd_fmo_unit_group(f, S, unit, n, W, units0);
		map[a] = (uint8_t)g;
	}
	return 0;
}

/* h264_next_mb_addr (ref src/h264_fmo.c:307-319): the next address above `cur`
 * of the same slice group, PicSizeInMbs if there is none, cur + 1 beyond the
 * picture.  64 addresses per look. */
__device__ __forceinline__ uint32_t sd_fmo_next(const uint8_t *map,
						uint32_t pic_mbs, uint32_t cur,
						uint32_t lane)
{
	uint32_t i = cur + 1u;
	if (i >= pic_mbs)
		return i;
	const uint32_t g = wb_uniform(map[cur]);
	for (;;) {
		uint32_t a = i + lane;
		uint32_t v = map[a < pic_mbs ? a : pic_mbs - 1u];
		wb_mask_t hit = __ballot(a < pic_mbs && v == g);
		if (hit)
			return i + (uint32_t)__builtin_ctzll(hit);
		i += 64u;
		if (i >= pic_mbs)
			return pic_mbs;
	}
}

/* ---- phase timers (diagnostic build only: -DSD_PHASE_TIMERS) -----------------
 * tools/phase_timers.py: cycles (s_memtime) between consecutive marks, summed
 * per phase in lane `phase` of one vector register, calls counted next to them;
 * the sums replace the first 32 T0 words of the slice.  Phase 15 follows phase
 * 14 directly: the cost of a mark itself. */
#ifdef SD_PHASE_TIMERS
struct SdTimers {
	uint32_t acc, cnt; /* per lane */
	uint32_t last; /* scalar */
	uint64_t t0, r0; /* scalar: shader clock and 100 MHz clock at the start */
};
#define SD_TARG , SdTimers &T
#define SD_TPASS , T
#define SD_T(k)                                                                \
	do {                                                                   \
		uint32_t t_ = (uint32_t)__builtin_readcyclecounter();          \
		uint32_t d_ = t_ - T.last;                                     \
		T.last = t_;                                                   \
		T.acc += lane == (uint32_t)(k) ? d_ : 0u;                      \
		T.cnt += lane == (uint32_t)(k) ? 1u : 0u;                      \
	} while (0)
#else
#define SD_TARG
#define SD_TPASS
#define SD_T(k)                                                                \
	do {                                                                   \
	} while (0)
#endif

/* RBSP bytes a wave spends at one issue priority before it steps to the next
 * (see the slice-data loop): 1 << shift, a multiple of the 128-byte window
 * slide */
#ifndef SD_PRIO_SHIFT
#define SD_PRIO_SHIFT 7
#endif

/* ---- macroblock bookkeeping ---------------------------------------------- */

/* Neighbour context of 9.2.1 (nC) kept incrementally.  Lane l = comp * 16 +
 * blkIdx stands for one 4x4 block.  `sum` holds nA + nB of that block as far as
 * they are known: the external (left / above macroblock) part is loaded when
 * the macroblock starts, the internal part is added when the neighbouring
 * block of the same macroblock has been decoded.  Reading nC is then one
 * v_readlane instead of the index arithmetic of
 * src/h264_slice_data.c:1319-1334. */
struct MbCtx {
	/* per lane, changing */
	uint32_t cur; /* nz_coeff of the current macroblock */
	uint32_t left; /* nz_coeff of the previous macroblock */
	uint32_t sum; /* nA + nB of this lane's block */
	/* per lane, constant for the slice */
	uint64_t dep; /* bits: the lanes of the A / B neighbours INSIDE the MB */
	uint32_t hasint; /* bit 0: has an inner A neighbour, bit 1: an inner B */
	uint32_t a_ext; /* lane (previous MB) of an external A neighbour or 0xff */
	int top_slot; /* row-context byte holding an external B neighbour or -1 */
	int row_slot; /* row-context byte this lane saves (bottom row) or -1 */
	uint32_t info_n, info_i16; /* lane = residual step: block parameters */
	uint32_t bothv; /* per lane: 1 when both neighbours are available */
	/* uniform */
	uint32_t availA, availB;
};

/* z-order index of the 4x4 luma block at (x, y), 6.4.13.1 */
__device__ __forceinline__ uint32_t blk_at(uint32_t x, uint32_t y)
{
	return (x & 1u) | ((y & 1u) << 1) | ((x >> 1) << 2) | ((y >> 1) << 3);
}

/* ctx: the lane whose table class applies; the block's own lane, except for
 * chroma DC blocks, which use the constant classes parked in lanes 48 / 49 */
#define SD_INFO(nzlane, maxnum, cdc, coef, ctx)                                \
	((nzlane) | ((maxnum) << 6) | ((cdc) << 11) | ((coef) << 13) |         \
	 ((ctx) << 24))

/* Per-lane constants: neighbour geometry (6.4.11.4, as restated by the
 * reference in src/h264_slice_data.c:1239-1334) and, with lane = step, the
 * parameters of every residual block in bitstream order
 * (src/h264_syntax_slice_data.h:334-419): for each luma-like component the
 * Intra16x16 DC block then its 16 4x4 blocks, then chroma DC Cb, Cr, chroma AC
 * Cb, Cr. */
__device__ __forceinline__ void sd_mb_consts(MbCtx &m, uint32_t lane, uint32_t cat)
{
	const uint32_t comp = lane >> 4, b = lane & 15u;
	const bool lumalike = lane < 48 && (comp == 0 || cat == 3);
	const bool chroma = lane < 48 && comp > 0 && (cat == 1 || cat == 2);
	const uint32_t nchroma = cat == 1 ? 4u : 8u, h4 = cat == 1 ? 2u : 4u;
	/* luma geometry */
	const uint32_t lx = (b & 1u) | ((b >> 1) & 2u);
	const uint32_t ly = ((b >> 1) & 1u) | ((b >> 2) & 2u);
	/* chroma geometry: 2 blocks wide */
	const uint32_t cx = b & 1u, cy = b >> 1;
	const bool cvalid = chroma && b < nchroma;

	uint32_t a_int = 0xff, b_int = 0xff, a_ext = 0xff;
	int top_slot = -1, row_slot = -1;
	if (lumalike) {
		a_int = lx ? 16u * comp + blk_at(lx - 1u, ly) : 0xffu;
		a_ext = lx ? 0xffu : 16u * comp + blk_at(3, ly);
		b_int = ly ? 16u * comp + blk_at(lx, ly - 1u) : 0xffu;
		top_slot = ly ? -1 : (int)(4u * comp + lx);
		row_slot = ly == 3 ? (int)(4u * comp + lx) : -1;
	}
	if (cvalid) {
		a_int = cx ? 16u * comp + 2u * cy : 0xffu;
		a_ext = cx ? 0xffu : 16u * comp + 2u * cy + 1u;
		b_int = cy ? 16u * comp + 2u * (cy - 1u) + cx : 0xffu;
		top_slot = cy ? -1 : (int)(4u * comp + cx);
		row_slot = cy == h4 - 1u ? (int)(4u * comp + cx) : -1;
	}
	m.dep = (a_int != 0xffu ? 1ull << a_int : 0ull) |
		(b_int != 0xffu ? 1ull << b_int : 0ull);
	m.hasint = (a_int != 0xffu ? 1u : 0u) | (b_int != 0xffu ? 2u : 0u);
	m.a_ext = a_ext;
	m.top_slot = top_slot;
	m.row_slot = row_slot;

	/* residual steps */
	const uint32_t luma_steps = 17u * (cat == 3 ? 3u : 1u);
	const uint32_t s = lane;
	uint32_t in = 0, ii = 0;
	if (s < luma_steps) {
		uint32_t c = s / 17u, sub = s % 17u;
		if (sub == 0) {
			in = ii = SD_INFO(16u * c, 16u, 0u, 256u * c, 16u * c);
		} else {
			uint32_t blk = sub - 1u;
			in = SD_INFO(16u * c + blk, 16u, 0u, 256u * c + 16u * blk,
				     16u * c + blk);
			ii = SD_INFO(16u * c + blk, 15u, 0u,
				     256u * c + 16u + 15u * blk, 16u * c + blk);
		}
	} else if (cat == 1 || cat == 2) {
		uint32_t c = s - luma_steps;
		if (c < 2) {
			in = SD_INFO(16u * (1u + c), nchroma, cat, 256u + nchroma * c,
				     48u + c);
		} else if (c < 2u + 2u * nchroma) {
			c -= 2u;
			uint32_t cc = c >= nchroma ? 1u : 0u;
			uint32_t blk = c - cc * nchroma;
			in = SD_INFO(16u * (1u + cc) + blk, 15u, 0u,
				     256u + 2u * nchroma + 15u * (nchroma * cc + blk),
				     16u * (1u + cc) + blk);
		}
		ii = in;
	}
	m.info_n = in;
	m.info_i16 = ii;
}

/* A macroblock starts: external neighbour sums and the both-available mask */
__device__ __forceinline__ void sd_mb_begin(MbCtx &m, const uint8_t *row,
					    uint32_t mbx, uint32_t lane)
{
	uint32_t lv = (uint32_t)__shfl((int)m.left, (int)(m.a_ext & 63u));
	uint32_t tv = row[16u * mbx + (uint32_t)(m.top_slot & 15)];
	bool extA = m.a_ext != 0xffu && m.availA != 0;
	bool extB = m.top_slot >= 0 && m.availB != 0;
	m.sum = (extA ? lv : 0u) + (extB ? tv : 0u);
	bool hasA = (m.hasint & 1u) != 0 || extA;
	bool hasB = (m.hasint & 2u) != 0 || extB;
	m.bothv = (hasA && hasB) ? 1u : 0u;
}

/* nC of every block (9.2.1), per lane: (nA + nB + 1) >> 1 with both
 * neighbours, the available one alone otherwise, 0 with none */
__device__ __forceinline__ uint32_t sd_nc_all(const MbCtx &m)
{
	return (m.sum + m.bothv) >> m.bothv;
}

__device__ __forceinline__ uint32_t sd_nc(const MbCtx &m, uint32_t l)
{
	return (uint32_t)__builtin_amdgcn_readlane((int)sd_nc_all(m), (int)l);
}

/* block at lane l now has TotalCoeff tc (it may have had another value: the
 * DC blocks park theirs at index 0, Appendix C.4) */
__device__ __forceinline__ void sd_nz_set(MbCtx &m, uint32_t l, uint32_t tc,
					  uint32_t lane)
{
	/* unconditional: a handful of vector instructions instead of a scalar
	 * compare and branch (the scalar port is the scarce one) */
	uint32_t old = (uint32_t)__builtin_amdgcn_readlane((int)m.cur, (int)l);
	uint32_t delta = tc - old;
	/* m.cur = lane == l ? tc : m.cur, the lane mask from the scalar side (no
	 * compare into VCC and wait before the select) */
	asm("v_cndmask_b32 %0, %0, %1, %2"
	    : "+v"(m.cur)
	    : "v"(tc), "s"(1ull << (l & 63u)));
	/* the blocks that have this one as their inner A or B neighbour: one
	 * shift of the lane's neighbour mask instead of two compare-select pairs */
	uint32_t bit = (uint32_t)(m.dep >> l);
	/* bit 0 spread over the word (v_bfe_i32), kept opaque: the optimiser
	 * would turn the and back into a compare and a select */
	int32_t msk = (int32_t)(bit << 31) >> 31;
	asm volatile("" : "+v"(msk));
	m.sum += (uint32_t)msk & delta;
}

/* The n (16 or 4) prev_intra_pred_mode_flag / rem_intra_pred_mode pairs of an
 * I_NxN macroblock (ref src/h264_syntax_slice_data.h:520-553), all at once.
 * Every symbol is '1' or '0xxx', so lane k speculates on a symbol starting at
 * bit k of the next 64 bits: next[k] = k + (1 or 4).  Pointer doubling turns
 * that into next^2, next^4 ... and lane j composes them along the bits of j to
 * land on the start of symbol j.  Seven cross-lane gathers instead of up to 32
 * dependent scalar reads.  Stores the packed nibbles (mode + 1, 0 = predicted)
 * into bytes 8..15 of the staged record and returns the bits consumed. */
__device__ __forceinline__ uint32_t sd_gather(uint32_t table, uint32_t idx)
{
	/* table[i] > i and table[i] <= i + 64 (a jump forward by 1 .. 64 bits):
	 * the maximum is the entry for idx < 64 and idx itself beyond, where
	 * the lane wraps -- one instruction where compare and select are two
	 * and a wait on VCC */
	uint32_t v = (uint32_t)__shfl((int)table, (int)(idx & 63u));
	return v > idx ? v : idx;
}

__device__ __forceinline__ uint32_t sd_intra_pred_modes(const VBits &vb,
							const WaveBits &wb,
							uint32_t lane, uint32_t n,
							uint8_t *stage)
{
	/* the 64 bits at the cursor (vo < 32: the caller has normalised): what W
	 * holds, then the top of the window dword behind it */
	uint32_t d2 = (uint32_t)__builtin_amdgcn_readlane((int)wb.wnext,
							   (int)(vb.sd + 1u));
	VB_VREG(d2);
	const uint64_t win = (vb.W << vb.vo) | ((d2 >> 1) >> (31u - vb.vo));
	uint32_t flag = (uint32_t)(win >> (63u - lane)) & 1u;
	uint32_t j1 = lane + (flag ? 1u : 4u);
	uint32_t j2 = sd_gather(j1, j1);
	uint32_t j4 = sd_gather(j2, j2);
	uint32_t j8 = sd_gather(j4, j4);
	uint32_t j16 = sd_gather(j8, j8);
	/* every gather runs in all lanes (a cross-lane read must not sit in a
	 * divergent branch); the lane then keeps or drops the result */
	uint32_t p = 0, g;
	g = sd_gather(j1, p);
	p = (lane & 1u) ? g : p;
	g = sd_gather(j2, p);
	p = (lane & 2u) ? g : p;
	g = sd_gather(j4, p);
	p = (lane & 4u) ? g : p;
	g = sd_gather(j8, p);
	p = (lane & 8u) ? g : p;
	g = sd_gather(j16, p);
	p = (lane & 16u) ? g : p;
	/* lane j < n: symbol j starts at bit p (< 64) */
	uint32_t sym = (uint32_t)((win << (p & 63u)) >> 60);
	uint32_t nib = (sym & 8u) ? 0u : (sym & 7u) + 1u;
	uint32_t nxt = (uint32_t)__shfl_down((int)nib, 1);
	bool mine = lane < n && (lane & 1u) == 0;
	*lds_sel(stage + 8u + (lane >> 1), stage - SD_DUMP_BYTES, mine, lane) =
		(uint8_t)(nib | (nxt << 4));
	return (uint32_t)__builtin_amdgcn_readlane((int)p, (int)n);
}

/* inclusive prefix sum over the 64 lanes with DPP row shifts and broadcasts
 * (register-to-register, no LDS round trip) */
__device__ __forceinline__ uint32_t sd_scan_add(uint32_t x)
{
	uint32_t v = x;
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
	/* lane 15 of each row into the next row (rows 1, 3), then lane 31 into
	 * rows 2 and 3 */
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
	return v;
}

/* One level the long way (9.2.2.1; ref src/h264_syntax_slice_data.h:156-202):
 * level_prefix of 14 and more with suffixLength 0, 15 and more otherwise (the
 * escape codes).  Rare, so it runs on the scalar cursor: returns levelCode
 * (first_adj included) and moves suffix_len on.  bad: level_prefix > 25. */
__device__ __forceinline__ uint32_t sd_level_escape(WaveBits &wb, uint32_t lane,
						    uint32_t &suffix_len,
						    uint32_t first_adj, uint32_t &bad)
{
	uint32_t bits = wb_peek(wb, lane);
	uint32_t prefix = (uint32_t)__builtin_clz(bits | 0x3fu);
	uint32_t code;
	bad |= bits < (1u << 6) ? 1u : 0u; /* level_prefix > 25 */
	if (prefix < 14u + (suffix_len ? 1u : 0u)) {
		/* not an escape after all (callers only come here for escapes, but
		 * the general form costs nothing to keep) */
		uint32_t suffix = ((bits << (prefix + 1u)) >> 1) >> (31u - suffix_len);
		code = (prefix << suffix_len) + suffix;
		wb_skip(wb, prefix + 1u + suffix_len);
	} else {
		wb_skip(wb, prefix + 1u);
		code = (prefix < 15u ? prefix : 15u) << suffix_len;
		uint32_t size = prefix == 14u ? 4u : prefix - 3u;
		code += wb_u(wb, lane, size);
		if (prefix >= 15 && suffix_len == 0)
			code += 15;
		if (prefix >= 16)
			code += (1u << (prefix - 3u)) - 4096u;
	}
	code += first_adj;
	/* even code -> +(code + 2) / 2, odd -> -(code + 1) / 2 */
	uint32_t mag = (code + 2u) >> 1;
	if (mag > 32767u) {
		/* levelVal is int16_t in the reference: after an escape the
		 * magnitude that drives suffix_len is the wrapped one */
		uint32_t neg = code & 1u;
		int32_t v16 = (int16_t)((mag ^ (0u - neg)) + neg);
		mag = (uint32_t)(v16 < 0 ? -v16 : v16);
	}
	suffix_len = suffix_len ? suffix_len : 1u;
	suffix_len += (mag > (3u << (suffix_len - 1u)) && suffix_len < 6u) ? 1u : 0u;
	return code;
}

/* 9.2 residual_block (ref src/h264_syntax_slice_data.h:103-242).  The decoded
 * levels go to stage[coef_off ...] (int16 units from the record start).
 * Returns TotalCoeff.  Malformed codes do not branch out: they raise a flag
 * (serr) and the walk goes on with benign values until
 * the caller looks at them.
 *
 * Division of labour (profiles/r02_decode_cost_model.md): the bit cursor and
 * everything decoded from the bits live in wave-uniform VECTOR registers
 * (vbits.hip.h); the scalar unit keeps the counts that steer loops (TotalCoeff,
 * TrailingOnes, zerosLeft).  A level is recorded as its levelCode in lane i and
 * all levels turn into values at once behind the loop; zerosLeft before run i
 * goes to lane i the same way.
 *
 * vb: normalised on entry (vo < 32). */
__device__ __forceinline__ uint32_t
sd_residual_block(VBits &vb, WaveBits &wb, const VlcTabs &vt, uint32_t lane,
		  uint8_t *stage, uint32_t cls, uint32_t bits, uint32_t maxnum,
		  uint32_t coef_off, uint32_t &serr SD_TARG)
{
	/* ---- coeff_token; bits: the peek at the cursor, which the block loop
	 * already has ---- */
	uint32_t tc, t1, len;
	{
		/* a malformed token comes back as "no coefficients, length 0" with
		 * DEV_VLC_BAD set */
		uint32_t cw = vlc_match_total(vt.ct + 512u * cls, vt.off64, bits);
		serr |= cw & DEV_VLC_BAD;
		len = cw & 31u;
		t1 = (cw >> 5) & 3u;
		tc = (cw >> 7) & 31u;
	}
	vb_skip(vb, len); /* <= 16 bits */
	SD_T(4);
	if (SD_RARE(tc == 0))
		return 0;
	/* ---- trailing ones: bit (t1 - 1 - i) of the sign field is level i, i.e.
	 * bit 31 - i of what the peek holds behind the token (at most 16 bits
	 * long: the peek still has the signs).  As levelCode: +1 = 0, -1 = 1 ---- */
	uint32_t codes;
	{
		uint32_t rest = bits << len;
		uint32_t sbit = (rest << lane) >> 31;
		codes = lane < t1 ? sbit : 0u;
	}
	vb_skip(vb, t1); /* vo <= 31 + 16 + 3 */
	uint32_t i = t1;
	if (i < tc) {
		/* ---- levels (9.2.2.1).  The FIRST one in full generality:
		 * suffixLength 0 or 1, and after fewer than three trailing ones it
		 * cannot be +-1, so its code is offset by 2 ---- */
		/* plain integer arithmetic: booleans would live in 64-bit lane
		 * masks and come back through the vector unit */
		const uint32_t few = (t1 - 3u) >> 31; /* t1 < 3 */
		const uint32_t first_adj = 2u * few;
		const uint32_t sl0 = few & ((10u - tc) >> 31); /* && tc > 10 */
		uint32_t sl; /* suffixLength, vector side */
		/* lane i of `codes` takes level i: the lane mask comes from the
		 * scalar side (one s_lshl_b64 per level) instead of a v_cmp into
		 * VCC with its wait states before the select */
		uint64_t imask = 1ull << i;
		vb_norm(vb, wb);
		{
			uint32_t b1 = vb_peek(vb);
			/* 15 stands for "15 or more": the escape path looks again */
			uint32_t p = (uint32_t)__builtin_clz(b1 | 0x10000u);
			if (SD_RARE(vb_true(p >= 14u + sl0))) {
				uint32_t ssl = sl0;
				vb_store(vb, wb);
				uint32_t code = sd_level_escape(wb, lane, ssl, first_adj,
								serr);
				vb_load(vb, wb);
				sl = ssl;
				VB_VREG(sl);
				VB_VREG(code);
				codes = lane == i ? code : codes;
			} else {
				/* prefix zeros, a one, sl0 suffix bits (the low sl0 of
				 * the code word's n):
				 * levelCode = (prefix << sl0) + suffix */
				uint32_t n = p + 1u + sl0;
				uint32_t code = (p << sl0) +
						__builtin_amdgcn_ubfe(b1, 32u - n, sl0) +
						first_adj;
				vb_skip(vb, n); /* <= 16 bits */
				/* suffixLength becomes 1, and 2 when the magnitude
				 * (code + 2) >> 1 exceeds 3 */
				sl = code >= 6u ? 2u : 1u;
				asm("v_cndmask_b32 %0, %0, %1, %2"
				    : "+v"(codes)
				    : "v"(code), "s"(imask));
			}
		}
		imask <<= 1;
		/* every later level finds 1 <= suffixLength <= 6 and no adjustment:
		 * levelCode lies below 3 << suffixLength exactly when prefix < 3, so
		 * suffixLength grows iff prefix >= 3 -- no magnitude needed */
		/* Window: a level takes at most 21 bits here, so the loop keeps
		 * vo <= 43 (64 - 21) at its top.  ONE test per level covers both
		 * rare events, the escape (prefix >= 15) and the window step: the
		 * usual updates are applied first and an escape takes them back. */
		vb_norm(vb, wb);
		const uint64_t iend = 1ull << tc; /* tc <= 16 */
		/* (a guarded do-while with the guard marked usual: the loop then sits
		 * in line behind the first level, and the way through it needs no
		 * jump in and no jump out) */
		if (SD_USUAL(imask != iend)) do {
			/* (raw peek: the test below is a 32-bit one already) */
			uint32_t b1 = (uint32_t)((vb.W << vb.vo) >> 32);
			uint32_t p = (uint32_t)__builtin_clz(b1 | 0x10000u);
			uint32_t n = p + 1u + sl;
			/* vo >= 44 behind the level: step.  p >= 15: vo is 17 at
			 * least and vo + 2 p >= 44 as well; a smaller p that
			 * trips the test early (vo >= 38) takes the step path a
			 * level before it must, which that path does not mind.
			 * The cursor moves first and an escape takes it back: no
			 * copy of the old offset is carried. */
			vb.vo += n; /* n <= 21 bits */
			const uint32_t rare = vb.vo + 2u * p;
			if (SD_RARE(vb_true(rare >= 44u))) {
				const uint32_t i = (uint32_t)__builtin_ctzll(imask);
				/* (opaque here, so that the sum above is made in
				 * place: otherwise the old offset stays live for
				 * the subtraction below and every level pays a
				 * register copy) */
				VB_VREG(vb.vo);
				if (vb_true(p >= 15u)) {
					uint32_t ssl = (uint32_t)
						__builtin_amdgcn_readfirstlane((int)sl);
					vb.vo -= n;
					vb_store(vb, wb);
					uint32_t c2 = sd_level_escape(wb, lane, ssl, 0u,
								      serr);
					vb_load(vb, wb);
					sl = ssl;
					VB_VREG(sl);
					VB_VREG(c2);
					codes = lane == i ? c2 : codes;
				} else {
					uint32_t code = ((p - 1u) << sl) +
							(b1 >> (32u - n));
					codes = lane == i ? code : codes;
					uint32_t grown = sl + (p >= 3u ? 1u : 0u);
					sl = grown < 6u ? grown : 6u;
					vb_norm(vb, wb);
				}
				continue;
			}
			uint32_t code = (p << sl) + __builtin_amdgcn_ubfe(b1, 32u - n, sl);
			/* + 1 iff p >= 3, as bit p of 0xfff8: a compare would put
			 * the carry through VCC with its wait states (and the
			 * optimiser reads "p >= 3" through the count of leading
			 * zeros as a 64-bit compare of the shifted window with a
			 * constant in spilled scalar registers) */
			uint32_t grown = sl + ((0xfff8u >> p) & 1u);
			sl = grown < 6u ? grown : 6u;
			asm("v_cndmask_b32 %0, %0, %1, %2"
			    : "+v"(codes)
			    : "v"(code), "s"(imask));
		} while ((imask <<= 1) != iend);
		i = tc;
	}
	SD_T(5);
	/* ---- total_zeros and the runs: zl = zerosLeft before run i, per lane
	 * (collecting them as nibbles of a scalar pair, two scalar operations a
	 * run instead of three vector ones, is slower at four waves a SIMD: the
	 * scalar unit is the fuller one there, profiles/r03_decode_ab_history) ---- */
	uint32_t zl = 0;
	uint32_t zeros_left = 0;
	if (SD_USUAL(tc < maxnum)) {
		/* no vb_norm() here: vo <= 50 on every way in (behind the token and
		 * the trailing ones 31 + 16 + 3, behind the first level 31 + 16,
		 * behind the loop 43, behind an escape 31), and a total_zeros code
		 * word has 9 bits at most */
		uint32_t b1 = vb_peek(vb);
		/* 1 <= tc < maxnum and maxnum in {4, 8, 15, 16} (the callers' block
		 * constants) keep k inside the 25 tables */
		uint32_t k = maxnum == 4 ? 14u + tc : maxnum == 8 ? 17u + tc : tc - 1u;
		uint32_t bad;
		int hit;
		uint32_t cw = vlc_match(vt.tz + 128u * k, vt.off16, b1, bad, hit);
		uint32_t okm = bad - 1u;
		serr |= bad;
		vb_skip(vb, cw & 31u & okm); /* <= 9 bits */
		zeros_left = (uint32_t)hit & okm;
	}
	/* run_before (Table 9-10).  zerosLeft > 6: three bits (111 .. 001 -> 0 ..
	 * 6) or a unary tail (0001 -> 7, ...), up to 11 bits, one at a time.
	 * zerosLeft <= 6: at most 3 bits, so ten of them fit behind one vb_norm();
	 * a byte table per class held by lane `class` (vt.rb_lo / rb_hi), entry =
	 * len << 4 | zerosLeft behind the run, picked by the next three bits. */
	SD_T(6);
	i = 0;
	/* (loop conditions as ONE compare of a minimum: "runs still to read and
	 * zeros still to place") */
	uint32_t left = tc - 1u;
	if (SD_USUAL((left < zeros_left ? left : zeros_left) != 0u)) do {
		vb_norm(vb, wb);
		if (zeros_left > 6) {
			zl = wb_setlane(zl, i, zeros_left, lane);
			uint32_t sb = vb_peek_s(vb);
			uint32_t t3 = sb >> 29;
			if (t3) {
				vb_skip(vb, 3);
				zeros_left -= 7u - t3;
			} else {
				uint32_t lz = (uint32_t)__builtin_clz(sb | 0x00100000u);
				uint32_t bad = (sb >> 21) == 0 ? 1u : 0u; /* > 11 bits */
				uint32_t run = bad ? 0u : lz + 4u;
				uint32_t over = run > zeros_left ? 1u : 0u;
				serr |= bad | over;
				vb_skip(vb, bad ? 0u : lz + 1u);
				zeros_left -= over ? zeros_left : run;
			}
			i++;
			left--;
			continue;
		}
		uint32_t batch = left < 10u ? left : 10u;
		left -= batch;
		do {
			zl = wb_setlane(zl, i, zeros_left, lane);
			uint32_t top3 = vb_peek(vb) >> 29;
			/* byte top3 of the lane's eight in byte 0 (the other
			 * three selectors are 0: copies of the lane's first
			 * byte, masked off below -- cheaper than making them
			 * zero selectors first) */
			uint32_t ev = __builtin_amdgcn_perm(vt.rb_hi, vt.rb_lo, top3);
			uint32_t e = (uint32_t)__builtin_amdgcn_readlane(
				(int)ev, (int)zeros_left);
			vb_skip(vb, (e >> 4) & 15u); /* <= 3 bits */
			zeros_left = e & 15u; /* the table has zerosLeft - run */
			i++;
			batch--;
		} while ((batch < zeros_left ? batch : zeros_left) != 0u);
		left += batch;
	} while ((left < zeros_left ? left : zeros_left) != 0u);
	zl = wb_setlane(zl, tc - 1u, zeros_left, lane);
	SD_T(7);
	/* ---- levelCode -> value (even code -> +(code + 2) / 2, odd ->
	 * -(code + 1) / 2, as int16), and lane i places coefficient i ---- */
	uint32_t ng = codes & 1u;
	uint32_t lv = ((((codes + 2u) >> 1) ^ (0u - ng)) + ng) & 0xffffu;
	uint32_t pos = zl + tc - 1u - lane;
	/* A position outside the block (damaged stream: index 15 of a
	 * 15-coefficient block) is dropped and the walk goes on, which is what
	 * the reference's unchecked store amounts to for everything it makes
	 * observable (ref src/h264_syntax_slice_data.h:222-226; the argument is
	 * in oracle/orc_cavlc.c at the same place). */
	bool mine = lane < tc && pos < maxnum;
	*reinterpret_cast<uint16_t *>(
		lds_sel(stage + H264_AMD_T2_HEAD + 2u * (coef_off + pos),
			stage - SD_DUMP_BYTES, mine, lane)) = (uint16_t)lv;
	SD_T(8);
	return tc;
}

/* checksum terms (h264_amd_checksum_term, h264_amd.h) of 4 consecutive dwords
 * starting at dword index g: per dword one v_dot2_u32_u16 against the packed
 * position weights, one for the two squares, one multiply by the 32-bit weight;
 * the weights step with one packed / one plain add */
typedef unsigned short sd_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void sd_checksum(const uint4 q, uint32_t g,
					    bool valid, uint32_t &lo,
					    uint32_t &hi)
{
	const uint32_t d[4] = {q.x, q.y, q.z, q.w};
	/* (4g + 1, 4g + 3) * KP, both modulo 2^16 */
	const uint32_t x = 4u * g + 1u;
	sd_u16x2 pos, kp, step;
	pos.x = (unsigned short)x;
	pos.y = (unsigned short)(x + 2u);
	kp.x = kp.y = (unsigned short)H264_AMD_CHECKSUM_KP;
	step.x = step.y = (unsigned short)(4u * H264_AMD_CHECKSUM_KP);
	sd_u16x2 p = pos * kp;
	uint32_t b = (2u * g + 1u) * H264_AMD_CHECKSUM_KB;
	uint32_t l = 0, h = 0;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		sd_u16x2 w;
		__builtin_memcpy(&w, &d[k], 4);
		l = __builtin_amdgcn_udot2(w, p, l, false);
		h += __builtin_amdgcn_udot2(w, w, 0u, false) * b;
		p += step;
		b += 2u * H264_AMD_CHECKSUM_KB;
	}
	lo += valid ? l : 0u;
	hi += valid ? h : 0u;
}

struct OutCtx {
	uint32_t tier, mb_stride, rec_bytes; /* rec_bytes = T2 stride for cat */
	uint8_t *mb_out; /* this slice's output area */
	uint32_t t0acc; /* per lane: T0 word of macroblock (idx & 63) */
	uint32_t lo, hi; /* per lane checksum partials */
};

/* T0: one word per macroblock, flushed 64 at a time (256-byte stores) */
__device__ __forceinline__ void sd_t0_put(OutCtx &o, uint32_t word0,
					  uint32_t mbidx, uint32_t lane)
{
	o.t0acc = wb_setlane(o.t0acc, mbidx & 63u, word0, lane);
	if (SD_RARE((mbidx & 63u) == 63u))
		*reinterpret_cast<uint32_t *>(
			o.mb_out + (size_t)(mbidx - 63u + lane) * o.mb_stride) = o.t0acc;
}

/* Finish one macroblock whose record is staged in LDS.  Lanes beyond the end
 * of the record repeat the store of its last 16 bytes (same address, same
 * data) instead of being masked off: no divergent branch. */
__device__ __forceinline__ void sd_emit_staged(OutCtx &o, const uint8_t *stage,
					       uint32_t mbidx, uint32_t lane)
{
	uint32_t word0 = wb_uniform(*reinterpret_cast<const uint32_t *>(stage));
	for (uint32_t part = 0; part * 1024u < o.rec_bytes; part++) {
		uint32_t want = part * 1024u + 16u * lane;
		bool valid = want < o.rec_bytes;
		uint32_t off = valid ? want : o.rec_bytes - 16u;
		uint4 q = *reinterpret_cast<const uint4 *>(stage + off);
		sd_checksum(q, mbidx * 512u + off / 4u, valid, o.lo, o.hi);
		if (o.tier == H264_AMD_TIER_T2)
			*reinterpret_cast<uint4 *>(
				o.mb_out + (size_t)mbidx * o.mb_stride + off) = q;
	}
	if (o.tier == H264_AMD_TIER_T1) {
		uint32_t l = lane & 15u;
		*reinterpret_cast<uint32_t *>(o.mb_out + (size_t)mbidx * o.mb_stride +
					      4u * l) =
			*reinterpret_cast<const uint32_t *>(stage + 4u * l);
	}
	if (o.tier == H264_AMD_TIER_T0)
		sd_t0_put(o, word0, mbidx, lane);
}

/* A skipped macroblock: the record is word0 followed by zeros. */
__device__ __forceinline__ void sd_emit_skipped(OutCtx &o, uint32_t word0,
						uint32_t mbidx, uint32_t lane)
{
	sd_checksum(make_uint4(word0, 0, 0, 0), mbidx * 512u, lane == 0, o.lo,
		    o.hi);
	if (o.tier == H264_AMD_TIER_T2) {
		for (uint32_t part = 0; part * 1024u < o.rec_bytes; part++) {
			uint32_t want = part * 1024u + 16u * lane;
			uint32_t off = want < o.rec_bytes ? want : o.rec_bytes - 16u;
			uint4 q = make_uint4(off == 0 ? word0 : 0u, 0, 0, 0);
			*reinterpret_cast<uint4 *>(
				o.mb_out + (size_t)mbidx * o.mb_stride + off) = q;
		}
	} else if (o.tier == H264_AMD_TIER_T1) {
		uint32_t l = lane & 15u;
		*reinterpret_cast<uint32_t *>(o.mb_out + (size_t)mbidx * o.mb_stride +
					      4u * l) = l == 0 ? word0 : 0u;
	} else if (o.tier == H264_AMD_TIER_T0) {
		sd_t0_put(o, word0, mbidx, lane);
	}
}

/* advance the neighbour context to the next macroblock */
__device__ __forceinline__ void sd_next_mb(MbCtx &m, uint8_t *row, uint8_t *dump,
					   uint32_t &mbx, uint32_t W, uint32_t lane)
{
	*lds_sel(row + 16u * mbx + (uint32_t)(m.row_slot & 15), dump,
		 m.row_slot >= 0, lane) = (uint8_t)m.cur;
	m.left = m.cur;
	m.cur = 0;
	mbx++;
	if (mbx == W)
		mbx = 0;
}

/* ---- MBAFF frames: macroblock pairs ----------------------------------------
 * (ref src/h264_macroblock.c:110-231 neighbouring locations 6.4.12.2,
 * src/h264_slice_data.c:1145-1206 field flag inference and ref_idx range,
 * src/h264_syntax_slice_data.h:754-761 where the flag is read.)
 * The neighbour context of MbCtx stays what it is -- per lane nA + nB of the
 * lane's block -- only where the external parts come from changes: the block
 * to the LEFT of a left-edge block sits in one of the TWO macroblocks of the
 * pair to the left, in a row that depends on frame / field coding of both
 * pairs (Table 6-4); the block ABOVE a top-edge block sits in the bottom row
 * of the pair above (either macroblock) or of this pair's own top macroblock.
 * So the lanes keep the nz_coeff of both macroblocks of the left pair and of
 * this pair's top macroblock, and a column of the row context holds the
 * bottom rows of both macroblocks of the pair above (32 bytes per column:
 * [0,12) top macroblock, [12,16) slice-group tag, [16,28) bottom macroblock,
 * [28] the pair's field flag). */
#define SD_ROW_STRIDE_MBAFF 32u

struct PairCtx {
	/* per lane, changing */
	uint32_t left_top, left_bot; /* nz_coeff of the pair to the left */
	uint32_t pair_top; /* nz_coeff of this pair's top macroblock */
	/* per lane, constant */
	uint32_t y4; /* sample row of the lane's block: 0, 4, 8, 12 */
	uint32_t maxh; /* 16, or MbHeightC for chroma blocks */
	uint32_t base; /* 16 * component */
	uint32_t chroma; /* 1: 4:2:0 / 4:2:2 chroma block */
	uint32_t own_src; /* lane of the block above in the pair's top MB */
	/* uniform */
	uint32_t left_field; /* field flag of the pair to the left */
	uint32_t pair_field; /* of this pair, as far as known */
	uint32_t top_skipped;
};

__device__ __forceinline__ void sd_pair_consts(PairCtx &P, uint32_t lane,
					       uint32_t cat)
{
	const uint32_t comp = lane >> 4, b = lane & 15u;
	const bool chroma = lane < 48 && comp > 0 && (cat == 1 || cat == 2);
	const uint32_t lx = (b & 1u) | ((b >> 1) & 2u);
	const uint32_t ly = ((b >> 1) & 1u) | ((b >> 2) & 2u);
	const uint32_t cx = b & 1u, cy = b >> 1;
	const uint32_t h4 = cat == 1 ? 2u : 4u;
	P.chroma = chroma ? 1u : 0u;
	P.base = 16u * comp;
	P.y4 = 4u * (chroma ? cy : ly);
	P.maxh = chroma ? 4u * h4 : 16u;
	P.own_src = chroma ? 16u * comp + 2u * (h4 - 1u) + cx
			   : 16u * comp + blk_at(lx, 3);
	P.left_top = P.left_bot = P.pair_top = 0;
	P.left_field = P.pair_field = P.top_skipped = 0;
}

/* a macroblock of a pair starts: external neighbour sums (see above).
 * top: 1 for the pair's top macroblock; field: this macroblock's flag;
 * availA / availB: the pair to the left / above is part of this slice;
 * above_field: field flag of the pair above */
__device__ __forceinline__ void sd_mb_begin_mbaff(MbCtx &m, const PairCtx &P,
						  const uint8_t *row, uint32_t mbx,
						  uint32_t top, uint32_t field,
						  uint32_t above_field)
{
	const uint32_t cur_frame = field ^ 1u, a_frame = P.left_field ^ 1u;
	/* ---- left (xN < 0): which macroblock of the left pair, which row ---- */
	uint32_t yN = P.y4, src_bot, yM;
	if (cur_frame) {
		if (a_frame) {
			src_bot = top ^ 1u;
			yM = yN;
		} else {
			/* block rows start on even sample rows: always the top
			 * field macroblock */
			src_bot = 0;
			yM = top ? yN >> 1 : (yN + P.maxh) >> 1;
		}
	} else {
		if (a_frame) {
			uint32_t lower = yN < (P.maxh >> 1) ? 0u : 1u; /* per lane */
			src_bot = lower;
			yM = (yN << 1) + (top ^ 1u) - (lower ? P.maxh : 0u);
		} else {
			src_bot = top ^ 1u;
			yM = yN;
		}
	}
	uint32_t r = yM >> 2;
	uint32_t src = P.chroma ? P.base + 2u * r + 1u
				: P.base + 5u + 2u * (r & 1u) + 8u * (r >> 1);
	uint32_t lt = (uint32_t)__shfl((int)P.left_top, (int)(src & 63u));
	uint32_t lb = (uint32_t)__shfl((int)P.left_bot, (int)(src & 63u));
	uint32_t lv = src_bot ? lb : lt;
	/* ---- above (yN < 0) ---- */
	uint32_t own = cur_frame & (top ^ 1u); /* the pair's own top MB */
	uint32_t use_top_mb = (cur_frame ^ 1u) & top & above_field;
	uint32_t tv_row = row[SD_ROW_STRIDE_MBAFF * mbx + (use_top_mb ? 0u : 16u) +
			      (uint32_t)(m.top_slot & 15)];
	uint32_t tv_own = (uint32_t)__shfl((int)P.pair_top, (int)(P.own_src & 63u));
	uint32_t tv = own ? tv_own : tv_row;
	bool extA = m.a_ext != 0xffu && m.availA != 0;
	bool extB = m.top_slot >= 0 && (own != 0 || m.availB != 0);
	m.sum = (extA ? lv : 0u) + (extB ? tv : 0u);
	bool hasA = (m.hasint & 1u) != 0 || extA;
	bool hasB = (m.hasint & 2u) != 0 || extB;
	m.bothv = (hasA && hasB) ? 1u : 0u;
}

/* a macroblock of a pair is done */
__device__ __forceinline__ void sd_next_mb_mbaff(MbCtx &m, PairCtx &P, uint8_t *row,
						 uint8_t *dump, uint32_t mbx,
						 uint32_t top, uint32_t field,
						 uint32_t lane)
{
	if (top) {
		P.pair_top = m.cur;
	} else {
		/* the pair is complete: its bottom rows replace those of the
		 * pair above in this column, and it becomes the left pair */
		uint8_t *col = row + SD_ROW_STRIDE_MBAFF * mbx;
		uint32_t slot = (uint32_t)(m.row_slot & 15);
		*lds_sel(col + slot, dump, m.row_slot >= 0, lane) =
			(uint8_t)P.pair_top;
		*lds_sel(col + 16u + slot, dump, m.row_slot >= 0, lane) =
			(uint8_t)m.cur;
		col[28] = (uint8_t)field; /* every lane, same byte */
		P.left_top = P.pair_top;
		P.left_bot = m.cur;
	}
	/* (kept out of the branch above: assigned there, the optimiser takes
	 * the value for lane-dependent and the whole cursor with it) */
	P.left_field = top ? P.left_field : field;
	m.left = m.cur;
	m.cur = 0;
}

/* ---- slice data (ref src/h264_syntax_slice_data.h:701-787) -------------- */
/* ONLY_I: instantiated for I slices (not SI) alone, so that everything the
 * P / B / SI syntax needs is compiled out of the intra loop instead of being
 * carried through it in registers. */
/* MBAFF: instantiated for the slices of MBAFF frames (never with ONLY_I):
 * macroblock pairs, see PairCtx */
/* vb_sync() with the priority step (see the slice-data loop) in its taken
 * branch */
__device__ __forceinline__ void sd_vb_sync(VBits &b, WaveBits &s, uint32_t lane,
					   const SliceState &S)
{
	if (SD_RARE(b.sd >= WB_SLIDE_BITS / 32u)) {
		vb_store(b, s);
		wb_sync(s, lane);
		vb_load(b, s);
#ifndef SD_NO_PRIO_ROTATION
		if (S.prio_on) {
			/* base: RBSP byte index of the window start, 128 more
			 * per slide */
			switch (((s.base >> SD_PRIO_SHIFT) + S.prio_slot) & 3u) {
			case 0:
				__builtin_amdgcn_s_setprio(0);
				break;
			case 1:
				__builtin_amdgcn_s_setprio(1);
				break;
			case 2:
				__builtin_amdgcn_s_setprio(2);
				break;
			default:
				__builtin_amdgcn_s_setprio(3);
				break;
			}
		}
#endif
	}
}

template <bool ONLY_I, bool MBAFF>
__device__ __forceinline__ int
sd_slice_data(WaveBits &wb, const VlcTabs &vt, uint32_t cbp_map, uint32_t lane,
	      uint8_t *stage, uint8_t *row, const SliceState &S, OutCtx &o,
	      uint32_t mb_capacity, uint32_t &mb_count, const uint8_t *fmo_map,
	      uint32_t pic_mbs SD_TARG)
{
	const uint32_t cat = S.cat, W = S.W;
	const bool is_si = !ONLY_I && S.slice_type == H264_SLICE_TYPE_SI;
	const bool is_i = ONLY_I || S.slice_type == H264_SLICE_TYPE_I || is_si;
	const bool is_b = !ONLY_I && S.slice_type == H264_SLICE_TYPE_B;
	const bool is_p = !ONLY_I && (S.slice_type == H264_SLICE_TYPE_P ||
				      S.slice_type == H264_SLICE_TYPE_SP);
	const bool chroma12 = cat == 1 || cat == 2;
	const uint32_t nchroma = cat == 1 ? 4u : 8u; /* 4x4 blocks per component */
	bool ok = true;
	VBits vb;
	vb_load(vb, wb);
#define VRD_UE(var)                                                            \
	do {                                                                   \
		var = vb_ue(vb, wb, ok);                                       \
		if (SD_RARE(!ok))                                  \
			return -EIO;                                           \
	} while (0)
#define VRD_SE(var)                                                            \
	do {                                                                   \
		var = vb_se(vb, wb, ok);                                       \
		if (SD_RARE(!ok))                                  \
			return -EIO;                                           \
	} while (0)
	MbCtx m;
	m.cur = m.left = m.sum = 0;
	m.availA = m.availB = 0;
	m.bothv = 0;
	sd_mb_consts(m, lane, cat);
	/* Residual blocks of a macroblock as a mask over the steps (bit = step in
	 * bitstream order, src/h264_syntax_slice_data.h:334-419), for every
	 * coded_block_pattern at once: lane = CodedBlockPatternLuma |
	 * CodedBlockPatternChroma << 4.  Luma bit k covers blocks 4k .. 4k + 3 of
	 * each luma-like component (steps 1 .. 16 behind the component's DC
	 * step); chroma 1 / 2: the two DC blocks, 2: the AC blocks too. */
	uint32_t todo_lo, todo_hi;
	{
		const uint32_t cl = lane & 15u, cc = (lane >> 4) & 3u;
		uint32_t e16 = ((cl & 1u) | ((cl & 2u) << 3) | ((cl & 4u) << 6) |
				((cl & 8u) << 9)) * 15u;
		uint64_t per_comp = (uint64_t)e16 << 1;
		uint64_t t = per_comp;
		if (cat == 3)
			t |= (per_comp << 17) | (per_comp << 34);
		if (chroma12) {
			uint64_t c = 0;
			if (cc & 3u)
				c |= 3u;
			if (cc & 2u)
				c |= ((1ull << (2u * nchroma)) - 1ull) << 2;
			t |= c << 17;
		}
		todo_lo = (uint32_t)t;
		todo_hi = (uint32_t)(t >> 32);
	}
	/* Table 7-11 (mb_type of I slices): lane = value; mb_type | prediction
	 * mode << 4 | partitions << 7 | Intra16x16PredMode << 8 |
	 * CodedBlockPatternLuma << 10 | CodedBlockPatternChroma << 14 */
	uint32_t itab;
	{
		const uint32_t t1 = lane - 1u;
		uint32_t i16e = (uint32_t)H264_MB_TYPE_I_16x16 |
				((uint32_t)PM_I16x16 << 4) | (1u << 7) |
				((t1 & 3u) << 8) | ((lane <= 12u ? 0u : 15u) << 10) |
				(((t1 >> 2) % 3u) << 14);
		itab = lane == 0   ? (uint32_t)H264_MB_TYPE_I_NxN |
					   ((uint32_t)PM_I4x4 << 4) | (1u << 7)
		       : lane <= 24 ? i16e
				    : (uint32_t)H264_MB_TYPE_I_PCM |
					      ((uint32_t)PM_I4x4 << 4);
	}
	/* the Intra16x16 DC step of every luma-like component */
	const uint64_t dc_steps = cat == 3 ? 1ull | (1ull << 17) | (1ull << 34) : 1ull;
	uint32_t mbx = S.first_mb % W;
	uint32_t mbidx = 0;
	/* bytes of row context per macroblock column */
	const uint32_t RS = MBAFF ? SD_ROW_STRIDE_MBAFF : 16u;
	PairCtx P;
	if (MBAFF)
		sd_pair_consts(P, lane, cat);
	/* Slice groups (generic instantiation only).  The slice walks the
	 * addresses of ONE group in increasing order, so "macroblock X is part
	 * of this slice and came before" (the reference's available flag,
	 * src/h264_macroblock.c:318-350) is: X is the previous address for the
	 * left neighbour, and for the one above, the last address this slice
	 * visited in that column (kept as a tag in the column's row context). */
	const bool fmo = !ONLY_I && S.fmo != 0;
	/* (first_mb_in_slice counts pairs in an MBAFF frame; with slice groups
	 * the tags name the TOP macroblock of the last pair of a column) */
	const uint32_t first_addr = S.first_mb * (MBAFF ? 2u : 1u);
	uint32_t cur_addr = first_addr, prev_addr = 0xffffffffu;
	if (fmo) {
		for (uint32_t c0 = 0; c0 < W; c0 += 64u) {
			uint32_t c = c0 + lane < W ? c0 + lane : W - 1u;
			*reinterpret_cast<uint32_t *>(row + RS * c + 12u) =
				0xffffffffu;
		}
	}
/* one macroblock further: tags, addresses, column */
#define SD_FMO_STEP()                                                          \
	do {                                                                   \
		if (fmo) {                                                     \
			if (!MBAFF || (cur_addr & 1u)) {                       \
				uint32_t t_ = MBAFF ? cur_addr - 1u : cur_addr; \
				uint32_t col_ = (MBAFF ? t_ >> 1 : t_) % W;    \
				*reinterpret_cast<uint32_t *>(row + RS * col_ + \
							      12u) = t_;       \
				prev_addr = t_;                                \
			}                                                      \
			cur_addr = sd_fmo_next(fmo_map, pic_mbs, cur_addr, lane); \
			mbx = (MBAFF ? cur_addr >> 1 : cur_addr) % W;          \
		} else if (MBAFF) {                                            \
			cur_addr++;                                            \
			if (!(cur_addr & 1u)) {                                \
				mbx++;                                         \
				if (mbx == W)                                  \
					mbx = 0;                               \
			}                                                      \
		}                                                              \
	} while (0)
	/* MBAFF: is the pair to the left / above part of this slice (ref
	 * src/h264_macroblock.c:327-350), and the field flag of the pair above */
#define SD_PAIR_NEIGHBOURS()                                                   \
	do {                                                                   \
		uint32_t pair_ = cur_addr >> 1;                                \
		uint32_t first_ = S.first_mb;                                  \
		m.availA = (mbx != 0 && pair_ >= first_ + 1u &&                \
			    (mbidx >> 1) >= 1u)                                \
				   ? 1u                                        \
				   : 0u;                                       \
		m.availB = (pair_ >= first_ + W && (mbidx >> 1) >= W) ? 1u : 0u; \
		if (fmo) {                                                     \
			uint32_t tag_ = wb_uniform(                            \
				*reinterpret_cast<const uint32_t *>(           \
					row + RS * mbx + 12u));                \
			uint32_t top_ = cur_addr & ~1u;                        \
			m.availA = (mbx != 0 && prev_addr + 2u == top_) ? 1u : 0u; \
			m.availB = (top_ >= 2u * W && tag_ == top_ - 2u * W)   \
					   ? 1u                                \
					   : 0u;                               \
		}                                                              \
		above_field = wb_uniform((uint32_t)row[RS * mbx + 28u]) & 1u;  \
	} while (0)
	/* field pictures: every macroblock carries the picture's flag
	 * (ref src/h264_slice_data.c:1145-1149) */
	const uint32_t field_flag = S.field_pic ? (uint32_t)H264_AMD_MB_FIELD : 0u;
	const uint32_t skip_word0 =
		(is_b ? (uint32_t)H264_MB_TYPE_B_SKIP : (uint32_t)H264_MB_TYPE_P_SKIP) |
		(((uint32_t)H264_AMD_MB_SKIPPED | field_flag) << 16);
	/* for coded macroblocks the flag waits in a vector register and joins
	 * the record's first word there: a scalar carried through the whole
	 * macroblock walk cost 0.7 % */
	uint32_t v_field = field_flag << 16;
	asm volatile("" : "+v"(v_field));

	/* Fair shares of the SIMD.  The instruction arbiter of a SIMD serves the
	 * OLDEST wave first when priorities are equal: of the four waves of a full
	 * launch the first dispatched then runs almost as if alone and the last
	 * one takes half as long again, which is when the kernel ends -- with
	 * three, two, one wave left on the SIMD for the last third of the time
	 * (tools/phase_timers.py prints the waves' own wall times by quarter of
	 * the batch: 50.8 / 56.5 / 64.4 / 75.0 ms).  So every wave steps its own
	 * priority round the four levels, offset by its age rank (workgroups are
	 * dispatched in order, one layer of them per pass over the compute
	 * units: S.prio_slot): a step whenever the bit window slides (sd_vb_sync:
	 * 1024 bits of slice data, about six macroblocks -- a branch that exists
	 * anyway), so every wave spends the same share of its
	 * bits at every level and all four finish together.  62.4 -> 54.4 ms for
	 * the bench batch. */
	do {
		uint32_t prev_mb_skipped = 0;
		/* the scalar cursor is in step with vb here (vb_store at the end of
		 * the previous macroblock / vb_load above) */
		sd_vb_sync(vb, wb, lane, S);
		if (SD_RARE(wb_overrun(wb)))
			return -EIO;
		if (!is_i) {
			uint32_t run;
			VRD_UE(run);
			/* a run read across the end of the data (zero padding) is
			 * the reference's -EIO on the ue itself: no skipped
			 * macroblock may come out of it */
			vb_store(vb, wb);
			if (SD_RARE(wb_overrun(wb)))
				return -EIO;
			prev_mb_skipped = run > 0;
			for (uint32_t r = 0; r < run; r++) {
				if (SD_RARE(mbidx >= mb_capacity))
					return H264_AMD_E_MB_OVERFLOW;
				/* under slice groups the address runs ahead of
				 * the count (h264_amd.h, H264_AMD_MAX_SLICE_MBS) */
				if (fmo && cur_addr - first_addr >=
						   H264_AMD_MAX_SLICE_MBS)
					return H264_AMD_E_MB_OVERFLOW;
				if (MBAFF) {
					/* field flag of a skipped macroblock (ref
					 * src/h264_slice_data.c:1150-1172) */
					uint32_t top = (cur_addr & 1u) ^ 1u, ff;
					uint32_t above_field;
					SD_PAIR_NEIGHBOURS();
					if (top) {
						ff = 0; /* waits for the bottom one */
						P.top_skipped = 1;
					} else if (!P.top_skipped) {
						ff = P.pair_field;
					} else {
						ff = m.availA   ? P.left_field
						     : m.availB ? above_field
								: 0u;
					}
					P.pair_field = ff;
					sd_emit_skipped(
						o,
						skip_word0 |
							(ff ? (uint32_t)H264_AMD_MB_FIELD
									<< 16
							    : 0u),
						mbidx, lane);
					sd_next_mb_mbaff(m, P, row,
							 stage - SD_DUMP_BYTES, mbx, top,
							 ff, lane);
				} else {
					sd_emit_skipped(o, skip_word0, mbidx, lane);
					sd_next_mb(m, row, stage - SD_DUMP_BYTES, mbx,
						   W, lane);
				}
				SD_FMO_STEP();
				mbidx++;
				mb_count = mbidx;
			}
			if (run > 0 && !wb_more_rbsp_data(wb, lane))
				break;
		}
		if (SD_RARE(mbidx >= mb_capacity))
			return H264_AMD_E_MB_OVERFLOW;
		if (SD_RARE(fmo && cur_addr - first_addr >= H264_AMD_MAX_SLICE_MBS))
			return H264_AMD_E_MB_OVERFLOW;

		/* ---- new macroblock: neighbours, clean record ---- */
		SD_T(0);
		uint32_t mb_field = 0, mb_top = 1;
		if (MBAFF) {
			/* mb_field_decoding_flag: with the top macroblock of a
			 * pair, or with the bottom one behind a skipped top (ref
			 * src/h264_syntax_slice_data.h:754-761) */
			uint32_t above_field;
			mb_top = (cur_addr & 1u) ^ 1u;
			if (mb_top || prev_mb_skipped)
				mb_field = vb_u(vb, wb, 1);
			else
				mb_field = P.pair_field;
			if (mb_top)
				P.top_skipped = 0;
			P.pair_field = mb_field;
			SD_PAIR_NEIGHBOURS();
			sd_mb_begin_mbaff(m, P, row, mbx, mb_top, mb_field,
					  above_field);
		}
		if (!MBAFF) {
		m.availA = (mbx != 0 && mbidx >= 1) ? 1u : 0u;
		m.availB = mbidx >= W ? 1u : 0u;
		}
		if (fmo && !MBAFF) {
			uint32_t tag = wb_uniform(*reinterpret_cast<const uint32_t *>(
				row + RS * mbx + 12u));
			m.availA = (mbx != 0 && prev_addr + 1u == cur_addr) ? 1u : 0u;
			m.availB = (cur_addr >= W && tag == cur_addr - W) ? 1u : 0u;
		}
		if (!MBAFF)
			sd_mb_begin(m, row, mbx, lane);
		/* clean record: the staging area is 4 KiB, zero what the
		 * largest record (1792 B) can touch */
		*reinterpret_cast<uint4 *>(stage + 16u * lane) = make_uint4(0, 0, 0, 0);
		if (o.rec_bytes > 1024u)
			*reinterpret_cast<uint4 *>(stage + 1024u + 16u * lane) =
				make_uint4(0, 0, 0, 0);

		/* ---- mb_type (ref src/h264_slice_data.c:839-969) ---- */
		/* ref_idx: present when there is more than one picture to choose
		 * from -- a field macroblock of an MBAFF frame always has (two
		 * fields per frame), with twice the range; max_ref_idx is a
		 * uint8_t in the reference (src/h264_macroblock.h:141-142,
		 * src/h264_slice_data.c:1199-1205) */
		const bool ref0_present = S.nref0 > 0 || (MBAFF && mb_field);
		const bool ref1_present = S.nref1 > 0 || (MBAFF && mb_field);
		const uint32_t ref0_range =
			((MBAFF && mb_field) ? 2u * S.nref0 + 1u : S.nref0) & 0xffu;
		const uint32_t ref1_range =
			((MBAFF && mb_field) ? 2u * S.nref1 + 1u : S.nref1) & 0xffu;
		uint32_t raw_type, t, mb_type = 0, nparts = 0;
		uint32_t pm0 = PM_I4x4, pm1 = PM_I4x4;
		uint32_t i16pred = 0, cbp_luma = 0, cbp_chroma = 0;
		bool intra = is_i; /* decode t with the I-slice table */
		VRD_UE(raw_type);
		t = raw_type;
		if (is_si) {
			if (t == 0) {
				mb_type = H264_MB_TYPE_SI;
				nparts = 1;
				pm0 = PM_I4x4;
				intra = false;
			} else {
				t -= 1;
			}
		} else if (is_p) {
			if (t <= 4) {
				mb_type = H264_MB_TYPE_P_16x16 + t;
				nparts = t == 0 ? 1u : t <= 2 ? 2u : 4u;
				pm0 = pm1 = nparts == 4 ? PM_I4x4 : PM_L0;
			} else {
				t -= 5;
				intra = true;
			}
		} else if (is_b) {
			if (t == 0) {
				mb_type = H264_MB_TYPE_B_Direct_16x16;
				nparts = 1;
				pm0 = PM_DIRECT;
			} else if (t <= 3) {
				mb_type = H264_MB_TYPE_B_16x16;
				nparts = 1;
				pm0 = PM_L0 + t - 1u;
			} else if (t <= 21) {
				/* Table 7-14: types come in (16x8, 8x16) pairs that
				 * share prediction modes; pair k = (t-4)/2:
				 * 0 L0L0 1 L1L1 2 L0L1 3 L1L0 4 L0Bi 5 L1Bi
				 * 6 BiL0 7 BiL1 8 BiBi; 2 bits each: 0 L0 1 L1 2 Bi */
				uint32_t k = (t - 4u) >> 1;
				mb_type = (t & 1u) ? H264_MB_TYPE_B_8x16
						   : H264_MB_TYPE_B_16x8;
				nparts = 2;
				pm0 = PM_L0 + ((0x2A444u >> (2u * k)) & 3u);
				pm1 = PM_L0 + ((0x24A14u >> (2u * k)) & 3u);
			} else if (t == 22) {
				mb_type = H264_MB_TYPE_B_8x8;
				nparts = 4;
			} else {
				t -= 23;
				intra = true;
			}
		}
		if (intra) {
			/* Table 7-11 from the table the lanes hold (lane = mb_type
			 * value): no chain of compares */
			if (SD_RARE(t > 25))
				return -EIO;
			uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)itab,
									  (int)t);
			mb_type = e & 15u;
			pm0 = (e >> 4) & 7u;
			nparts = (e >> 7) & 1u;
			i16pred = (e >> 8) & 3u;
			cbp_luma = (e >> 10) & 15u;
			cbp_chroma = (e >> 14) & 3u;
		}

		uint32_t flags = 0, icp = 0;
		int32_t qpd = 0;
		uint32_t pred_lo = 0, pred_hi = 0; /* pred[8] of the summary */
		SD_T(1);

		if (SD_RARE(mb_type == H264_MB_TYPE_I_PCM)) {
			/* pcm_alignment_zero_bit, then raw samples (:622-645) */
			vb_store(vb, wb);
			uint32_t p = wb_pos(wb) & 7u;
			if (SD_RARE(p && wb_u(wb, lane, 8u - p) != 0))
				return -EIO;
			/* 64 samples per step, one per lane: the sample counts (256
			 * luma, 0 / 64 / 128 / 256 per chroma plane) are multiples of
			 * 64, so a step is all luma or all chroma.  Lane k fetches the
			 * two window dwords its sample straddles with a cross-lane
			 * gather.  The window is 2048 bits, slides below bit 1024 and
			 * a step spans at most 64 * 14 bits. */
			const uint32_t total = 256u + 2u * S.pcm_chroma;
			for (uint32_t k0 = 0; k0 < total; k0 += 64u) {
				wb_sync(wb, lane);
				if (SD_RARE(wb_overrun(wb)))
					return -EIO;
				const uint32_t bd = k0 < 256u ? S.bdl : S.bdc;
				uint32_t off = wb.bitpos + lane * bd;
				uint32_t hi = (uint32_t)__shfl((int)wb.wreg, (int)(off >> 5));
				uint32_t lo = (uint32_t)__shfl((int)wb.wreg,
							       (int)((off >> 5) + 1u));
				uint64_t v = (((uint64_t)hi << 32) | lo) << (off & 31u);
				uint32_t smp = (uint32_t)(v >> 32) >> (32u - bd);
				*reinterpret_cast<uint16_t *>(
					stage + H264_AMD_T2_HEAD + 2u * (k0 + lane)) =
					(uint16_t)smp;
				wb_skip(wb, 64u * bd);
			}
			wb_sync(wb, lane);
			vb_load(vb, wb);
			m.cur = lane < 48 ? 16u : 0u;
		} else {
			bool no_sub8 = true;
			if (mb_type != H264_MB_TYPE_I_NxN && pm0 != PM_I16x16 &&
			    nparts == 4) {
				/* ---- sub_mb_pred (:422-503) ---- */
				uint32_t sub_pm[4], sub_n[4];
#pragma unroll
				for (uint32_t p = 0; p < 4; p++) {
					uint32_t st;
					VRD_UE(st);
					pred_lo |= (st & 0xffu) << (8u * p);
					if (is_b) {
						if (SD_RARE(st >= 13))
							return -EIO;
						/* Table 7-18 */
						sub_pm[p] = st == 0   ? (uint32_t)PM_DIRECT
							    : st <= 3 ? PM_L0 + st - 1u
							    : st <= 9 ? PM_L0 + ((st - 4u) >> 1)
								      : PM_L0 + st - 10u;
						sub_n[p] = st == 0   ? 4u
							   : st <= 3 ? 1u
							   : st <= 9 ? 2u
								     : 4u;
					} else {
						if (SD_RARE(st >= 4))
							return -EIO;
						sub_pm[p] = PM_L0;
						sub_n[p] = st == 0 ? 1u : st == 3 ? 4u : 2u;
					}
					if (sub_pm[p] != PM_DIRECT) {
						if (sub_n[p] > 1)
							no_sub8 = false;
					} else if (!S.direct8x8) {
						no_sub8 = false;
					}
				}
				if (ref0_present && mb_type != H264_MB_TYPE_P_8x8ref0) {
#pragma unroll
					for (uint32_t p = 0; p < 4; p++)
						if (sub_pm[p] != PM_DIRECT &&
						    sub_pm[p] != PM_L1) {
							uint32_t r = vb_te(vb, wb, ref0_range, ok);
							if (SD_RARE(!ok))
								return -EIO;
							lds_w8(stage, 64 + p, r, lane);
						}
				}
				if (ref1_present) {
#pragma unroll
					for (uint32_t p = 0; p < 4; p++)
						if (sub_pm[p] != PM_DIRECT &&
						    sub_pm[p] != PM_L0) {
							uint32_t r = vb_te(vb, wb, ref1_range, ok);
							if (SD_RARE(!ok))
								return -EIO;
							lds_w8(stage, 68 + p, r, lane);
						}
				}
#pragma unroll
				for (uint32_t l = 0; l < 2; l++) {
#pragma unroll
					for (uint32_t p = 0; p < 4; p++) {
						if (sub_pm[p] == PM_DIRECT ||
						    sub_pm[p] == (l ? PM_L0 : PM_L1))
							continue;
						sd_vb_sync(vb, wb, lane, S);
						for (uint32_t s = 0; s < sub_n[p]; s++)
							for (uint32_t c = 0; c < 2; c++) {
								int32_t d;
								VRD_SE(d);
								lds_w16(stage,
									128 + 64 * l + 16 * p + 4 * s + 2 * c,
									(uint32_t)d, lane);
							}
					}
				}
			} else {
				/* ---- mb_pred (:506-601) ---- */
				if (S.t8x8_mode && mb_type == H264_MB_TYPE_I_NxN) {
					if (vb_u(vb, wb, 1)) {
						flags |= H264_AMD_MB_TRANSFORM_8X8;
						pm0 = PM_I8x8;
					}
				}
				if (pm0 == PM_I4x4 || pm0 == PM_I8x8 || pm0 == PM_I16x16) {
					if (pm0 != PM_I16x16) {
						uint32_t n = pm0 == PM_I4x4 ? 16u : 4u;
						vb_norm(vb, wb);
						vb_skip(vb, sd_intra_pred_modes(vb, wb, lane,
										n, stage));
						/* up to 64 bits: two steps */
						vb_norm(vb, wb);
						vb_norm(vb, wb);
					}
					if (chroma12)
						VRD_UE(icp);
				} else if (pm0 != PM_DIRECT) {
					if (ref0_present) {
						for (uint32_t p = 0; p < nparts; p++)
							if ((p ? pm1 : pm0) != PM_L1) {
								uint32_t r = vb_te(vb, wb, ref0_range, ok);
								if (SD_RARE(!ok))
									return -EIO;
								lds_w8(stage, 64 + p, r, lane);
							}
					}
					if (ref1_present) {
						for (uint32_t p = 0; p < nparts; p++)
							if ((p ? pm1 : pm0) != PM_L0) {
								uint32_t r = vb_te(vb, wb, ref1_range, ok);
								if (SD_RARE(!ok))
									return -EIO;
								lds_w8(stage, 68 + p, r, lane);
							}
					}
					for (uint32_t l = 0; l < 2; l++)
						for (uint32_t p = 0; p < nparts; p++) {
							if ((p ? pm1 : pm0) == (l ? PM_L0 : PM_L1))
								continue;
							for (uint32_t c = 0; c < 2; c++) {
								int32_t d;
								VRD_SE(d);
								lds_w16(stage, 128 + 64 * l + 16 * p + 2 * c,
									(uint32_t)d, lane);
							}
						}
				}
			}
			SD_T(2);
			sd_vb_sync(vb, wb, lane, S);
			if (pm0 != PM_I16x16) {
				/* coded_block_pattern, me(v) mapping of Table 9-4
				 * (src/h264_slice_data.c:1041-1080); cbp_map lane =
				 * codeNum, bytes: cat 1/2 intra, inter, cat 0/3
				 * intra, inter */
				uint32_t code;
				VRD_UE(code);
				if (SD_RARE(cat > 3 || code >= (chroma12 ? 48u : 16u)))
					return -EIO;
				bool intra_col = mb_type == H264_MB_TYPE_I_NxN ||
						 mb_type == H264_MB_TYPE_SI;
				uint32_t ent = (uint32_t)__builtin_amdgcn_readlane(
					(int)cbp_map, (int)code);
				uint32_t cbp = (ent >> (8u * ((chroma12 ? 0u : 2u) +
							      (intra_col ? 0u : 1u)))) &
					       0xffu;
				cbp_luma = cbp & 15u;
				cbp_chroma = cbp >> 4;
				if (cbp_luma > 0 && S.t8x8_mode &&
				    mb_type != H264_MB_TYPE_I_NxN && no_sub8 &&
				    (mb_type != H264_MB_TYPE_B_Direct_16x16 ||
				     S.direct8x8)) {
					if (vb_u(vb, wb, 1))
						flags |= H264_AMD_MB_TRANSFORM_8X8;
				}
			}
			SD_T(3);
			if (SD_USUAL(cbp_luma > 0 || cbp_chroma > 0 || pm0 == PM_I16x16)) {
				VRD_SE(qpd);
				/* ---- residual (:334-419).  The coded blocks
				 * are the set bits of `todo` (bit = step, in
				 * bitstream order); blocks that are not coded cost
				 * nothing ---- */
				const bool i16 = pm0 == PM_I16x16;
				/* the block mask of this coded_block_pattern from
				 * the table the lanes hold (lane = pattern), the
				 * Intra16x16 DC block(s) on top */
				const uint32_t pat = cbp_luma | (cbp_chroma << 4);
				uint64_t todo =
					((uint64_t)(uint32_t)__builtin_amdgcn_readlane(
						 (int)todo_hi, (int)pat)
					 << 32) |
					(uint32_t)__builtin_amdgcn_readlane((int)todo_lo,
									    (int)pat);
				todo |= i16 ? dc_steps : 0ull;
				const uint32_t vinfo = i16 ? m.info_i16 : m.info_n;
				uint32_t serr = 0;
				SD_T(9);
				while (todo) {
					sd_vb_sync(vb, wb, lane, S);
					vb_norm(vb, wb);
					/* Runs of empty blocks.  For nC < 8 the code word
					 * for "no coefficient" is a string of 1, 2 or 4
					 * one bits (Table 9-5: nC < 2, < 4, < 8), and an
					 * empty block changes no neighbour sum, so nC of
					 * the blocks behind it is already known.  With
					 * the blocks still to come laid out over the
					 * lanes in bitstream order, a prefix sum of their
					 * code lengths against the number of leading
					 * one bits at the cursor gives the length of the
					 * run at once.  A block with a non-zero nz slot
					 * (a parked DC count), nC >= 8 or a chroma DC
					 * block ends the run. */
					uint32_t peek = vb_peek(vb);
					/* (one block in three starts with a 1 bit: no
					 * hint -- out of line this pass pays two taken
					 * branches, measured slower) */
					if (vb_true((int32_t)peek < 0)) {
						/* 1, 2 or 4 bits for nC < 2, < 4, < 8; anything
						 * above 32 stands for "ends the run" (nC >= 8,
						 * a parked count, a chroma DC block).  In
						 * arithmetic: three compare-select pairs are
						 * six instructions and as many wait states on
						 * VCC, which a lone wave pays in full. */
						uint32_t ncv = sd_nc_all(m);
						uint32_t k = ncv >> 1;
						k = k < 2u ? k : 2u;
						uint32_t lb = (1u << k) | ((ncv >> 3) << 6) |
							      ((m.cur + 63u) & 64u);
						uint32_t ls = (uint32_t)__shfl(
							(int)lb, (int)(vinfo & 63u));
						ls |= (vinfo >> 5) & 0xc0u; /* kind << 6 */
						uint32_t in_todo =
							(uint32_t)(todo >> lane) & 1u;
						uint32_t incl = sd_scan_add(in_todo ? ls : 0u);
						uint32_t ones = (uint32_t)__builtin_clz(
							~peek | 1u);
						/* (the lanes "in todo" are the bits of todo) */
						uint64_t fit = __ballot(incl <= ones) & todo;
						if (fit) {
							uint32_t last = 63u - (uint32_t)
								__builtin_clzll(fit);
							/* <= 31 bits */
							vb_skip(vb, (uint32_t)
								__builtin_amdgcn_readlane(
									(int)incl, (int)last));
							todo &= ~fit;
							SD_T(10);
							continue;
						}
					}
					uint32_t step = (uint32_t)__builtin_ctzll(todo);
					asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(step));
					uint32_t info = (uint32_t)__builtin_amdgcn_readlane(
						(int)vinfo, (int)step);
					uint32_t nzlane = info & 63u;
					/* table class from nC of the block's context
					 * lane (scalar side: the vector port is the
					 * fuller one); chroma DC blocks have their own */
					uint32_t cls = vlc_ct_class(sd_nc(m, info >> 24));
					{
						/* cls = ((info >> 11) & 3) ? 3 + cat : cls,
						 * on the condition code of the s_and (the
						 * backend adds a compare) */
						uint32_t dc = 3u + cat, t;
						asm("s_and_b32 %1, %2, 0x1800\n\t"
						    "s_cselect_b32 %0, %3, %0"
						    : "+s"(cls), "=&s"(t)
						    : "s"(info), "s"(dc)
						    : "scc");
					}
					SD_T(11);
					uint32_t tc = sd_residual_block(
						vb, wb, vt, lane, stage, cls, peek,
						(info >> 6) & 31u,
						(info >> 13) & 0x7ffu, serr SD_TPASS);
					/* TotalCoeff is recorded for every class, DC
					 * blocks included (Appendix C.4) */
					sd_nz_set(m, nzlane, tc, lane);
					SD_T(12);
				}
				if (SD_RARE(serr != 0))
					return -EIO;
			}
		}
		vb_store(vb, wb);
		if (SD_RARE(wb_overrun(wb)))
			return -EIO;
		SD_T(13);

		/* ---- summary (struct h264_amd_mb_summary) ---- */
		{
			uint32_t word0 = mb_type | ((cbp_luma | (cbp_chroma << 4)) << 8) |
					 (flags << 16) | ((raw_type & 0xffu) << 24);
			uint32_t word1 = ((uint32_t)qpd & 0xffffu) |
					 ((icp & 0xffu) << 16) | (i16pred << 24);
			if (MBAFF)
				word0 |= mb_field ? (uint32_t)H264_AMD_MB_FIELD << 16
						  : 0u;
			lds_w32(stage, 0, word0 | v_field, lane);
			lds_w32(stage, 4, word1, lane);
			/* intra prediction modes are already in place */
			if (pred_lo | pred_hi) {
				lds_w32(stage, 8, pred_lo, lane);
				lds_w32(stage, 12, pred_hi, lane);
			}
			*lds_sel(stage + 16u + lane, stage - SD_DUMP_BYTES, lane < 48,
				 lane) = (uint8_t)m.cur;
		}
		sd_emit_staged(o, stage, mbidx, lane);
		if (MBAFF)
			sd_next_mb_mbaff(m, P, row, stage - SD_DUMP_BYTES, mbx, mb_top,
					 mb_field, lane);
		else
			sd_next_mb(m, row, stage - SD_DUMP_BYTES, mbx, W, lane);
		SD_FMO_STEP();
		mbidx++;
		mb_count = mbidx;
		SD_T(14);
		SD_T(15);
	} while (wb_more_rbsp_data(wb, lane));
#undef SD_FMO_STEP
#undef SD_PAIR_NEIGHBOURS
#undef VRD_UE
#undef VRD_SE
	return 0;
}

/* ---- kernel ---------------------------------------------------------------- */

struct sd_args {
	const uint8_t *buf;
	const struct h264_amd_slice_desc *descs;
	uint32_t n;
	const dev_sps *spsv;
	const dev_pps *ppsv;
	const dev_epoch *epochs;
	uint32_t n_sps, n_pps, n_epochs;
	const dev_vlc_tables *vlc;
	uint32_t reader_flags, tier, mb_capacity, mb_stride;
	struct h264_amd_slice_result *results;
	struct h264_slice_header *headers;
	uint8_t *mb_out;
	uint64_t buf_len; /* bytes behind buf, 0 = not checked */
	/* slice groups: per PPS version the map parameters, and room for one
	 * macroblock-to-slice-group map per slice (fmo_stride bytes each) */
	const dev_fmo *fmov;
	uint8_t *fmo_map;
	uint32_t fmo_stride;
	uint32_t row_pitch; /* macroblock columns of row context in LDS */
	uint32_t row_mbaff; /* 1: 32 bytes per column (MBAFF frames possible) */
	/* LDS of one wave: SD_STAGE_OFF + max(SD_STAGE_BYTES, SD_REC_BYTES +
	 * (16 or 32) * row_pitch) */
	uint32_t wave_lds_bytes;
	/* workgroups per dispatch layer (the number of compute units): the
	 * age rank of a wave on its SIMD is blockIdx.x / prio_div; 0: the launch
	 * fits one layer, no priority rotation */
	uint32_t prio_div;
	/* launch position -> slice, longest first (h264_amd_k_order_slices);
	 * NULL: position = slice */
	const uint32_t *order;
};

/* one slice NAL unit, start to finish, by this wavefront */
template <bool WITH_MBAFF>
__device__ __forceinline__ void sd_one_slice(const sd_args &a, const uint32_t slice,
					     uint8_t *lds_all, uint8_t *lds,
					     const uint32_t lane)
{
	uint8_t *ring = lds + SD_RING_OFF;
	uint8_t *stage = lds + SD_STAGE_OFF;
	uint8_t *row = lds + SD_ROW_OFF;

	struct h264_amd_slice_desc d;
	sd_load_uniform(&d, a.descs + slice);
	/* a descriptor reaching past the buffer is answered, not followed */
	const bool bad_desc = a.buf_len != 0 &&
			      (d.offset > a.buf_len || d.size > a.buf_len - d.offset);
	const uint32_t len = bad_desc ? 0u : d.size;
	const uint8_t *nal = a.buf + (bad_desc ? 0u : d.offset);

	SliceState S;
	S.cat = 1;
	S.W = 1;
	S.bdl = S.bdc = 8;
	S.pcm_chroma = 0;
	S.t8x8_mode = S.direct8x8 = S.entropy = S.unsupported = 0;
	S.mbaff = 0;
	/* a launch of one workgroup per compute unit or less has every wave
	 * alone on its SIMD: nothing to share (and stepping through the low
	 * priorities there costs 2 %: the scalar unit and LDS are shared with
	 * the other SIMDs' waves) */
	S.prio_slot = a.prio_div ? (blockIdx.x / a.prio_div) & 3u : 0u;
	S.prio_on = a.prio_div ? 1u : 0u;
	S.fmo = S.fmo_type = S.pps_version = S.frame_mbs_only = 0;
	S.map_units_h = S.change_cycle = 0;
	S.change_rate = 1;
	S.slice_type = 0;
	S.first_mb = S.nref0 = S.nref1 = S.field_pic = 0;
	S.slice_qp = 0;
	S.nal_ref_idc = S.nal_unit_type = 0;
	S.hdr_len_bits = S.rawdata_off = S.rawdata_len = S.partial = S.partialbits = 0;

	int status = 0;
	uint32_t rflags = 0, mb_count = 0, end_bit = 0;
#ifdef SD_PHASE_TIMERS
	SdTimers T;
	T.acc = T.cnt = T.last = 0;
	T.t0 = T.r0 = 0;
#endif
	OutCtx o;
	o.tier = a.tier;
	o.mb_stride = a.mb_stride;
	o.rec_bytes = 1024;
	o.mb_out = a.mb_out ? a.mb_out + (size_t)slice * a.mb_capacity * a.mb_stride
			    : nullptr;
	if (o.mb_out == nullptr)
		o.tier = H264_AMD_TIER_NONE;
	o.t0acc = 0;
	o.lo = o.hi = 0;

	/* header image starts as zeros (the reference memset()s it,
	 * src/h264_syntax.h:1408-1410) */
	for (uint32_t part = 0; part < SD_STAGE_BYTES / 1024u; part++)
		*reinterpret_cast<uint4 *>(stage + part * 1024u + 16u * lane) =
			make_uint4(0, 0, 0, 0);

	WaveBits wb;
	wb_open(wb, nal, len, ring, reinterpret_cast<uint32_t *>(lds + SD_STATE_OFF), lane);

	/* NAL unit header (ref src/h264_syntax.h:1434-1443) */
	if (len == 0) {
		status = bad_desc ? H264_AMD_E_BAD_DESC : -EIO;
	} else {
		uint32_t h = wb_u(wb, lane, 8);
		if (h & 0x80u)
			status = -EIO;
		S.nal_ref_idc = (h >> 5) & 3u;
		S.nal_unit_type = h & 31u;
		if (status == 0 && S.nal_unit_type != 1 && S.nal_unit_type != 5)
			status = -EINVAL; /* not a slice: caller's mistake */
	}

	if (status == 0) {
		uint32_t e = d.epoch < a.n_epochs ? d.epoch : 0u;
		status = sd_slice_header(wb, lane, stage, S, a.spsv, a.ppsv,
					 a.epochs + e, a.n_sps, a.n_pps);
		if (status == 0 && wb_overrun(wb))
			status = -EIO;
	}
	if (status == 0) {
		/* ctx->slice.hdr_len and rawdata (src/h264_syntax.h:1383,
		 * src/h264_syntax_slice_data.h:803-806) */
		uint32_t P = wb_pos(wb);
		uint32_t k = (P + 7u) / 8u - 1u; /* last RBSP byte touched */
		uint32_t raw_after = wb_raw_of_clean(wb, k, lane) + 1u;
		S.partialbits = (8u - (P & 7u)) & 7u;
		/* bs->cache holds the whole last byte (it may lie just before
		 * the window, but the ring still has it) */
		S.partial = wb_uniform(ring[k & (WB_RING_BYTES - 1)]);
		S.hdr_len_bits = 8u * raw_after - S.partialbits;
		S.rawdata_off = raw_after;
		S.rawdata_len = len - raw_after;
		rflags |= H264_AMD_SR_HEADER_OK;
		if (S.entropy)
			rflags |= H264_AMD_SR_CABAC;
		end_bit = P;
		o.rec_bytes = sd_t2_stride(S.cat);
		if (a.headers) {
			uint8_t *dst = reinterpret_cast<uint8_t *>(a.headers + slice);
			const uint32_t hb = (uint32_t)sizeof(struct h264_slice_header);
			for (uint32_t part = 0; part * 1024u < hb; part++) {
				uint32_t want = part * 1024u + 16u * lane;
				uint32_t off = want < hb ? want : hb - 16u;
				*reinterpret_cast<uint4 *>(dst + off) =
					*reinterpret_cast<const uint4 *>(stage + off);
			}
		}
		if ((a.reader_flags & H264_READER_FLAGS_SLICE_DATA) && !S.entropy) {
			if (S.unsupported) {
				status = H264_AMD_E_UNSUPPORTED;
			} else if (S.W == 0 || S.W > a.row_pitch ||
				   (S.mbaff && !a.row_mbaff)) {
				status = H264_AMD_E_UNSUPPORTED;
			} else {
				/* slice groups: this slice's macroblock-to-slice-
				 * group map first (the header image has left the
				 * staging area, which the box-out walk borrows) */
				uint8_t *fmo_map = nullptr;
				uint32_t pic_mbs = 0;
				if (S.fmo) {
					if (a.fmo_map == nullptr || a.fmov == nullptr) {
						status = H264_AMD_E_UNSUPPORTED;
					} else {
						fmo_map = a.fmo_map +
							  (size_t)slice * a.fmo_stride;
						status = sd_fmo_build(
							a.fmov + S.pps_version, S, stage,
							fmo_map, a.fmo_stride, lane,
							pic_mbs);
						/* the map is read back by this wave:
						 * stores out, L1 refreshed */
						__threadfence();
					}
				}
				if (status == 0) {
					rflags |= H264_AMD_SR_DATA_PARSED;
					VlcTabs vt;
					vlc_open(vt, lds_all, lane);
					uint32_t cbp_map = a.vlc->cbp[lane];
#ifdef SD_PHASE_TIMERS
					T.last = (uint32_t)__builtin_readcyclecounter();
					T.t0 = __builtin_readcyclecounter();
					T.r0 = __builtin_amdgcn_s_memrealtime();
#endif
					if (S.mbaff) {
						/* the engine launches the pair-aware
						 * build whenever an SPS allows MBAFF */
						if constexpr (WITH_MBAFF)
							status = sd_slice_data<false, true>(
								wb, vt, cbp_map, lane, stage,
								row, S, o, a.mb_capacity,
								mb_count, fmo_map, pic_mbs
									SD_TPASS);
						else
							status = -ENOSYS;
					} else if (S.slice_type == H264_SLICE_TYPE_I &&
						 !S.fmo)
						status = sd_slice_data<true, false>(
							wb, vt, cbp_map, lane, stage, row,
							S, o, a.mb_capacity, mb_count,
							nullptr, 0 SD_TPASS);
					else
						status = sd_slice_data<false, false>(
							wb, vt, cbp_map, lane, stage, row,
							S, o, a.mb_capacity, mb_count,
							fmo_map, pic_mbs SD_TPASS);
					end_bit = wb_pos(wb);
				}
			}
		}
	}

	/* T0: flush the partial last group of 64 words */
	if (o.tier == H264_AMD_TIER_T0 && (mb_count & 63u) != 0) {
		uint32_t base = mb_count & ~63u;
		uint32_t l = lane < (mb_count & 63u) ? lane : (mb_count & 63u) - 1u;
		uint32_t v = (uint32_t)__shfl((int)o.t0acc, (int)l);
		*reinterpret_cast<uint32_t *>(o.mb_out + (size_t)(base + l) * o.mb_stride) = v;
	}

#ifdef SD_PHASE_TIMERS
	if (o.tier == H264_AMD_TIER_T0 && mb_count >= 64u) {
		if (lane < 16u) {
			*reinterpret_cast<uint32_t *>(o.mb_out + (size_t)lane * o.mb_stride) = T.acc;
			*reinterpret_cast<uint32_t *>(o.mb_out + (size_t)(16u + lane) * o.mb_stride) = T.cnt;
		}
		/* words 32 / 33: the wave's shader-clock ticks and 100 MHz ticks
		 * over the slice data: the clock it really ran at */
		uint32_t dt = (uint32_t)(__builtin_readcyclecounter() - T.t0);
		uint32_t dr = (uint32_t)(__builtin_amdgcn_s_memrealtime() - T.r0);
		if (lane < 3u)
			*reinterpret_cast<uint32_t *>(o.mb_out + (size_t)(32u + lane) * o.mb_stride) =
				lane == 0 ? dt : lane == 1 ? dr : (uint32_t)T.r0;
	}
#endif

	/* wave reduction of the checksum halves */
	uint32_t lo = o.lo, hi = o.hi;
#pragma unroll
	for (int sft = 32; sft >= 1; sft >>= 1) {
		lo += (uint32_t)__shfl_xor((int)lo, sft);
		hi += (uint32_t)__shfl_xor((int)hi, sft);
	}

	{
		struct h264_amd_slice_result r;
		r.status = status;
		r.flags = rflags;
		r.nal_ref_idc = S.nal_ref_idc;
		r.nal_unit_type = S.nal_unit_type;
		r.slice_type = (rflags & H264_AMD_SR_HEADER_OK) ? (int32_t)S.slice_type : -1;
		r.first_mb_in_slice = (rflags & H264_AMD_SR_HEADER_OK) ? S.first_mb : 0u;
		r.mb_count = mb_count;
		r.slice_qp = (rflags & H264_AMD_SR_HEADER_OK) ? S.slice_qp : 0;
		r.hdr_len_bits = S.hdr_len_bits;
		r.rawdata_off = S.rawdata_off;
		r.rawdata_len = S.rawdata_len;
		r.partial = (uint8_t)S.partial;
		r.partialbits = (uint8_t)S.partialbits;
		r.chroma_array_type =
			(rflags & H264_AMD_SR_HEADER_OK) ? (uint16_t)S.cat : (uint16_t)0;
		r.end_bit = status == 0 ? end_bit : 0u;
		r.pic_width_in_mbs = (rflags & H264_AMD_SR_HEADER_OK) ? S.W : 0u;
		r.checksum = ((uint64_t)hi << 32) | lo;
		/* every lane holds the same record: lane k stores dword k & 15 */
		const uint32_t *src = reinterpret_cast<const uint32_t *>(&r);
		uint32_t *dst = reinterpret_cast<uint32_t *>(a.results + slice);
		uint32_t k = lane & 15u, v = src[0];
#pragma unroll
		for (uint32_t j = 1; j < 16; j++)
			v = k == j ? src[j] : v;
		dst[k] = v;
	}
}

/* Work distribution inside a launch (hard part 5): the
 * hardware's workgroup dispatcher is the queue.  Workgroups start in index order
 * as wave slots come free; position q of the launch stands for slice order[q],
 * LONGEST FIRST (h264_amd_k_order_slices on the device: no caller has to sort),
 * so the four slices of a workgroup are of a size and end together, and what
 * starts last is what is shortest.  Results go out under the slice's own index.
 * A loop per wave over an atomic counter -- the persistent-wave form of the
 * same idea -- was built and measured: bit-exact, but the loop around the
 * slice body costs the register allocator 104 spilled vector registers (0
 * without it) and the bench batch 54.5 -> 61.8 ms, 5632 slices 101 -> 148 ms
 * (profiles/r03_decode_ab_history.jsonl); laundering the table pointers against
 * hoisting, -disable-machine-licm and an out-of-line slice function did not
 * bring them back. */
template <bool WITH_MBAFF>
__device__ __forceinline__ void sd_kernel_body(const sd_args &a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
	vlc_stage_tables(lds_all, a.vlc, threadIdx.x, 64u * SD_WAVES_PER_BLOCK);
	__syncthreads();
	/* from here on the waves of the workgroup are independent */
	const uint32_t wave = wb_uniform(threadIdx.x >> 6);
	uint8_t *lds = lds_all + SD_TABS_BYTES + wave * a.wave_lds_bytes;
	const uint32_t lane = wb_lane();
	const uint32_t q = blockIdx.x * SD_WAVES_PER_BLOCK + wave;
	if (q >= a.n)
		return;
	const uint32_t slice = a.order ? wb_uniform(a.order[q]) : q;
	if (slice < a.n)
		sd_one_slice<WITH_MBAFF>(a, slice, lds_all, lds, lane);
}

/* Longest-first order of a batch: slices bucketed by size (5 bits of exponent, 5
 * of mantissa: 3 % resolution), buckets laid out from the largest down; inside
 * a bucket slices keep no particular order.  One workgroup. */
__device__ __forceinline__ uint32_t sd_size_bucket(uint32_t size)
{
	if (size < 32u)
		return size;
	const uint32_t e = 31u - (uint32_t)__builtin_clz(size);
	return (e - 4u) * 32u + ((size >> (e - 5u)) & 31u);
}

extern "C" __global__ void __launch_bounds__(1024)
h264_amd_k_order_slices(const struct h264_amd_slice_desc *descs, uint32_t n,
			uint32_t *order)
{
	__shared__ uint32_t hist[1024], scan[1024];
	const uint32_t t = threadIdx.x;
	hist[t] = 0;
	__syncthreads();
	for (uint32_t i = t; i < n; i += 1024u)
		atomicAdd(&hist[sd_size_bucket(descs[i].size)], 1u);
	__syncthreads();
	/* scan[t] = slices in buckets above 1023 - t ... i.e. a prefix sum over
	 * the buckets in descending order */
	scan[t] = hist[1023u - t];
	__syncthreads();
	for (uint32_t d = 1; d < 1024u; d <<= 1) {
		uint32_t v = t >= d ? scan[t - d] : 0u;
		__syncthreads();
		scan[t] += v;
		__syncthreads();
	}
	/* first position of bucket b: the sum over the buckets above it */
	hist[1023u - t] = scan[t] - hist[1023u - t];
	__syncthreads();
	for (uint32_t i = t; i < n; i += 1024u) {
		uint32_t pos = atomicAdd(&hist[sd_size_bucket(descs[i].size)], 1u);
		order[pos] = i;
	}
}

/* 128 registers, 4 waves per SIMD: the fastest wave; batches that fill the
 * chip once (up to 4096 slices) */
extern "C" __global__ void __launch_bounds__(64 * SD_WAVES_PER_BLOCK, 4)
h264_amd_k_decode_slices(sd_args a)
{
	sd_kernel_body<false>(a);
}

/* 96 registers, 5 waves per SIMD: more waves to hide one another's latency,
 * for batches of more than one round */
extern "C" __global__ void __launch_bounds__(64 * SD_WAVES_PER_BLOCK, 5)
h264_amd_k_decode_slices_w5(sd_args a)
{
	sd_kernel_body<false>(a);
}

/* the build that also holds the macroblock-pair walk (MBAFF): launched only
 * when a parameter set allows it, so that its registers and code are not
 * carried by the two above */
extern "C" __global__ void __launch_bounds__(64 * SD_WAVES_PER_BLOCK, 4)
h264_amd_k_decode_slices_mbaff(sd_args a)
{
	sd_kernel_body<true>(a);
}

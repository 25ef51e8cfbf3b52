/**
 * slice_write.hip -- device-side serialisation of slice NAL units for gfx950.
 *
 * GPU counterparts of the reference's writer entry points, one slice per
 * wavefront:
 *   h264_amd_k_write_slices   <- h264_write_nalu for nal_unit_type 1/5
 *        NAL header + slice header re-encoded field by field
 *            (ref src/h264_syntax.h:1014-1387,1434-1443, WRITE instantiation
 *             src/h264_writer.c:30-46),
 *        then the saved partial byte and the raw slice-data bytes verbatim
 *            (ref src/h264_syntax_slice_data.h:813-830,
 *             src/h264_bitstream.c:428-440) -- an HBM-to-HBM copy.
 *   h264_amd_k_encode_synth   <- h264_write_grey_i_slice /
 *        h264_write_skipped_p_slice, CAVLC and CABAC
 *            (ref src/h264_writer.c:49-309, src/h264_bac.c:150-358,
 *             src/h264_cabac.c:61-220,230-470,568-975).
 *   h264_amd_k_cabac_init     <- h264_cabac_init_states
 *            (ref src/h264_cabac_ctx_tables.c:4441-4468, src/h264_bac.c:216-230)
 *
 * The bit writer mirrors h264_bs_write_bits / h264_bs_flush
 * (ref src/h264_bitstream.c:54-81,211-239): a completed byte is preceded by
 * 0x03 when the two previous OUTPUT bytes are zero and the byte is <= 3.  Its
 * state is wave-uniform (SGPRs); bytes are staged in LDS and leave for HBM in
 * coalesced 256-byte rows.  As in slice_decode.hip the code contains no
 * divergent branches.
 */
#ifndef H264_AMD_SLICE_WRITE_HIP
#define H264_AMD_SLICE_WRITE_HIP

#include <hip/hip_runtime.h>
#include <errno.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <h264_amd.h>

#include "dev_params.h"
#include "wave_bits.hip.h"

/* CABAC constants: Table 9-44 / 9-45 and the (m, n) pairs of Tables 9-12..33,
 * [ctxIdx][I, P idc0, idc1, idc2][m, n] */
struct cabac_tables {
	uint8_t range_lps[64][4];
	uint8_t next_lps[64];
	uint8_t next_mps[64];
	int8_t mn[1024][4][2];
};

static void cabac_build_tables(cabac_tables *t)
{
	memset(t, 0, sizeof(*t));
#define H264_LPS(i, a, b, c, d)                                                \
	t->range_lps[i][0] = a;                                                \
	t->range_lps[i][1] = b;                                                \
	t->range_lps[i][2] = c;                                                \
	t->range_lps[i][3] = d;
#define H264_TRANS(i, l, m)                                                    \
	t->next_lps[i] = l;                                                    \
	t->next_mps[i] = m;
#define H264_MN(i, mi, ni, m0, n0, m1, n1, m2, n2)                             \
	t->mn[i][0][0] = mi;                                                   \
	t->mn[i][0][1] = ni;                                                   \
	t->mn[i][1][0] = m0;                                                   \
	t->mn[i][1][1] = n0;                                                   \
	t->mn[i][2][0] = m1;                                                   \
	t->mn[i][2][1] = n1;                                                   \
	t->mn[i][3][0] = m2;                                                   \
	t->mn[i][3][1] = n2;
#include "../common/h264_cabac_tables.inc"
#undef H264_LPS
#undef H264_TRANS
#undef H264_MN
}

/* ---- bit writer ------------------------------------------------------------- */

#define SW_STAGE_BYTES 1024u /* LDS ring of output bytes, per wave */

struct BitW {
	uint8_t *stage; /* LDS */
	uint8_t *dst; /* global, 16-byte aligned start of this NAL */
	uint32_t cap; /* bytes available at dst */
	uint32_t acc, nbits; /* partially filled byte */
	uint32_t pos; /* bytes produced */
	uint32_t flushed; /* bytes already in global memory (multiple of 256) */
	uint32_t zeros; /* trailing zero OUTPUT bytes, saturating at 2 */
	uint32_t escape; /* insert emulation-prevention bytes */
	uint32_t overflow;
};

/* wflags: H264_AMD_W_* of <h264_amd.h> */
__device__ __forceinline__ void bw_open(BitW &w, uint8_t *stage, uint8_t *dst,
					uint32_t cap, uint32_t wflags)
{
	w.stage = stage;
	w.dst = dst;
	w.cap = cap;
	w.acc = w.nbits = w.pos = w.flushed = w.overflow = 0;
	w.zeros = (wflags & 3u) > 2u ? 2u : (wflags & 3u);
	w.escape = (wflags & H264_AMD_W_NO_ESCAPE) ? 0u : 1u;
}

/* one 256-byte row from the LDS ring to HBM; out-of-capacity lanes repeat
 * lane 0's store */
__device__ __forceinline__ void bw_flush_row(BitW &w, uint32_t lane)
{
	uint32_t off = w.flushed + 4u * lane;
	bool ok = off + 4u <= w.cap;
	uint32_t o = ok ? off : w.flushed;
	uint32_t v = *reinterpret_cast<const uint32_t *>(
		w.stage + (o & (SW_STAGE_BYTES - 1)));
	if (w.flushed + 4u <= w.cap)
		*reinterpret_cast<uint32_t *>(w.dst + o) = v;
	w.flushed += 256u;
}

__device__ __forceinline__ void bw_emit(BitW &w, uint32_t byte, uint32_t lane)
{
	/* every lane stores the same byte to the same address */
	w.stage[w.pos & (SW_STAGE_BYTES - 1)] = (uint8_t)byte;
	w.pos++;
	if (w.pos > w.cap)
		w.overflow = 1;
	if (w.pos - w.flushed >= 512u)
		bw_flush_row(w, lane);
}

/* completed byte leaves the cache (ref src/h264_bitstream.c:54-81) */
__device__ __forceinline__ void bw_byte(BitW &w, uint32_t byte, uint32_t lane)
{
	if (w.escape && w.zeros >= 2 && byte <= 3) {
		bw_emit(w, 3, lane);
		w.zeros = 0;
	}
	bw_emit(w, byte, lane);
	w.zeros = byte == 0 ? (w.zeros < 2 ? w.zeros + 1 : 2) : 0;
}

/* the low n bits of v, MSB first, 0 <= n <= 32 */
__device__ __forceinline__ void bw_u(BitW &w, uint32_t v, uint32_t n,
				     uint32_t lane)
{
	while (n > 0) {
		uint32_t k = 8u - w.nbits;
		k = k < n ? k : n;
		w.acc = (w.acc << k) | ((v >> (n - k)) & ((1u << k) - 1u));
		w.nbits += k;
		n -= k;
		if (w.nbits == 8) {
			bw_byte(w, w.acc & 0xffu, lane);
			w.acc = 0;
			w.nbits = 0;
		}
	}
}

/* u(n) for any n, as the reference's writer behaves for a width taken from a
 * damaged SPS (src/h264_bitstream.c:210-240): the value is a 32-bit field
 * widened to 64 bits, so bits above 32 are zeros; n > 64 is refused (-EINVAL). */
__device__ __forceinline__ int bw_u_any(BitW &w, uint32_t v, uint32_t n,
					uint32_t lane)
{
	if (n > 64)
		return -EINVAL;
	if (n > 32) {
		bw_u(w, 0, n - 32u, lane);
		n = 32;
	}
	bw_u(w, v, n, lane);
	return 0;
}

/* ue(v) as the reference writes it (ref src/h264_bitstream.c:245-287): v + 1
 * on 2 * bits - 1 bits; v = 0xffffffff wraps to a single 0 bit there too */
__device__ __forceinline__ void bw_ue(BitW &w, uint32_t v, uint32_t lane)
{
	uint32_t x = v + 1u;
	uint32_t n = x ? 32u - (uint32_t)__builtin_clz(x) : 1u;
	if (n > 1)
		bw_u(w, 0, n - 1u, lane);
	bw_u(w, x, n, lane);
}

__device__ __forceinline__ void bw_se(BitW &w, int32_t v, uint32_t lane)
{
	bw_ue(w, v <= 0 ? (uint32_t)(-2 * v) : (uint32_t)(2 * v - 1), lane);
}

__device__ __forceinline__ void bw_trailing(BitW &w, uint32_t lane)
{
	bw_u(w, 1, 1, lane);
	if (w.nbits)
		bw_u(w, 0, 8u - w.nbits, lane);
}

/* write what is still in the ring; bytes, not rows */
__device__ __forceinline__ void bw_drain(BitW &w, uint32_t lane)
{
	while (w.pos - w.flushed >= 256u)
		bw_flush_row(w, lane);
	uint32_t left = w.pos - w.flushed; /* < 256 */
	for (uint32_t base = 0; base < left; base += 64u) {
		uint32_t i = base + lane;
		/* spare lanes repeat the row's first byte (same address, data) */
		uint32_t j = (i < left && w.flushed + i < w.cap) ? i : base;
		uint32_t o = w.flushed + j;
		uint8_t b = w.stage[o & (SW_STAGE_BYTES - 1)];
		if (w.flushed + base < w.cap)
			w.dst[o] = b;
	}
	w.flushed = w.pos;
}

/* ---- slice header, WRITE direction ---------------------------------------- */

#define WOFF(f) ((uint32_t)offsetof(struct h264_slice_header, f))

__device__ __forceinline__ uint32_t hg(const uint8_t *hdr, uint32_t off)
{
	return wb_uniform(*reinterpret_cast<const uint32_t *>(hdr + off));
}
#define HG(f) hg(hdr, WOFF(f))

template <typename T>
__device__ __forceinline__ void sw_load_uniform(T *dst, const T *src)
{
	const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
	uint32_t *d = reinterpret_cast<uint32_t *>(dst);
#pragma unroll
	for (uint32_t i = 0; i < sizeof(T) / 4; i++)
		d[i] = wb_uniform(s[i]);
}

__device__ __forceinline__ uint32_t sw_intlog2(uint32_t x)
{
	uint32_t r = x ? 32u - (uint32_t)__builtin_clz(x) : 0u;
	if (r > 0 && x == (1u << (r - 1u)))
		r--;
	return r;
}

struct SwParams {
	dev_sps sps;
	dev_pps pps;
};

/* h264_ctx_set_active_pps for the writer; 0 or -EINVAL */
__device__ __forceinline__ int sw_activate(SwParams &P, uint32_t pps_id,
					   const dev_sps *spsv,
					   const dev_pps *ppsv, const dev_epoch *ep,
					   uint32_t n_sps, uint32_t n_pps)
{
	if (pps_id >= 256)
		return -EINVAL;
	uint32_t pv = wb_uniform(ep->pps_ver[pps_id]);
	if (pv == DEV_NO_VERSION || pv >= n_pps)
		return -EINVAL;
	sw_load_uniform(&P.pps, ppsv + pv);
	if (P.pps.seq_parameter_set_id >= 32)
		return -EINVAL;
	uint32_t sv = wb_uniform(ep->sps_ver[P.pps.seq_parameter_set_id]);
	if (sv == DEV_NO_VERSION || sv >= n_sps)
		return -EINVAL;
	sw_load_uniform(&P.sps, spsv + sv);
	return 0;
}

/* NAL header + slice header.  hdr: LDS image of struct h264_slice_header. */
__device__ __forceinline__ int sw_slice_header(BitW &w, const uint8_t *hdr,
					       const SwParams &P,
					       uint32_t nal_ref_idc,
					       uint32_t nal_unit_type,
					       uint32_t lane)
{
	const dev_sps &sps = P.sps;
	const dev_pps &pps = P.pps;
	bw_u(w, 0, 1, lane);
	bw_u(w, nal_ref_idc, 2, lane);
	bw_u(w, nal_unit_type, 5, lane);

	uint32_t raw_type = HG(slice_type), type = raw_type % 5u;
	bw_ue(w, HG(first_mb_in_slice), lane);
	bw_ue(w, raw_type, lane);
	bw_ue(w, HG(pic_parameter_set_id), lane);
	if (sps.separate_colour_plane_flag)
		bw_u(w, HG(colour_plane_id), 2, lane);
	if (bw_u_any(w, HG(frame_num), sps.log2_max_frame_num, lane))
		return -EINVAL;
	uint32_t field_pic = 0;
	if (!sps.frame_mbs_only_flag) {
		field_pic = HG(field_pic_flag);
		bw_u(w, field_pic, 1, lane);
		if (field_pic)
			bw_u(w, HG(bottom_field_flag), 1, lane);
	}
	if (nal_unit_type == 5)
		bw_ue(w, HG(idr_pic_id), lane);
	if (sps.pic_order_cnt_type == 0) {
		if (bw_u_any(w, HG(pic_order_cnt_lsb), sps.log2_max_pic_order_cnt_lsb,
			     lane))
			return -EINVAL;
		if (pps.bottom_field_pic_order_in_frame_present_flag && !field_pic)
			bw_se(w, (int32_t)HG(delta_pic_order_cnt_bottom), lane);
	}
	if (sps.pic_order_cnt_type == 1 && !sps.delta_pic_order_always_zero_flag) {
		bw_se(w, (int32_t)HG(delta_pic_order_cnt[0]), lane);
		if (pps.bottom_field_pic_order_in_frame_present_flag && !field_pic)
			bw_se(w, (int32_t)HG(delta_pic_order_cnt[1]), lane);
	}
	if (pps.redundant_pic_cnt_present_flag)
		bw_ue(w, HG(redundant_pic_cnt), lane);
	if (type == H264_SLICE_TYPE_B)
		bw_u(w, HG(direct_spatial_mv_pred_flag), 1, lane);
	uint32_t nref0 = HG(num_ref_idx_l0_active_minus1);
	uint32_t nref1 = HG(num_ref_idx_l1_active_minus1);
	if (type == H264_SLICE_TYPE_P || type == H264_SLICE_TYPE_SP ||
	    type == H264_SLICE_TYPE_B) {
		uint32_t ov = HG(num_ref_idx_active_override_flag);
		bw_u(w, ov, 1, lane);
		if (ov) {
			bw_ue(w, nref0, lane);
			if (type == H264_SLICE_TYPE_B)
				bw_ue(w, nref1, lane);
		}
	}
	/* ref_pic_list_modification */
	for (uint32_t l = 0; l < 2; l++) {
		if (l == 0 ? (type == H264_SLICE_TYPE_I || type == H264_SLICE_TYPE_SI)
			   : type != H264_SLICE_TYPE_B)
			continue;
		uint32_t fo = l ? WOFF(rplm.ref_pic_list_modification_flag_l1)
				: WOFF(rplm.ref_pic_list_modification_flag_l0);
		uint32_t io = l ? WOFF(rplm.pic_num_l1) : WOFF(rplm.pic_num_l0);
		uint32_t f = hg(hdr, fo);
		bw_u(w, f, 1, lane);
		if (f) {
			uint32_t i = 0, val;
			do {
				if (i >= 32)
					return -EIO;
				val = hg(hdr, io + 8u * i);
				bw_ue(w, val, lane);
				if (val == 0 || val == 1 || val == 2 || val == 4 ||
				    val == 5)
					bw_ue(w, hg(hdr, io + 8u * i + 4u), lane);
				i++;
			} while (val != 3);
		}
	}
	/* pred_weight_table */
	if ((pps.weighted_pred_flag &&
	     (type == H264_SLICE_TYPE_P || type == H264_SLICE_TYPE_SP)) ||
	    (pps.weighted_bipred_idc == 1 && type == H264_SLICE_TYPE_B)) {
		uint32_t cat = sps.separate_colour_plane_flag ? 0u
							       : sps.chroma_format_idc;
		bw_ue(w, HG(pwt.luma_log2_weight_denom), lane);
		if (cat != 0)
			bw_ue(w, HG(pwt.chroma_log2_weight_denom), lane);
		for (uint32_t l = 0; l < (type == H264_SLICE_TYPE_B ? 2u : 1u); l++) {
			uint32_t n = l ? nref1 : nref0;
			uint32_t base = l ? WOFF(pwt.l1) : WOFF(pwt.l0);
			if (n > 32)
				return -EIO;
			for (uint32_t i = 0; i <= n; i++) {
				uint32_t o = base + 32u * i;
				uint32_t f = hg(hdr, o);
				bw_u(w, f, 1, lane);
				if (f) {
					bw_se(w, (int32_t)hg(hdr, o + 4), lane);
					bw_se(w, (int32_t)hg(hdr, o + 8), lane);
				}
				if (cat != 0) {
					f = hg(hdr, o + 12);
					bw_u(w, f, 1, lane);
					if (f) {
						bw_se(w, (int32_t)hg(hdr, o + 16), lane);
						bw_se(w, (int32_t)hg(hdr, o + 24), lane);
						bw_se(w, (int32_t)hg(hdr, o + 20), lane);
						bw_se(w, (int32_t)hg(hdr, o + 28), lane);
					}
				}
			}
		}
	}
	/* dec_ref_pic_marking */
	if (nal_ref_idc != 0) {
		if (nal_unit_type == 5) {
			bw_u(w, HG(drpm.no_output_of_prior_pics_flag), 1, lane);
			bw_u(w, HG(drpm.long_term_reference_flag), 1, lane);
		} else {
			uint32_t f = HG(drpm.adaptive_ref_pic_marking_mode_flag);
			bw_u(w, f, 1, lane);
			if (f) {
				uint32_t i = 0, op;
				do {
					uint32_t o = WOFF(drpm.mm) + 20u * i;
					if (i >= 64)
						return -EIO;
					op = hg(hdr, o);
					bw_ue(w, op, lane);
					if (op == 1 || op == 3)
						bw_ue(w, hg(hdr, o + 4), lane);
					if (op == 2)
						bw_ue(w, hg(hdr, o + 8), lane);
					if (op == 3 || op == 6)
						bw_ue(w, hg(hdr, o + 12), lane);
					if (op == 4)
						bw_ue(w, hg(hdr, o + 16), lane);
					i++;
				} while (op != 0);
			}
		}
	}
	if (pps.entropy_coding_mode_flag && type != H264_SLICE_TYPE_I &&
	    type != H264_SLICE_TYPE_SI)
		bw_ue(w, HG(cabac_init_idc), lane);
	bw_se(w, (int32_t)HG(slice_qp_delta), lane);
	if (type == H264_SLICE_TYPE_SP || type == H264_SLICE_TYPE_SI) {
		if (type == H264_SLICE_TYPE_SP)
			bw_u(w, HG(sp_for_switch_flag), 1, lane);
		bw_se(w, (int32_t)HG(slice_qs_delta), lane);
	}
	if (pps.deblocking_filter_control_present_flag) {
		uint32_t idc = HG(disable_deblocking_filter_idc);
		bw_ue(w, idc, lane);
		if (idc != 1) {
			bw_se(w, (int32_t)HG(slice_alpha_c0_offset_div2), lane);
			bw_se(w, (int32_t)HG(slice_beta_offset_div2), lane);
		}
	}
	if (pps.num_slice_groups_minus1 > 0 && pps.slice_group_map_type >= 3 &&
	    pps.slice_group_map_type <= 5) {
		uint32_t units = sps.pic_width_in_mbs * sps.pic_height_in_map_units;
		uint32_t n = sw_intlog2(
			units / (pps.slice_group_change_rate_minus1 + 1u) + 1u);
		if (n)
			bw_u(w, HG(slice_group_change_cycle), n, lane);
	}
	return 0;
}

/* load a 3968-byte header image into LDS */
__device__ __forceinline__ void sw_load_header(uint8_t *hdr,
					       const struct h264_slice_header *src,
					       uint32_t lane)
{
	const uint32_t hb = (uint32_t)sizeof(struct h264_slice_header);
	const uint8_t *s = reinterpret_cast<const uint8_t *>(src);
	for (uint32_t part = 0; part * 1024u < hb; part++) {
		uint32_t want = part * 1024u + 16u * lane;
		uint32_t off = want < hb ? want : hb - 16u;
		*reinterpret_cast<uint4 *>(hdr + off) =
			*reinterpret_cast<const uint4 *>(s + off);
	}
}

/* ---- write-back kernel ------------------------------------------------------ */

struct sw_args {
	const uint8_t *src;
	const struct h264_amd_slice_desc *descs;
	const struct h264_amd_slice_result *results;
	const struct h264_slice_header *headers;
	uint32_t n;
	const dev_sps *spsv;
	const dev_pps *ppsv;
	const dev_epoch *epochs;
	uint32_t n_sps, n_pps, n_epochs;
	uint8_t *out;
	const uint64_t *out_offsets;
	uint32_t *out_sizes;
	uint32_t out_capacity;
	uint32_t wflags;
	uint32_t *out_hdr_bits;
	uint64_t src_len; /* bytes behind src, 0 = not checked */
};

#define SW_HDR_OFF 0u
#define SW_STAGE_OFF 4096u
#define SW_LDS_BYTES (SW_STAGE_OFF + SW_STAGE_BYTES)

/* Copy n bytes from src (any alignment) to dst (any alignment), whole wave.
 * Body: 16-byte stores to aligned destination rows fed by aligned dword loads
 * that are byte-shifted with v_alignbyte; head and tail bytewise. */
__device__ __forceinline__ void sw_copy(uint8_t *dst, const uint8_t *src,
					uint32_t n, uint32_t lane)
{
	/* head: up to 15 bytes until dst is 16-byte aligned */
	uint32_t head = (uint32_t)((16u - (reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u);
	head = head < n ? head : n;
	if (head) {
		uint32_t j = lane < head ? lane : 0u;
		dst[j] = src[j];
	}
	dst += head;
	src += head;
	n -= head;
	uint32_t rows = n / 16u;
	uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u);
	const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src - sh);
	for (uint32_t r0 = 0; r0 < rows; r0 += 64u) {
		uint32_t r = r0 + lane;
		r = r < rows ? r : r0; /* spare lanes repeat row r0 */
		const uint32_t *p = s32 + 4u * r;
		uint32_t a = p[0], b = p[1], c = p[2], d = p[3];
		/* the fifth dword only exists when the source is misaligned */
		uint32_t e = sh ? p[4] : 0u;
		uint4 q;
		q.x = __builtin_amdgcn_alignbyte(b, a, sh);
		q.y = __builtin_amdgcn_alignbyte(c, b, sh);
		q.z = __builtin_amdgcn_alignbyte(d, c, sh);
		q.w = __builtin_amdgcn_alignbyte(e, d, sh);
		*reinterpret_cast<uint4 *>(dst + 16u * r) = q;
	}
	uint32_t done = rows * 16u, tail = n - done; /* < 16 */
	if (tail) {
		uint32_t j = done + (lane < tail ? lane : 0u);
		dst[j] = src[j];
	}
}

extern "C" __global__ void __launch_bounds__(64)
h264_amd_k_write_slices(sw_args a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint8_t *hdr = lds + SW_HDR_OFF;
	uint8_t *stage = lds + SW_STAGE_OFF;
	const uint32_t lane = wb_lane();
	const uint32_t slice = blockIdx.x;
	if (slice >= a.n)
		return;
	struct h264_amd_slice_desc d;
	sw_load_uniform(&d, a.descs + slice);
	struct h264_amd_slice_result res;
	sw_load_uniform(&res, a.results + slice);
	uint64_t oo = a.out_offsets[slice];
	oo = ((uint64_t)wb_uniform((uint32_t)(oo >> 32)) << 32) |
	     wb_uniform((uint32_t)oo);
	uint8_t *dst = a.out + oo;

	int status = 0;
	uint32_t total = 0;
	if (!(res.flags & H264_AMD_SR_HEADER_OK)) {
		status = -EINVAL; /* nothing to serialise */
	} else {
		sw_load_header(hdr, a.headers + slice, lane);
		SwParams P;
		uint32_t e = d.epoch < a.n_epochs ? d.epoch : 0u;
		status = sw_activate(P, hg(hdr, WOFF(pic_parameter_set_id)),
				     a.spsv, a.ppsv, a.epochs + e, a.n_sps, a.n_pps);
		BitW w;
		bw_open(w, stage, dst, a.out_capacity, a.wflags);
		if (status == 0)
			status = sw_slice_header(w, hdr, P, res.nal_ref_idc,
						 res.nal_unit_type, lane);
		if (status == 0 && a.out_hdr_bits)
			a.out_hdr_bits[slice] = 8u * w.pos + w.nbits;
		if (status == 0 && (res.partialbits || res.rawdata_len)) {
			/* partial byte, then raw bytes (must be aligned now) */
			if (res.partialbits)
				bw_u(w, res.partial, res.partialbits, lane);
			if (w.nbits != 0)
				status = -EIO;
		}
		if (status == 0 && w.nbits != 0) {
			/* header only, ending inside a byte: hand the pending
			 * bits back, left aligned, behind the whole bytes */
			uint32_t pend = (w.acc << (8u - w.nbits)) & 0xffu;
			w.stage[w.pos & (SW_STAGE_BYTES - 1)] = (uint8_t)pend;
			w.pos++;
			bw_drain(w, lane);
			w.pos--;
			if (w.overflow || w.pos + 1u > a.out_capacity)
				status = -EIO;
			total = w.pos;
		} else if (status == 0) {
			bw_drain(w, lane);
			if (a.src_len != 0 &&
			    (d.offset > a.src_len || d.size > a.src_len - d.offset)) {
				status = H264_AMD_E_BAD_DESC;
			} else if (w.overflow ||
				   w.pos + res.rawdata_len > a.out_capacity ||
				   res.rawdata_off + res.rawdata_len > d.size) {
				status = -EIO;
			} else {
				sw_copy(dst + w.pos, a.src + d.offset + res.rawdata_off,
					res.rawdata_len, lane);
				total = w.pos + res.rawdata_len;
			}
		}
	}
	a.out_sizes[slice] = status ? (uint32_t)status : total;
}

/* ---- CABAC ------------------------------------------------------------------- */

/* 9.3.1.1 with the reference's clipping (ref src/h264_bac.c:216-230):
 * returns pStateIdx | valMPS << 8 */
__device__ __forceinline__ uint32_t cabac_state_init_q(int32_t qp, int32_t m,
						       int32_t n);

__device__ __forceinline__ uint32_t cabac_state_init(int32_t qp, int32_t m,
						     int32_t n)
{
	return cabac_state_init_q(qp < 1 ? 1 : qp > 51 ? 51 : qp, m, n);
}

/* the same with the QP already clipped by the caller */
__device__ __forceinline__ uint32_t cabac_state_init_q(int32_t qp, int32_t m,
						       int32_t n)
{
	int32_t idx = ((m * qp) >> 4) + n;
	if (idx <= 63)
		return (uint32_t)(63 - (idx > 1 ? idx : 1));
	return (uint32_t)((idx < 126 ? idx : 126) - 64) | 0x100u;
}

/* states[1024][2] for n (slice_type, cabac_init_idc, SliceQPLuma) triples;
 * one 256-thread block per triple, the (m,n) table staged in LDS */
extern "C" __global__ void __launch_bounds__(256)
h264_amd_k_cabac_init(const cabac_tables *tab, const int32_t *triples,
		      uint32_t n, uint8_t *states)
{
	__shared__ int8_t mn[1024][4][2];
	const uint32_t t = threadIdx.x;
	{
		const uint4 *s = reinterpret_cast<const uint4 *>(&tab->mn[0][0][0]);
		uint4 *d = reinterpret_cast<uint4 *>(&mn[0][0][0]);
		d[t] = s[t];
		d[t + 256] = s[t + 256];
	}
	__syncthreads();
	for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
		int32_t st = triples[3 * i], idc = triples[3 * i + 1];
		int32_t qp = triples[3 * i + 2];
		bool intra = st == H264_SLICE_TYPE_I || st == H264_SLICE_TYPE_SI;
		bool valid = intra || (idc >= 0 && idc <= 2);
		uint32_t col = intra ? 0u : (uint32_t)(1 + (valid ? idc : 0));
		for (uint32_t c = t; c < 1024; c += 256) {
			uint32_t s = cabac_state_init(qp, mn[c][col][0], mn[c][col][1]);
			s = valid ? s : 0u;
			states[(size_t)i * 2048 + 2 * c] = (uint8_t)(s & 0xff);
			states[(size_t)i * 2048 + 2 * c + 1] = (uint8_t)(s >> 8);
		}
	}
}

/* arithmetic encoder state, wave-uniform (ref struct h264_bac_enc) */
struct Bac {
	uint32_t range, low, outstanding, first;
};

__device__ __forceinline__ void bac_put_bit(Bac &e, BitW &w, uint32_t bit,
					    uint32_t lane)
{
	if (e.first)
		e.first = 0;
	else
		bw_u(w, bit, 1, lane);
	while (e.outstanding > 0) {
		bw_u(w, bit ^ 1u, 1, lane);
		e.outstanding--;
	}
}

__device__ __forceinline__ void bac_renorm(Bac &e, BitW &w, uint32_t lane)
{
	while (e.range < 256) {
		if (e.low < 256) {
			bac_put_bit(e, w, 0, lane);
		} else if (e.low < 512) {
			e.low -= 256;
			e.outstanding++;
		} else {
			e.low -= 512;
			bac_put_bit(e, w, 1, lane);
		}
		e.range <<= 1;
		e.low <<= 1;
	}
}

/* Lane tables: lane s holds rangeTabLPS[s][0..3] packed, and
 * transIdxLPS | transIdxMPS << 8 */
struct BacTabs {
	uint32_t lps4, trans;
};

/* one decision; ctx state lives in LDS as idx | mps << 8 */
__device__ __forceinline__ void bac_decision(Bac &e, BitW &w,
					     const BacTabs &t, uint16_t *ctx,
					     uint32_t ci, uint32_t bin,
					     uint32_t lane)
{
	uint32_t st = wb_uniform(ctx[ci]);
	uint32_t idx = st & 0xffu, mps = st >> 8;
	uint32_t q = (e.range >> 6) & 3u;
	uint32_t r4 = (uint32_t)__builtin_amdgcn_readlane((int)t.lps4, (int)idx);
	uint32_t r_lps = (r4 >> (8u * q)) & 0xffu;
	uint32_t tr = (uint32_t)__builtin_amdgcn_readlane((int)t.trans, (int)idx);
	e.range -= r_lps;
	if (bin == mps) {
		idx = tr >> 8;
	} else {
		e.low += e.range;
		e.range = r_lps;
		if (idx == 0)
			mps ^= 1u;
		idx = tr & 0xffu;
	}
	ctx[ci] = (uint16_t)(idx | (mps << 8)); /* every lane, same value */
	bac_renorm(e, w, lane);
}

__device__ __forceinline__ void bac_terminate(Bac &e, BitW &w, uint32_t bin,
					      uint32_t lane)
{
	e.range -= 2;
	if (!bin) {
		bac_renorm(e, w, lane);
		return;
	}
	e.low += e.range;
	e.range = 2;
	bac_renorm(e, w, lane);
	bac_put_bit(e, w, (e.low >> 9) & 1u, lane);
	bw_u(w, ((e.low >> 7) & 3u) | 1u, 2, lane);
}

struct synth_args {
	const struct h264_amd_synth_desc *descs;
	const struct h264_slice_header *headers;
	uint32_t n;
	const dev_sps *spsv;
	const dev_pps *ppsv;
	const dev_epoch *epochs;
	uint32_t n_sps, n_pps, n_epochs;
	const cabac_tables *tab;
	uint8_t *out;
	const uint64_t *out_offsets;
	int32_t *out_sizes;
	uint32_t out_capacity;
	uint32_t wflags;
	uint32_t *out_hdr_bits;
};

#define SY_HDR_OFF 0u
#define SY_STAGE_OFF 4096u
#define SY_CTX_OFF (SY_STAGE_OFF + SW_STAGE_BYTES)
#define SY_MN_OFF (SY_CTX_OFF + 2048u)
#define SY_LDS_BYTES (SY_MN_OFF + 8192u)

/* One synthetic slice per wave (ref src/h264_writer.c:49-309).  The (m,n)
 * context tables are staged in LDS and the 1024 context variables are
 * initialised by the lanes 16 at a time. */
extern "C" __global__ void __launch_bounds__(64)
h264_amd_k_encode_synth(synth_args a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint8_t *hdr = lds + SY_HDR_OFF;
	uint8_t *stage = lds + SY_STAGE_OFF;
	uint16_t *ctx = reinterpret_cast<uint16_t *>(lds + SY_CTX_OFF);
	int8_t *mn = reinterpret_cast<int8_t *>(lds + SY_MN_OFF);
	const uint32_t lane = wb_lane();
	const uint32_t slice = blockIdx.x;
	if (slice >= a.n)
		return;
	struct h264_amd_synth_desc d;
	sw_load_uniform(&d, a.descs + slice);
	uint64_t oo = a.out_offsets[slice];
	oo = ((uint64_t)wb_uniform((uint32_t)(oo >> 32)) << 32) |
	     wb_uniform((uint32_t)oo);
	uint8_t *dst = a.out + oo;

	/* (m,n) tables -> LDS: 8 KiB, 8 x 16 B per lane */
	{
		const uint4 *s = reinterpret_cast<const uint4 *>(&a.tab->mn[0][0][0]);
		uint4 *m = reinterpret_cast<uint4 *>(mn);
#pragma unroll
		for (uint32_t k = 0; k < 8; k++)
			m[64u * k + lane] = s[64u * k + lane];
	}
	sw_load_header(hdr, a.headers + slice, lane);

	int status = 0;
	SwParams P;
	BitW w;
	bw_open(w, stage, dst, a.out_capacity, a.wflags);
	if (d.mb_count == 0 || d.kind > H264_AMD_SYNTH_SKIP_P) {
		status = -EINVAL;
	} else {
		uint32_t e = d.epoch < a.n_epochs ? d.epoch : 0u;
		status = sw_activate(P, hg(hdr, WOFF(pic_parameter_set_id)), a.spsv,
				     a.ppsv, a.epochs + e, a.n_sps, a.n_pps);
	}
	if (status == 0)
		status = sw_slice_header(w, hdr, P, d.nal_ref_idc, d.nal_unit_type,
					 lane);
	if (status == 0 && a.out_hdr_bits)
		a.out_hdr_bits[slice] = 8u * w.pos + w.nbits;
	if (status == 0) {
		const uint32_t type = hg(hdr, WOFF(slice_type)) % 5u;
		const uint32_t first_mb = hg(hdr, WOFF(first_mb_in_slice));
		const uint32_t W = P.sps.pic_width_in_mbs ? P.sps.pic_width_in_mbs : 1u;
		if (!P.pps.entropy_coding_mode_flag) {
			if (d.kind == H264_AMD_SYNTH_GREY_I) {
				/* ue(3) mb_type I_16x16_2_0_0, ue(0) chroma DC,
				 * se(0) qp delta, coeff_token "1" */
				for (uint32_t i = 0; i < d.mb_count; i++)
					bw_u(w, 0x27u, 8, lane);
			} else {
				bw_ue(w, d.mb_count, lane); /* one mb_skip_run */
			}
			bw_trailing(w, lane);
		} else {
			/* cabac_alignment_one_bit */
			if (w.nbits)
				bw_u(w, (1u << (8u - w.nbits)) - 1u, 8u - w.nbits, lane);
			/* 9.3.1.1: all 1024 context variables */
			{
				uint32_t idc = hg(hdr, WOFF(cabac_init_idc));
				bool intra = type == H264_SLICE_TYPE_I ||
					     type == H264_SLICE_TYPE_SI;
				bool valid = intra || idc <= 2;
				uint32_t col = intra ? 0u : 1u + (valid ? idc : 0u);
				int32_t qp = P.pps.pic_init_qp_minus26 + 26 +
					     (int32_t)hg(hdr, WOFF(slice_qp_delta));
				for (uint32_t c = lane; c < 1024; c += 64) {
					uint32_t s = cabac_state_init(
						qp, mn[8u * c + 2u * col],
						mn[8u * c + 2u * col + 1u]);
					ctx[c] = (uint16_t)(valid ? s : 0u);
				}
			}
			BacTabs bt;
			{
				const uint8_t *r = &a.tab->range_lps[lane][0];
				bt.lps4 = (uint32_t)r[0] | ((uint32_t)r[1] << 8) |
					  ((uint32_t)r[2] << 16) | ((uint32_t)r[3] << 24);
				bt.trans = (uint32_t)a.tab->next_lps[lane] |
					   ((uint32_t)a.tab->next_mps[lane] << 8);
			}
			Bac e;
			e.low = 0;
			e.range = 510;
			e.first = 1;
			e.outstanding = 0;
			const uint32_t skip_off =
				type == H264_SLICE_TYPE_B ? 24u : 11u;
			if (d.kind == H264_AMD_SYNTH_SKIP_P &&
			    !(type == H264_SLICE_TYPE_P || type == H264_SLICE_TYPE_SP ||
			      type == H264_SLICE_TYPE_B))
				status = -EIO;
			if (d.kind == H264_AMD_SYNTH_GREY_I && type != H264_SLICE_TYPE_I)
				status = -EIO;
			for (uint32_t i = 0; i < d.mb_count && status == 0; i++) {
				uint32_t addr = first_mb + i;
				/* neighbour availability inside the slice */
				uint32_t availA = (addr % W != 0 && i >= 1) ? 1u : 0u;
				uint32_t availB = i >= W ? 1u : 0u;
				uint32_t last = i + 1 == d.mb_count ? 1u : 0u;
				if (d.kind == H264_AMD_SYNTH_SKIP_P) {
					/* mb_skip_flag = 1: neighbours are
					 * skipped too, ctxIdxInc 0 */
					bac_decision(e, w, bt, ctx, skip_off, 1, lane);
				} else {
					/* mb_type 3 -> bins 1 0 0 0 1 0 (Table 9-36);
					 * neighbours are I_16x16: condTerm 1 */
					bac_decision(e, w, bt, ctx, 3u + availA + availB, 1, lane);
					bac_terminate(e, w, 0, lane);
					bac_decision(e, w, bt, ctx, 3u + 3u, 0, lane);
					bac_decision(e, w, bt, ctx, 3u + 4u, 0, lane);
					bac_decision(e, w, bt, ctx, 3u + 6u, 1, lane);
					bac_decision(e, w, bt, ctx, 3u + 7u, 0, lane);
					/* intra_chroma_pred_mode 0: neighbours' is 0 */
					bac_decision(e, w, bt, ctx, 64u, 0, lane);
					/* mb_qp_delta 0 */
					bac_decision(e, w, bt, ctx, 60u, 0, lane);
					/* coded_block_flag(Intra16x16DC) = 0; condTerm
					 * is 1 only for a missing neighbour */
					bac_decision(e, w, bt, ctx,
						     85u + (1u - availA) + 2u * (1u - availB),
						     0, lane);
				}
				bac_terminate(e, w, last, lane);
			}
			/* stop bit is out: zero-align */
			if (w.nbits)
				bw_u(w, 0, 8u - w.nbits, lane);
		}
	}
	if (status == 0) {
		bw_drain(w, lane);
		if (w.overflow)
			status = -EIO;
	}
	a.out_sizes[slice] = status ? status : (int32_t)w.pos;
}

#endif /* H264_AMD_SLICE_WRITE_HIP */

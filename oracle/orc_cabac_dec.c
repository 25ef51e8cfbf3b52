/**
 * oracle/orc_cabac_dec.c -- TEST INFRASTRUCTURE (see orc.h).
 *
 * CABAC slice data, DECODE direction -- an ENGINE EXTENSION, not reference
 * behaviour: the reference returns before any CABAC slice data
 * (src/h264_syntax_slice_data.h:715-717; its src/h264_bac.c:236-247 decoder
 * stub is dead code and off the standard), and so do oracle and product unless
 * H264_AMD_READER_FLAGS_CABAC is given.  PARITY UNPINNED (spec-only): what
 * follows is written from the standard --
 *   7.3.4, 7.3.5.x     slice_data / macroblock_layer / mb_pred / sub_mb_pred /
 *                      residual / residual_block_cabac syntax
 *   9.3.1.1, 9.3.1.2   initialisation of context variables and decoding engine
 *   9.3.2.x            binarisations (U, TU, UEGk, FL, mb_type / sub_mb_type
 *                      trees, coded_block_pattern, mb_qp_delta)
 *   9.3.3.1.x          ctxIdx assignment, ctxIdxInc from neighbouring data
 *   9.3.3.2.x          DecodeDecision / DecodeBypass / DecodeTerminate
 * -- bit-serially and with the neighbour derivations of 6.4.11 by block index,
 * the way the rest of the oracle restates the reference.  What pins it:
 *   * it is the inverse of the reference's own ENCODER on everything that
 *     encoder emits (src/h264_bac.c:253-358, src/h264_cabac.c:61-975,
 *     src/h264_writer.c:79-129,177-219): decode(h264_write_grey_i_slice(x)) and
 *     decode(h264_write_skipped_p_slice(x)) give back x (tests/test_cabac_*.py;
 *     the reference's shortcuts -- mb_qp_delta ctxIdxInc 0, transBlock
 *     unavailable -- coincide with the standard on those streams);
 *   * context initialisation is the reference's (orc_cabac_init_states: SliceQPY
 *     clipped to 1..51 where the standard says 0..51), so that the library's
 *     own writers round-trip at SliceQPY 0; a reader flag switches to
 *     the standard's clip for streams of other encoders (x264 lossless sits at
 *     QP 0), the engine's H264_AMD_READER_FLAGS_CABAC_STD_QP;
 *   * an independent ENCODER for the whole macroblock layer written from the
 *     standard (libh264_amd/csrc/gen/h264gen_cabac.inc), whose recorded
 *     macroblock syntax must equal what this decoder reads back.
 * Conventions of the records (h264_amd.h, H264_AMD_READER_FLAGS_CABAC) and the
 * defined behaviour on damaged input are shared with the kernel
 * (libh264_amd/csrc/kernels/cabac_slice.hip).  MBAFF frames and slice groups
 * with CABAC: H264_AMD_E_UNSUPPORTED.
 */
#include "orc.h"

#define CHK(x)                                                                 \
	do {                                                                   \
		int _r = (x);                                                  \
		if (_r < 0)                                                    \
			return _r;                                             \
	} while (0)

#define EGK_MAX 24u /* prefix bits of an Exp-Golomb escape before -EIO */

uint64_t orc_stat_cabac_bins;
int orc_quirk_x264_444_cbf; /* see cbf_cond() */
/* bins by kind (bench statistics): [ctxIdx] for decisions, [1024] bypass,
 * [1025] terminate */
uint64_t orc_stat_cabac_hist[1026];

/* ---- constants ------------------------------------------------------------ */
static uint8_t s_lps[64][4], s_next_lps[64], s_next_mps[64];
static int s_ready;

#define H264_CRT(type, name, dims, ...) static const type crt_##name dims = __VA_ARGS__;
#include "../libh264_amd/csrc/common/h264_cabac_residual_tables.inc"
#undef H264_CRT

static void tables_init(void)
{
	if (s_ready)
		return;
#define H264_LPS(i, a, b, c, d)                                                \
	s_lps[i][0] = a;                                                       \
	s_lps[i][1] = b;                                                       \
	s_lps[i][2] = c;                                                       \
	s_lps[i][3] = d;
#define H264_TRANS(i, l, m)                                                    \
	s_next_lps[i] = l;                                                     \
	s_next_mps[i] = m;
#include "../libh264_amd/csrc/common/h264_cabac_tables.inc"
#undef H264_LPS
#undef H264_TRANS
	s_ready = 1;
}

/* ---- decoding engine (9.3.1.2, 9.3.3.2) ------------------------------------- */
struct cd {
	struct h264_bitstream *bs;
	uint32_t range, offset;
	struct {
		uint8_t idx, mps;
	} st[1024];
	struct orc_ctx *ctx;
	struct orc_mb *mb;
	struct orc_mb_info *info; /* of the current macroblock */
	int prev_qpd_nz; /* previous macroblock in decoding order: mb_qp_delta != 0 */
	uint32_t field; /* field decoding: the field context sets */
	uint32_t cat; /* ChromaArrayType */
};

static int d_init(struct cd *c)
{
	c->range = 510;
	if (orc_bs_read_bits(c->bs, &c->offset, 9) < 0)
		return -EIO;
	/* 9.3.1.2: codIOffset 510 and 511 are illegal.  From a legal start
	 * codIOffset < codIRange holds for good; from an illegal one it never
	 * does again and codIOffset grows without bound -- an error here, in
	 * oracle and product alike, instead of arithmetic on a state the
	 * decoding engine is not defined for (the volume fuzz found the two
	 * diverging there: the kernel keeps codIOffset scaled in one register) */
	return c->offset >= c->range ? -EIO : 0;
}

static int d_renorm(struct cd *c)
{
	while (c->range < 256) {
		uint32_t b = 0;
		if (orc_bs_read_bits(c->bs, &b, 1) < 0)
			return -EIO;
		c->range <<= 1;
		c->offset = (c->offset << 1) | b;
	}
	return 0;
}

static int dd(struct cd *c, uint32_t ci, uint32_t *bin)
{
	uint32_t q = (c->range >> 6) & 3;
	uint32_t r_lps = s_lps[c->st[ci].idx][q];
	orc_stat_cabac_bins++;
	orc_stat_cabac_hist[ci & 1023u]++;
	c->range -= r_lps;
	if (c->offset >= c->range) {
		*bin = !c->st[ci].mps;
		c->offset -= c->range;
		c->range = r_lps;
		if (c->st[ci].idx == 0)
			c->st[ci].mps = !c->st[ci].mps;
		c->st[ci].idx = s_next_lps[c->st[ci].idx];
	} else {
		*bin = c->st[ci].mps;
		c->st[ci].idx = s_next_mps[c->st[ci].idx];
	}
	return d_renorm(c);
}

static int d_bypass(struct cd *c, uint32_t *bin)
{
	uint32_t b = 0;
	orc_stat_cabac_bins++;
	orc_stat_cabac_hist[1024]++;
	if (orc_bs_read_bits(c->bs, &b, 1) < 0)
		return -EIO;
	c->offset = (c->offset << 1) | b;
	if (c->offset >= c->range) {
		*bin = 1;
		c->offset -= c->range;
	} else {
		*bin = 0;
	}
	return 0;
}

static int d_term(struct cd *c, uint32_t *bin)
{
	orc_stat_cabac_bins++;
	orc_stat_cabac_hist[1025]++;
	c->range -= 2;
	if (c->offset >= c->range) {
		*bin = 1;
		return 0;
	}
	*bin = 0;
	return d_renorm(c);
}

/* suffix of UEGk (9.3.2.3) */
static int d_egk(struct cd *c, uint32_t k, uint32_t *val)
{
	uint32_t v = 0, b = 0, n = 0;
	for (;;) {
		CHK(d_bypass(c, &b));
		if (!b)
			break;
		v += 1u << k;
		k++;
		if (++n >= EGK_MAX)
			return -EIO;
	}
	while (k--) {
		CHK(d_bypass(c, &b));
		v += b << k;
	}
	*val = v;
	return 0;
}

/* ---- macroblock-level neighbours (6.4.9) ----------------------------------- */
static int is_intra_type(uint32_t t)
{
	return t == H264_MB_TYPE_I_NxN || t == H264_MB_TYPE_I_16x16 ||
	       t == H264_MB_TYPE_I_PCM || t == H264_MB_TYPE_SI;
}

static const struct orc_mb_info *info_of(const struct cd *c, uint32_t addr)
{
	if (addr == ORC_MB_ADDR_INVALID)
		return NULL;
	return &c->ctx->slice.mb_table.info[orc_mb_off(c->ctx, addr)];
}

/* ---- mb_type (9.3.2.5, Tables 9-36, 9-37, 9-39) ----------------------------- */

/* the bin string of the I slice types, with the contexts of ctxIdxOffset 3
 * (c0 = ctxIdx of the first bin) or of the suffixes 17 / 32 */
static int d_mb_type_intra(struct cd *c, uint32_t off, uint32_t c0, uint32_t *t)
{
	uint32_t b = 0, luma, chroma, pred;
	const uint32_t i = off == 3;
	CHK(dd(c, c0, &b));
	if (!b) {
		*t = 0; /* I_NxN */
		return 0;
	}
	CHK(d_term(c, &b));
	if (b) {
		*t = 25; /* I_PCM */
		return 0;
	}
	CHK(dd(c, i ? 3 + 3 : off + 1, &luma));
	CHK(dd(c, i ? 3 + 4 : off + 2, &chroma));
	if (chroma) {
		CHK(dd(c, i ? 3 + 5 : off + 2, &b));
		chroma += b;
	}
	CHK(dd(c, i ? 3 + 6 : off + 3, &pred));
	CHK(dd(c, i ? 3 + 7 : off + 3, &b));
	pred = 2 * pred + b;
	*t = 1 + pred + 4 * chroma + 12 * luma;
	return 0;
}

static uint32_t inc_mb_type(const struct orc_mb_info *n, uint32_t off)
{
	if (n == NULL)
		return 0;
	if (off == 0)
		return n->mbt != H264_MB_TYPE_SI;
	if (off == 3)
		return n->mbt != H264_MB_TYPE_I_NxN;
	return n->mbt != H264_MB_TYPE_B_SKIP &&
	       n->mbt != H264_MB_TYPE_B_Direct_16x16;
}

/* mb_type numbered as in Tables 7-11, 7-13, 7-14 (what a CAVLC stream carries) */
static int d_mb_type(struct cd *c, uint32_t *type)
{
	const struct orc_mb *mb = c->mb;
	uint32_t b = 0, t = 0;
	switch (c->ctx->slice.type) {
	case H264_SLICE_TYPE_I:
		return d_mb_type_intra(c, 3,
				       3 + inc_mb_type(info_of(c, mb->mbAddrA), 3) +
					       inc_mb_type(info_of(c, mb->mbAddrB), 3),
				       type);
	case H264_SLICE_TYPE_SI:
		CHK(dd(c,
		       0 + inc_mb_type(info_of(c, mb->mbAddrA), 0) +
			       inc_mb_type(info_of(c, mb->mbAddrB), 0),
		       &b));
		if (!b) {
			*type = 0;
			return 0;
		}
		CHK(d_mb_type_intra(c, 3,
				    3 + inc_mb_type(info_of(c, mb->mbAddrA), 3) +
					    inc_mb_type(info_of(c, mb->mbAddrB), 3),
				    &t));
		*type = 1 + t;
		return 0;
	case H264_SLICE_TYPE_P:
	case H264_SLICE_TYPE_SP:
		CHK(dd(c, 14, &b));
		if (b) {
			CHK(d_mb_type_intra(c, 17, 17, &t));
			*type = 5 + t;
			return 0;
		}
		CHK(dd(c, 15, &b));
		if (!b) {
			CHK(dd(c, 16, &b));
			*type = b ? 3 : 0; /* P_8x8 : P_L0_16x16 */
		} else {
			CHK(dd(c, 17, &b));
			*type = b ? 1 : 2; /* P_L0_L0_16x8 : P_L0_L0_8x16 */
		}
		return 0;
	case H264_SLICE_TYPE_B: {
		uint32_t bits;
		CHK(dd(c,
		       27 + inc_mb_type(info_of(c, mb->mbAddrA), 27) +
			       inc_mb_type(info_of(c, mb->mbAddrB), 27),
		       &b));
		if (!b) {
			*type = 0; /* B_Direct_16x16 */
			return 0;
		}
		CHK(dd(c, 27 + 3, &b));
		if (!b) {
			CHK(dd(c, 27 + 5, &b));
			*type = 1 + b; /* B_L0_16x16, B_L1_16x16 */
			return 0;
		}
		CHK(dd(c, 27 + 4, &b));
		bits = b << 3;
		CHK(dd(c, 27 + 5, &b));
		bits |= b << 2;
		CHK(dd(c, 27 + 5, &b));
		bits |= b << 1;
		CHK(dd(c, 27 + 5, &b));
		bits |= b;
		if (bits < 8) {
			*type = 3 + bits;
		} else if (bits == 13) {
			CHK(d_mb_type_intra(c, 32, 32, &t));
			*type = 23 + t;
		} else if (bits == 14) {
			*type = 11; /* B_L1_L0_8x16 */
		} else if (bits == 15) {
			*type = 22; /* B_8x8 */
		} else {
			CHK(dd(c, 27 + 5, &b));
			*type = 2 * bits + b - 4;
		}
		return 0;
	}
	default:
		return -EIO;
	}
}

static int d_sub_mb_type(struct cd *c, uint32_t *type)
{
	uint32_t b = 0, t;
	if (c->ctx->slice.type != H264_SLICE_TYPE_B) {
		CHK(dd(c, 21, &b));
		if (b) {
			*type = 0;
			return 0;
		}
		CHK(dd(c, 22, &b));
		if (!b) {
			*type = 1;
			return 0;
		}
		CHK(dd(c, 23, &b));
		*type = b ? 2 : 3;
		return 0;
	}
	CHK(dd(c, 36, &b));
	if (!b) {
		*type = 0;
		return 0;
	}
	CHK(dd(c, 37, &b));
	if (!b) {
		CHK(dd(c, 39, &b));
		*type = 1 + b;
		return 0;
	}
	t = 3;
	CHK(dd(c, 38, &b));
	if (b) {
		CHK(dd(c, 39, &b));
		if (b) {
			CHK(dd(c, 39, &b));
			*type = 11 + b;
			return 0;
		}
		t += 4;
	}
	CHK(dd(c, 39, &b));
	t += 2 * b;
	CHK(dd(c, 39, &b));
	*type = t + b;
	return 0;
}

/* ---- prediction syntax ------------------------------------------------------- */

/* upper-left luma sample of macroblock partition p (6.4.2.1) and its size */
static void part_rect(const struct orc_mb *mb, uint32_t p, uint32_t *x, uint32_t *y,
		      uint32_t *w, uint32_t *h)
{
	switch (mb->mb_type) {
	case H264_MB_TYPE_P_16x8:
	case H264_MB_TYPE_B_16x8:
		*w = 16;
		*h = 8;
		break;
	case H264_MB_TYPE_P_8x16:
	case H264_MB_TYPE_B_8x16:
		*w = 8;
		*h = 16;
		break;
	case H264_MB_TYPE_P_8x8:
	case H264_MB_TYPE_P_8x8ref0:
	case H264_MB_TYPE_B_8x8:
		*w = 8;
		*h = 8;
		break;
	default:
		*w = 16;
		*h = 16;
		break;
	}
	/* InverseRasterScan(p, w, h, 16, e) */
	*x = (p % (16 / *w)) * *w;
	*y = (p / (16 / *w)) * *h;
}

/* the same for sub-macroblock partition s of sub-macroblock p (6.4.2.2) */
static void sub_rect(const struct orc_mb *mb, uint32_t p, uint32_t s, uint32_t *x,
		     uint32_t *y, uint32_t *w, uint32_t *h)
{
	switch (mb->sub_mb_type[p]) {
	case ORC_SUB_P_8x4:
	case ORC_SUB_B_8x4:
		*w = 8;
		*h = 4;
		break;
	case ORC_SUB_P_4x8:
	case ORC_SUB_B_4x8:
		*w = 4;
		*h = 8;
		break;
	case ORC_SUB_P_4x4:
	case ORC_SUB_B_4x4:
		*w = 4;
		*h = 4;
		break;
	default:
		*w = 8;
		*h = 8;
		break;
	}
	*x = (p % 2) * 8 + (s % (8 / *w)) * *w;
	*y = (p / 2) * 8 + (s / (8 / *w)) * *h;
}

/* 6.4.13.1: luma4x4BlkIdx of the block covering (x, y) */
static uint32_t blk_of(uint32_t x, uint32_t y)
{
	return 8 * (y / 8) + 4 * (x / 8) + 2 * ((y % 8) / 4) + ((x % 8) / 4);
}

/* ref_idx_lX (9.3.3.1.1.6; U binarisation) of the partition at (x, y) */
static int d_ref_idx(struct cd *c, uint32_t list, uint32_t x, uint32_t y, uint32_t w,
		     uint32_t h, uint32_t *val)
{
	uint32_t addrN, xW, yW, ra = 0, rb = 0, v = 0, b = 0, i, j;
	orc_neigh_loc(c->ctx, c->mb, 16, 16, (int32_t)x - 1, (int32_t)y, &addrN, &xW,
		      &yW);
	if (addrN != ORC_MB_ADDR_INVALID)
		ra = info_of(c, addrN)->ref_idx[list][2 * (yW / 8) + xW / 8];
	orc_neigh_loc(c->ctx, c->mb, 16, 16, (int32_t)x, (int32_t)y - 1, &addrN, &xW,
		      &yW);
	if (addrN != ORC_MB_ADDR_INVALID)
		rb = info_of(c, addrN)->ref_idx[list][2 * (yW / 8) + xW / 8];
	CHK(dd(c, 54 + (ra > 0) + 2 * (rb > 0), &b));
	if (b) {
		v = 1;
		CHK(dd(c, 54 + 4, &b));
		while (b) {
			if (++v > 255)
				return -EIO;
			CHK(dd(c, 54 + 5, &b));
		}
	}
	for (j = y / 8; j < (y + h) / 8; j++)
		for (i = x / 8; i < (x + w) / 8; i++)
			c->info->ref_idx[list][2 * j + i] = (uint8_t)v;
	*val = v;
	return 0;
}

/* mvd_lX[][][0..1] (9.3.3.1.1.7; UEG3, uCoff 9, signed) of the (sub-)partition
 * at (x, y) */
static int d_mvd(struct cd *c, uint32_t list, uint32_t x, uint32_t y, uint32_t w,
		 uint32_t h, int16_t out[2])
{
	uint32_t comp, i, j;
	int32_t mv[2];
	for (comp = 0; comp < 2; comp++) {
		const uint32_t off = comp ? 47 : 40;
		uint32_t addrN, xW, yW, sum = 0, inc, m = 0, b = 0;
		orc_neigh_loc(c->ctx, c->mb, 16, 16, (int32_t)x - 1, (int32_t)y, &addrN,
			      &xW, &yW);
		if (addrN != ORC_MB_ADDR_INVALID) {
			int32_t a = info_of(c, addrN)->mvd[list][blk_of(xW, yW)][comp];
			sum += (uint32_t)(a < 0 ? -a : a);
		}
		orc_neigh_loc(c->ctx, c->mb, 16, 16, (int32_t)x, (int32_t)y - 1, &addrN,
			      &xW, &yW);
		if (addrN != ORC_MB_ADDR_INVALID) {
			int32_t a = info_of(c, addrN)->mvd[list][blk_of(xW, yW)][comp];
			sum += (uint32_t)(a < 0 ? -a : a);
		}
		inc = sum < 3 ? 0 : sum > 32 ? 2 : 1;
		/* prefix: TU with cMax = 9 */
		CHK(dd(c, off + inc, &b));
		while (b) {
			m++;
			if (m == 9)
				break;
			CHK(dd(c, off + (m < 4 ? m + 2 : 6), &b));
		}
		if (m == 9) {
			uint32_t suf = 0;
			CHK(d_egk(c, 3, &suf));
			m += suf;
		}
		mv[comp] = (int32_t)m;
		if (m) {
			CHK(d_bypass(c, &b));
			if (b)
				mv[comp] = -(int32_t)m;
		}
	}
	for (comp = 0; comp < 2; comp++) {
		out[comp] = (int16_t)mv[comp];
		for (j = y; j < y + h; j += 4)
			for (i = x; i < x + w; i += 4)
				c->info->mvd[list][blk_of(i, j)][comp] = (int16_t)mv[comp];
	}
	return 0;
}

/* 7.3.5.2 */
static int c_sub_mb_pred(struct cd *c)
{
	struct orc_mb *mb = c->mb;
	const struct h264_slice_header *sh = &c->ctx->slice.hdr;
	uint32_t p, s, x, y, w, h, v = 0;
	for (p = 0; p < 4; p++) {
		CHK(d_sub_mb_type(c, &v));
		CHK(orc_apply_sub_mb_type(c->ctx, mb, p, v));
	}
	if ((sh->num_ref_idx_l0_active_minus1 > 0 ||
	     mb->mb_field_decoding_flag != (int)sh->field_pic_flag) &&
	    mb->mb_type != H264_MB_TYPE_P_8x8ref0) {
		for (p = 0; p < 4; p++)
			if (mb->sub_mb_type[p] != ORC_SUB_B_Direct_8x8 &&
			    mb->SubMbPredMode[p] != ORC_PRED_L1) {
				CHK(d_ref_idx(c, 0, (p % 2) * 8, (p / 2) * 8, 8, 8, &v));
				mb->ref_idx_l0[p] = (uint8_t)v;
			}
	}
	if (sh->num_ref_idx_l1_active_minus1 > 0 ||
	    mb->mb_field_decoding_flag != (int)sh->field_pic_flag) {
		for (p = 0; p < 4; p++)
			if (mb->sub_mb_type[p] != ORC_SUB_B_Direct_8x8 &&
			    mb->SubMbPredMode[p] != ORC_PRED_L0) {
				CHK(d_ref_idx(c, 1, (p % 2) * 8, (p / 2) * 8, 8, 8, &v));
				mb->ref_idx_l1[p] = (uint8_t)v;
			}
	}
	for (p = 0; p < 4; p++) {
		if (mb->sub_mb_type[p] == ORC_SUB_B_Direct_8x8 ||
		    mb->SubMbPredMode[p] == ORC_PRED_L1)
			continue;
		for (s = 0; s < mb->NumSubMbPart[p]; s++) {
			sub_rect(mb, p, s, &x, &y, &w, &h);
			CHK(d_mvd(c, 0, x, y, w, h, mb->mvd_l0[p][s]));
		}
	}
	for (p = 0; p < 4; p++) {
		if (mb->sub_mb_type[p] == ORC_SUB_B_Direct_8x8 ||
		    mb->SubMbPredMode[p] == ORC_PRED_L0)
			continue;
		for (s = 0; s < mb->NumSubMbPart[p]; s++) {
			sub_rect(mb, p, s, &x, &y, &w, &h);
			CHK(d_mvd(c, 1, x, y, w, h, mb->mvd_l1[p][s]));
		}
	}
	return 0;
}

/* 7.3.5.1 */
static int c_mb_pred(struct cd *c)
{
	struct orc_mb *mb = c->mb;
	const struct h264_slice_header *sh = &c->ctx->slice.hdr;
	const uint32_t m0 = mb->MbPartPredMode[0];
	uint32_t i, p, x, y, w, h, b = 0, v = 0;

	if (m0 == ORC_PRED_Intra_4x4 || m0 == ORC_PRED_Intra_8x8 ||
	    m0 == ORC_PRED_Intra_16x16) {
		const uint32_t n = m0 == ORC_PRED_Intra_4x4   ? 16
				   : m0 == ORC_PRED_Intra_8x8 ? 4
							      : 0;
		for (i = 0; i < n; i++) {
			int8_t mode = -1;
			CHK(dd(c, 68, &b)); /* prev_intra{4x4,8x8}_pred_mode_flag */
			if (!b) {
				/* rem_intra..._pred_mode: FL, 3 bins, LSB first */
				CHK(dd(c, 69, &b));
				v = b;
				CHK(dd(c, 69, &b));
				v |= b << 1;
				CHK(dd(c, 69, &b));
				v |= b << 2;
				mode = (int8_t)v;
			}
			if (n == 16)
				mb->intra4x4_pred_mode[i] = mode;
			else
				mb->intra8x8_pred_mode[i] = mode;
		}
		if (c->cat == 1 || c->cat == 2) {
			/* intra_chroma_pred_mode: TU, cMax 3 (9.3.3.1.1.8) */
			const struct orc_mb_info *a = info_of(c, mb->mbAddrA);
			const struct orc_mb_info *bb = info_of(c, mb->mbAddrB);
			uint32_t inc = 0;
			if (a != NULL && is_intra_type(a->mbt) &&
			    a->mbt != H264_MB_TYPE_I_PCM && a->intra_chroma_pred_mode != 0)
				inc++;
			if (bb != NULL && is_intra_type(bb->mbt) &&
			    bb->mbt != H264_MB_TYPE_I_PCM && bb->intra_chroma_pred_mode != 0)
				inc++;
			CHK(dd(c, 64 + inc, &b));
			v = b;
			if (v == 1) {
				CHK(dd(c, 64 + 3, &b));
				v += b;
			}
			if (v == 2) {
				CHK(dd(c, 64 + 3, &b));
				v += b;
			}
			mb->intra_chroma_pred_mode = (uint8_t)v;
			c->info->intra_chroma_pred_mode = (uint8_t)v;
		}
		return 0;
	}
	if (m0 == ORC_PRED_Direct)
		return 0;
	if (sh->num_ref_idx_l0_active_minus1 > 0 ||
	    mb->mb_field_decoding_flag != (int)sh->field_pic_flag) {
		for (p = 0; p < mb->NumMbPart; p++)
			if (mb->MbPartPredMode[p] != ORC_PRED_L1) {
				part_rect(mb, p, &x, &y, &w, &h);
				CHK(d_ref_idx(c, 0, x, y, w, h, &v));
				mb->ref_idx_l0[p] = (uint8_t)v;
			}
	}
	if (sh->num_ref_idx_l1_active_minus1 > 0 ||
	    mb->mb_field_decoding_flag != (int)sh->field_pic_flag) {
		for (p = 0; p < mb->NumMbPart; p++)
			if (mb->MbPartPredMode[p] != ORC_PRED_L0) {
				part_rect(mb, p, &x, &y, &w, &h);
				CHK(d_ref_idx(c, 1, x, y, w, h, &v));
				mb->ref_idx_l1[p] = (uint8_t)v;
			}
	}
	for (p = 0; p < mb->NumMbPart; p++) {
		if (mb->MbPartPredMode[p] == ORC_PRED_L1)
			continue;
		part_rect(mb, p, &x, &y, &w, &h);
		CHK(d_mvd(c, 0, x, y, w, h, mb->mvd_l0[p][0]));
	}
	for (p = 0; p < mb->NumMbPart; p++) {
		if (mb->MbPartPredMode[p] == ORC_PRED_L0)
			continue;
		part_rect(mb, p, &x, &y, &w, &h);
		CHK(d_mvd(c, 1, x, y, w, h, mb->mvd_l1[p][0]));
	}
	return 0;
}

/* coded_block_pattern (9.3.2.6, 9.3.3.1.1.4) */
static int d_cbp(struct cd *c)
{
	struct orc_mb *mb = c->mb;
	uint32_t b8, luma = 0, chroma = 0, b = 0;
	for (b8 = 0; b8 < 4; b8++) {
		uint32_t cond[2], k;
		for (k = 0; k < 2; k++) {
			/* 6.4.11.2: the 8x8 block to the left / above */
			uint32_t addrN, xW, yW, b8N;
			orc_neigh_loc(c->ctx, mb, 16, 16,
				      (int32_t)(b8 % 2) * 8 - (k == 0),
				      (int32_t)(b8 / 2) * 8 - (k == 1), &addrN, &xW,
				      &yW);
			b8N = 2 * (yW / 8) + xW / 8;
			if (addrN == ORC_MB_ADDR_INVALID) {
				cond[k] = 0;
			} else if (addrN == mb->mbAddr) {
				cond[k] = !((luma >> b8N) & 1); /* prior bins */
			} else {
				const struct orc_mb_info *n = info_of(c, addrN);
				if (n->mbt == H264_MB_TYPE_I_PCM)
					cond[k] = 0;
				else if (n->skipped)
					cond[k] = 1;
				else
					cond[k] = !((n->cbp >> b8N) & 1);
			}
		}
		CHK(dd(c, 73 + cond[0] + 2 * cond[1], &b));
		luma |= b << b8;
	}
	if (c->cat == 1 || c->cat == 2) {
		const struct orc_mb_info *n[2] = {info_of(c, mb->mbAddrA),
						  info_of(c, mb->mbAddrB)};
		uint32_t cond[2], k;
		for (k = 0; k < 2; k++)
			cond[k] = n[k] != NULL &&
				  (n[k]->mbt == H264_MB_TYPE_I_PCM ||
				   (!n[k]->skipped && (n[k]->cbp >> 4) != 0));
		CHK(dd(c, 77 + cond[0] + 2 * cond[1], &b));
		if (b) {
			for (k = 0; k < 2; k++)
				cond[k] = n[k] != NULL &&
					  (n[k]->mbt == H264_MB_TYPE_I_PCM ||
					   (!n[k]->skipped && (n[k]->cbp >> 4) == 2));
			CHK(dd(c, 77 + 4 + cond[0] + 2 * cond[1], &b));
			chroma = 1 + b;
		}
	}
	mb->coded_block_pattern = (uint8_t)(luma | (chroma << 4));
	mb->CodedBlockPatternLuma = (uint8_t)luma;
	mb->CodedBlockPatternChroma = (uint8_t)chroma;
	return 0;
}

static int d_t8(struct cd *c)
{
	const struct orc_mb_info *a = info_of(c, c->mb->mbAddrA);
	const struct orc_mb_info *b = info_of(c, c->mb->mbAddrB);
	uint32_t v = 0;
	CHK(dd(c, 399 + (a != NULL && a->t8) + (b != NULL && b->t8), &v));
	c->mb->transform_size_8x8_flag = (int)v;
	c->info->t8 = (uint8_t)v;
	return 0;
}

/* mb_qp_delta (9.3.2.7, 9.3.3.1.1.5) */
static int d_qpd(struct cd *c)
{
	uint32_t b = 0, k = 0;
	CHK(dd(c, 60 + (c->prev_qpd_nz ? 1 : 0), &b));
	if (b) {
		k = 1;
		CHK(dd(c, 60 + 2, &b));
		while (b) {
			if (++k > 255)
				return -EIO;
			CHK(dd(c, 60 + 3, &b));
		}
	}
	c->mb->mb_qp_delta = (k & 1) ? (int32_t)((k + 1) / 2) : -(int32_t)(k / 2);
	c->prev_qpd_nz = k != 0;
	return 0;
}

/* ---- residual (7.3.5.3.x, 9.3.3.1.1.9, 9.3.3.1.3) --------------------------- */

enum { TB_DC, TB_4x4, TB_8x8, TB_CDC, TB_CAC };

/* condTermFlagN of coded_block_flag: kind of block, colour component (TB_DC,
 * TB_4x4, TB_8x8: 0 Y 1 Cb 2 Cr; chroma: 0 Cb 1 Cr), block index in mbAddrN */
static uint32_t cbf_cond(const struct cd *c, uint32_t addrN, uint32_t kind,
			 uint32_t comp, uint32_t idxN)
{
	const struct orc_mb_info *n;
	if (addrN == ORC_MB_ADDR_INVALID)
		return is_intra_type(c->mb->mb_type) ? 1 : 0;
	n = info_of(c, addrN);
	/* x264 before build 151 coded the coded_block_flag of an 8x8 block of a
	 * 4:4:4 macroblock next to a macroblock WITHOUT the 8x8 transform as if
	 * that neighbour were not there (what libavcodec's h264_cabac.c
	 * reproduces for streams of those builds).  Not the standard: off unless
	 * a test that decodes such a stream switches it on. */
	if (orc_quirk_x264_444_cbf && kind == TB_8x8 && !n->t8)
		return is_intra_type(c->mb->mb_type) ? 1 : 0;
	if (n->mbt == H264_MB_TYPE_I_PCM)
		return 1;
	if (n->skipped)
		return 0; /* available, transBlockN is not */
	switch (kind) {
	case TB_DC:
		if (n->mbt != H264_MB_TYPE_I_16x16)
			return 0;
		return (n->cbf_dc >> comp) & 1;
	case TB_4x4:
		/* the 4x4 block, or the 8x8 block holding it when the macroblock
		 * was transformed 8x8: its four bits are one flag */
		if (!((n->cbp >> (idxN >> 2)) & 1))
			return 0;
		return (n->cbf[comp] >> idxN) & 1;
	case TB_8x8:
		if (!((n->cbp >> idxN) & 1) || !n->t8)
			return 0;
		return (n->cbf[comp] >> (4 * idxN)) & 1;
	case TB_CDC:
		if ((n->cbp >> 4) == 0)
			return 0;
		return (n->cbf_dc >> (1 + comp)) & 1;
	default:
		if ((n->cbp >> 4) != 2)
			return 0;
		return (n->cbf[1 + comp] >> idxN) & 1;
	}
}

/* residual_block_cabac: numc coefficients into lvl[]; cbf_inc < 0: no
 * coded_block_flag in the stream (inferred 1) */
static int c_block(struct cd *c, uint32_t cat, int16_t *lvl, uint32_t numc,
		   int32_t cbf_inc, uint32_t *coded)
{
	const uint32_t nc8 = c->cat == 2 ? 2 : 1; /* NumC8x8 */
	const int is8 = cat == 5 || cat == 9 || cat == 13;
	uint8_t sig[64];
	uint32_t i, b = 1, n = numc, eq1 = 0, gt1 = 0;
	for (i = 0; i < numc; i++)
		lvl[i] = 0;
	if (cbf_inc >= 0)
		CHK(dd(c, crt_cbf_base[cat] + (uint32_t)cbf_inc, &b));
	*coded = b;
	if (!b)
		return 0;
	memset(sig, 0, sizeof(sig));
	for (i = 0; i + 1 < n; i++) {
		uint32_t si = is8	? crt_sig8[c->field][i]
			      : cat == 3 ? (i / nc8 < 2 ? i / nc8 : 2)
					 : i;
		uint32_t li = is8	? crt_last8[i]
			      : cat == 3 ? (i / nc8 < 2 ? i / nc8 : 2)
					 : i;
		CHK(dd(c, crt_sig_base[c->field][cat] + si, &b));
		sig[i] = (uint8_t)b;
		if (b) {
			CHK(dd(c, crt_last_base[c->field][cat] + li, &b));
			if (b)
				n = i + 1;
		}
	}
	sig[n - 1] = 1;
	for (i = n; i-- > 0;) {
		uint32_t a = 0;
		int32_t v;
		if (!sig[i])
			continue;
		/* coeff_abs_level_minus1: prefix TU (cMax 14), suffix UEG0 */
		CHK(dd(c, crt_abs_base[cat] + (gt1 ? 0 : (1 + eq1 < 4 ? 1 + eq1 : 4)), &b));
		if (b) {
			const uint32_t lim = 4 - (cat == 3);
			const uint32_t cn = crt_abs_base[cat] + 5 + (gt1 < lim ? gt1 : lim);
			a = 1;
			while (a < 14) {
				CHK(dd(c, cn, &b));
				if (!b)
					break;
				a++;
			}
			if (a == 14) {
				uint32_t suf = 0;
				CHK(d_egk(c, 0, &suf));
				a += suf;
			}
			gt1++;
		} else {
			eq1++;
		}
		CHK(d_bypass(c, &b)); /* coeff_sign_flag */
		v = (int32_t)a + 1;
		lvl[i] = (int16_t)(b ? -v : v);
	}
	return 0;
}

/* 7.3.5.3.1 for one colour component (0 Y, 1 Cb, 2 Cr) */
static int c_residual_luma(struct cd *c, uint32_t comp)
{
	struct orc_mb *mb = c->mb;
	const int i16 = mb->MbPartPredMode[0] == ORC_PRED_Intra_16x16;
	/* Table 9-42: DC / AC / 4x4 blocks of a component are ctxBlockCat cat0 +
	 * 0 / 1 / 2, its 8x8 block is 5, 9 or 13 */
	const uint32_t cat0 = comp == 0 ? 0 : comp == 1 ? 6 : 10;
	const uint32_t cat8 = comp == 0 ? 5 : comp == 1 ? 9 : 13;
	uint32_t i8, i4, k, coded = 0;
	uint32_t addrA, idxA, addrB, idxB;

	if (i16) {
		int32_t inc = (int32_t)(cbf_cond(c, mb->mbAddrA, TB_DC, comp, 0) +
					2 * cbf_cond(c, mb->mbAddrB, TB_DC, comp, 0));
		CHK(c_block(c, cat0, mb->I16DC[comp], 16, inc, &coded));
		if (coded)
			c->info->cbf_dc |= (uint8_t)(1u << comp);
	}
	for (i8 = 0; i8 < 4; i8++) {
		if (mb->transform_size_8x8_flag) {
			int32_t inc = -1;
			if (!((mb->CodedBlockPatternLuma >> i8) & 1))
				continue;
			if (c->cat == 3) {
				uint32_t xW, yW, ca, cb;
				orc_neigh_loc(c->ctx, mb, 16, 16,
					      (int32_t)(i8 % 2) * 8 - 1,
					      (int32_t)(i8 / 2) * 8, &addrA, &xW, &yW);
				ca = cbf_cond(c, addrA, TB_8x8, comp,
					      2 * (yW / 8) + xW / 8);
				orc_neigh_loc(c->ctx, mb, 16, 16, (int32_t)(i8 % 2) * 8,
					      (int32_t)(i8 / 2) * 8 - 1, &addrB, &xW,
					      &yW);
				cb = cbf_cond(c, addrB, TB_8x8, comp,
					      2 * (yW / 8) + xW / 8);
				inc = (int32_t)(ca + 2 * cb);
			}
			CHK(c_block(c, cat8, mb->Lvl8x8[comp][i8], 64, inc, &coded));
			if (coded)
				c->info->cbf[comp] |= (uint16_t)(0xfu << (4 * i8));
			/* the record keeps the 4x4 view CAVLC reads
			 * (ref src/h264_syntax_slice_data.h:323-326) */
			for (k = 0; k < 64; k++)
				mb->Lvl4x4[comp][4 * i8 + (k & 3)][k >> 2] =
					mb->Lvl8x8[comp][i8][k];
			continue;
		}
		for (i4 = 0; i4 < 4; i4++) {
			const uint32_t b = 4 * i8 + i4;
			const uint32_t cat = cat0 + (i16 ? 1 : 2);
			int32_t inc;
			if (!((mb->CodedBlockPatternLuma >> i8) & 1))
				continue;
			orc_neigh_luma4x4(c->ctx, mb, b, &addrA, &idxA, &addrB, &idxB);
			inc = (int32_t)(cbf_cond(c, addrA, TB_4x4, comp, idxA) +
					2 * cbf_cond(c, addrB, TB_4x4, comp, idxB));
			if (i16)
				CHK(c_block(c, cat, mb->I16AC[comp][b], 15, inc, &coded));
			else
				CHK(c_block(c, cat, mb->Lvl4x4[comp][b], 16, inc, &coded));
			if (coded)
				c->info->cbf[comp] |= (uint16_t)(1u << b);
		}
	}
	return 0;
}

/* 7.3.5.3 */
static int c_residual(struct cd *c)
{
	struct orc_mb *mb = c->mb;
	uint32_t coded = 0;
	CHK(c_residual_luma(c, 0));
	if (c->cat == 1 || c->cat == 2) {
		const uint32_t n = 4 * (c->cat == 1 ? 1 : 2); /* 4 * NumC8x8 */
		uint32_t cc, b;
		if (mb->CodedBlockPatternChroma & 3)
			for (cc = 0; cc < 2; cc++) {
				int32_t inc = (int32_t)(
					cbf_cond(c, mb->mbAddrA, TB_CDC, cc, 0) +
					2 * cbf_cond(c, mb->mbAddrB, TB_CDC, cc, 0));
				CHK(c_block(c, 3, mb->ChromaDC[cc], n, inc, &coded));
				if (coded)
					c->info->cbf_dc |= (uint8_t)(2u << cc);
			}
		if (mb->CodedBlockPatternChroma & 2)
			for (cc = 0; cc < 2; cc++)
				for (b = 0; b < n; b++) {
					uint32_t addrA, idxA, addrB, idxB;
					int32_t inc;
					orc_neigh_chroma4x4(c->ctx, mb, b, &addrA, &idxA,
							    &addrB, &idxB);
					inc = (int32_t)(
						cbf_cond(c, addrA, TB_CAC, cc, idxA) +
						2 * cbf_cond(c, addrB, TB_CAC, cc, idxB));
					CHK(c_block(c, 4, mb->ChromaAC[cc][b], 15, inc,
						    &coded));
					if (coded)
						c->info->cbf[1 + cc] |= (uint16_t)(1u << b);
				}
	} else if (c->cat == 3) {
		CHK(c_residual_luma(c, 1));
		CHK(c_residual_luma(c, 2));
	}
	return 0;
}

/* nz_coeff of a CABAC macroblock: non-zero coefficients of each 4x4 array of
 * the record, DC blocks not counted (h264_amd.h) */
static void count_nz(struct cd *c)
{
	const struct orc_mb *mb = c->mb;
	const int i16 = mb->MbPartPredMode[0] == ORC_PRED_Intra_16x16;
	uint32_t comp, b, k;
	memset(c->info->nz_coeff, 0, 48);
	for (comp = 0; comp < (c->cat == 3 ? 3u : 1u); comp++)
		for (b = 0; b < 16; b++)
			for (k = 0; k < (i16 ? 15u : 16u); k++)
				c->info->nz_coeff[16 * comp + b] +=
					(i16 ? mb->I16AC[comp][b][k]
					     : mb->Lvl4x4[comp][b][k]) != 0;
	if (c->cat == 1 || c->cat == 2)
		for (comp = 0; comp < 2; comp++)
			for (b = 0; b < (c->cat == 1 ? 4u : 8u); b++)
				for (k = 0; k < 15; k++)
					c->info->nz_coeff[16 * (1 + comp) + b] +=
						mb->ChromaAC[comp][b][k] != 0;
}

/* ---- 7.3.5 macroblock_layer --------------------------------------------------- */
static int c_macroblock_layer(struct cd *c)
{
	struct orc_ctx *ctx = c->ctx;
	struct orc_mb *mb = c->mb;
	const int t8x8 = ctx->pps->transform_8x8_mode_flag;
	const int d8x8 = ctx->sps->direct_8x8_inference_flag;
	uint32_t type = 0, i, cc, v = 0;
	int noSub8x8 = 1;

	CHK(d_mb_type(c, &type));
	CHK(orc_apply_mb_type(ctx, mb, type));
	c->info->mbt = (uint8_t)mb->mb_type;
	if (mb->mb_type == H264_MB_TYPE_I_PCM) {
		/* the terminating bin of mb_type ended the arithmetic codeword;
		 * pcm_alignment_zero_bit, the samples, and a fresh start of the
		 * decoding engine behind them (9.3.1.2) */
		while (c->bs->cachebits % 8 != 0) {
			CHK(orc_bs_read_bits(c->bs, &v, 1) < 0 ? -EIO : 0);
			if (v != 0)
				return -EIO;
		}
		for (i = 0; i < 256; i++) {
			CHK(orc_bs_read_bits(c->bs, &v, ctx->sps_derived.BitDepthLuma) < 0
				    ? -EIO
				    : 0);
			mb->pcm_sample_luma[i] = (uint16_t)v;
		}
		for (cc = 0; cc < 2; cc++)
			for (i = 0; i < ctx->sps_derived.MbWidthC *
						ctx->sps_derived.MbHeightC;
			     i++) {
				CHK(orc_bs_read_bits(c->bs, &v,
						     ctx->sps_derived.BitDepthChroma) < 0
					    ? -EIO
					    : 0);
				mb->pcm_sample_chroma[cc][i] = (uint16_t)v;
			}
		memset(c->info->nz_coeff, 16, 48);
		c->info->cbp = 0x2f;
		c->info->cbf_dc = 7;
		c->info->cbf[0] = c->info->cbf[1] = c->info->cbf[2] = 0xffff;
		c->prev_qpd_nz = 0;
		return d_init(c);
	}

	if (mb->mb_type != H264_MB_TYPE_I_NxN &&
	    mb->MbPartPredMode[0] != ORC_PRED_Intra_16x16 && mb->NumMbPart == 4) {
		uint32_t p;
		CHK(c_sub_mb_pred(c));
		for (p = 0; p < 4; p++) {
			if (mb->sub_mb_type[p] != ORC_SUB_B_Direct_8x8) {
				if (mb->NumSubMbPart[p] > 1)
					noSub8x8 = 0;
			} else if (!d8x8) {
				noSub8x8 = 0;
			}
		}
	} else {
		if (t8x8 && mb->mb_type == H264_MB_TYPE_I_NxN) {
			CHK(d_t8(c));
			if (mb->transform_size_8x8_flag)
				mb->MbPartPredMode[0] = ORC_PRED_Intra_8x8;
		}
		CHK(c_mb_pred(c));
	}
	if (mb->MbPartPredMode[0] != ORC_PRED_Intra_16x16) {
		CHK(d_cbp(c));
		if (mb->CodedBlockPatternLuma > 0 && t8x8 &&
		    mb->mb_type != H264_MB_TYPE_I_NxN && noSub8x8 &&
		    (mb->mb_type != H264_MB_TYPE_B_Direct_16x16 || d8x8))
			CHK(d_t8(c));
	}
	c->info->cbp = (uint8_t)(mb->CodedBlockPatternLuma |
				 (mb->CodedBlockPatternChroma << 4));
	if (mb->CodedBlockPatternLuma > 0 || mb->CodedBlockPatternChroma > 0 ||
	    mb->MbPartPredMode[0] == ORC_PRED_Intra_16x16) {
		CHK(d_qpd(c));
		CHK(c_residual(c));
	} else {
		c->prev_qpd_nz = 0;
	}
	count_nz(c);
	return 0;
}

/* ---- 7.3.4 slice_data with CABAC ---------------------------------------------- */
int orc_cabac_slice_data(struct h264_bitstream *bs, struct orc_reader *rd)
{
	struct orc_ctx *ctx = rd->ctx;
	const struct h264_slice_header *sh = &ctx->slice.hdr;
	const uint32_t type = (uint32_t)ctx->slice.type;
	struct cd cds;
	struct cd *c = &cds;
	uint8_t init[1024][2];
	uint32_t addr, mb_count = 0, i, v = 0, bin = 0;

	if (ctx->derived.MbaffFrameFlag || ctx->pps->num_slice_groups_minus1 > 0)
		return H264_AMD_E_UNSUPPORTED;
	if (ctx->sps_derived.PicWidthInMbs == 0 ||
	    ctx->sps_derived.PicWidthInMbs > H264_AMD_MAX_CABAC_WIDTH_MBS)
		return H264_AMD_E_UNSUPPORTED;
	tables_init();
	if (rd->cbs.slice_data_begin)
		rd->cbs.slice_data_begin(ctx, sh, rd->userdata);
	orc_clear_macroblock_table(ctx);
	while (bs->cachebits % 8 != 0) { /* cabac_alignment_one_bit */
		CHK(orc_bs_read_bits(bs, &v, 1) < 0 ? -EIO : 0);
		if (v != 1)
			return -EIO;
	}
	memset(c, 0, sizeof(*c));
	c->bs = bs;
	c->ctx = ctx;
	c->field = sh->field_pic_flag ? 1 : 0;
	c->cat = ctx->sps_derived.ChromaArrayType;
	orc_cabac_init_states_floor((int)type, sh->cabac_init_idc,
				    ctx->derived.SliceQPLuma,
				    (rd->flags & H264_AMD_READER_FLAGS_CABAC_STD_QP) ? 0 : 1,
				    init);
	for (i = 0; i < 1024; i++) {
		c->st[i].idx = init[i][0];
		c->st[i].mps = init[i][1];
	}
	CHK(d_init(c));

	addr = sh->first_mb_in_slice;
	for (;;) {
		if (mb_count >= H264_AMD_MAX_SLICE_MBS)
			return H264_AMD_E_MB_OVERFLOW;
		CHK(orc_new_macroblock(ctx, addr, 0, -1));
		c->mb = ctx->mb;
		c->info = &ctx->slice.mb_table.info[orc_mb_off(ctx, addr)];
		if (type != H264_SLICE_TYPE_I && type != H264_SLICE_TYPE_SI) {
			/* mb_skip_flag (9.3.3.1.1.1) */
			const struct orc_mb_info *a = info_of(c, c->mb->mbAddrA);
			const struct orc_mb_info *b = info_of(c, c->mb->mbAddrB);
			CHK(dd(c,
			       (type == H264_SLICE_TYPE_B ? 24 : 11) +
				       (a != NULL && !a->skipped) +
				       (b != NULL && !b->skipped),
			       &bin));
		} else {
			bin = 0;
		}
		if (bin) {
			CHK(orc_new_macroblock(ctx, addr, 1, -1));
			c->mb = ctx->mb;
			c->info->mbt = (uint8_t)c->mb->mb_type;
			c->prev_qpd_nz = 0;
		} else {
			CHK(c_macroblock_layer(c));
		}
		if (rd->cbs.slice_data_mb)
			rd->cbs.slice_data_mb(ctx, sh, c->mb->mbAddr, c->mb->mb_type,
					      rd->userdata);
		addr++;
		mb_count++;
		CHK(d_term(c, &bin)); /* end_of_slice_flag */
		if (bin)
			break;
	}
	if (rd->cbs.slice_data_end)
		rd->cbs.slice_data_end(ctx, sh, mb_count, rd->userdata);
	return 0;
}

/**
 * oracle/orc_cabac.c -- TEST INFRASTRUCTURE (see orc.h).
 * Restatement of the CABAC pieces the reference actually has -- the ENCODER
 * (there is no CABAC decoder in the reference, SURVEY.md finding 1) -- and of
 * the synthetic-slice writers built on it:
 *   src/h264_bac.c:150-210   PutBit / renormalisation / flush
 *   src/h264_bac.c:216-230   context-variable initialisation (9.3.1.1)
 *   src/h264_bac.c:253-358   encode init / decision / bypass / terminate
 *   src/h264_cabac.c:42-220  binstrings, ctxIdxInc derivations (with the
 *                            reference's shortcuts, Appendix C.14)
 *   src/h264_cabac.c:230-470 ctxIdx assignment (subset used by the writers)
 *   src/h264_cabac.c:568-975 U/TU binarisation, the six write_* elements
 *   src/h264_cabac_ctx_tables.c:4441-4468 table selection
 *   src/h264_writer.c:49-370 grey-I / skipped-P slices, header rewrite
 */
#include "orc.h"

#define CHK(x)                                                                 \
	do {                                                                   \
		int _r = (x);                                                  \
		if (_r < 0)                                                    \
			return _r;                                             \
	} while (0)

/* ---- constants --------------------------------------------------------- */
static uint8_t s_lps[64][4];
static uint8_t s_next_lps[64], s_next_mps[64];
static int8_t s_mn[1024][4][2];
static int s_ready;

static void cabac_tables_init(void)
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
#define H264_MN(i, mi, ni, m0, n0, m1, n1, m2, n2)                             \
	s_mn[i][0][0] = mi;                                                    \
	s_mn[i][0][1] = ni;                                                    \
	s_mn[i][1][0] = m0;                                                    \
	s_mn[i][1][1] = n0;                                                    \
	s_mn[i][2][0] = m1;                                                    \
	s_mn[i][2][1] = n1;                                                    \
	s_mn[i][3][0] = m2;                                                    \
	s_mn[i][3][1] = n2;
#include "../libh264_amd/csrc/common/h264_cabac_tables.inc"
#undef H264_LPS
#undef H264_TRANS
#undef H264_MN
	s_ready = 1;
}

struct bac_state {
	uint8_t idx, mps;
};

struct bac_enc {
	struct h264_bitstream *bs;
	uint32_t range, low;
	int first_bit;
	uint32_t outstanding;
};

struct cabac {
	struct bac_enc enc;
	struct bac_state st[1024];
};

/* ref src/h264_bac.c:216-230 (qp clipped to 1..51, not 0..51: qp_floor 1;
 * the standard's 9.3.1.1 is qp_floor 0, which only the CABAC decoder -- an
 * extension -- can be asked for) */
static void state_init(struct bac_state *s, int32_t qp, int32_t qp_floor,
		       int8_t m, int8_t n)
{
	int32_t idx;
	qp = qp < qp_floor ? qp_floor : qp > 51 ? 51 : qp;
	idx = ((m * qp) >> 4) + n;
	if (idx <= 63) {
		s->idx = 63 - (idx > 1 ? idx : 1);
		s->mps = 0;
	} else {
		s->idx = (idx < 126 ? idx : 126) - 64;
		s->mps = 1;
	}
}

/* table column: 0 for I/SI, 1 + cabac_init_idc otherwise; an idc above 2
 * leaves the states untouched in the reference (zeroed by its caller) */
void orc_cabac_init_states(int slice_type, uint32_t cabac_init_idc,
			   int32_t SliceQPLuma, uint8_t states[1024][2])
{
	orc_cabac_init_states_floor(slice_type, cabac_init_idc, SliceQPLuma, 1,
				    states);
}

void orc_cabac_init_states_floor(int slice_type, uint32_t cabac_init_idc,
				 int32_t SliceQPLuma, int32_t qp_floor,
				 uint8_t states[1024][2])
{
	uint32_t col, i;
	cabac_tables_init();
	memset(states, 0, 2048);
	if (slice_type == H264_SLICE_TYPE_I || slice_type == H264_SLICE_TYPE_SI)
		col = 0;
	else if (cabac_init_idc <= 2)
		col = 1 + cabac_init_idc;
	else
		return;
	for (i = 0; i < 1024; i++) {
		struct bac_state s;
		state_init(&s, SliceQPLuma, qp_floor, s_mn[i][col][0],
			   s_mn[i][col][1]);
		states[i][0] = s.idx;
		states[i][1] = s.mps;
	}
}

/* ---- arithmetic encoder (ref src/h264_bac.c:150-210,253-358) ----------- */
static int put_bit(struct bac_enc *e, int bit)
{
	if (e->first_bit)
		e->first_bit = 0;
	else
		CHK(orc_bs_write_bits(e->bs, bit, 1));
	while (e->outstanding > 0) {
		CHK(orc_bs_write_bits(e->bs, !bit, 1));
		e->outstanding--;
	}
	return 0;
}

static int renorm(struct bac_enc *e)
{
	while (e->range < 256) {
		if (e->low < 256) {
			CHK(put_bit(e, 0));
		} else if (e->low < 512) {
			e->low -= 256;
			e->outstanding++;
		} else {
			e->low -= 512;
			CHK(put_bit(e, 1));
		}
		e->range <<= 1;
		e->low <<= 1;
	}
	return 0;
}

static int enc_decision(struct bac_enc *e, struct bac_state *s, int bin)
{
	uint32_t q = (e->range >> 6) & 3;
	uint32_t r_lps = s_lps[s->idx][q];
	e->range -= r_lps;
	bin = !!bin;
	if (bin == s->mps) {
		s->idx = s_next_mps[s->idx];
	} else {
		e->low += e->range;
		e->range = r_lps;
		if (s->idx == 0)
			s->mps = !s->mps;
		s->idx = s_next_lps[s->idx];
	}
	return renorm(e);
}

static int enc_terminate(struct bac_enc *e, int bin)
{
	e->range -= 2;
	if (!bin)
		return renorm(e);
	e->low += e->range;
	/* flush: the final two bits carry the rbsp_stop_one_bit */
	e->range = 2;
	CHK(renorm(e));
	CHK(put_bit(e, (e->low >> 9) & 1));
	return orc_bs_write_bits(e->bs, ((e->low >> 7) & 3) | 1, 2);
}

static void cabac_init_enc(struct cabac *c, struct orc_ctx *ctx,
			   struct h264_bitstream *bs)
{
	uint8_t st[1024][2];
	uint32_t i;
	memset(c, 0, sizeof(*c));
	c->enc.bs = bs;
	c->enc.low = 0;
	c->enc.range = 510;
	c->enc.first_bit = 1;
	c->enc.outstanding = 0;
	orc_cabac_init_states(ctx->slice.type, ctx->slice.hdr.cabac_init_idc,
			      ctx->derived.SliceQPLuma, st);
	for (i = 0; i < 1024; i++) {
		c->st[i].idx = st[i][0];
		c->st[i].mps = st[i][1];
	}
}

/* ---- binarisation + ctxIdx (ref src/h264_cabac.c) ---------------------- */
struct binstring {
	uint8_t value, numbits;
	uint16_t ctxIdxOffset;
};

static int bin_at(const struct binstring *b, uint32_t idx)
{
	return (b->value >> (b->numbits - idx - 1)) & 1;
}

static int is_inter(uint32_t t)
{
	return t >= H264_MB_TYPE_P_16x16 && t <= H264_MB_TYPE_B_SKIP;
}

static int is_intra(uint32_t t)
{
	return t >= H264_MB_TYPE_I_NxN && t <= H264_MB_TYPE_SI;
}

static uint32_t inc_skip(const struct orc_mb_info *i)
{
	return (i == NULL || i->skipped) ? 0 : 1;
}

static uint32_t inc_mb_type(const struct orc_mb_info *i, uint32_t off)
{
	if (i == NULL)
		return 0;
	if (off == 0 && i->mb_type == H264_MB_TYPE_SI)
		return 0;
	if (off == 3 && i->mb_type == H264_MB_TYPE_I_NxN)
		return 0;
	/* the reference also tests for B_SKIP (16) here, which its 4-bit
	 * mb_type field can never hold (Appendix C.3): dead comparison */
	if (off == 27 && i->mb_type == H264_MB_TYPE_B_Direct_16x16)
		return 0;
	return 1;
}

static uint32_t inc_chroma_pred(const struct orc_mb_info *i)
{
	if (i == NULL || is_inter(i->mb_type) ||
	    i->mb_type == H264_MB_TYPE_I_PCM || i->intra_chroma_pred_mode == 0)
		return 0;
	return 1;
}

/* ref src/h264_cabac.c:230-470, the offsets the writers reach */
static uint32_t ctx_idx(const struct orc_mb *mb, const struct binstring *b,
			uint32_t binIdx)
{
	uint32_t off = b->ctxIdxOffset;
	switch (off) {
	case 3: /* mb_type, I slices */
		switch (binIdx) {
		case 0:
			return off + inc_mb_type(mb->infoA, off) +
			       inc_mb_type(mb->infoB, off);
		case 1:
			return 276;
		case 2:
			return off + 3;
		case 3:
			return off + 4;
		case 4:
			return off + (bin_at(b, 3) ? 5 : 6);
		case 5:
			return off + (bin_at(b, 3) ? 6 : 7);
		default:
			return 7;
		}
	case 11: /* mb_skip_flag P/SP */
	case 24: /* mb_skip_flag B */
		if (binIdx == 0)
			return off + inc_skip(mb->infoA) + inc_skip(mb->infoB);
		break;
	case 60: /* mb_qp_delta: ctxIdxInc hard-wired to 0 for bin 0 */
		return binIdx == 0 ? off : binIdx == 1 ? off + 2 : off + 3;
	case 64: /* intra_chroma_pred_mode */
		if (binIdx == 0)
			return off + inc_chroma_pred(mb->infoA) +
			       inc_chroma_pred(mb->infoB);
		if (binIdx <= 2)
			return off + 3;
		break;
	case 276:
		if (binIdx == 0)
			return off;
		break;
	default:
		break;
	}
	return 0;
}

static int write_binstring(struct cabac *c, const struct binstring *b,
			   const struct orc_mb *mb)
{
	uint32_t i;
	for (i = 0; i < b->numbits; i++) {
		uint32_t ci = ctx_idx(mb, b, i);
		if (ci == 276)
			CHK(enc_terminate(&c->enc, bin_at(b, i)));
		else
			CHK(enc_decision(&c->enc, &c->st[ci], bin_at(b, i)));
	}
	return 0;
}

static void binarize_u(uint32_t val, struct binstring *b)
{
	b->value = (uint8_t)(((1u << val) - 1) << 1);
	b->numbits = (uint8_t)(val + 1);
}

static void binarize_tu(uint32_t cmax, uint32_t val, struct binstring *b)
{
	if (val < cmax) {
		binarize_u(val, b);
	} else {
		b->value = (uint8_t)((1u << val) - 1);
		b->numbits = (uint8_t)val;
	}
}

/* Table 9-36 binarisation of I-slice mb_type: value, bit count.
 * 0 -> "0"; 25 -> "11"; 1..24 -> "1 0" + cbpLuma!=0 + chroma TU + 2-bit mode */
static void binarize_mb_type_i(uint32_t t, struct binstring *b)
{
	if (t == 0) {
		b->value = 0;
		b->numbits = 1;
	} else if (t == 25) {
		b->value = 3;
		b->numbits = 2;
	} else {
		uint32_t k = t - 1, mode = k % 4, chroma = (k / 4) % 3,
			 luma = k / 12;
		uint32_t v = (1u << 2) | luma; /* bins "1", "0", luma != 0 */
		uint32_t n = 3;
		if (chroma == 0) {
			v = (v << 1) | 0;
			n += 1;
		} else {
			v = (v << 2) | 2 | (chroma == 2 ? 1 : 0);
			n += 2;
		}
		v = (v << 2) | mode;
		n += 2;
		b->value = (uint8_t)v;
		b->numbits = (uint8_t)n;
	}
}

/* ref src/h264_cabac.c:630-785 -- only the I-slice branch is reachable from
 * the grey-I writer */
static int write_mb_type(struct cabac *c, struct orc_ctx *ctx,
			 const struct orc_mb *mb)
{
	struct binstring b;
	if (ctx->slice.type != H264_SLICE_TYPE_I)
		return -EIO;
	if (mb->raw_mb_type > 25)
		return -EIO;
	binarize_mb_type_i(mb->raw_mb_type, &b);
	b.ctxIdxOffset = 3;
	return write_binstring(c, &b, mb);
}

/* ref src/h264_cabac.c:853-925: coded_block_flag, ctxBlockCat 0 only.
 * transBlock availability is hard-wired off in the reference, which leaves
 *   condTerm = 1 if neighbour MB missing and current MB intra,
 *              1 if neighbour is I_PCM, else 0 */
static uint32_t cbf_cond(const struct orc_mb *mb, const struct orc_mb_info *i)
{
	if (i == NULL && is_inter(mb->mb_type))
		return 0;
	if (i != NULL && i->mb_type != H264_MB_TYPE_I_PCM)
		return 0;
	if (i == NULL && is_intra(mb->mb_type))
		return 1;
	if (i != NULL && i->mb_type == H264_MB_TYPE_I_PCM)
		return 1;
	return 0;
}

static int write_coded_block_flag_i16dc(struct cabac *c,
					const struct orc_mb *mb, int flag)
{
	uint32_t ci = 85 + 0 + cbf_cond(mb, mb->infoA) +
		      2 * cbf_cond(mb, mb->infoB);
	return enc_decision(&c->enc, &c->st[ci], flag);
}

/* ---- synthetic slices (ref src/h264_writer.c:49-309) ------------------- */
static int setup_grey_mb(struct orc_ctx *ctx, uint32_t i)
{
	struct orc_mb *mb;
	CHK(orc_new_macroblock(ctx, ctx->slice.hdr.first_mb_in_slice + i, 0, -1));
	mb = ctx->mb;
	mb->raw_mb_type = 3; /* I_16x16_2_0_0 */
	mb->mb_type = H264_MB_TYPE_I_16x16;
	mb->NumMbPart = 1;
	mb->MbPartPredMode[0] = ORC_PRED_Intra_16x16;
	mb->intra_chroma_pred_mode = 0;
	ctx->slice.mb_table.info[i].mb_type = mb->mb_type;
	ctx->slice.mb_table.info[i].intra_chroma_pred_mode = 0;
	return 0;
}

static int align_with(struct h264_bitstream *bs, int bit)
{
	while (bs->cachebits % 8 != 0)
		CHK(orc_bs_write_bits(bs, bit, 1));
	return 0;
}

static int grey_i_cabac(struct h264_bitstream *bs, struct orc_ctx *ctx,
			uint32_t mb_count)
{
	struct cabac c;
	struct binstring b;
	uint32_t i;
	CHK(align_with(bs, 1)); /* cabac_alignment_one_bit */
	cabac_init_enc(&c, ctx, bs);
	for (i = 0; i < mb_count; i++) {
		CHK(setup_grey_mb(ctx, i));
		CHK(write_mb_type(&c, ctx, ctx->mb));
		binarize_tu(3, ctx->mb->intra_chroma_pred_mode, &b);
		b.ctxIdxOffset = 64;
		CHK(write_binstring(&c, &b, ctx->mb));
		binarize_u(0, &b); /* mb_qp_delta = 0 */
		b.ctxIdxOffset = 60;
		CHK(write_binstring(&c, &b, ctx->mb));
		CHK(write_coded_block_flag_i16dc(&c, ctx->mb, 0));
		CHK(enc_terminate(&c.enc, i == mb_count - 1));
	}
	return align_with(bs, 0);
}

static int grey_i_cavlc(struct h264_bitstream *bs, uint32_t mb_count)
{
	uint32_t i;
	for (i = 0; i < mb_count; i++) {
		CHK(orc_bs_write_ue(bs, 3)); /* mb_type I_16x16_2_0_0 */
		CHK(orc_bs_write_ue(bs, 0)); /* intra_chroma_pred_mode DC */
		CHK(orc_bs_write_se(bs, 0)); /* mb_qp_delta */
		CHK(orc_bs_write_bits(bs, 1, 1)); /* coeff_token 0/0, nC 0 */
	}
	return orc_bs_write_trailing(bs);
}

static int skip_p_cabac(struct h264_bitstream *bs, struct orc_ctx *ctx,
			uint32_t mb_count)
{
	struct cabac c;
	struct binstring b;
	uint32_t i;
	CHK(align_with(bs, 1));
	cabac_init_enc(&c, ctx, bs);
	for (i = 0; i < mb_count; i++) {
		CHK(orc_new_macroblock(
			ctx, ctx->slice.hdr.first_mb_in_slice + i, 1, -1));
		if (ctx->slice.type == H264_SLICE_TYPE_P ||
		    ctx->slice.type == H264_SLICE_TYPE_SP)
			b.ctxIdxOffset = 11;
		else if (ctx->slice.type == H264_SLICE_TYPE_B)
			b.ctxIdxOffset = 24;
		else
			return -EIO;
		b.value = 1;
		b.numbits = 1;
		CHK(write_binstring(&c, &b, ctx->mb));
		CHK(enc_terminate(&c.enc, i == mb_count - 1));
	}
	return align_with(bs, 0);
}

int orc_write_grey_i_slice(struct h264_bitstream *bs, struct orc_ctx *ctx,
			   uint32_t mb_count)
{
	if (bs == NULL || ctx == NULL || mb_count == 0)
		return -EINVAL;
	cabac_tables_init();
	CHK(orc_ctx_set_active_pps(ctx, ctx->slice.hdr.pic_parameter_set_id));
	CHK(orc_write_nalu(bs, ctx));
	if (ctx->pps->entropy_coding_mode_flag)
		return grey_i_cabac(bs, ctx, mb_count);
	return grey_i_cavlc(bs, mb_count);
}

int orc_write_skipped_p_slice(struct h264_bitstream *bs, struct orc_ctx *ctx,
			      uint32_t mb_count)
{
	if (bs == NULL || ctx == NULL || mb_count == 0)
		return -EINVAL;
	cabac_tables_init();
	CHK(orc_ctx_set_active_pps(ctx, ctx->slice.hdr.pic_parameter_set_id));
	CHK(orc_write_nalu(bs, ctx));
	if (ctx->pps->entropy_coding_mode_flag)
		return skip_p_cabac(bs, ctx, mb_count);
	CHK(orc_bs_write_ue(bs, mb_count)); /* one mb_skip_run */
	return orc_bs_write_trailing(bs);
}

/* ref src/h264_writer.c:312-370 */
int orc_rewrite_slice_header(struct h264_bitstream *bs, struct orc_ctx *ctx,
			     const struct h264_slice_header *sh)
{
	struct h264_bitstream tmp;
	uint8_t data[64];
	size_t saved_len;
	int res;
	if (bs == NULL || ctx == NULL || sh == NULL)
		return -EINVAL;
	h264_bs_init(&tmp, data, sizeof(data), 1);
	ctx->slice.saved_hdr = ctx->slice.hdr;
	saved_len = ctx->slice.hdr_len;
	CHK(orc_ctx_set_slice_header(ctx, sh));
	res = orc_write_nalu(&tmp, ctx);
	if (res >= 0 && ctx->slice.hdr_len != saved_len)
		res = -EPROTO;
	if (res < 0) {
		ctx->slice.hdr = ctx->slice.saved_hdr;
		ctx->slice.hdr_len = saved_len;
		return res;
	}
	memcpy(bs->data, tmp.data, tmp.off);
	if (tmp.cachebits != 0) {
		uint8_t mask = (uint8_t)((1u << (8 - tmp.cachebits)) - 1);
		bs->data[tmp.off] = (uint8_t)((tmp.cache & ~mask) |
					      (bs->data[tmp.off] & mask));
	}
	return 0;
}

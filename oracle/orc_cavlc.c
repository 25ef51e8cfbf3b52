/**
 * oracle/orc_cavlc.c -- TEST INFRASTRUCTURE (see orc.h).
 * Bit-serial restatement of the reference's CAVLC slice_data path:
 *   src/h264_slice_data.c:33-72      prefix-code reader, one bit per step
 *   src/h264_slice_data.c:839-1080   mb_type / sub_mb_type / coded_block_pattern
 *   src/h264_slice_data.c:1083-1232  macroblock table, nz_coeff bookkeeping
 *   src/h264_slice_data.c:1239-1416  coeff_token (nC rule), total_zeros, run_before
 *   src/h264_macroblock.c:35-104,275-433  neighbour derivation (non-MBAFF)
 *   src/h264_syntax_slice_data.h:103-419  residual_block / residual_luma / residual
 *   src/h264_syntax_slice_data.h:422-696  sub_mb_pred / mb_pred / macroblock_layer
 *   src/h264_syntax_slice_data.h:701-811  slice_data loop, rawdata extent
 *   src/h264_fmo.c:307-319           next_mb_addr without slice groups
 * MBAFF frames and slice groups (FMO) are outside the engine's scope: both
 * this restatement and the kernels report H264_AMD_E_UNSUPPORTED for them.
 */
#include "orc.h"

#define CHK(x)                                                                 \
	do {                                                                   \
		int _r = (x);                                                  \
		if (_r < 0)                                                    \
			return _r;                                             \
	} while (0)

/* ---- prefix-code tables built from the standard's bit strings ---------- */

struct vlc {
	uint8_t maxbits, maxcode;
	uint8_t hit[17][16]; /* [numbits][rawcode] = 0x80 | value, 0 = none */
};

/* test instrumentation: coefficients dropped because a damaged stream placed
 * them outside their block (see residual_block) */
uint64_t orc_stat_dropped_coeffs;

static struct vlc s_ct[6]; /* coeff_token columns (col 3 = 6-bit FLC) */
static struct vlc s_tz[16], s_tzc420[4], s_tzc422[8], s_rb[8];
static uint8_t s_cbp[2][48][2];
static int s_tables_ready;

static void vlc_add(struct vlc *t, const char *bits, uint8_t val)
{
	uint32_t n = 0, code = 0;
	for (; *bits; bits++) {
		if (*bits == '0' || *bits == '1') {
			code = (code << 1) | (uint32_t)(*bits - '0');
			n++;
		}
	}
	t->hit[n][code] = 0x80 | val;
}

static void tables_init(void)
{
	int i;
	if (s_tables_ready)
		return;
	for (i = 0; i < 6; i++) {
		s_ct[i].maxbits = 16;
		s_ct[i].maxcode = 15;
	}
	for (i = 0; i < 16; i++) {
		s_tz[i].maxbits = 9;
		s_tz[i].maxcode = 7;
	}
	for (i = 0; i < 4; i++) {
		s_tzc420[i].maxbits = 3;
		s_tzc420[i].maxcode = 1;
	}
	for (i = 0; i < 8; i++) {
		s_tzc422[i].maxbits = 5;
		s_tzc422[i].maxcode = 7;
		s_rb[i].maxbits = 11;
		s_rb[i].maxcode = 7;
	}
#define H264_CT(col, bits, t1, tc)                                             \
	if ((col) != 3)                                                        \
		vlc_add(&s_ct[col], bits, (uint8_t)(((t1) << 5) | (tc)));
#define H264_TZ(idx, bits, v) vlc_add(&s_tz[idx], bits, v);
#define H264_TZ_C420(idx, bits, v) vlc_add(&s_tzc420[idx], bits, v);
#define H264_TZ_C422(idx, bits, v) vlc_add(&s_tzc422[idx], bits, v);
#define H264_RB(idx, bits, v) vlc_add(&s_rb[idx], bits, v);
#define H264_CBP(cls, code, intra, inter)                                      \
	s_cbp[cls][code][0] = intra;                                           \
	s_cbp[cls][code][1] = inter;
#include "../libh264_amd/csrc/common/h264_cavlc_tables.inc"
#undef H264_CT
#undef H264_TZ
#undef H264_TZ_C420
#undef H264_TZ_C422
#undef H264_RB
#undef H264_CBP
	/* column 3 (nC >= 8) is a fixed 6-bit code: see flc_init() */
	s_tables_ready = 1;
}

static uint8_t s_ct_flc[64];
static void flc_init(void)
{
	static int done;
	if (done)
		return;
#define H264_CT(col, bits, t1, tc)                                             \
	if ((col) == 3) {                                                      \
		uint32_t c = 0;                                                \
		const char *p = bits;                                          \
		for (; *p; p++)                                                \
			if (*p == '0' || *p == '1')                            \
				c = (c << 1) | (uint32_t)(*p - '0');           \
		s_ct_flc[c] = (uint8_t)(0x80 | ((t1) << 5) | (tc));            \
	}
#include "../libh264_amd/csrc/common/h264_cavlc_tables.inc"
#undef H264_CT
	done = 1;
}

/* ref src/h264_slice_data.c:33-72 */
static int read_vlc(struct h264_bitstream *bs, const struct vlc *t,
		    uint32_t *val)
{
	uint32_t bit = 0, numbits = 0, raw = 0;
	while (numbits < t->maxbits) {
		CHK(orc_bs_read_bits(bs, &bit, 1));
		raw = (raw << 1) | bit;
		numbits++;
		if (raw > t->maxcode)
			return -EIO;
		if (t->hit[numbits][raw] != 0) {
			*val = t->hit[numbits][raw];
			return 0;
		}
	}
	return -EIO;
}

/* ---- macroblock table (ref src/h264_slice_data.c:1098-1232) ------------ */

uint32_t orc_mb_off(const struct orc_ctx *ctx, uint32_t mbAddr)
{
	return mbAddr - ctx->slice.hdr.first_mb_in_slice *
				(1 + ctx->derived.MbaffFrameFlag);
}

/* ref src/h264_macroblock.c:306-351 */
static void compute_neighbours(struct orc_ctx *ctx, struct orc_mb *mb)
{
	uint32_t W = ctx->sps_derived.PicWidthInMbs;
	uint32_t first = ctx->slice.hdr.first_mb_in_slice;
	mb->mbAddrA = mb->mbAddrB = ORC_MB_ADDR_INVALID;
	mb->infoA = mb->infoB = NULL;
	if (!ctx->derived.MbaffFrameFlag) {
		if (mb->mbAddr >= first + 1 && mb->mbAddr % W != 0)
			mb->mbAddrA = mb->mbAddr - 1;
		if (mb->mbAddr >= first + W)
			mb->mbAddrB = mb->mbAddr - W;
	} else {
		/* pairs: A / B name the TOP macroblock of the neighbouring pair
		 * (ref :327-335) */
		uint32_t half = mb->mbAddr / 2;
		if (half >= first + 1 && half % W != 0)
			mb->mbAddrA = 2 * (half - 1);
		if (half >= first + W)
			mb->mbAddrB = 2 * (half - W);
	}
	if (mb->mbAddrA != ORC_MB_ADDR_INVALID) {
		struct orc_mb_info *i =
			&ctx->slice.mb_table.info[orc_mb_off(ctx, mb->mbAddrA)];
		if (!i->available)
			mb->mbAddrA = ORC_MB_ADDR_INVALID;
		else
			mb->infoA = i;
	}
	if (mb->mbAddrB != ORC_MB_ADDR_INVALID) {
		struct orc_mb_info *i =
			&ctx->slice.mb_table.info[orc_mb_off(ctx, mb->mbAddrB)];
		if (!i->available)
			mb->mbAddrB = ORC_MB_ADDR_INVALID;
		else
			mb->infoB = i;
	}
}

int orc_new_macroblock(struct orc_ctx *ctx, uint32_t mbAddr, int skipped,
		       int field_flag)
{
	const struct h264_slice_header *sh = &ctx->slice.hdr;
	uint32_t off = orc_mb_off(ctx, mbAddr);
	struct orc_mb *mb;

	/* an MBAFF frame slice with first_mb_in_slice > 0 reaches here through the
	 * synthetic-slice writers with mbAddr below first_mb_in_slice * 2: the
	 * offset wraps and the reference asks realloc for ~200 GB, which fails
	 * on one machine and grinds for minutes on another.  Here: -ENOMEM. */
	if (off >= H264_AMD_MAX_SLICE_MBS) {
		/* under slice groups the address runs ahead of the count of
		 * macroblocks delivered: the slice ends like one that has
		 * delivered H264_AMD_MAX_SLICE_MBS (h264_amd.h) */
		uint32_t base = sh->first_mb_in_slice *
				(1u + (uint32_t)ctx->derived.MbaffFrameFlag);
		return mbAddr >= base ? H264_AMD_E_MB_OVERFLOW : -ENOMEM;
	}
	if ((size_t)off + 1 > ctx->slice.mb_table.maxlen) {
		size_t nl = ((size_t)off + 1 + 127) & ~(size_t)127;
		struct orc_mb_info *ni = realloc(ctx->slice.mb_table.info,
						 nl * sizeof(*ni));
		if (ni == NULL)
			return -ENOMEM;
		memset(ni + ctx->slice.mb_table.maxlen, 0,
		       (nl - ctx->slice.mb_table.maxlen) * sizeof(*ni));
		ctx->slice.mb_table.info = ni;
		ctx->slice.mb_table.maxlen = nl;
	}
	if ((size_t)off + 1 >= ctx->slice.mb_table.len)
		ctx->slice.mb_table.len = (size_t)off + 1;
	ctx->slice.mb_table.info[off].available = 1;
	ctx->slice.mb_table.info[off].skipped = skipped;

	mb = ctx->mb = &ctx->_mb;
	memset(mb, 0, sizeof(*mb));
	mb->mbAddr = mbAddr;
	mb->mb_type = !skipped ? H264_MB_TYPE_UNKNOWN
		      : ctx->slice.type == H264_SLICE_TYPE_B
			      ? H264_MB_TYPE_B_SKIP
			      : H264_MB_TYPE_P_SKIP;
	compute_neighbours(ctx, mb);
	/* field flag (ref src/h264_slice_data.c:1145-1194) */
	if (!ctx->derived.MbaffFrameFlag) {
		if (field_flag != -1)
			return -EINVAL;
		mb->mb_field_decoding_flag = sh->field_pic_flag;
	} else if (skipped) {
		struct orc_mb_info *info = ctx->slice.mb_table.info;
		if (mbAddr % 2 == 0) {
			/* waits for the bottom macroblock */
		} else if (!info[off - 1].skipped) {
			mb->mb_field_decoding_flag = info[off - 1].field_flag;
		} else {
			/* the whole pair is skipped: inferred from the pair to
			 * the left, else from the one above, else frame */
			if (mb->mbAddrA != ORC_MB_ADDR_INVALID)
				mb->mb_field_decoding_flag = mb->infoA->field_flag;
			else if (mb->mbAddrB != ORC_MB_ADDR_INVALID)
				mb->mb_field_decoding_flag = mb->infoB->field_flag;
			else
				mb->mb_field_decoding_flag = 0;
			info[off - 1].field_flag =
				(uint8_t)mb->mb_field_decoding_flag;
		}
	} else if (mbAddr % 2 == 0) {
		if (field_flag == -1)
			return -EINVAL;
		mb->mb_field_decoding_flag = field_flag;
	} else if (field_flag != -1) {
		/* bottom macroblock with a flag of its own: the top one was
		 * skipped and takes it too */
		mb->mb_field_decoding_flag = field_flag;
		if (!ctx->slice.mb_table.info[off - 1].skipped)
			return -EINVAL;
		ctx->slice.mb_table.info[off - 1].field_flag = (uint8_t)field_flag;
	} else {
		if (ctx->slice.mb_table.info[off - 1].skipped)
			return -EINVAL;
		mb->mb_field_decoding_flag =
			ctx->slice.mb_table.info[off - 1].field_flag;
	}
	ctx->slice.mb_table.info[off].field_flag =
		(uint8_t)mb->mb_field_decoding_flag;
	if (!ctx->derived.MbaffFrameFlag || !mb->mb_field_decoding_flag) {
		mb->max_ref_idx_0 = (uint8_t)sh->num_ref_idx_l0_active_minus1;
		mb->max_ref_idx_1 = (uint8_t)sh->num_ref_idx_l1_active_minus1;
	} else {
		/* uint8_t in the reference too (src/h264_macroblock.h) */
		mb->max_ref_idx_0 =
			(uint8_t)(2 * sh->num_ref_idx_l0_active_minus1 + 1);
		mb->max_ref_idx_1 =
			(uint8_t)(2 * sh->num_ref_idx_l1_active_minus1 + 1);
	}
	return 0;
}

static void set_nz(struct orc_ctx *ctx, uint32_t mbAddr, uint32_t comp,
		   uint32_t idx, uint32_t n)
{
	ctx->slice.mb_table.info[orc_mb_off(ctx, mbAddr)].nz_coeff[comp * 16 + idx] =
		n;
}

static uint32_t get_nz(struct orc_ctx *ctx, uint32_t mbAddr, uint32_t comp,
		       uint32_t idx)
{
	return ctx->slice.mb_table.info[orc_mb_off(ctx, mbAddr)]
		.nz_coeff[comp * 16 + idx];
}

/* ---- 4x4 block neighbours (ref src/h264_macroblock.c:35-104,275-433) --- */

/* 6.4.12.1 / 6.4.12.2 (ref src/h264_macroblock.c:84-231) */
void orc_neigh_loc(const struct orc_ctx *ctx, const struct orc_mb *mb,
		      uint32_t maxW, uint32_t maxH, int32_t xN, int32_t yN,
		      uint32_t *addrN, uint32_t *xW, uint32_t *yW)
{
	int32_t yM = yN;
	if (!ctx->derived.MbaffFrameFlag) {
		if (xN < 0)
			*addrN = mb->mbAddrA;
		else if (yN < 0)
			*addrN = mb->mbAddrB;
		else
			*addrN = mb->mbAddr;
	} else {
		int curFrame = !mb->mb_field_decoding_flag;
		int top = mb->mbAddr % 2 == 0;
		if (xN < 0) {
			int aFrame;
			if (mb->mbAddrA == ORC_MB_ADDR_INVALID) {
				*addrN = ORC_MB_ADDR_INVALID;
				*xW = *yW = 0;
				return;
			}
			aFrame = !mb->infoA->field_flag;
			if (curFrame) {
				if (top) {
					if (aFrame) {
						*addrN = mb->mbAddrA;
						yM = yN;
					} else {
						*addrN = mb->mbAddrA + (yN % 2 != 0);
						yM = yN >> 1;
					}
				} else {
					if (aFrame) {
						*addrN = mb->mbAddrA + 1;
						yM = yN;
					} else {
						*addrN = mb->mbAddrA + (yN % 2 != 0);
						yM = (yN + (int32_t)maxH) >> 1;
					}
				}
			} else {
				if (top) {
					if (aFrame) {
						if ((uint32_t)yN < maxH / 2) {
							*addrN = mb->mbAddrA;
							yM = yN << 1;
						} else {
							*addrN = mb->mbAddrA + 1;
							yM = (yN << 1) - (int32_t)maxH;
						}
					} else {
						*addrN = mb->mbAddrA;
						yM = yN;
					}
				} else {
					if (aFrame) {
						if ((uint32_t)yN < maxH / 2) {
							*addrN = mb->mbAddrA;
							yM = (yN << 1) + 1;
						} else {
							*addrN = mb->mbAddrA + 1;
							yM = (yN << 1) + 1 -
							     (int32_t)maxH;
						}
					} else {
						*addrN = mb->mbAddrA + 1;
						yM = yN;
					}
				}
			}
		} else if (yN < 0) {
			if (curFrame) {
				if (top) {
					if (mb->mbAddrB == ORC_MB_ADDR_INVALID) {
						*addrN = ORC_MB_ADDR_INVALID;
						*xW = *yW = 0;
						return;
					}
					*addrN = mb->mbAddrB + 1;
				} else {
					*addrN = mb->mbAddr - 1;
				}
				yM = yN;
			} else {
				int bFrame;
				if (mb->mbAddrB == ORC_MB_ADDR_INVALID) {
					*addrN = ORC_MB_ADDR_INVALID;
					*xW = *yW = 0;
					return;
				}
				bFrame = !mb->infoB->field_flag;
				if (top) {
					if (bFrame) {
						*addrN = mb->mbAddrB + 1;
						yM = 2 * yN;
					} else {
						*addrN = mb->mbAddrB;
						yM = yN;
					}
				} else {
					*addrN = mb->mbAddrB + 1;
					yM = yN;
				}
			}
		} else {
			*addrN = mb->mbAddr;
		}
	}
	*xW = xN < 0 ? (uint32_t)(xN + (int32_t)maxW) : (uint32_t)xN;
	*yW = yM < 0 ? (uint32_t)(yM + (int32_t)maxH) : (uint32_t)yM;
}

void orc_neigh_luma4x4(const struct orc_ctx *ctx, const struct orc_mb *mb,
			  uint32_t idx,
			  uint32_t *addrA, uint32_t *idxA, uint32_t *addrB,
			  uint32_t *idxB)
{
	/* 6.4.3 inverse scan and 6.4.13.1 index from position */
	static const uint8_t pos[16][2] = {
		{0, 0}, {4, 0},	 {0, 4},  {4, 4},  {8, 0}, {12, 0},
		{8, 4}, {12, 4}, {0, 8},  {4, 8},  {0, 12}, {4, 12},
		{8, 8}, {12, 8}, {8, 12}, {12, 12},
	};
	static const uint8_t at[4][4] = {
		{0, 2, 8, 10}, {1, 3, 9, 11}, {4, 6, 12, 14}, {5, 7, 13, 15}};
	uint32_t xW, yW;
	int32_t x = pos[idx][0], y = pos[idx][1];
	orc_neigh_loc(ctx, mb, 16, 16, x - 1, y, addrA, &xW, &yW);
	*idxA = at[xW / 4][yW / 4];
	orc_neigh_loc(ctx, mb, 16, 16, x, y - 1, addrB, &xW, &yW);
	*idxB = at[xW / 4][yW / 4];
}

void orc_neigh_chroma4x4(const struct orc_ctx *ctx, const struct orc_mb *mb,
			    uint32_t idx, uint32_t *addrA, uint32_t *idxA,
			    uint32_t *addrB, uint32_t *idxB)
{
	uint32_t xW, yW;
	int32_t x = 4 * (idx & 1), y = 4 * (idx >> 1);
	uint32_t mw = ctx->sps_derived.MbWidthC, mh = ctx->sps_derived.MbHeightC;
	orc_neigh_loc(ctx, mb, mw, mh, x - 1, y, addrA, &xW, &yW);
	*idxA = 2 * (yW / 4) + xW / 4;
	orc_neigh_loc(ctx, mb, mw, mh, x, y - 1, addrB, &xW, &yW);
	*idxB = 2 * (yW / 4) + xW / 4;
}

/* ---- 9.2.1 coeff_token (ref src/h264_slice_data.c:1239-1360) ----------- */
static int read_coeff_token(struct h264_bitstream *bs, struct orc_ctx *ctx,
			    struct orc_mb *mb, uint32_t kind, uint32_t comp,
			    uint32_t blkIdx, uint32_t *t1, uint32_t *tc)
{
	uint32_t addrA = ORC_MB_ADDR_INVALID, addrB = ORC_MB_ADDR_INVALID;
	uint32_t idxA = 0, idxB = 0, nA = 0, nB = 0, nC, tok = 0;
	int availA, availB;

	if (kind == ORC_LVL_CHROMA_DC) {
		CHK(read_vlc(bs,
			     ctx->sps_derived.ChromaArrayType == 1 ? &s_ct[4]
								   : &s_ct[5],
			     &tok));
		goto done;
	}
	if (kind == ORC_LVL_CHROMA_AC) {
		if (blkIdx >= 8)
			return -EPROTO;
		orc_neigh_chroma4x4(ctx, mb, blkIdx, &addrA, &idxA, &addrB, &idxB);
	} else {
		orc_neigh_luma4x4(ctx, mb, blkIdx, &addrA, &idxA, &addrB, &idxB);
	}
	availA = addrA != ORC_MB_ADDR_INVALID;
	availB = addrB != ORC_MB_ADDR_INVALID;
	if (availA)
		nA = get_nz(ctx, addrA, comp, idxA);
	if (availB)
		nB = get_nz(ctx, addrB, comp, idxB);
	if (availA && availB)
		nC = (nA + nB + 1) >> 1;
	else if (availA)
		nC = nA;
	else if (availB)
		nC = nB;
	else
		nC = 0;

	if (nC < 2) {
		CHK(read_vlc(bs, &s_ct[0], &tok));
	} else if (nC < 4) {
		CHK(read_vlc(bs, &s_ct[1], &tok));
	} else if (nC < 8) {
		CHK(read_vlc(bs, &s_ct[2], &tok));
	} else {
		uint32_t code = 0;
		CHK(orc_bs_read_bits(bs, &code, 6));
		tok = s_ct_flc[code];
		if (tok == 0)
			return -EIO;
	}
done:
	*t1 = (tok >> 5) & 3;
	*tc = tok & 0x1f;
	/* stored for every class, DC blocks included (Appendix C.4) */
	set_nz(ctx, mb->mbAddr, comp, blkIdx, *tc);
	return 0;
}

/* ref src/h264_slice_data.c:1366-1391 */
static int read_total_zeros(struct h264_bitstream *bs, uint32_t tc,
			    uint32_t maxnum, uint32_t *tz)
{
	uint32_t code = 0;
	if (tc >= maxnum) {
		*tz = 0;
		return 0;
	}
	if (maxnum == 4)
		CHK(read_vlc(bs, &s_tzc420[tc], &code));
	else if (maxnum == 8)
		CHK(read_vlc(bs, &s_tzc422[tc], &code));
	else if (maxnum <= 16)
		CHK(read_vlc(bs, &s_tz[tc], &code));
	else
		return -EIO;
	*tz = code & 0x7f;
	return 0;
}

/* ref src/h264_slice_data.c:1397-1416 */
static int read_run_before(struct h264_bitstream *bs, uint32_t zeros_left,
			   uint32_t *run)
{
	uint32_t code = 0;
	if (zeros_left == 0) {
		*run = 0;
		return 0;
	}
	CHK(read_vlc(bs, &s_rb[zeros_left <= 6 ? zeros_left : 7], &code));
	*run = code & 0x7f;
	return 0;
}

/* ---- 9.2 residual_block (ref src/h264_syntax_slice_data.h:103-242) ----- */
static int residual_block(struct h264_bitstream *bs, struct orc_ctx *ctx,
			  struct orc_mb *mb, int16_t *coeffLevel,
			  uint32_t startIdx, uint32_t endIdx, uint32_t maxNum,
			  uint32_t kind, uint32_t comp, uint32_t blkIdx)
{
	int16_t levelVal[64];
	int runVal[64];
	uint32_t tc = 0, t1 = 0, suffixLength, i, bit = 0;
	uint32_t zerosLeft = 0, total_zeros = 0;
	int32_t coeffNum;

	for (i = 0; i < maxNum; i++)
		coeffLevel[i] = 0;
	CHK(read_coeff_token(bs, ctx, mb, kind, comp, blkIdx, &t1, &tc));
	if (tc == 0)
		return 0;
	suffixLength = (tc > 10 && t1 < 3) ? 1 : 0;

	for (i = 0; i < tc; i++) {
		uint32_t prefix = 0, levelCode;
		if (i < t1) {
			CHK(orc_bs_read_bits(bs, &bit, 1));
			levelVal[i] = 1 - 2 * (int)bit;
			continue;
		}
		CHK(orc_bs_read_bits(bs, &bit, 1));
		while (bit == 0) {
			prefix++;
			if (prefix > 25)
				return -EIO;
			CHK(orc_bs_read_bits(bs, &bit, 1));
		}
		levelCode = (prefix < 15 ? prefix : 15) << suffixLength;
		if (suffixLength > 0 || prefix >= 14) {
			uint32_t size;
			if (prefix == 14 && suffixLength == 0)
				size = 4;
			else if (prefix >= 15)
				size = prefix - 3;
			else
				size = suffixLength;
			if (size != 0) {
				uint32_t suffix = 0;
				CHK(orc_bs_read_bits(bs, &suffix, size));
				levelCode += suffix;
			}
		}
		if (prefix >= 15 && suffixLength == 0)
			levelCode += 15;
		if (prefix >= 16)
			levelCode += (1u << (prefix - 3)) - 4096;
		if (i == t1 && t1 < 3)
			levelCode += 2;
		if (levelCode % 2 == 0)
			levelVal[i] = (int16_t)((levelCode + 2) >> 1);
		else
			levelVal[i] = (int16_t)((-(int32_t)levelCode - 1) >> 1);
		if (suffixLength == 0)
			suffixLength = 1;
		{
			int a = levelVal[i] >= 0 ? levelVal[i] : -levelVal[i];
			if (a > (3 << (suffixLength - 1)) && suffixLength < 6)
				suffixLength++;
		}
	}

	CHK(read_total_zeros(bs, tc, endIdx - startIdx + 1, &total_zeros));
	zerosLeft = total_zeros;
	for (i = 0; i + 1 < tc; i++) {
		uint32_t run = 0;
		CHK(read_run_before(bs, zerosLeft, &run));
		runVal[i] = (int)run;
		if (run > zerosLeft)
			return -EIO;
		zerosLeft -= run;
	}
	runVal[tc - 1] = (int)zerosLeft;
	coeffNum = -1;
	for (i = tc; i-- > 0;) {
		coeffNum += runVal[i] + 1;
		/* The reference stores unchecked and goes on (ref :222-226).  A
		 * position past the block only arises in a damaged stream and
		 * only as index 15 of a 15-coefficient block (Intra16x16 AC,
		 * chroma AC: total_zeros comes from the 4x4 tables, TotalCoeff
		 * may be 16).  In the reference that element is the first one of
		 * the NEXT block of the same array, which is cleared when that
		 * block is visited (:129-130, and every uncoded block is
		 * zero-filled), or -- behind the last block -- lies in a member
		 * no caller or dump of such a macroblock shows.  Dropping the
		 * coefficient and going on therefore reproduces everything the
		 * reference makes observable, callbacks included (an earlier
		 * restatement failed here with -EIO and delivered fewer
		 * macroblocks than the reference on damaged slices). */
		if ((uint32_t)(startIdx + coeffNum) >= maxNum) {
			orc_stat_dropped_coeffs++;
			continue;
		}
		coeffLevel[startIdx + coeffNum] = levelVal[i];
	}
	return 0;
}

/* ref src/h264_syntax_slice_data.h:247-331; comp 0 Y, 1 Cb, 2 Cr */
static int residual_luma(struct h264_bitstream *bs, struct orc_ctx *ctx,
			 struct orc_mb *mb, uint32_t startIdx, uint32_t endIdx,
			 uint32_t comp)
{
	static const uint32_t dc_kind[3] = {ORC_LVL_I16DC, ORC_LVL_CB_I16DC,
					    ORC_LVL_CR_I16DC};
	static const uint32_t ac_kind[3] = {ORC_LVL_I16AC, ORC_LVL_CB_I16AC,
					    ORC_LVL_CR_I16AC};
	static const uint32_t l4_kind[3] = {ORC_LVL_LUMA4x4, ORC_LVL_CB4x4,
					    ORC_LVL_CR4x4};
	int i16 = mb->MbPartPredMode[0] == ORC_PRED_Intra_16x16;
	uint32_t i8, i4, i;

	if (startIdx == 0 && i16)
		CHK(residual_block(bs, ctx, mb, mb->I16DC[comp], 0, 15, 16,
				   dc_kind[comp], comp, 0));
	for (i8 = 0; i8 < 4; i8++) {
		for (i4 = 0; i4 < 4; i4++) {
			uint32_t b = i8 * 4 + i4;
			if (mb->CodedBlockPatternLuma & (1 << i8)) {
				if (i16)
					CHK(residual_block(
						bs, ctx, mb, mb->I16AC[comp][b],
						startIdx > 0 ? startIdx - 1 : 0,
						endIdx - 1, 15, ac_kind[comp],
						comp, b));
				else
					CHK(residual_block(
						bs, ctx, mb,
						mb->Lvl4x4[comp][b], startIdx,
						endIdx, 16, l4_kind[comp], comp,
						b));
			} else if (i16) {
				for (i = 0; i < 15; i++)
					mb->I16AC[comp][b][i] = 0;
			} else {
				for (i = 0; i < 16; i++)
					mb->Lvl4x4[comp][b][i] = 0;
			}
			if (mb->transform_size_8x8_flag) {
				for (i = 0; i < 16; i++)
					mb->Lvl8x8[comp][i8][4 * i + i4] =
						mb->Lvl4x4[comp][b][i];
			}
		}
	}
	return 0;
}

/* ref src/h264_syntax_slice_data.h:334-419 */
static int residual(struct h264_bitstream *bs, struct orc_ctx *ctx,
		    struct orc_mb *mb, uint32_t startIdx, uint32_t endIdx)
{
	uint32_t cat = ctx->sps_derived.ChromaArrayType;
	uint32_t c, i8, i4, i;
	CHK(residual_luma(bs, ctx, mb, startIdx, endIdx, 0));
	if (cat == 1 || cat == 2) {
		uint32_t n8 = 4 / (ctx->sps_derived.SubWidthC *
				   ctx->sps_derived.SubHeightC);
		for (c = 0; c < 2; c++) {
			if ((mb->CodedBlockPatternChroma & 3) && startIdx == 0) {
				CHK(residual_block(bs, ctx, mb, mb->ChromaDC[c],
						   0, 4 * n8 - 1, 4 * n8,
						   ORC_LVL_CHROMA_DC, 1 + c, 0));
			} else {
				for (i = 0; i < 4 * n8; i++)
					mb->ChromaDC[c][i] = 0;
			}
		}
		for (c = 0; c < 2; c++) {
			for (i8 = 0; i8 < n8; i8++) {
				for (i4 = 0; i4 < 4; i4++) {
					uint32_t b = i8 * 4 + i4;
					if (mb->CodedBlockPatternChroma & 2) {
						CHK(residual_block(
							bs, ctx, mb,
							mb->ChromaAC[c][b],
							startIdx > 0
								? startIdx - 1
								: 0,
							endIdx - 1, 15,
							ORC_LVL_CHROMA_AC,
							1 + c, b));
					} else {
						for (i = 0; i < 15; i++)
							mb->ChromaAC[c][b][i] =
								0;
					}
				}
			}
		}
	} else if (cat == 3) {
		CHK(residual_luma(bs, ctx, mb, startIdx, endIdx, 1));
		CHK(residual_luma(bs, ctx, mb, startIdx, endIdx, 2));
	}
	return 0;
}

/* ---- 7.4.5 mb_type (ref src/h264_slice_data.c:839-969) ----------------- */
int orc_apply_mb_type(struct orc_ctx *ctx, struct orc_mb *mb, uint32_t type)
{
	/* B_16x8 / B_8x16 partition prediction modes, mb_type 4..21 */
	static const uint8_t bmode[18][2] = {
		{ORC_PRED_L0, ORC_PRED_L0}, {ORC_PRED_L0, ORC_PRED_L0},
		{ORC_PRED_L1, ORC_PRED_L1}, {ORC_PRED_L1, ORC_PRED_L1},
		{ORC_PRED_L0, ORC_PRED_L1}, {ORC_PRED_L0, ORC_PRED_L1},
		{ORC_PRED_L1, ORC_PRED_L0}, {ORC_PRED_L1, ORC_PRED_L0},
		{ORC_PRED_L0, ORC_PRED_Bi}, {ORC_PRED_L0, ORC_PRED_Bi},
		{ORC_PRED_L1, ORC_PRED_Bi}, {ORC_PRED_L1, ORC_PRED_Bi},
		{ORC_PRED_Bi, ORC_PRED_L0}, {ORC_PRED_Bi, ORC_PRED_L0},
		{ORC_PRED_Bi, ORC_PRED_L1}, {ORC_PRED_Bi, ORC_PRED_L1},
		{ORC_PRED_Bi, ORC_PRED_Bi}, {ORC_PRED_Bi, ORC_PRED_Bi},
	};
	int intra = 0;
	mb->raw_mb_type = type;

	switch (ctx->slice.type) {
	case H264_SLICE_TYPE_I:
		intra = 1;
		break;
	case H264_SLICE_TYPE_SI:
		if (type == 0) {
			mb->mb_type = H264_MB_TYPE_SI;
			mb->NumMbPart = 1;
			mb->MbPartPredMode[0] = ORC_PRED_Intra_4x4;
		} else {
			type -= 1;
			intra = 1;
		}
		break;
	case H264_SLICE_TYPE_P:
	case H264_SLICE_TYPE_SP:
		if (type == 0) {
			mb->mb_type = H264_MB_TYPE_P_16x16;
			mb->NumMbPart = 1;
			mb->MbPartPredMode[0] = ORC_PRED_L0;
		} else if (type == 1 || type == 2) {
			mb->mb_type = type == 1 ? H264_MB_TYPE_P_16x8
						: H264_MB_TYPE_P_8x16;
			mb->NumMbPart = 2;
			mb->MbPartPredMode[0] = ORC_PRED_L0;
			mb->MbPartPredMode[1] = ORC_PRED_L0;
		} else if (type == 3) {
			mb->mb_type = H264_MB_TYPE_P_8x8;
			mb->NumMbPart = 4;
		} else if (type == 4) {
			mb->mb_type = H264_MB_TYPE_P_8x8ref0;
			mb->NumMbPart = 4;
		} else {
			type -= 5;
			intra = 1;
		}
		break;
	case H264_SLICE_TYPE_B:
		if (type == 0) {
			mb->mb_type = H264_MB_TYPE_B_Direct_16x16;
			mb->NumMbPart = 1;
			mb->MbPartPredMode[0] = ORC_PRED_Direct;
		} else if (type <= 3) {
			mb->mb_type = H264_MB_TYPE_B_16x16;
			mb->NumMbPart = 1;
			mb->MbPartPredMode[0] = type == 1   ? ORC_PRED_L0
						: type == 2 ? ORC_PRED_L1
							    : ORC_PRED_Bi;
		} else if (type <= 21) {
			mb->mb_type = (type & 1) ? H264_MB_TYPE_B_8x16
						 : H264_MB_TYPE_B_16x8;
			mb->NumMbPart = 2;
			mb->MbPartPredMode[0] = bmode[type - 4][0];
			mb->MbPartPredMode[1] = bmode[type - 4][1];
		} else if (type == 22) {
			mb->mb_type = H264_MB_TYPE_B_8x8;
			mb->NumMbPart = 4;
		} else {
			type -= 23;
			intra = 1;
		}
		break;
	default:
		break;
	}
	if (intra) {
		if (type == 0) {
			mb->mb_type = H264_MB_TYPE_I_NxN;
			mb->NumMbPart = 1;
			mb->MbPartPredMode[0] = ORC_PRED_Intra_4x4;
		} else if (type <= 24) {
			mb->mb_type = H264_MB_TYPE_I_16x16;
			mb->NumMbPart = 1;
			mb->MbPartPredMode[0] = ORC_PRED_Intra_16x16;
			mb->Intra16x16PredMode = (type - 1) % 4;
			mb->CodedBlockPatternLuma = type <= 12 ? 0 : 15;
			mb->CodedBlockPatternChroma = ((type - 1) / 4) % 3;
		} else if (type == 25) {
			mb->mb_type = H264_MB_TYPE_I_PCM;
			mb->NumMbPart = 0;
		} else {
			return -EIO;
		}
	}
	/* 4-bit field in the reference (Appendix C.3) */
	ctx->slice.mb_table.info[orc_mb_off(ctx, mb->mbAddr)].mb_type =
		mb->mb_type & 0xf;
	return 0;
}

static int read_mb_type(struct h264_bitstream *bs, struct orc_ctx *ctx,
			struct orc_mb *mb)
{
	uint32_t type = 0;
	CHK(orc_bs_read_ue(bs, &type));
	return orc_apply_mb_type(ctx, mb, type);
}

/* ref src/h264_slice_data.c:975-1035; the table step for one sub-macroblock */
int orc_apply_sub_mb_type(struct orc_ctx *ctx, struct orc_mb *mb, uint32_t p,
			  uint32_t type)
{
	static const uint8_t tp[4][3] = {
		{ORC_SUB_P_8x8, 1, ORC_PRED_L0},
		{ORC_SUB_P_8x4, 2, ORC_PRED_L0},
		{ORC_SUB_P_4x8, 2, ORC_PRED_L0},
		{ORC_SUB_P_4x4, 4, ORC_PRED_L0},
	};
	static const uint8_t tb[13][3] = {
		{ORC_SUB_B_Direct_8x8, 4, ORC_PRED_Direct},
		{ORC_SUB_B_8x8, 1, ORC_PRED_L0},
		{ORC_SUB_B_8x8, 1, ORC_PRED_L1},
		{ORC_SUB_B_8x8, 1, ORC_PRED_Bi},
		{ORC_SUB_B_8x4, 2, ORC_PRED_L0},
		{ORC_SUB_B_4x8, 2, ORC_PRED_L0},
		{ORC_SUB_B_8x4, 2, ORC_PRED_L1},
		{ORC_SUB_B_4x8, 2, ORC_PRED_L1},
		{ORC_SUB_B_8x4, 2, ORC_PRED_Bi},
		{ORC_SUB_B_4x8, 2, ORC_PRED_Bi},
		{ORC_SUB_B_4x4, 4, ORC_PRED_L0},
		{ORC_SUB_B_4x4, 4, ORC_PRED_L1},
		{ORC_SUB_B_4x4, 4, ORC_PRED_Bi},
	};
	{
		const uint8_t *row = NULL;
		mb->raw_sub_mb_type[p] = type;
		if (ctx->slice.type == H264_SLICE_TYPE_P ||
		    ctx->slice.type == H264_SLICE_TYPE_SP) {
			if (type >= 4)
				return -EIO;
			row = tp[type];
		} else if (ctx->slice.type == H264_SLICE_TYPE_B) {
			if (type >= 13)
				return -EIO;
			row = tb[type];
		}
		if (row != NULL) {
			mb->sub_mb_type[p] = row[0];
			mb->NumSubMbPart[p] = row[1];
			mb->SubMbPredMode[p] = row[2];
		}
	}
	return 0;
}

static int read_sub_mb_type(struct h264_bitstream *bs, struct orc_ctx *ctx,
			    struct orc_mb *mb)
{
	uint32_t p, type = 0;
	for (p = 0; p < 4; p++) {
		CHK(orc_bs_read_ue(bs, &type));
		CHK(orc_apply_sub_mb_type(ctx, mb, p, type));
	}
	return 0;
}

/* ref src/h264_slice_data.c:1041-1080 */
static int read_cbp(struct h264_bitstream *bs, struct orc_ctx *ctx,
		    struct orc_mb *mb)
{
	uint32_t code = 0, cat = ctx->sps_derived.ChromaArrayType;
	uint32_t cls, lim, col;
	CHK(orc_bs_read_ue(bs, &code));
	if (cat == 1 || cat == 2) {
		cls = 0;
		lim = 48;
	} else if (cat == 0 || cat == 3) {
		cls = 1;
		lim = 16;
	} else {
		return -EIO;
	}
	if (code >= lim)
		return -EIO;
	col = (mb->mb_type == H264_MB_TYPE_I_NxN ||
	       mb->mb_type == H264_MB_TYPE_I_16x16 ||
	       mb->mb_type == H264_MB_TYPE_SI)
		      ? 0
		      : 1;
	mb->coded_block_pattern = s_cbp[cls][code][col];
	mb->CodedBlockPatternLuma = mb->coded_block_pattern % 16;
	mb->CodedBlockPatternChroma = mb->coded_block_pattern / 16;
	return 0;
}

/* ---- 7.3.5.2 sub_mb_pred (ref src/h264_syntax_slice_data.h:422-503) ---- */
static int sub_mb_pred(struct h264_bitstream *bs, struct orc_ctx *ctx,
		       struct orc_mb *mb)
{
	const struct h264_slice_header *sh = &ctx->slice.hdr;
	uint32_t p, s, c, v;
	int32_t sv;
	CHK(read_sub_mb_type(bs, ctx, mb));
	if ((sh->num_ref_idx_l0_active_minus1 > 0 ||
	     mb->mb_field_decoding_flag != sh->field_pic_flag) &&
	    mb->mb_type != H264_MB_TYPE_P_8x8ref0) {
		for (p = 0; p < 4; p++) {
			if (mb->sub_mb_type[p] != ORC_SUB_B_Direct_8x8 &&
			    mb->SubMbPredMode[p] != ORC_PRED_L1) {
				CHK(orc_bs_read_te(bs, &v, mb->max_ref_idx_0));
				mb->ref_idx_l0[p] = v;
			}
		}
	}
	if (sh->num_ref_idx_l1_active_minus1 > 0 ||
	    mb->mb_field_decoding_flag != sh->field_pic_flag) {
		for (p = 0; p < 4; p++) {
			if (mb->sub_mb_type[p] != ORC_SUB_B_Direct_8x8 &&
			    mb->SubMbPredMode[p] != ORC_PRED_L0) {
				CHK(orc_bs_read_te(bs, &v, mb->max_ref_idx_1));
				mb->ref_idx_l1[p] = v;
			}
		}
	}
	for (p = 0; p < 4; p++) {
		if (mb->sub_mb_type[p] == ORC_SUB_B_Direct_8x8 ||
		    mb->SubMbPredMode[p] == ORC_PRED_L1)
			continue;
		for (s = 0; s < mb->NumSubMbPart[p]; s++)
			for (c = 0; c < 2; c++) {
				CHK(orc_bs_read_se(bs, &sv));
				mb->mvd_l0[p][s][c] = sv;
			}
	}
	for (p = 0; p < 4; p++) {
		if (mb->sub_mb_type[p] == ORC_SUB_B_Direct_8x8 ||
		    mb->SubMbPredMode[p] == ORC_PRED_L0)
			continue;
		for (s = 0; s < mb->NumSubMbPart[p]; s++)
			for (c = 0; c < 2; c++) {
				CHK(orc_bs_read_se(bs, &sv));
				mb->mvd_l1[p][s][c] = sv;
			}
	}
	return 0;
}

/* ---- 7.3.5.1 mb_pred (ref src/h264_syntax_slice_data.h:506-601) -------- */
static int mb_pred(struct h264_bitstream *bs, struct orc_ctx *ctx,
		   struct orc_mb *mb)
{
	const struct h264_slice_header *sh = &ctx->slice.hdr;
	uint32_t i, p, c, flag = 0, v = 0;
	int32_t sv;
	uint32_t m0 = mb->MbPartPredMode[0];

	if (m0 == ORC_PRED_Intra_4x4 || m0 == ORC_PRED_Intra_8x8 ||
	    m0 == ORC_PRED_Intra_16x16) {
		if (m0 == ORC_PRED_Intra_4x4) {
			for (i = 0; i < 16; i++) {
				CHK(orc_bs_read_bits(bs, &flag, 1));
				if (!flag) {
					CHK(orc_bs_read_bits(bs, &v, 3));
					mb->intra4x4_pred_mode[i] = (int8_t)v;
				} else {
					mb->intra4x4_pred_mode[i] = -1;
				}
			}
		}
		if (m0 == ORC_PRED_Intra_8x8) {
			for (i = 0; i < 4; i++) {
				CHK(orc_bs_read_bits(bs, &flag, 1));
				if (!flag) {
					CHK(orc_bs_read_bits(bs, &v, 3));
					mb->intra8x8_pred_mode[i] = (int8_t)v;
				} else {
					mb->intra8x8_pred_mode[i] = -1;
				}
			}
		}
		if (ctx->sps_derived.ChromaArrayType == 1 ||
		    ctx->sps_derived.ChromaArrayType == 2) {
			CHK(orc_bs_read_ue(bs, &v));
			mb->intra_chroma_pred_mode = (uint8_t)v;
		}
	} else if (m0 != ORC_PRED_Direct) {
		if (sh->num_ref_idx_l0_active_minus1 > 0 ||
		    mb->mb_field_decoding_flag != sh->field_pic_flag) {
			for (p = 0; p < mb->NumMbPart; p++) {
				if (mb->MbPartPredMode[p] != ORC_PRED_L1) {
					CHK(orc_bs_read_te(bs, &v,
							   mb->max_ref_idx_0));
					mb->ref_idx_l0[p] = v;
				}
			}
		}
		if (sh->num_ref_idx_l1_active_minus1 > 0 ||
		    mb->mb_field_decoding_flag != sh->field_pic_flag) {
			for (p = 0; p < mb->NumMbPart; p++) {
				if (mb->MbPartPredMode[p] != ORC_PRED_L0) {
					CHK(orc_bs_read_te(bs, &v,
							   mb->max_ref_idx_1));
					mb->ref_idx_l1[p] = v;
				}
			}
		}
		for (p = 0; p < mb->NumMbPart; p++) {
			if (mb->MbPartPredMode[p] == ORC_PRED_L1)
				continue;
			for (c = 0; c < 2; c++) {
				CHK(orc_bs_read_se(bs, &sv));
				mb->mvd_l0[p][0][c] = sv;
			}
		}
		for (p = 0; p < mb->NumMbPart; p++) {
			if (mb->MbPartPredMode[p] == ORC_PRED_L0)
				continue;
			for (c = 0; c < 2; c++) {
				CHK(orc_bs_read_se(bs, &sv));
				mb->mvd_l1[p][0][c] = sv;
			}
		}
	}
	return 0;
}

/* ---- 7.3.5 macroblock_layer (ref src/h264_syntax_slice_data.h:604-696) - */
static int macroblock_layer(struct h264_bitstream *bs, struct orc_ctx *ctx,
			    struct orc_mb *mb)
{
	int t8x8 = ctx->pps->transform_8x8_mode_flag;
	int d8x8 = ctx->sps->direct_8x8_inference_flag;
	uint32_t i, c, v = 0;
	int32_t sv;

	CHK(read_mb_type(bs, ctx, mb));
	if (mb->mb_type == H264_MB_TYPE_I_PCM) {
		uint32_t comp, blk;
		while (bs->cachebits % 8 != 0) {
			CHK(orc_bs_read_bits(bs, &v, 1));
			if (v != 0)
				return -EIO;
		}
		for (i = 0; i < 256; i++) {
			CHK(orc_bs_read_bits(bs, &v,
					     ctx->sps_derived.BitDepthLuma));
			mb->pcm_sample_luma[i] = (uint16_t)v;
		}
		for (c = 0; c < 2; c++) {
			for (i = 0; i < ctx->sps_derived.MbWidthC *
						ctx->sps_derived.MbHeightC;
			     i++) {
				CHK(orc_bs_read_bits(
					bs, &v, ctx->sps_derived.BitDepthChroma));
				mb->pcm_sample_chroma[c][i] = (uint16_t)v;
			}
		}
		for (comp = 0; comp < 3; comp++)
			for (blk = 0; blk < 16; blk++)
				set_nz(ctx, mb->mbAddr, comp, blk, 16);
		return 0;
	}

	{
		int noSub8x8 = 1;
		if (mb->mb_type != H264_MB_TYPE_I_NxN &&
		    mb->MbPartPredMode[0] != ORC_PRED_Intra_16x16 &&
		    mb->NumMbPart == 4) {
			uint32_t p;
			CHK(sub_mb_pred(bs, ctx, mb));
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
				CHK(orc_bs_read_bits(bs, &v, 1));
				mb->transform_size_8x8_flag = (int)v;
				if (v)
					mb->MbPartPredMode[0] =
						ORC_PRED_Intra_8x8;
			}
			CHK(mb_pred(bs, ctx, mb));
		}
		if (mb->MbPartPredMode[0] != ORC_PRED_Intra_16x16) {
			CHK(read_cbp(bs, ctx, mb));
			if (mb->CodedBlockPatternLuma > 0 && t8x8 &&
			    mb->mb_type != H264_MB_TYPE_I_NxN && noSub8x8 &&
			    (mb->mb_type != H264_MB_TYPE_B_Direct_16x16 ||
			     d8x8)) {
				CHK(orc_bs_read_bits(bs, &v, 1));
				mb->transform_size_8x8_flag = (int)v;
			}
		}
		if (mb->CodedBlockPatternLuma > 0 ||
		    mb->CodedBlockPatternChroma > 0 ||
		    mb->MbPartPredMode[0] == ORC_PRED_Intra_16x16) {
			CHK(orc_bs_read_se(bs, &sv));
			mb->mb_qp_delta = sv;
			CHK(residual(bs, ctx, mb, 0, 15));
		}
	}
	return 0;
}

/* ---- 7.3.4 slice_data (ref src/h264_syntax_slice_data.h:701-811) ------- */
static int slice_data_internal(struct h264_bitstream *bs, struct orc_reader *rd)
{
	struct orc_ctx *ctx = rd->ctx;
	const struct h264_slice_header *sh = &ctx->slice.hdr;
	uint32_t addr, mb_count = 0, skip_run = 0, i;
	int res;

	if (ctx->pps->entropy_coding_mode_flag) {
		/* CABAC is not parsed (Appendix C.1) unless the engine's CABAC
		 * decoder is asked for */
		if (rd->flags & H264_AMD_READER_FLAGS_CABAC)
			return orc_cabac_slice_data(bs, rd);
		return 0;
	}
	/* deviation shared with the product (h264_amd.h): the reference takes
	 * any width and divides by a width of 0 */
	if (ctx->sps_derived.PicWidthInMbs == 0 ||
	    ctx->sps_derived.PicWidthInMbs > H264_AMD_MAX_PIC_WIDTH_MBS)
		return H264_AMD_E_UNSUPPORTED;
	/* and builds a slice group map of any size (h264_amd.h,
	 * H264_AMD_FMO_MAX_PIC_MBS) */
	if (ctx->pps->num_slice_groups_minus1 > 0) {
		uint64_t units = (uint64_t)ctx->sps_derived.PicWidthInMbs *
				 (ctx->sps->pic_height_in_map_units_minus1 + 1ull);
		uint64_t mbs = (ctx->sps->frame_mbs_only_flag ||
				sh->field_pic_flag)
				       ? units
				       : 2 * units;
		if (mbs > H264_AMD_FMO_MAX_PIC_MBS || units == 0)
			return H264_AMD_E_UNSUPPORTED;
		if (ctx->pps->slice_group_map_type == 3 &&
		    units > H264_AMD_FMO_MAX_BOX_OUT_UNITS)
			return H264_AMD_E_UNSUPPORTED;
	}

	if (rd->cbs.slice_data_begin)
		rd->cbs.slice_data_begin(ctx, sh, rd->userdata);
	orc_clear_macroblock_table(ctx);
	orc_gen_slice_group_map(ctx); /* status ignored, ref :723 */
	addr = sh->first_mb_in_slice * (1 + ctx->derived.MbaffFrameFlag);
	do {
		int field_flag = -1, prev_mb_skipped = 0;
		if (ctx->slice.type != H264_SLICE_TYPE_I &&
		    ctx->slice.type != H264_SLICE_TYPE_SI) {
			CHK(orc_bs_read_ue(bs, &skip_run));
			prev_mb_skipped = skip_run > 0;
			for (i = 0; i < skip_run; i++) {
				/* a damaged mb_skip_run can ask for 2^32
				 * macroblocks (and the table for them): product
				 * and oracle stop at H264_AMD_MAX_SLICE_MBS */
				if (mb_count >= H264_AMD_MAX_SLICE_MBS)
					return H264_AMD_E_MB_OVERFLOW;
				CHK(orc_new_macroblock(ctx, addr, 1, -1));
				if (rd->cbs.slice_data_mb)
					rd->cbs.slice_data_mb(ctx, sh,
							      ctx->mb->mbAddr,
							      ctx->mb->mb_type,
							      rd->userdata);
				addr = orc_next_mb_addr(ctx, addr);
				mb_count++;
			}
			if (skip_run > 0 && !orc_bs_more_rbsp_data(bs))
				break;
		}
		if (mb_count >= H264_AMD_MAX_SLICE_MBS)
			return H264_AMD_E_MB_OVERFLOW;
		/* mb_field_decoding_flag: with the top macroblock of a pair, or
		 * with the bottom one behind a skipped top (ref :754-761) */
		if (ctx->derived.MbaffFrameFlag &&
		    (addr % 2 == 0 || prev_mb_skipped)) {
			uint32_t v = 0;
			CHK(orc_bs_read_bits(bs, &v, 1));
			field_flag = (int)v;
		}
		/* the reference does not look at h264_new_macroblock's status
		 * (ref :763); its -EINVAL cases cannot arise from this loop, the
		 * table limit can */
		res = orc_new_macroblock(ctx, addr, 0, field_flag);
		if (res == -ENOMEM || res == H264_AMD_E_MB_OVERFLOW)
			return res;
		res = macroblock_layer(bs, ctx, ctx->mb);
		if (res < 0)
			return res;
		if (rd->cbs.slice_data_mb)
			rd->cbs.slice_data_mb(ctx, sh, ctx->mb->mbAddr,
					      ctx->mb->mb_type, rd->userdata);
		addr = orc_next_mb_addr(ctx, addr);
		mb_count++;
	} while (orc_bs_more_rbsp_data(bs));

	if (rd->cbs.slice_data_end)
		rd->cbs.slice_data_end(ctx, sh, mb_count, rd->userdata);
	return 0;
}

int orc_read_slice_data(struct h264_bitstream *bs, struct orc_reader *rd)
{
	struct orc_ctx *ctx = rd->ctx;
	tables_init();
	flc_init();
	ctx->slice.rawdata.partial = bs->cache;
	ctx->slice.rawdata.partialbits = bs->cachebits;
	ctx->slice.rawdata.buf = bs->cdata + bs->off;
	ctx->slice.rawdata.len = bs->len - bs->off;
	if (rd->flags & H264_READER_FLAGS_SLICE_DATA)
		return slice_data_internal(bs, rd);
	return 0;
}

/* ---- T2 record (format: include/h264_amd.h) ---------------------------- */
void orc_mb_to_record(const struct orc_ctx *ctx, uint8_t *rec, uint32_t stride)
{
	const struct orc_mb *mb = ctx->mb;
	const struct orc_mb_info *info =
		&ctx->slice.mb_table.info[orc_mb_off(ctx, mb->mbAddr)];
	struct h264_amd_mb_summary *s = (struct h264_amd_mb_summary *)rec;
	uint32_t cat = ctx->sps_derived.ChromaArrayType;
	int16_t *co = (int16_t *)(rec + H264_AMD_T2_HEAD);
	int16_t *mv;
	uint32_t i, b, c, n;
	int skipped = info->skipped;

	memset(rec, 0, stride);
	s->mb_type = (uint8_t)mb->mb_type;
	s->cbp = (uint8_t)(mb->CodedBlockPatternLuma |
			   (mb->CodedBlockPatternChroma << 4));
	s->flags = (skipped ? H264_AMD_MB_SKIPPED : 0) |
		   (mb->transform_size_8x8_flag ? H264_AMD_MB_TRANSFORM_8X8 : 0) |
		   (mb->mb_field_decoding_flag ? H264_AMD_MB_FIELD : 0);
	s->raw_mb_type = (uint8_t)mb->raw_mb_type;
	s->mb_qp_delta = (int16_t)mb->mb_qp_delta;
	s->intra_chroma_pred_mode = mb->intra_chroma_pred_mode;
	s->intra16x16_pred_mode = mb->Intra16x16PredMode;
	if (mb->mb_type == H264_MB_TYPE_I_NxN ||
	    mb->mb_type == H264_MB_TYPE_SI) {
		if (mb->MbPartPredMode[0] == ORC_PRED_Intra_8x8) {
			for (i = 0; i < 4; i++)
				s->pred[i / 2] |=
					(uint8_t)((mb->intra8x8_pred_mode[i] + 1)
						  << (4 * (i & 1)));
		} else {
			for (i = 0; i < 16; i++)
				s->pred[i / 2] |=
					(uint8_t)((mb->intra4x4_pred_mode[i] + 1)
						  << (4 * (i & 1)));
		}
	} else if (mb->NumMbPart == 4) {
		for (i = 0; i < 4; i++)
			s->pred[i] = (uint8_t)mb->raw_sub_mb_type[i];
	}
	memcpy(s->nz_coeff, info->nz_coeff, 48);
	memcpy(rec + 64, mb->ref_idx_l0, 4);
	memcpy(rec + 68, mb->ref_idx_l1, 4);
	mv = (int16_t *)(rec + 128);
	memcpy(mv, mb->mvd_l0, 64);
	memcpy(mv + 32, mb->mvd_l1, 64);

	if (mb->mb_type == H264_MB_TYPE_I_PCM) {
		uint16_t *pc = (uint16_t *)co;
		n = ctx->sps_derived.MbWidthC * ctx->sps_derived.MbHeightC;
		memcpy(pc, mb->pcm_sample_luma, 512);
		memcpy(pc + 256, mb->pcm_sample_chroma[0], 2 * n);
		memcpy(pc + 256 + n, mb->pcm_sample_chroma[1], 2 * n);
		return;
	}
	for (c = 0; c < (cat == 3 ? 3u : 1u); c++) {
		int16_t *dst = co + 256 * c;
		if (mb->MbPartPredMode[0] == ORC_PRED_Intra_16x16) {
			memcpy(dst, mb->I16DC[c], 32);
			for (b = 0; b < 16; b++)
				memcpy(dst + 16 + 15 * b, mb->I16AC[c][b], 30);
		} else {
			for (b = 0; b < 16; b++)
				memcpy(dst + 16 * b, mb->Lvl4x4[c][b], 32);
		}
	}
	if (cat == 1 || cat == 2) {
		n = cat == 1 ? 4 : 8;
		for (c = 0; c < 2; c++)
			memcpy(co + 256 + n * c, mb->ChromaDC[c], 2 * n);
		for (c = 0; c < 2; c++)
			for (b = 0; b < n; b++)
				memcpy(co + 256 + 2 * n + 15 * (n * c + b),
				       mb->ChromaAC[c][b], 30);
	}
}

/**
 * h264gen.c -- seeded synthetic H.264 stream generator (see h264gen.h).
 *
 * An ENCODER for the syntax the engine parses: SPS/PPS (7.3.2.1, 7.3.2.2),
 * slice header (7.3.3), slice_data / macroblock_layer / mb_pred / sub_mb_pred
 * (7.3.4, 7.3.5) and CAVLC residual blocks (9.2), driven by a xorshift
 * generator.  Code tables come from the standard's bit strings in
 * ../common/h264_cavlc_tables.inc, inverted here for encoding.
 */
#define _GNU_SOURCE
#include "h264gen.h"

#include <errno.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- T2 record layout constants (mirror of <h264_amd.h>) --------------- */
#define T2_HEAD 256u
#define MBF_SKIPPED 0x01
#define MBF_T8X8 0x02

enum { MB_UNKNOWN, MB_I_NxN, MB_I_16x16, MB_I_PCM, MB_SI, MB_P_16x16,
       MB_P_16x8, MB_P_8x16, MB_P_8x8, MB_P_8x8ref0, MB_P_SKIP,
       MB_B_Direct_16x16, MB_B_16x16, MB_B_16x8, MB_B_8x16, MB_B_8x8,
       MB_B_SKIP };

enum { PM_L0 = 1, PM_L1 = 2, PM_BI = 3, PM_DIRECT = 4 };

/* ---- random numbers ----------------------------------------------------- */
struct rng {
	uint64_t s;
};

static uint64_t rng_next(struct rng *r)
{
	uint64_t x = r->s;
	x ^= x >> 12;
	x ^= x << 25;
	x ^= x >> 27;
	r->s = x;
	return x * 0x2545F4914F6CDD1Dull;
}

static uint32_t rng_below(struct rng *r, uint32_t n)
{
	return n ? (uint32_t)((rng_next(r) >> 32) % n) : 0;
}

static int rng_pm(struct rng *r, uint32_t per_mille)
{
	return rng_below(r, 1000) < per_mille;
}

/* ---- emulation-preventing bit writer ------------------------------------ */
struct bw {
	uint8_t *p;
	size_t cap, off;
	uint32_t acc;
	int nbits;
	int zeros;
	int overflow;
};

static void bw_init(struct bw *w, uint8_t *p, size_t cap)
{
	memset(w, 0, sizeof(*w));
	w->p = p;
	w->cap = cap;
}

static void bw_byte(struct bw *w, uint8_t b)
{
	if (w->zeros >= 2 && b <= 3) {
		if (w->off < w->cap)
			w->p[w->off] = 3;
		else
			w->overflow = 1;
		w->off++;
		w->zeros = 0;
	}
	if (w->off < w->cap)
		w->p[w->off] = b;
	else
		w->overflow = 1;
	w->off++;
	w->zeros = b == 0 ? w->zeros + 1 : 0;
}

static void bw_u(struct bw *w, uint32_t v, int n)
{
	while (n > 0) {
		int k = 8 - w->nbits;
		if (k > n)
			k = n;
		w->acc = (w->acc << k) | ((v >> (n - k)) & ((1u << k) - 1));
		w->nbits += k;
		n -= k;
		if (w->nbits == 8) {
			bw_byte(w, (uint8_t)w->acc);
			w->acc = 0;
			w->nbits = 0;
		}
	}
}

/* a field whose width comes from the SPS: any 32-bit width.  Above 32 bits the
 * value is preceded by zeros (what the reference's writer produces for a
 * 32-bit field); above 40000 bits only 2048 zero bits are written, which makes
 * the slice too short on purpose (hostile-SPS tests). */
static void bw_u_wide(struct bw *w, uint32_t v, uint32_t n)
{
	uint32_t z;
	if (n > 40000) {
		for (z = 0; z < 2048; z += 32)
			bw_u(w, 0, 32);
		return;
	}
	for (z = n > 32 ? n - 32 : 0; z > 0; z -= z < 32 ? z : 32)
		bw_u(w, 0, z < 32 ? (int)z : 32);
	if (n)
		bw_u(w, v, n < 32 ? (int)n : 32);
}

static void bw_ue(struct bw *w, uint32_t v)
{
	uint32_t x = v + 1;
	int n = 0;
	uint32_t t;
	for (t = x; t; t >>= 1)
		n++;
	if (n > 1)
		bw_u(w, 0, n - 1);
	bw_u(w, x, n);
}

static void bw_se(struct bw *w, int32_t v)
{
	bw_ue(w, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v));
}

static void bw_te(struct bw *w, uint32_t v, uint32_t range)
{
	if (range == 1)
		bw_u(w, !v, 1);
	else
		bw_ue(w, v);
}

static void bw_trailing(struct bw *w)
{
	bw_u(w, 1, 1);
	while (w->nbits != 0)
		bw_u(w, 0, 1);
}

/* ---- code tables, inverted for encoding --------------------------------- */
struct code {
	uint16_t bits;
	uint8_t len;
};

static struct code s_ct[6][4][17];
static struct code s_tz[16][16], s_tzc420[4][4], s_tzc422[8][8], s_rb[8][15];
static uint8_t s_cbp_code[2][2][48]; /* [class][intra/inter][cbp] -> codeNum */
static pthread_once_t s_once = PTHREAD_ONCE_INIT;

static struct code mk(const char *s)
{
	struct code c = {0, 0};
	for (; *s; s++) {
		if (*s == '0' || *s == '1') {
			c.bits = (uint16_t)((c.bits << 1) | (*s - '0'));
			c.len++;
		}
	}
	return c;
}

static void cabac_tables_init(void);

static void tables_init(void)
{
#define H264_CT(col, bits, t1, tc) s_ct[col][t1][tc] = mk(bits);
#define H264_TZ(i, bits, v) s_tz[i][v] = mk(bits);
#define H264_TZ_C420(i, bits, v) s_tzc420[i][v] = mk(bits);
#define H264_TZ_C422(i, bits, v) s_tzc422[i][v] = mk(bits);
#define H264_RB(i, bits, v) s_rb[i][v] = mk(bits);
#define H264_CBP(cls, code, intra, inter)                                      \
	s_cbp_code[cls][0][intra] = code;                                      \
	s_cbp_code[cls][1][inter] = code;
#include "../common/h264_cavlc_tables.inc"
	cabac_tables_init();
}

static void put_code(struct bw *w, struct code c)
{
	bw_u(w, c.bits, c.len);
}

/* ---- parameter sets ------------------------------------------------------ */
static int is_high(uint32_t p)
{
	return p == 100 || p == 110 || p == 122 || p == 244 || p == 44;
}

static uint32_t chroma_array_type(const struct h264gen_cfg *c)
{
	return is_high(c->profile_idc) ? c->chroma_format_idc : 1;
}

static int finish(struct bw *w)
{
	if (w->overflow)
		return -ENOBUFS;
	return (int)w->off;
}

int h264gen_sps(const struct h264gen_cfg *c, uint8_t *out, size_t cap)
{
	struct bw w;
	bw_init(&w, out, cap);
	bw_u(&w, 0x67, 8);
	bw_u(&w, c->profile_idc, 8);
	bw_u(&w, 0, 8); /* constraint flags + reserved */
	bw_u(&w, c->level_idc, 8);
	bw_ue(&w, c->sps_id);
	if (is_high(c->profile_idc)) {
		bw_ue(&w, c->chroma_format_idc);
		if (c->chroma_format_idc == 3)
			bw_u(&w, 0, 1); /* separate_colour_plane_flag */
		bw_ue(&w, c->bit_depth_luma_minus8);
		bw_ue(&w, c->bit_depth_chroma_minus8);
		bw_u(&w, 0, 1); /* qpprime_y_zero_transform_bypass_flag */
		bw_u(&w, 0, 1); /* seq_scaling_matrix_present_flag */
	}
	bw_ue(&w, c->log2_max_frame_num_minus4);
	bw_ue(&w, c->pic_order_cnt_type);
	if (c->pic_order_cnt_type == 0) {
		bw_ue(&w, c->log2_max_poc_lsb_minus4);
	} else if (c->pic_order_cnt_type == 1) {
		bw_u(&w, c->delta_pic_order_always_zero_flag, 1);
		bw_se(&w, -3); /* offset_for_non_ref_pic */
		bw_se(&w, 1); /* offset_for_top_to_bottom_field */
		bw_ue(&w, 3); /* num_ref_frames_in_pic_order_cnt_cycle */
		bw_se(&w, 2);
		bw_se(&w, -1);
		bw_se(&w, 4);
	}
	bw_ue(&w, c->max_num_ref_frames);
	bw_u(&w, 0, 1); /* gaps_in_frame_num_value_allowed_flag */
	bw_ue(&w, c->width_mbs - 1);
	bw_ue(&w, c->height_mbs - 1);
	bw_u(&w, c->field_coding ? 0 : 1, 1); /* frame_mbs_only_flag */
	if (c->field_coding)
		bw_u(&w, c->field_coding == 2, 1); /* mb_adaptive_frame_field_flag */
	bw_u(&w, c->direct_8x8_inference_flag, 1);
	if (c->crop_bottom) {
		bw_u(&w, 1, 1);
		bw_ue(&w, 0);
		bw_ue(&w, 0);
		bw_ue(&w, 0);
		bw_ue(&w, c->crop_bottom);
	} else {
		bw_u(&w, 0, 1);
	}
	if (c->vui) {
		bw_u(&w, 1, 1);
		bw_u(&w, 1, 1); /* aspect_ratio_info_present_flag */
		bw_u(&w, 1, 8); /* 1:1 */
		bw_u(&w, 0, 1); /* overscan */
		bw_u(&w, 1, 1); /* video_signal_type_present_flag */
		bw_u(&w, 5, 3);
		bw_u(&w, 1, 1); /* full range */
		bw_u(&w, 1, 1); /* colour description */
		bw_u(&w, 1, 8);
		bw_u(&w, 1, 8);
		bw_u(&w, 1, 8);
		bw_u(&w, 0, 1); /* chroma loc */
		bw_u(&w, 1, 1); /* timing_info_present_flag */
		bw_u(&w, 1001, 32);
		bw_u(&w, 60000, 32);
		bw_u(&w, 1, 1);
		bw_u(&w, 0, 1); /* nal hrd */
		bw_u(&w, 0, 1); /* vcl hrd */
		bw_u(&w, 0, 1); /* pic_struct_present_flag */
		bw_u(&w, 1, 1); /* bitstream_restriction_flag */
		bw_u(&w, 1, 1);
		bw_ue(&w, 0);
		bw_ue(&w, 0);
		bw_ue(&w, 10);
		bw_ue(&w, 10);
		bw_ue(&w, 0);
		bw_ue(&w, c->max_num_ref_frames);
	} else {
		bw_u(&w, 0, 1);
	}
	bw_trailing(&w);
	return finish(&w);
}

int h264gen_pps(const struct h264gen_cfg *c, uint8_t *out, size_t cap)
{
	struct bw w;
	bw_init(&w, out, cap);
	bw_u(&w, 0x68, 8);
	bw_ue(&w, c->pps_id);
	bw_ue(&w, c->sps_id);
	bw_u(&w, c->entropy_coding_mode_flag, 1);
	bw_u(&w, c->bottom_field_pic_order_in_frame_present_flag, 1);
	bw_ue(&w, c->num_slice_groups_minus1);
	if (c->num_slice_groups_minus1 > 0) {
		uint32_t i, units = c->width_mbs * c->height_mbs;
		bw_ue(&w, c->slice_group_map_type);
		switch (c->slice_group_map_type) {
		case 0:
			for (i = 0; i <= c->num_slice_groups_minus1; i++)
				bw_ue(&w, c->run_length_minus1[i]);
			break;
		case 2:
			for (i = 0; i < c->num_slice_groups_minus1; i++) {
				bw_ue(&w, c->top_left[i]);
				bw_ue(&w, c->bottom_right[i]);
			}
			break;
		case 3:
		case 4:
		case 5:
			bw_u(&w, c->slice_group_change_direction_flag, 1);
			bw_ue(&w, c->slice_group_change_rate_minus1);
			break;
		case 6: {
			/* Ceil(Log2(num_slice_groups_minus1 + 1)) bits per id */
			uint32_t bits = 0;
			while ((1u << bits) < c->num_slice_groups_minus1 + 1u)
				bits++;
			bw_ue(&w, units - 1u); /* pic_size_in_map_units_minus1 */
			for (i = 0; i < units; i++)
				if (bits)
					bw_u(&w, c->slice_group_id[i & 255u], (int)bits);
			break;
		}
		default:
			break;
		}
	}
	bw_ue(&w, c->num_ref_idx_l0_default_minus1);
	bw_ue(&w, c->num_ref_idx_l1_default_minus1);
	bw_u(&w, c->weighted_pred_flag, 1);
	bw_u(&w, c->weighted_bipred_idc, 2);
	bw_se(&w, c->pic_init_qp_minus26);
	bw_se(&w, 0);
	bw_se(&w, 0);
	bw_u(&w, c->deblocking_filter_control_present_flag, 1);
	bw_u(&w, c->constrained_intra_pred_flag, 1);
	bw_u(&w, c->redundant_pic_cnt_present_flag, 1);
	if (c->transform_8x8_mode_flag) {
		bw_u(&w, 1, 1);
		bw_u(&w, 0, 1); /* pic_scaling_matrix_present_flag */
		bw_se(&w, 0); /* second_chroma_qp_index_offset */
	}
	bw_trailing(&w);
	return finish(&w);
}

int h264gen_aud(uint32_t primary_pic_type, uint8_t *out, size_t cap)
{
	struct bw w;
	bw_init(&w, out, cap);
	bw_u(&w, 0x09, 8);
	bw_u(&w, primary_pic_type, 3);
	bw_trailing(&w);
	return finish(&w);
}

int h264gen_filler(uint32_t n_ff, uint8_t *out, size_t cap)
{
	struct bw w;
	uint32_t i;
	bw_init(&w, out, cap);
	bw_u(&w, 0x0c, 8);
	for (i = 0; i < n_ff; i++)
		bw_u(&w, 0xff, 8);
	bw_trailing(&w);
	return finish(&w);
}

int h264gen_sei_user_data(uint32_t n, uint64_t seed, uint8_t *out, size_t cap)
{
	struct bw w;
	struct rng r = {seed | 1};
	uint32_t size = 16 + n, i, s;
	bw_init(&w, out, cap);
	bw_u(&w, 0x06, 8);
	bw_u(&w, 5, 8); /* user_data_unregistered */
	for (s = size; s >= 255; s -= 255)
		bw_u(&w, 0xff, 8);
	bw_u(&w, s, 8);
	for (i = 0; i < size; i++)
		bw_u(&w, rng_below(&r, 8) == 0 ? 0 : rng_below(&r, 256), 8);
	bw_trailing(&w);
	return finish(&w);
}

/* ---- slice state --------------------------------------------------------- */
/* CABAC: the arithmetic encoder and what later macroblocks look at
 * (h264gen_cabac.inc) */
struct cenc {
	struct bw *w;
	uint32_t low, range, outstanding;
	int first;
	uint8_t st[1024]; /* pStateIdx | valMPS << 6 */
	uint64_t bins;
};

struct cmb {
	uint8_t mbt; /* MB_* */
	uint8_t skip, pcm, intra, t8;
	uint8_t cbp; /* luma | chroma << 4; I_PCM: 0x2f */
	uint8_t icp; /* intra_chroma_pred_mode */
	uint8_t dc; /* coded_block_flag of DC blocks: bit 0 Y (Intra16x16), 1 Cb, 2 Cr */
	/* coded_block_flag per 4x4 block, bit y * w + x: [0] luma (w = 4),
	 * [1] [2] Cb Cr -- like luma with ChromaArrayType 3, else the chroma AC
	 * blocks (w = 2).  8x8 blocks own four bits */
	uint16_t cbf[3];
	uint8_t ref[2][4]; /* ref_idx as transmitted per 8x8 block, y * 2 + x */
	uint16_t amvd[2][16][2]; /* |mvd| as transmitted per 4x4 block, y * 4 + x */
};

struct gen {
	const struct h264gen_cfg *c;
	const struct h264gen_slice *s;
	struct bw w;
	struct rng r;
	uint32_t cat, W;
	uint32_t st; /* syntax class: 0 P (and SP), 1 B, 2 I (and SI) */
	int sp, si; /* slice_type % 5 == 3 / == 4 */
	uint32_t nref0, nref1; /* num_ref_idx_lX_active_minus1 */
	uint8_t *nz; /* [num_mbs][48] */
	uint8_t *t2;
	uint32_t t2_stride;
	/* slice groups: address of macroblock i, and which i an address has
	 * (-1: not part of this slice); NULL without slice groups */
	uint32_t *addr_of;
	int32_t *idx_of;
	uint32_t n_addr;
	/* MBAFF frame (field_coding 2, frame picture): macroblock pairs */
	int mbaff;
	uint8_t *ff; /* [num_mbs] mb_field_decoding_flag as finally inferred */
	/* CABAC (entropy_coding_mode_flag with cabac_syntax) */
	int cabac;
	struct cenc ce;
	struct cmb *ci; /* [num_mbs] */
	uint32_t prev_qpd_nz; /* previous macroblock carried mb_qp_delta != 0 */
};

/* ---- slice groups (8.2.2), written from the standard ------------------------- */

static uint32_t fmo_units0(const struct h264gen_cfg *c, uint32_t cycle)
{
	/* MapUnitsInSliceGroup0 = Min(slice_group_change_cycle *
	 * SliceGroupChangeRate, PicSizeInMapUnits) (7-34) */
	uint64_t v = (uint64_t)cycle * (c->slice_group_change_rate_minus1 + 1u);
	uint32_t units = c->width_mbs * c->height_mbs;
	return v < units ? (uint32_t)v : units;
}

static void fmo_map_units(const struct h264gen_cfg *c, uint32_t cycle,
			  uint8_t *m)
{
	const uint32_t W = c->width_mbs, H = c->height_mbs, n = W * H;
	const uint32_t G = c->num_slice_groups_minus1 + 1u;
	const uint32_t dir = c->slice_group_change_direction_flag ? 1u : 0u;
	const uint32_t u0 = fmo_units0(c, cycle);
	const uint32_t upper = dir ? n - u0 : u0; /* sizeOfUpperLeftGroup */
	uint32_t i, j, k, g;
	switch (c->slice_group_map_type) {
	case 0: /* 8.2.2.1 interleaved */
		i = 0;
		do {
			for (g = 0; g < G && i < n;
			     i += c->run_length_minus1[g++] + 1u)
				for (j = 0; j <= c->run_length_minus1[g] &&
					    i + j < n;
				     j++)
					m[i + j] = (uint8_t)g;
		} while (i < n);
		break;
	case 1: /* 8.2.2.2 dispersed */
		for (i = 0; i < n; i++)
			m[i] = (uint8_t)(((i % W) + (((i / W) * G) / 2u)) % G);
		break;
	case 2: /* 8.2.2.3 foreground with left-over */
		for (i = 0; i < n; i++)
			m[i] = (uint8_t)(G - 1u);
		for (g = G - 1u; g-- > 0;) {
			uint32_t yt = c->top_left[g] / W, xl = c->top_left[g] % W;
			uint32_t yb = c->bottom_right[g] / W;
			uint32_t xr = c->bottom_right[g] % W, x, y;
			for (y = yt; y <= yb && y < H; y++)
				for (x = xl; x <= xr; x++)
					m[y * W + x] = (uint8_t)g;
		}
		break;
	case 3: { /* 8.2.2.4 box-out */
		int32_t x = (int32_t)((W - dir) / 2u), y = (int32_t)((H - dir) / 2u);
		int32_t left = x, right = x, top = y, bottom = y;
		int32_t xd = (int32_t)dir - 1, yd = (int32_t)dir;
		uint32_t vacant;
		for (i = 0; i < n; i++)
			m[i] = 1;
		for (k = 0; k < u0; k += vacant) {
			vacant = m[(uint32_t)y * W + (uint32_t)x] == 1;
			if (vacant)
				m[(uint32_t)y * W + (uint32_t)x] = 0;
			if (xd == -1 && x == left) {
				left = left - 1 > 0 ? left - 1 : 0;
				x = left;
				xd = 0;
				yd = 2 * (int32_t)dir - 1;
			} else if (xd == 1 && x == right) {
				right = right + 1 < (int32_t)W - 1 ? right + 1
								   : (int32_t)W - 1;
				x = right;
				xd = 0;
				yd = 1 - 2 * (int32_t)dir;
			} else if (yd == -1 && y == top) {
				top = top - 1 > 0 ? top - 1 : 0;
				y = top;
				xd = 1 - 2 * (int32_t)dir;
				yd = 0;
			} else if (yd == 1 && y == bottom) {
				bottom = bottom + 1 < (int32_t)H - 1 ? bottom + 1
								     : (int32_t)H - 1;
				y = bottom;
				xd = 2 * (int32_t)dir - 1;
				yd = 0;
			} else {
				x += xd;
				y += yd;
			}
		}
		break;
	}
	case 4: /* 8.2.2.5 raster scan */
		for (i = 0; i < n; i++)
			m[i] = (uint8_t)(i < upper ? dir : 1u - dir);
		break;
	case 5: /* 8.2.2.6 wipe */
		k = 0;
		for (j = 0; j < W; j++)
			for (i = 0; i < H; i++)
				m[i * W + j] = (uint8_t)(k++ < upper ? dir : 1u - dir);
		break;
	case 6: /* 8.2.2.7 explicit */
		for (i = 0; i < n; i++)
			m[i] = c->slice_group_id[i & 255u];
		break;
	default:
		memset(m, 0, n);
		break;
	}
}

long h264gen_mb_to_slice_group_map(const struct h264gen_cfg *c, uint32_t cycle,
				   uint32_t field_pic, uint8_t *map, size_t cap)
{
	const uint32_t W = c->width_mbs, units = W * c->height_mbs;
	/* macroblocks of the picture: a frame of interlaced material has two
	 * map-unit rows' worth per unit row (7-27 ff.) */
	const uint32_t n = (c->field_coding && !field_pic) ? 2u * units : units;
	uint8_t *m;
	uint32_t a;
	if (W == 0 || units == 0 || units > (1u << 22) || n > cap)
		return -EINVAL;
	if (c->num_slice_groups_minus1 == 0) {
		memset(map, 0, n);
		return (long)n;
	}
	m = malloc(units);
	if (m == NULL)
		return -ENOMEM;
	fmo_map_units(c, cycle, m);
	/* 8.2.2.8: map units are macroblocks for frame_mbs_only and for field
	 * pictures, macroblock pairs with MBAFF, and otherwise (frame picture
	 * of interlaced material) a unit covers two vertically adjacent rows */
	for (a = 0; a < n; a++) {
		if (!c->field_coding || field_pic)
			map[a] = m[a];
		else if (c->field_coding == 2)
			map[a] = m[a / 2u];
		else
			map[a] = m[(a / (2u * W)) * W + (a % W)];
	}
	free(m);
	return (long)n;
}

static uint8_t *rec_of(struct gen *g, uint32_t i)
{
	return g->t2 ? g->t2 + (size_t)i * g->t2_stride : NULL;
}

static int16_t *coef_of(uint8_t *rec)
{
	return (int16_t *)(rec + T2_HEAD);
}

/* ---- MBAFF (7.3.4, 7.4.4, 6.4.10, 6.4.12.2), written from the standard ------ */

static uint32_t mb_address(const struct gen *g, uint32_t i)
{
	if (g->addr_of)
		return g->addr_of[i];
	return g->s->first_mb * (g->mbaff ? 2u : 1u) + i;
}

/* index in this slice of the TOP macroblock of pair `pair`, if the pair came
 * before macroblock i (6.4.8: available = same slice, decoded earlier) */
static int64_t pair_top_index(const struct gen *g, uint32_t i, int64_t pair)
{
	int64_t k;
	if (pair < 0)
		return -1;
	if (g->addr_of) {
		if ((uint64_t)(2 * pair) >= g->n_addr)
			return -1;
		k = g->idx_of[2 * pair];
	} else {
		k = 2 * (pair - (int64_t)g->s->first_mb);
	}
	/* an earlier pair of this slice */
	return (k >= 0 && k < (int64_t)(i & ~1u)) ? k : -1;
}

/* 6.4.12.2, Table 6-4 for one neighbouring luma / chroma location of a
 * macroblock in an MBAFF frame: which macroblock (index in this slice, -1
 * not available) and which sample row yM of it holds (xN, yN), xN = -1 or
 * yN = -1. */
static int64_t mbaff_neighbour(const struct gen *g, uint32_t i, int left,
			       int32_t yN, int32_t maxH, int32_t *yM)
{
	uint32_t addr = mb_address(g, i);
	int64_t pair = addr / 2;
	int top = (addr & 1u) == 0;
	int cur_frame = !g->ff[i];
	uint32_t col = (uint32_t)(pair % g->W);
	if (left) {
		int64_t a = col != 0 ? pair_top_index(g, i, pair - 1) : -1;
		int a_frame;
		if (a < 0)
			return -1;
		a_frame = !g->ff[a];
		if (cur_frame) {
			if (a_frame) {
				*yM = yN;
				return top ? a : a + 1;
			}
			/* frame macroblock beside a field pair: rows alternate
			 * between the two field macroblocks */
			*yM = top ? yN >> 1 : (yN + maxH) >> 1;
			return (yN & 1) ? a + 1 : a;
		}
		if (a_frame) {
			/* field macroblock beside a frame pair */
			int32_t y2 = (yN << 1) + (top ? 0 : 1);
			if (yN < maxH / 2) {
				*yM = y2;
				return a;
			}
			*yM = y2 - maxH;
			return a + 1;
		}
		*yM = yN;
		return top ? a : a + 1;
	} else {
		int64_t b;
		if (cur_frame && !top) {
			*yM = -1; /* the pair's own top macroblock */
			return (int64_t)i - 1;
		}
		b = pair_top_index(g, i, pair - (int64_t)g->W);
		if (b < 0)
			return -1;
		if (cur_frame) {
			*yM = -1;
			return b + 1;
		}
		if (top) {
			if (!g->ff[b]) {
				*yM = -2; /* 2 * yN */
				return b + 1;
			}
			*yM = -1;
			return b;
		}
		*yM = -1;
		return b + 1;
	}
}

static uint32_t ref_range(const struct gen *g, uint32_t i, uint32_t nref)
{
	/* field macroblocks of an MBAFF frame address fields: twice the range
	 * (7.4.5.1); kept in 8 bits like the reference keeps it */
	if (g->mbaff && g->ff[i])
		return (2u * nref + 1u) & 0xffu;
	return nref;
}

static int ref_present(const struct gen *g, uint32_t i, uint32_t nref)
{
	return nref > 0 || (g->mbaff && g->ff[i]);
}

/* availability + TotalCoeff of the neighbouring 4x4 block (6.4.11.4) */
static uint32_t neighbour_nc(struct gen *g, uint32_t i, uint32_t comp,
			     uint32_t blk, int chroma)
{
	uint32_t addr = mb_address(g, i);
	int availA, availB;
	int64_t ia = (int64_t)i - 1, ib = (int64_t)i - (int64_t)g->W;
	uint32_t nA = 0, nB = 0, x, y, bw4, bh4;
	const uint8_t *cur = g->nz + (size_t)i * 48 + comp * 16;
	if (g->addr_of) {
		/* a neighbour counts when it is part of this slice and came
		 * before (6.4.8: same slice, smaller address) */
		ia = addr >= 1 ? g->idx_of[addr - 1] : -1;
		ib = addr >= g->W ? g->idx_of[addr - g->W] : -1;
	}
	static const uint8_t lx[16] = {0, 1, 0, 1, 2, 3, 2, 3,
				       0, 1, 0, 1, 2, 3, 2, 3};
	static const uint8_t ly[16] = {0, 0, 1, 1, 0, 0, 1, 1,
				       2, 2, 3, 3, 2, 2, 3, 3};
	static const uint8_t lidx[4][4] = {/* [y][x] */
					   {0, 1, 4, 5},
					   {2, 3, 6, 7},
					   {8, 9, 12, 13},
					   {10, 11, 14, 15}};
	if (g->mbaff) {
		/* pairs: the neighbouring block may sit in either macroblock of
		 * the neighbouring pair and in another row (6.4.12.2) */
		int32_t maxH, yM = 0;
		int64_t n;
		if (chroma) {
			x = blk & 1;
			y = blk >> 1;
			bw4 = 2;
			maxH = g->cat == 1 ? 8 : 16;
		} else {
			x = lx[blk];
			y = ly[blk];
			bw4 = 4;
			maxH = 16;
		}
		availA = availB = 1;
		if (x > 0) {
			nA = chroma ? cur[2 * y + x - 1] : cur[lidx[y][x - 1]];
		} else {
			n = mbaff_neighbour(g, i, 1, (int32_t)(4 * y), maxH, &yM);
			availA = n >= 0;
			if (availA) {
				const uint8_t *l = g->nz + (size_t)n * 48 + comp * 16;
				uint32_t r = (uint32_t)(yM < 0 ? yM + maxH : yM) / 4;
				nA = chroma ? l[2 * r + bw4 - 1] : l[lidx[r][bw4 - 1]];
			}
		}
		if (y > 0) {
			nB = chroma ? cur[2 * (y - 1) + x] : cur[lidx[y - 1][x]];
		} else {
			n = mbaff_neighbour(g, i, 0, -1, maxH, &yM);
			availB = n >= 0;
			if (availB) {
				const uint8_t *t = g->nz + (size_t)n * 48 + comp * 16;
				uint32_t r = (uint32_t)(yM + maxH) / 4;
				nB = chroma ? t[2 * r + x] : t[lidx[r][x]];
			}
		}
		if (availA && availB)
			return (nA + nB + 1) >> 1;
		return availA ? nA : availB ? nB : 0;
	}
	if (chroma) {
		x = blk & 1;
		y = blk >> 1;
		bw4 = 2;
		bh4 = g->cat == 1 ? 2 : 4;
	} else {
		x = lx[blk];
		y = ly[blk];
		bw4 = 4;
		bh4 = 4;
	}
	/* left */
	if (x > 0) {
		availA = 1;
		nA = chroma ? cur[2 * y + x - 1] : cur[lidx[y][x - 1]];
	} else {
		availA = (addr % g->W != 0) && ia >= 0;
		if (availA) {
			const uint8_t *l = g->nz + (size_t)ia * 48 + comp * 16;
			nA = chroma ? l[2 * y + bw4 - 1] : l[lidx[y][bw4 - 1]];
		}
	}
	/* above */
	if (y > 0) {
		availB = 1;
		nB = chroma ? cur[2 * (y - 1) + x] : cur[lidx[y - 1][x]];
	} else {
		availB = ib >= 0;
		if (availB) {
			const uint8_t *t = g->nz + (size_t)ib * 48 + comp * 16;
			nB = chroma ? t[2 * (bh4 - 1) + x] : t[lidx[bh4 - 1][x]];
		}
	}
	if (availA && availB)
		return (nA + nB + 1) >> 1;
	if (availA)
		return nA;
	if (availB)
		return nB;
	return 0;
}

/* one level with the running suffixLength (inverse of 9.2.2.1) */
static void put_level(struct bw *w, int32_t level, uint32_t *sl, int first_adj)
{
	uint32_t a = level < 0 ? (uint32_t)-level : (uint32_t)level;
	uint32_t code = level > 0 ? 2 * a - 2 : 2 * a - 1;
	uint32_t s = *sl, prefix, esc;
	if (first_adj)
		code -= 2;
	if (s == 0 && code < 14) {
		bw_u(w, 1, (int)code + 1);
	} else if (s == 0 && code < 30) {
		bw_u(w, 1, 15);
		bw_u(w, code - 14, 4);
	} else if (s > 0 && (code >> s) < 15) {
		bw_u(w, 1, (int)(code >> s) + 1);
		bw_u(w, code & ((1u << s) - 1), (int)s);
	} else {
		esc = code - (s == 0 ? 30 : (15u << s));
		if (esc < 4096) {
			prefix = 15;
		} else {
			/* level_prefix p >= 16 covers
			 * [2^(p-3) - 4096, 2^(p-2) - 4096) */
			prefix = 16;
			while (esc >= (1u << (prefix - 2)) - 4096)
				prefix++;
			esc -= (1u << (prefix - 3)) - 4096;
		}
		bw_u(w, 0, (int)prefix);
		bw_u(w, 1, 1);
		bw_u(w, esc, (int)prefix - 3);
	}
	if (s == 0)
		s = 1;
	if (a > (3u << (s - 1)) && s < 6)
		s++;
	*sl = s;
}

/* Encode one residual block with random content.
 * coefs: where the expected coefficients go (max entries), may be NULL.
 * returns TotalCoeff */
static uint32_t put_block(struct gen *g, uint32_t maxnum, uint32_t col,
			  int16_t *coefs)
{
	struct bw *w = &g->w;
	uint32_t tc, t1, i, sl, tz, zleft;
	int32_t lv[16];
	uint32_t run[16];
	uint32_t d = g->s->density;

	/* TotalCoeff: mostly small, occasionally full */
	if (d == 0 || rng_below(&g->r, 16) >= d + 2) {
		tc = 0;
	} else {
		uint32_t span = rng_below(&g->r, 8) == 0 ? maxnum : (d < maxnum ? d : maxnum);
		tc = rng_below(&g->r, span + 1);
	}
	if (tc > maxnum)
		tc = maxnum;
	t1 = rng_below(&g->r, (tc < 3 ? tc : 3) + 1);
	for (i = 0; i < tc; i++) {
		int32_t mag;
		if (i < t1) {
			mag = 1;
		} else if (rng_pm(&g->r, g->s->pm_escape)) {
			mag = 16 + (int32_t)rng_below(&g->r, 3000);
			if (rng_below(&g->r, 8) == 0)
				mag = 3000 + (int32_t)rng_below(&g->r, 12000);
		} else {
			uint32_t k = rng_below(&g->r, 16);
			mag = k < 9 ? 1 : k < 13 ? 2 : 3 + (int32_t)rng_below(&g->r, 12);
		}
		if (i == t1 && t1 < 3 && mag < 2)
			mag = 2;
		lv[i] = rng_below(&g->r, 2) ? -mag : mag;
	}
	put_code(w, s_ct[col][t1][tc]);
	if (tc == 0)
		return 0;
	sl = (tc > 10 && t1 < 3) ? 1 : 0;
	for (i = 0; i < tc; i++) {
		if (i < t1)
			bw_u(w, lv[i] < 0, 1);
		else
			put_level(w, lv[i], &sl, i == t1 && t1 < 3);
	}
	tz = 0;
	if (tc < maxnum) {
		tz = rng_below(&g->r, maxnum - tc + 1);
		if (rng_below(&g->r, 3) == 0)
			tz = 0;
		if (maxnum == 4)
			put_code(w, s_tzc420[tc][tz]);
		else if (maxnum == 8)
			put_code(w, s_tzc422[tc][tz]);
		else
			put_code(w, s_tz[tc][tz]);
	}
	zleft = tz;
	for (i = 0; i + 1 < tc; i++) {
		run[i] = 0;
		if (zleft > 0) {
			run[i] = rng_below(&g->r, zleft + 1);
			put_code(w, s_rb[zleft <= 6 ? zleft : 7][run[i]]);
			zleft -= run[i];
		}
	}
	run[tc - 1] = zleft;
	if (coefs) {
		int32_t pos = -1;
		for (i = tc; i-- > 0;) {
			pos += (int32_t)run[i] + 1;
			coefs[pos] = (int16_t)lv[i];
		}
	}
	return tc;
}

static uint32_t ct_col(uint32_t nc)
{
	return nc < 2 ? 0 : nc < 4 ? 1 : nc < 8 ? 2 : 3;
}

/* residual_luma for one colour component (7.3.5.3.1) */
static void put_residual_luma(struct gen *g, uint32_t i, uint32_t comp,
			      int i16, uint32_t cbp_luma, uint8_t *rec)
{
	uint8_t *nz = g->nz + (size_t)i * 48 + comp * 16;
	int16_t *co = rec ? coef_of(rec) + 256 * comp : NULL;
	uint32_t b;
	if (i16) {
		uint32_t nc = neighbour_nc(g, i, comp, 0, 0);
		nz[0] = (uint8_t)put_block(g, 16, ct_col(nc), co);
	}
	for (b = 0; b < 16; b++) {
		uint32_t nc;
		if (!(cbp_luma & (1u << (b / 4))))
			continue;
		nc = neighbour_nc(g, i, comp, b, 0);
		if (i16)
			nz[b] = (uint8_t)put_block(g, 15, ct_col(nc),
						   co ? co + 16 + 15 * b : NULL);
		else
			nz[b] = (uint8_t)put_block(g, 16, ct_col(nc),
						   co ? co + 16 * b : NULL);
	}
}

static void put_residual(struct gen *g, uint32_t i, int i16, uint32_t cbp_luma,
			 uint32_t cbp_chroma, uint8_t *rec)
{
	put_residual_luma(g, i, 0, i16, cbp_luma, rec);
	if (g->cat == 1 || g->cat == 2) {
		uint32_t n = g->cat == 1 ? 4 : 8, c, b;
		int16_t *co = rec ? coef_of(rec) + 256 : NULL;
		for (c = 0; c < 2; c++) {
			uint8_t *nz = g->nz + (size_t)i * 48 + (1 + c) * 16;
			if (cbp_chroma & 3)
				nz[0] = (uint8_t)put_block(
					g, n, g->cat == 1 ? 4 : 5,
					co ? co + n * c : NULL);
		}
		for (c = 0; c < 2; c++) {
			uint8_t *nz = g->nz + (size_t)i * 48 + (1 + c) * 16;
			if (!(cbp_chroma & 2))
				continue;
			for (b = 0; b < n; b++) {
				uint32_t nc = neighbour_nc(g, i, 1 + c, b, 1);
				nz[b] = (uint8_t)put_block(
					g, 15, ct_col(nc),
					co ? co + 2 * n + 15 * (n * c + b)
					   : NULL);
			}
		}
	} else if (g->cat == 3) {
		put_residual_luma(g, i, 1, i16, cbp_luma, rec);
		put_residual_luma(g, i, 2, i16, cbp_luma, rec);
	}
}

/* field pictures: every macroblock record carries the field flag (0x04) */
static __thread uint32_t t_field_flag;

static void rec_head(uint8_t *rec, uint32_t mb_type, uint32_t cbp,
		     uint32_t flags, uint32_t raw, int32_t qpd)
{
	if (!rec)
		return;
	flags |= t_field_flag;
	rec[0] = (uint8_t)mb_type;
	rec[1] = (uint8_t)cbp;
	rec[2] = (uint8_t)flags;
	rec[3] = (uint8_t)raw;
	rec[4] = (uint8_t)(qpd & 0xff);
	rec[5] = (uint8_t)((qpd >> 8) & 0xff);
}

static int32_t pick_qpd(struct gen *g)
{
	return rng_below(&g->r, 4) == 0 ? (int32_t)rng_below(&g->r, 13) - 6 : 0;
}

#include "h264gen_cabac.inc"

/* intra macroblock; `base` is the mb_type offset of the slice type */
static void put_intra_mb(struct gen *g, uint32_t i, uint32_t base, uint8_t *rec)
{
	struct bw *w = &g->w;
	uint32_t n, k, b;
	if (rng_pm(&g->r, g->s->pm_pcm)) {
		uint32_t bdl = 8 + g->c->bit_depth_luma_minus8;
		uint32_t bdc = 8 + g->c->bit_depth_chroma_minus8;
		uint32_t nch = g->cat == 0 ? 0 : g->cat == 1 ? 64 : g->cat == 2 ? 128 : 256;
		int zero = rng_pm(&g->r, g->s->pm_zero_pcm);
		uint16_t *pc = rec ? (uint16_t *)coef_of(rec) : NULL;
		if (g->cabac) {
			/* mb_type ends in a terminating bin that flushes the
			 * arithmetic encoder; it restarts behind the samples
			 * (9.3.1.2), the context variables carry on */
			g->ci[i].mbt = MB_I_PCM;
			g->ci[i].intra = 1;
			c_mb_type(g, i, base + 25, base != 0 && !g->si);
			g->ci[i].pcm = 1;
			g->ci[i].cbp = 0x2f;
			g->ci[i].dc = 7;
			g->ci[i].cbf[0] = g->ci[i].cbf[1] = g->ci[i].cbf[2] = 0xffff;
			g->prev_qpd_nz = 0;
		} else {
			bw_ue(w, base + 25);
		}
		while (w->nbits != 0)
			bw_u(w, 0, 1);
		for (k = 0; k < 256 + 2 * nch; k++) {
			uint32_t bd = k < 256 ? bdl : bdc;
			uint32_t v = zero ? (rng_below(&g->r, 64) == 0 ? rng_below(&g->r, 4) : 0)
					  : rng_below(&g->r, 1u << bd);
			bw_u(w, v, (int)bd);
			if (pc)
				pc[k] = (uint16_t)v;
		}
		if (g->cabac)
			ce_start(&g->ce);
		memset(g->nz + (size_t)i * 48, 16, 48);
		rec_head(rec, MB_I_PCM, 0, 0, base + 25, 0);
		return;
	}
	if (rng_pm(&g->r, g->s->pm_i16x16)) {
		uint32_t pred = rng_below(&g->r, 4);
		uint32_t chroma = (g->cat == 1 || g->cat == 2) ? rng_below(&g->r, 3) : 0;
		uint32_t luma = rng_below(&g->r, 2) ? 15 : 0;
		uint32_t t = 1 + pred + 4 * chroma + (luma ? 12 : 0);
		uint32_t icp = 0;
		int32_t qpd = pick_qpd(g);
		if (g->cabac) {
			g->ci[i].mbt = MB_I_16x16;
			g->ci[i].intra = 1;
			c_mb_type(g, i, base + t, base != 0 && !g->si);
		} else {
			bw_ue(w, base + t);
		}
		if (g->cat == 1 || g->cat == 2) {
			icp = rng_below(&g->r, 4);
			if (g->cabac)
				c_icp(g, i, icp);
			else
				bw_ue(w, icp);
		}
		if (g->cabac)
			c_qpd(g, qpd);
		else
			bw_se(w, qpd);
		rec_head(rec, MB_I_16x16, luma | (chroma << 4), 0, base + t, qpd);
		if (rec) {
			rec[6] = (uint8_t)icp;
			rec[7] = (uint8_t)pred;
		}
		if (g->cabac) {
			g->ci[i].cbp = (uint8_t)(luma | (chroma << 4));
			c_residual(g, i, 1, luma, chroma, 0, rec);
			c_count_nz(g, i, 1, rec);
		} else {
			put_residual(g, i, 1, luma, chroma, rec);
		}
		return;
	}
	/* I_NxN, or the SI macroblock (mb_type 0) of an SI slice: Intra_4x4
	 * prediction, no transform flag in front of the prediction modes but,
	 * as for every macroblock that is not I_NxN, behind a coded_block_pattern
	 * with luma (7.3.5; ref src/h264_syntax_slice_data.h:662-682) */
	{
		int si_mb = g->si && rng_pm(&g->r, 300);
		int t8 = !si_mb && g->c->transform_8x8_mode_flag &&
			 rng_pm(&g->r, g->s->pm_t8x8);
		uint32_t luma, chroma, cbp, icp = 0;
		int32_t qpd = 0;
		int no_res = rng_pm(&g->r, g->s->pm_cbp_zero);
		if (g->cabac) {
			c_mb_type(g, i, si_mb ? 0 : base + 0, base != 0 && !g->si);
			g->ci[i].mbt = si_mb ? MB_SI : MB_I_NxN;
			g->ci[i].intra = 1;
		} else {
			bw_ue(w, si_mb ? 0 : base + 0);
		}
		if (!si_mb && g->c->transform_8x8_mode_flag) {
			if (g->cabac)
				c_t8(g, i, (uint32_t)t8);
			else
				bw_u(w, (uint32_t)t8, 1);
		}
		n = t8 ? 4 : 16;
		for (b = 0; b < n; b++) {
			if (rng_below(&g->r, 2)) {
				if (g->cabac)
					c_ipred(g, 1, 0);
				else
					bw_u(w, 1, 1);
			} else {
				uint32_t m = rng_below(&g->r, 8);
				if (g->cabac) {
					c_ipred(g, 0, m);
				} else {
					bw_u(w, 0, 1);
					bw_u(w, m, 3);
				}
				if (rec)
					rec[8 + b / 2] |= (uint8_t)((m + 1) << (4 * (b & 1)));
			}
		}
		if (g->cat == 1 || g->cat == 2) {
			icp = rng_below(&g->r, 4);
			if (g->cabac)
				c_icp(g, i, icp);
			else
				bw_ue(w, icp);
		}
		luma = no_res ? 0 : rng_below(&g->r, 16);
		chroma = (no_res || !(g->cat == 1 || g->cat == 2)) ? 0 : rng_below(&g->r, 3);
		cbp = luma | (chroma << 4);
		if (g->cabac)
			c_cbp(g, i, cbp);
		else
			bw_ue(w, s_cbp_code[(g->cat == 1 || g->cat == 2) ? 0 : 1][0][cbp]);
		if (si_mb && luma > 0 && g->c->transform_8x8_mode_flag) {
			t8 = rng_pm(&g->r, g->s->pm_t8x8);
			if (g->cabac)
				c_t8(g, i, (uint32_t)t8);
			else
				bw_u(w, (uint32_t)t8, 1);
		}
		if (cbp) {
			qpd = pick_qpd(g);
			if (g->cabac)
				c_qpd(g, qpd);
			else
				bw_se(w, qpd);
		} else if (g->cabac) {
			g->prev_qpd_nz = 0;
		}
		rec_head(rec, si_mb ? MB_SI : MB_I_NxN, cbp, t8 ? MBF_T8X8 : 0,
			 si_mb ? 0 : base + 0, qpd);
		if (rec)
			rec[6] = (uint8_t)icp;
		if (g->cabac) {
			g->ci[i].cbp = (uint8_t)cbp;
			if (cbp)
				c_residual(g, i, 0, luma, chroma, t8, rec);
			c_count_nz(g, i, 0, rec);
		} else if (cbp) {
			put_residual(g, i, 0, luma, chroma, rec);
		}
	}
}

/* where a (sub-)macroblock partition lies: x4 | y4 << 4 | w4 << 8 | h4 << 12,
 * in units of 4x4 blocks (6.4.2) */
static uint32_t part_geom(uint32_t mbt, uint32_t p)
{
	if (mbt == MB_P_16x8 || mbt == MB_B_16x8)
		return 0u | (2u * p) << 4 | 4u << 8 | 2u << 12;
	if (mbt == MB_P_8x16 || mbt == MB_B_8x16)
		return 2u * p | 0u << 4 | 2u << 8 | 4u << 12;
	return 0u | 0u << 4 | 4u << 8 | 4u << 12; /* 16x16 */
}
#define PART_X8(mbt, p) ((part_geom(mbt, p) & 15u) / 2u)
#define PART_Y8(mbt, p) (((part_geom(mbt, p) >> 4) & 15u) / 2u)
#define PART_W8(mbt) (((part_geom(mbt, 0) >> 8) & 15u) / 2u)
#define PART_H8(mbt) (((part_geom(mbt, 0) >> 12) & 15u) / 2u)

static uint32_t sub_geom(int isb, uint32_t sub, uint32_t p, uint32_t s)
{
	/* shape of the sub-macroblock partitions: 0 8x8, 1 8x4, 2 4x8, 3 4x4 */
	static const uint8_t bshape[13] = {3, 0, 0, 0, 1, 2, 1, 2, 1, 2, 3, 3, 3};
	uint32_t shape = isb ? bshape[sub] : sub;
	uint32_t x = 2u * (p & 1u), y = 2u * (p >> 1), w = 2, h = 2;
	if (shape == 1) {
		h = 1;
		y += s;
	} else if (shape == 2) {
		w = 1;
		x += s;
	} else if (shape == 3) {
		w = h = 1;
		x += s & 1u;
		y += s >> 1;
	}
	return x | y << 4 | w << 8 | h << 12;
}

static void put_mvd(struct gen *g, uint32_t i, uint32_t list, uint32_t geom,
		    int16_t *dst)
{
	uint32_t c;
	int32_t mv[2];
	for (c = 0; c < 2; c++) {
		int32_t v = rng_below(&g->r, 3) == 0
				    ? 0
				    : (int32_t)rng_below(&g->r, 65) - 32;
		if (rng_below(&g->r, 64) == 0)
			v = (int32_t)rng_below(&g->r, 4001) - 2000;
		if (!g->cabac)
			bw_se(&g->w, v);
		mv[c] = v;
		if (dst)
			dst[c] = (int16_t)v;
	}
	if (g->cabac)
		c_mvd(g, i, list, geom & 15u, (geom >> 4) & 15u, (geom >> 8) & 15u,
		      geom >> 12, mv);
}

/* inter macroblock of a P or B slice (7.3.5.1, 7.3.5.2) */
static void put_inter_mb(struct gen *g, uint32_t i, uint8_t *rec)
{
	struct bw *w = &g->w;
	int isb = g->st == 1;
	uint32_t t, nparts, p, s, mbt;
	uint32_t pm[4] = {0, 0, 0, 0};
	uint32_t sub[4] = {0, 0, 0, 0}, nsub[4] = {1, 1, 1, 1};
	int no_sub8 = 1, t8 = 0;
	uint32_t luma, chroma, cbp;
	int32_t qpd = 0;
	int16_t *mv0 = rec ? (int16_t *)(rec + 128) : NULL;
	int16_t *mv1 = rec ? (int16_t *)(rec + 192) : NULL;
	static const uint8_t bpm[18][2] = {
		{PM_L0, PM_L0}, {PM_L0, PM_L0}, {PM_L1, PM_L1}, {PM_L1, PM_L1},
		{PM_L0, PM_L1}, {PM_L0, PM_L1}, {PM_L1, PM_L0}, {PM_L1, PM_L0},
		{PM_L0, PM_BI}, {PM_L0, PM_BI}, {PM_L1, PM_BI}, {PM_L1, PM_BI},
		{PM_BI, PM_L0}, {PM_BI, PM_L0}, {PM_BI, PM_L1}, {PM_BI, PM_L1},
		{PM_BI, PM_BI}, {PM_BI, PM_BI}};
	static const uint8_t bsub_pm[13] = {PM_DIRECT, PM_L0, PM_L1, PM_BI, PM_L0,
					    PM_L0, PM_L1, PM_L1, PM_BI, PM_BI,
					    PM_L0, PM_L1, PM_BI};
	static const uint8_t bsub_n[13] = {4, 1, 1, 1, 2, 2, 2, 2, 2, 2, 4, 4, 4};
	static const uint8_t psub_n[4] = {1, 2, 2, 4};

	if (!isb) {
		/* P_8x8ref0 has no CABAC bin string (Table 9-37) */
		t = rng_below(&g->r, g->cabac ? 4 : 5);
		nparts = t == 0 ? 1 : t <= 2 ? 2 : 4;
		mbt = MB_P_16x16 + t;
		for (p = 0; p < nparts; p++)
			pm[p] = PM_L0;
	} else {
		t = rng_below(&g->r, 23);
		if (t == 0) {
			nparts = 1;
			pm[0] = PM_DIRECT;
			mbt = MB_B_Direct_16x16;
		} else if (t <= 3) {
			nparts = 1;
			pm[0] = t;
			mbt = MB_B_16x16;
		} else if (t <= 21) {
			nparts = 2;
			pm[0] = bpm[t - 4][0];
			pm[1] = bpm[t - 4][1];
			mbt = (t & 1) ? MB_B_8x16 : MB_B_16x8;
		} else {
			nparts = 4;
			mbt = MB_B_8x8;
		}
	}
	if (g->cabac) {
		g->ci[i].mbt = (uint8_t)mbt;
		c_mb_type(g, i, t, 0);
	} else {
		bw_ue(w, t);
	}

	if (nparts == 4) {
		for (p = 0; p < 4; p++) {
			sub[p] = rng_below(&g->r, isb ? 13 : 4);
			if (g->cabac)
				c_sub_mb_type(g, sub[p]);
			else
				bw_ue(w, sub[p]);
			if (isb) {
				pm[p] = bsub_pm[sub[p]];
				nsub[p] = bsub_n[sub[p]];
			} else {
				pm[p] = PM_L0;
				nsub[p] = psub_n[sub[p]];
			}
			if (pm[p] != PM_DIRECT) {
				if (nsub[p] > 1)
					no_sub8 = 0;
			} else if (!g->c->direct_8x8_inference_flag) {
				no_sub8 = 0;
			}
			if (rec)
				rec[8 + p] = (uint8_t)sub[p];
		}
		if (ref_present(g, i, g->nref0) && mbt != MB_P_8x8ref0) {
			for (p = 0; p < 4; p++)
				if (pm[p] != PM_DIRECT && pm[p] != PM_L1) {
					uint32_t v = rng_below(
						&g->r, ref_range(g, i, g->nref0) + 1);
					if (g->cabac)
						c_ref_idx(g, i, 0, p & 1u, p >> 1, 1, 1, v);
					else
						bw_te(w, v, ref_range(g, i, g->nref0));
					if (rec)
						rec[64 + p] = (uint8_t)v;
				}
		}
		if (ref_present(g, i, g->nref1)) {
			for (p = 0; p < 4; p++)
				if (pm[p] != PM_DIRECT && pm[p] != PM_L0) {
					uint32_t v = rng_below(
						&g->r, ref_range(g, i, g->nref1) + 1);
					if (g->cabac)
						c_ref_idx(g, i, 1, p & 1u, p >> 1, 1, 1, v);
					else
						bw_te(w, v, ref_range(g, i, g->nref1));
					if (rec)
						rec[68 + p] = (uint8_t)v;
				}
		}
		for (p = 0; p < 4; p++)
			if (pm[p] != PM_DIRECT && pm[p] != PM_L1)
				for (s = 0; s < nsub[p]; s++)
					put_mvd(g, i, 0, sub_geom(isb, sub[p], p, s),
						mv0 ? mv0 + 8 * p + 2 * s : NULL);
		for (p = 0; p < 4; p++)
			if (pm[p] != PM_DIRECT && pm[p] != PM_L0)
				for (s = 0; s < nsub[p]; s++)
					put_mvd(g, i, 1, sub_geom(isb, sub[p], p, s),
						mv1 ? mv1 + 8 * p + 2 * s : NULL);
	} else if (pm[0] != PM_DIRECT) {
		if (ref_present(g, i, g->nref0))
			for (p = 0; p < nparts; p++)
				if (pm[p] != PM_L1) {
					uint32_t v = rng_below(
						&g->r, ref_range(g, i, g->nref0) + 1);
					if (g->cabac)
						c_ref_idx(g, i, 0, PART_X8(mbt, p),
							  PART_Y8(mbt, p), PART_W8(mbt),
							  PART_H8(mbt), v);
					else
						bw_te(w, v, ref_range(g, i, g->nref0));
					if (rec)
						rec[64 + p] = (uint8_t)v;
				}
		if (ref_present(g, i, g->nref1))
			for (p = 0; p < nparts; p++)
				if (pm[p] != PM_L0) {
					uint32_t v = rng_below(
						&g->r, ref_range(g, i, g->nref1) + 1);
					if (g->cabac)
						c_ref_idx(g, i, 1, PART_X8(mbt, p),
							  PART_Y8(mbt, p), PART_W8(mbt),
							  PART_H8(mbt), v);
					else
						bw_te(w, v, ref_range(g, i, g->nref1));
					if (rec)
						rec[68 + p] = (uint8_t)v;
				}
		for (p = 0; p < nparts; p++)
			if (pm[p] != PM_L1)
				put_mvd(g, i, 0, part_geom(mbt, p), mv0 ? mv0 + 8 * p : NULL);
		for (p = 0; p < nparts; p++)
			if (pm[p] != PM_L0)
				put_mvd(g, i, 1, part_geom(mbt, p), mv1 ? mv1 + 8 * p : NULL);
	}

	if (rng_pm(&g->r, g->s->pm_cbp_zero)) {
		luma = chroma = 0;
	} else {
		luma = rng_below(&g->r, 16);
		chroma = (g->cat == 1 || g->cat == 2) ? rng_below(&g->r, 3) : 0;
	}
	cbp = luma | (chroma << 4);
	if (g->cabac)
		c_cbp(g, i, cbp);
	else
		bw_ue(w, s_cbp_code[(g->cat == 1 || g->cat == 2) ? 0 : 1][1][cbp]);
	if (luma > 0 && g->c->transform_8x8_mode_flag && no_sub8 &&
	    (mbt != MB_B_Direct_16x16 || g->c->direct_8x8_inference_flag)) {
		t8 = rng_pm(&g->r, g->s->pm_t8x8);
		if (g->cabac)
			c_t8(g, i, (uint32_t)t8);
		else
			bw_u(w, (uint32_t)t8, 1);
	}
	if (cbp) {
		qpd = pick_qpd(g);
		if (g->cabac)
			c_qpd(g, qpd);
		else
			bw_se(w, qpd);
	} else if (g->cabac) {
		g->prev_qpd_nz = 0;
	}
	rec_head(rec, mbt, cbp, t8 ? MBF_T8X8 : 0, t, qpd);
	if (g->cabac) {
		g->ci[i].cbp = (uint8_t)cbp;
		if (cbp)
			c_residual(g, i, 0, luma, chroma, t8, rec);
		c_count_nz(g, i, 0, rec);
	} else if (cbp) {
		put_residual(g, i, 0, luma, chroma, rec);
	}
}

/* ---- slice header (7.3.3) ------------------------------------------------ */
static void put_slice_header(struct gen *g)
{
	const struct h264gen_cfg *c = g->c;
	const struct h264gen_slice *s = g->s;
	struct bw *w = &g->w;
	uint32_t i;
	bw_u(w, (s->nal_ref_idc << 5) | (s->idr ? 5 : 1), 8);
	bw_ue(w, s->first_mb);
	bw_ue(w, s->slice_type);
	bw_ue(w, c->pps_id);
	bw_u_wide(w, s->frame_num, c->log2_max_frame_num_minus4 + 4u);
	if (c->field_coding) {
		bw_u(w, s->field_pic != 0, 1); /* field_pic_flag */
		if (s->field_pic)
			bw_u(w, s->field_pic == 2, 1); /* bottom_field_flag */
	}
	if (s->idr)
		bw_ue(w, s->idr_pic_id);
	if (c->pic_order_cnt_type == 0) {
		bw_u_wide(w, s->poc_lsb, c->log2_max_poc_lsb_minus4 + 4u);
		if (c->bottom_field_pic_order_in_frame_present_flag &&
		    !(c->field_coding && s->field_pic))
			bw_se(w, (int32_t)rng_below(&g->r, 9) - 4);
	}
	if (c->pic_order_cnt_type == 1 && !c->delta_pic_order_always_zero_flag) {
		bw_se(w, (int32_t)rng_below(&g->r, 41) - 20);
		if (c->bottom_field_pic_order_in_frame_present_flag &&
		    !(c->field_coding && s->field_pic))
			bw_se(w, (int32_t)rng_below(&g->r, 41) - 20);
	}
	if (c->redundant_pic_cnt_present_flag)
		bw_ue(w, 0);
	if (g->st == 1)
		bw_u(w, 1, 1); /* direct_spatial_mv_pred_flag */
	if (g->st == 0 || g->st == 1) {
		if (s->hdr_extras & H264GEN_HDR_OVERRIDE) {
			bw_u(w, 1, 1);
			bw_ue(w, s->num_ref_idx_l0_minus1);
			if (g->st == 1)
				bw_ue(w, s->num_ref_idx_l1_minus1);
		} else {
			bw_u(w, 0, 1);
		}
		/* ref_pic_list_modification */
		for (i = 0; i < (g->st == 1 ? 2u : 1u); i++) {
			if (s->hdr_extras & H264GEN_HDR_RPLM) {
				bw_u(w, 1, 1);
				bw_ue(w, 0);
				bw_ue(w, rng_below(&g->r, 4));
				bw_ue(w, 1);
				bw_ue(w, rng_below(&g->r, 4));
				bw_ue(w, 2);
				bw_ue(w, rng_below(&g->r, 3));
				bw_ue(w, 3);
			} else {
				bw_u(w, 0, 1);
			}
		}
	}
	if ((c->weighted_pred_flag && g->st == 0) ||
	    (c->weighted_bipred_idc == 1 && g->st == 1)) {
		uint32_t l, k;
		bw_ue(w, rng_below(&g->r, 8)); /* luma_log2_weight_denom */
		if (g->cat != 0)
			bw_ue(w, rng_below(&g->r, 8));
		for (l = 0; l < (g->st == 1 ? 2u : 1u); l++) {
			uint32_t n = (l == 0 ? g->nref0 : g->nref1) + 1;
			for (k = 0; k < n; k++) {
				int f = (int)rng_below(&g->r, 2);
				bw_u(w, (uint32_t)f, 1);
				if (f) {
					bw_se(w, (int32_t)rng_below(&g->r, 256) - 128);
					bw_se(w, (int32_t)rng_below(&g->r, 256) - 128);
				}
				if (g->cat != 0) {
					f = (int)rng_below(&g->r, 2);
					bw_u(w, (uint32_t)f, 1);
					if (f) {
						bw_se(w, (int32_t)rng_below(&g->r, 256) - 128);
						bw_se(w, (int32_t)rng_below(&g->r, 256) - 128);
						bw_se(w, (int32_t)rng_below(&g->r, 256) - 128);
						bw_se(w, (int32_t)rng_below(&g->r, 256) - 128);
					}
				}
			}
		}
	}
	if (s->nal_ref_idc != 0) {
		if (s->idr) {
			bw_u(w, 0, 1);
			bw_u(w, 0, 1);
		} else if (s->hdr_extras & H264GEN_HDR_MMCO) {
			bw_u(w, 1, 1);
			bw_ue(w, 1);
			bw_ue(w, rng_below(&g->r, 4));
			bw_ue(w, 3);
			bw_ue(w, rng_below(&g->r, 4));
			bw_ue(w, rng_below(&g->r, 2));
			bw_ue(w, 2);
			bw_ue(w, rng_below(&g->r, 4));
			bw_ue(w, 4);
			bw_ue(w, rng_below(&g->r, 3));
			bw_ue(w, 6);
			bw_ue(w, rng_below(&g->r, 2));
			bw_ue(w, 0);
		} else {
			bw_u(w, 0, 1);
		}
	}
	if (c->entropy_coding_mode_flag && g->st != 2)
		bw_ue(w, s->cabac_init_idc);
	bw_se(w, s->slice_qp_delta);
	if (g->sp)
		bw_u(w, rng_below(&g->r, 2), 1); /* sp_for_switch_flag */
	if (g->sp || g->si)
		bw_se(w, (int32_t)rng_below(&g->r, 21) - 10); /* slice_qs_delta */
	if (c->deblocking_filter_control_present_flag) {
		bw_ue(w, s->disable_deblocking_filter_idc);
		if (s->disable_deblocking_filter_idc != 1) {
			bw_se(w, s->alpha_c0_offset_div2);
			bw_se(w, s->beta_offset_div2);
		}
	}
	if (c->num_slice_groups_minus1 > 0 && c->slice_group_map_type >= 3 &&
	    c->slice_group_map_type <= 5) {
		/* Ceil(Log2(PicSizeInMapUnits / SliceGroupChangeRate + 1)) bits */
		uint32_t v = c->width_mbs * c->height_mbs /
				     (c->slice_group_change_rate_minus1 + 1u) + 1u;
		uint32_t bits = 0;
		while (bits < 32 && (1ull << bits) < v)
			bits++;
		if (bits)
			bw_u_wide(w, s->slice_group_change_cycle, bits);
	}
}

/* bins (context coded, bypass, terminating) of all CABAC slices generated so far */
static uint64_t s_stat_bins;

uint64_t h264gen_stat_cabac_bins(int reset)
{
	uint64_t v = __atomic_load_n(&s_stat_bins, __ATOMIC_RELAXED);
	if (reset)
		__atomic_store_n(&s_stat_bins, 0, __ATOMIC_RELAXED);
	return v;
}

size_t h264gen_slice_bound(const struct h264gen_cfg *cfg,
			   const struct h264gen_slice *sl)
{
	uint32_t cat = chroma_array_type(cfg);
	/* worst case is all-PCM at full bit depth, escapes doubled */
	size_t per_mb = (size_t)(256 + (cat == 0 ? 0 : cat == 1 ? 128 : cat == 2 ? 256 : 512)) * 2 * 2 + 64;
	if (sl->pm_pcm == 0 && sl->pm_escape == 0)
		per_mb = 48 * 16 * 2 + 256; /* dense small levels */
	else if (sl->pm_pcm == 0)
		per_mb = 48 * 16 * 6 + 256;
	return 12288 + (size_t)sl->num_mbs * per_mb + sl->cabac_payload * 2;
}

int h264gen_slice_nal(const struct h264gen_cfg *cfg,
		      const struct h264gen_slice *sl, uint8_t *out, size_t cap,
		      uint8_t *t2, uint32_t t2_stride)
{
	struct gen g;
	uint32_t i, run = 0;
	int skipped_prev = 0;
	pthread_once(&s_once, tables_init);
	if (cfg->width_mbs == 0 || sl->num_mbs == 0)
		return -EINVAL;
	memset(&g, 0, sizeof(g));
	g.c = cfg;
	g.s = sl;
	g.r.s = sl->seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
	if (g.r.s == 0)
		g.r.s = 1;
	g.cat = chroma_array_type(cfg);
	g.W = cfg->width_mbs;
	g.st = sl->slice_type % 5;
	g.sp = g.st == 3;
	g.si = g.st == 4;
	g.st = g.sp ? 0 : g.si ? 2 : g.st;
	g.nref0 = (sl->hdr_extras & H264GEN_HDR_OVERRIDE)
			  ? sl->num_ref_idx_l0_minus1
			  : cfg->num_ref_idx_l0_default_minus1;
	g.nref1 = ((sl->hdr_extras & H264GEN_HDR_OVERRIDE) && g.st == 1)
			  ? sl->num_ref_idx_l1_minus1
			  : cfg->num_ref_idx_l1_default_minus1;
	g.t2 = t2;
	g.t2_stride = t2_stride;
	t_field_flag = (cfg->field_coding && sl->field_pic) ? 0x04u : 0u;
	bw_init(&g.w, out, cap);
	put_slice_header(&g);

	if (cfg->entropy_coding_mode_flag && !sl->cabac_syntax) {
		/* opaque CABAC payload: the reader only records its extent */
		while (g.w.nbits != 0)
			bw_u(&g.w, 1, 1);
		for (i = 0; i < sl->cabac_payload; i++)
			bw_u(&g.w, rng_below(&g.r, 6) == 0 ? 0 : rng_below(&g.r, 256), 8);
		bw_u(&g.w, 0x80, 8);
		return finish(&g.w);
	}

	g.mbaff = cfg->field_coding == 2 && !sl->field_pic;
	if (g.mbaff && (sl->num_mbs & 1u))
		return -EINVAL; /* whole macroblock pairs */
	g.cabac = cfg->entropy_coding_mode_flag != 0;
	if (g.cabac && g.mbaff)
		return -EINVAL; /* no CABAC generator for MBAFF frames */
	g.nz = calloc((size_t)sl->num_mbs, 48);
	g.ff = calloc((size_t)sl->num_mbs, 1);
	if (g.cabac)
		g.ci = calloc((size_t)sl->num_mbs, sizeof(*g.ci));
	if (g.nz == NULL || g.ff == NULL || (g.cabac && g.ci == NULL)) {
		free(g.nz);
		free(g.ff);
		free(g.ci);
		return -ENOMEM;
	}
	if (g.cabac) {
		/* cabac_alignment_one_bit, then the arithmetic code (7.3.4) */
		while (g.w.nbits != 0)
			bw_u(&g.w, 1, 1);
		g.ce.w = &g.w;
		ce_start(&g.ce);
		ce_init_states(&g.ce, g.st == 2, sl->cabac_init_idc,
			       26 + cfg->pic_init_qp_minus26 + sl->slice_qp_delta,
			       sl->cabac_syntax == 2 ? 0 : 1);
	}
	if (cfg->num_slice_groups_minus1 > 0) {
		/* the slice codes macroblocks of first_mb's group, in address
		 * order (8.2.2: nextMbAddress) */
		uint32_t units = cfg->width_mbs * cfg->height_mbs;
		uint32_t total = (cfg->field_coding && !sl->field_pic) ? 2u * units
								       : units;
		uint8_t *map = malloc(total ? total : 1);
		uint32_t a, k = 0;
		long nm;
		g.addr_of = malloc((size_t)sl->num_mbs * sizeof(uint32_t));
		g.idx_of = malloc((size_t)(total ? total : 1) * sizeof(int32_t));
		nm = map ? h264gen_mb_to_slice_group_map(
				   cfg, sl->slice_group_change_cycle,
				   sl->field_pic, map, total)
			 : -ENOMEM;
		if (nm < 0 || !g.addr_of || !g.idx_of ||
		    sl->first_mb * (g.mbaff ? 2u : 1u) >= total) {
			free(map);
			free(g.addr_of);
			free(g.idx_of);
			free(g.nz);
			free(g.ff);
			free(g.ci);
			return nm < 0 ? (int)nm : -EINVAL;
		}
		for (a = 0; a < total; a++)
			g.idx_of[a] = -1;
		/* first_mb_in_slice counts pairs in an MBAFF frame */
		for (a = sl->first_mb * (g.mbaff ? 2u : 1u);
		     a < total && k < sl->num_mbs; a++)
			if (map[a] == map[sl->first_mb * (g.mbaff ? 2u : 1u)]) {
				g.idx_of[a] = (int32_t)k;
				g.addr_of[k++] = a;
			}
		free(map);
		if (k != sl->num_mbs) { /* the group has fewer macroblocks left */
			free(g.addr_of);
			free(g.idx_of);
			free(g.nz);
			free(g.ff);
			free(g.ci);
			return -ERANGE;
		}
		g.n_addr = total;
	}
	if (t2)
		memset(t2, 0, (size_t)sl->num_mbs * t2_stride);
	for (i = 0; i < sl->num_mbs; i++) {
		uint8_t *rec = rec_of(&g, i);
		const int top = !(i & 1u);
		if (g.st == 2) {
			if (g.mbaff) {
				/* mb_field_decoding_flag with the top macroblock
				 * of every pair (7.3.4) */
				if (top) {
					g.ff[i] = g.ff[i + 1] =
						(uint8_t)rng_below(&g.r, 2);
					bw_u(&g.w, g.ff[i], 1);
				}
				t_field_flag = g.ff[i] ? 0x04u : 0u;
			}
			put_intra_mb(&g, i, g.si ? 1 : 0, rec);
			if (rec)
				memcpy(rec + 16, g.nz + (size_t)i * 48, 48);
			if (g.cabac) /* end_of_slice_flag */
				ce_term(&g.ce, i + 1 == sl->num_mbs);
			continue;
		}
		/* P / B */
		{
			int last = i + 1 == sl->num_mbs;
			int skip = rng_pm(&g.r, sl->pm_skip);
			if (last)
				skip = sl->end_with_skip ? 1 : 0;
			if (g.cabac) {
				/* mb_skip_flag per macroblock instead of runs */
				c_skip_flag(&g, i, (uint32_t)skip);
				if (skip) {
					g.ci[i].skip = 1;
					g.ci[i].mbt = g.st == 1 ? MB_B_SKIP : MB_P_SKIP;
					g.prev_qpd_nz = 0;
					rec_head(rec, g.st == 1 ? MB_B_SKIP : MB_P_SKIP,
						 0, MBF_SKIPPED, 0, 0);
				} else {
					if (rng_pm(&g.r, sl->pm_intra_in_inter))
						put_intra_mb(&g, i, g.st == 1 ? 23 : 5, rec);
					else
						put_inter_mb(&g, i, rec);
					if (rec)
						memcpy(rec + 16, g.nz + (size_t)i * 48, 48);
				}
				ce_term(&g.ce, (uint32_t)last);
				continue;
			}
			if (skip) {
				run++;
				if (g.mbaff) {
					if (top) {
						/* decided with the bottom macroblock;
						 * until then it counts as frame */
						g.ff[i] = 0;
					} else if (!skipped_prev) {
						g.ff[i] = g.ff[i - 1];
					} else {
						/* both skipped: inferred from the
						 * pair to the left, else the one
						 * above, else frame (7.4.4) */
						uint32_t addr = mb_address(&g, i);
						int64_t pair = addr / 2, n = -1;
						if (pair % g.W != 0)
							n = pair_top_index(&g, i, pair - 1);
						if (n < 0)
							n = pair_top_index(
								&g, i,
								pair - (int64_t)g.W);
						g.ff[i] = g.ff[i - 1] =
							n >= 0 ? g.ff[n] : 0;
					}
					t_field_flag = g.ff[i] ? 0x04u : 0u;
				}
				skipped_prev = 1;
				rec_head(rec, g.st == 1 ? MB_B_SKIP : MB_P_SKIP,
					 0, MBF_SKIPPED, 0, 0);
				if (last)
					bw_ue(&g.w, run);
				continue;
			}
			bw_ue(&g.w, run);
			if (g.mbaff) {
				if (top) {
					g.ff[i] = (uint8_t)rng_below(&g.r, 2);
					bw_u(&g.w, g.ff[i], 1);
				} else if (skipped_prev) {
					/* the top macroblock was skipped: the flag
					 * comes with this one and counts for both */
					g.ff[i] = g.ff[i - 1] =
						(uint8_t)rng_below(&g.r, 2);
					bw_u(&g.w, g.ff[i], 1);
				} else {
					g.ff[i] = g.ff[i - 1];
				}
				t_field_flag = g.ff[i] ? 0x04u : 0u;
			}
			run = 0;
			skipped_prev = 0;
			if (rng_pm(&g.r, sl->pm_intra_in_inter))
				put_intra_mb(&g, i, g.st == 1 ? 23 : 5, rec);
			else
				put_inter_mb(&g, i, rec);
			if (rec)
				memcpy(rec + 16, g.nz + (size_t)i * 48, 48);
		}
	}
	if (g.cabac) {
		__atomic_add_fetch(&s_stat_bins, g.ce.bins, __ATOMIC_RELAXED);
		/* the encoder's flush wrote the stop bit (9.3.4.5) */
		while (g.w.nbits != 0)
			bw_u(&g.w, 0, 1);
	} else {
		bw_trailing(&g.w);
	}
	free(g.nz);
	free(g.ff);
	free(g.ci);
	free(g.addr_of);
	free(g.idx_of);
	return finish(&g.w);
}

/* ---- threaded batch ------------------------------------------------------ */
struct batch {
	const struct h264gen_cfg *cfg;
	const struct h264gen_slice *slices;
	uint32_t n, first, step;
	uint8_t *out;
	const uint64_t *offs;
	uint32_t *sizes;
	const size_t *caps;
	int err;
};

static void *batch_worker(void *arg)
{
	struct batch *b = arg;
	uint32_t i;
	for (i = b->first; i < b->n; i += b->step) {
		int r = h264gen_slice_nal(b->cfg, &b->slices[i],
					  b->out + b->offs[i], b->caps[i], NULL,
					  0);
		if (r < 0) {
			b->err = r;
			b->sizes[i] = 0;
		} else {
			b->sizes[i] = (uint32_t)r;
		}
	}
	return NULL;
}

/* Two passes: slots sized by the bound, then compacted in order. */
long h264gen_batch(const struct h264gen_cfg *cfg,
		   const struct h264gen_slice *slices, uint32_t n, int threads,
		   uint8_t *out, size_t cap, uint64_t *offs, uint32_t *sizes)
{
	struct batch *jobs;
	pthread_t *tids;
	size_t *caps;
	uint64_t pos = 0;
	uint32_t i;
	int t, err = 0;
	if (threads < 1)
		threads = 1;
	caps = calloc(n ? n : 1, sizeof(*caps));
	jobs = calloc((size_t)threads, sizeof(*jobs));
	tids = calloc((size_t)threads, sizeof(*tids));
	if (!caps || !jobs || !tids)
		return -ENOMEM;
	for (i = 0; i < n; i++) {
		caps[i] = (h264gen_slice_bound(cfg, &slices[i]) + 15) & ~(size_t)15;
		offs[i] = pos;
		pos += caps[i];
	}
	if (pos > cap) {
		free(caps);
		free(jobs);
		free(tids);
		return -ENOBUFS;
	}
	for (t = 0; t < threads; t++) {
		jobs[t].cfg = cfg;
		jobs[t].slices = slices;
		jobs[t].n = n;
		jobs[t].first = (uint32_t)t;
		jobs[t].step = (uint32_t)threads;
		jobs[t].out = out;
		jobs[t].offs = offs;
		jobs[t].sizes = sizes;
		jobs[t].caps = caps;
		pthread_create(&tids[t], NULL, batch_worker, &jobs[t]);
	}
	for (t = 0; t < threads; t++) {
		pthread_join(tids[t], NULL);
		if (jobs[t].err)
			err = jobs[t].err;
	}
	/* compact to 16-byte aligned, back-to-back layout */
	pos = 0;
	for (i = 0; i < n && !err; i++) {
		if (offs[i] != pos)
			memmove(out + pos, out + offs[i], sizes[i]);
		offs[i] = pos;
		pos = (pos + sizes[i] + 15) & ~(uint64_t)15;
	}
	free(caps);
	free(jobs);
	free(tids);
	return err ? err : (long)pos;
}

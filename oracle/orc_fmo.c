/**
 * oracle/orc_fmo.c -- TEST INFRASTRUCTURE: CPU restatement of the reference's
 * slice-group code (ref src/h264_fmo.c:33-319: the seven map types of 8.2.2.1
 * - 8.2.2.7, h264_gen_slice_group_map, h264_next_mb_addr).  Used by tests/ as
 * the checker of the slice-decode kernel's FMO path; nothing under
 * libh264_amd/ links it.
 *
 * Where the reference indexes without bounds, a restatement has to pick:
 *  - type 2 (:77-99): rectangles come straight from the PPS; a cell outside
 *    the map (bottom_right beyond the picture) is a heap overflow there, and
 *    is SKIPPED here;
 *  - type 6 (:215-218): slice_group_id[] holds 256 entries and the PPS parser
 *    refuses more (product and oracle: -EIO for pic_size_in_map_units_minus1
 *    >= 256); a picture of more map units than the PPS carries ids for reads
 *    past the ids it has: group 0 here (the reference reads whatever follows
 *    in the struct);
 *  - type 3 (:105-160): the box-out walk is cut off after
 *    H264_AMD_FMO_BOX_OUT_STEPS(map units) steps (see map_type_3);
 *  - h264_next_mb_addr (:307-319) looks the current address up even when a
 *    slice has run off the picture; the value is only compared inside a loop
 *    that cannot run then, so "beyond the picture the next address is
 *    mbAddr + 1" is exact, not a choice.
 * A map type above 6 is -EIO from h264_gen_slice_group_map; the reference
 * ignores that status at its call sites (src/h264_syntax_slice_data.h:723,
 * src/h264_ctx.c:123) and goes on with the map it had (zeros after
 * h264_clear_slice_group_map): all macroblocks in group 0.  Same here.
 */
#include "orc.h"

#include <stdlib.h>
#include <string.h>

static uint32_t umin(uint32_t a, uint32_t b)
{
	return a < b ? a : b;
}

/* ref :33-52 */
static void map_type_0(struct orc_ctx *ctx, uint32_t *map, uint32_t n)
{
	const struct h264_pps *pps = ctx->pps;
	/* 64-bit position: run_length_minus1 + 1 is added in 32 bits by the
	 * reference, wraps for 2^32 - 2 and its loop then need not end
	 * (h264_amd.h, H264_AMD_FMO_MAX_PIC_MBS) */
	uint64_t i = 0, j;
	uint32_t grp;
	do {
		for (grp = 0; grp <= pps->num_slice_groups_minus1 && i < n;
		     i += (uint64_t)pps->run_length_minus1[grp++] + 1) {
			for (j = 0; j <= pps->run_length_minus1[grp] && i + j < n;
			     j++)
				map[i + j] = grp;
		}
	} while (i < n);
}

/* ref :58-72 */
static void map_type_1(struct orc_ctx *ctx, uint32_t *map, uint32_t n)
{
	uint32_t W = ctx->sps_derived.PicWidthInMbs;
	uint32_t g = ctx->pps->num_slice_groups_minus1 + 1, i;
	for (i = 0; i < n; i++)
		map[i] = ((i % W) + (((i / W) * g) / 2)) % g;
}

/* ref :77-99 */
static void map_type_2(struct orc_ctx *ctx, uint32_t *map, uint32_t n)
{
	const struct h264_pps *pps = ctx->pps;
	uint32_t W = ctx->sps_derived.PicWidthInMbs, i, x, y;
	int32_t grp;
	for (i = 0; i < n; i++)
		map[i] = pps->num_slice_groups_minus1;
	for (grp = (int32_t)pps->num_slice_groups_minus1 - 1; grp >= 0; grp--) {
		uint32_t yTL = pps->top_left[grp] / W, xTL = pps->top_left[grp] % W;
		uint32_t yBR = pps->bottom_right[grp] / W;
		uint32_t xBR = pps->bottom_right[grp] % W;
		for (y = yTL; y <= yBR; y++) {
			uint64_t base = (uint64_t)y * W;
			if (base >= n)
				break; /* outside the map: see the header */
			for (x = xTL; x <= xBR; x++)
				if (base + x < n)
					map[base + x] = (uint32_t)grp;
		}
	}
}

/* ref :105-160 */
static void map_type_3(struct orc_ctx *ctx, uint32_t *map, uint32_t n)
{
	int dir = ctx->pps->slice_group_change_direction_flag;
	uint32_t W = ctx->sps_derived.PicWidthInMbs;
	uint32_t H = ctx->sps_derived.PicHeightInMapUnits;
	uint32_t units0 = ctx->derived.MapUnitsInSliceGroup0;
	uint32_t i, k, vacant = 0;
	uint32_t x = (W - (uint32_t)dir) / 2, y = (H - (uint32_t)dir) / 2;
	uint32_t left = x, top = y, right = x, bottom = y;
	int32_t xDir = dir - 1, yDir = dir;
	/* the walk revisits filled units along a box that the picture's edge
	 * has clipped: in a picture a few rows high and a thousand columns wide
	 * a lap round the box costs two thousand steps and fills a dozen units.
	 * Product and oracle give up after H264_AMD_FMO_BOX_OUT_STEPS(n) steps
	 * (h264_amd.h) and leave what was not reached in group 1; the reference
	 * walks on, up to n^2 steps */
	uint64_t steps = H264_AMD_FMO_BOX_OUT_STEPS(n);
	for (i = 0; i < n; i++)
		map[i] = 1;
	for (k = 0; k < units0 && steps-- > 0; k += vacant) {
		if (map[y * W + x] == 1) {
			vacant = 1;
			map[y * W + x] = 0;
		} else {
			vacant = 0;
		}
		if (xDir == -1 && x == left) {
			left = left > 0 ? left - 1 : 0;
			x = left;
			xDir = 0;
			yDir = 2 * dir - 1;
		} else if (xDir == 1 && x == right) {
			right = umin(right + 1, W - 1);
			x = right;
			xDir = 0;
			yDir = 1 - 2 * dir;
		} else if (yDir == -1 && y == top) {
			top = top > 0 ? top - 1 : 0;
			y = top;
			xDir = 1 - 2 * dir;
			yDir = 0;
		} else if (yDir == 1 && y == bottom) {
			bottom = umin(bottom + 1, H - 1);
			y = bottom;
			xDir = 2 * dir - 1;
			yDir = 0;
		} else {
			x = (uint32_t)((int32_t)x + xDir);
			y = (uint32_t)((int32_t)y + yDir);
		}
	}
}

/* ref :166-181 */
static void map_type_4(struct orc_ctx *ctx, uint32_t *map, uint32_t n)
{
	uint32_t dir = (uint32_t)ctx->pps->slice_group_change_direction_flag;
	uint32_t units0 = ctx->derived.MapUnitsInSliceGroup0;
	uint32_t upper = dir ? n - units0 : units0, i;
	for (i = 0; i < n; i++)
		map[i] = i < upper ? dir : 1 - dir;
}

/* ref :187-210 */
static void map_type_5(struct orc_ctx *ctx, uint32_t *map, uint32_t n)
{
	uint32_t dir = (uint32_t)ctx->pps->slice_group_change_direction_flag;
	uint32_t W = ctx->sps_derived.PicWidthInMbs;
	uint32_t H = ctx->sps_derived.PicHeightInMapUnits;
	uint32_t units0 = ctx->derived.MapUnitsInSliceGroup0;
	uint32_t upper = dir ? n - units0 : units0, i, j, k = 0;
	for (j = 0; j < W; j++)
		for (i = 0; i < H; i++)
			map[i * W + j] = k++ < upper ? dir : 1 - dir;
}

/* ref :215-218 */
static void map_type_6(struct orc_ctx *ctx, uint32_t *map, uint32_t n)
{
	uint32_t i;
	uint32_t have = umin(ctx->pps->pic_size_in_map_units_minus1 + 1u, 256u);
	for (i = 0; i < n; i++)
		map[i] = i < have ? ctx->pps->slice_group_id[i] : 0u;
}

/* ref :244-291 */
int orc_gen_slice_group_map(struct orc_ctx *ctx)
{
	uint32_t n = ctx->sps_derived.PicSizeInMapUnits;
	if (ctx->pps->num_slice_groups_minus1 == 0)
		return 0;
	/* slice data of such a picture is not walked (h264_amd.h,
	 * H264_AMD_FMO_MAX_PIC_MBS): nothing ever looks at its map */
	{
		/* the reference multiplies width and height in 32 bits and
		 * walks both separately (types 3 and 5): a product that wraps
		 * is a heap overflow there */
		uint64_t n64 = (uint64_t)ctx->sps_derived.PicWidthInMbs *
			       ctx->sps_derived.PicHeightInMapUnits;
		if (n64 != n || n > H264_AMD_FMO_MAX_PIC_MBS ||
		    (ctx->pps->slice_group_map_type == 3 &&
		     n > H264_AMD_FMO_MAX_BOX_OUT_UNITS))
			return -E2BIG;
	}
	if (n > ctx->slice.group_map_maxlen) {
		uint32_t *nm = realloc(ctx->slice.group_map,
				       (size_t)n * sizeof(uint32_t));
		if (nm == NULL)
			return -ENOMEM;
		memset(nm + ctx->slice.group_map_maxlen, 0,
		       (size_t)(n - ctx->slice.group_map_maxlen) * sizeof(uint32_t));
		ctx->slice.group_map = nm;
		ctx->slice.group_map_maxlen = n;
	}
	switch (ctx->pps->slice_group_map_type) {
	case 0:
		map_type_0(ctx, ctx->slice.group_map, n);
		break;
	case 1:
		map_type_1(ctx, ctx->slice.group_map, n);
		break;
	case 2:
		map_type_2(ctx, ctx->slice.group_map, n);
		break;
	case 3:
		map_type_3(ctx, ctx->slice.group_map, n);
		break;
	case 4:
		map_type_4(ctx, ctx->slice.group_map, n);
		break;
	case 5:
		map_type_5(ctx, ctx->slice.group_map, n);
		break;
	case 6:
		map_type_6(ctx, ctx->slice.group_map, n);
		break;
	default:
		return -EIO;
	}
	return 0;
}

/* ref :294-301 */
void orc_clear_slice_group_map(struct orc_ctx *ctx)
{
	if (ctx->slice.group_map != NULL)
		memset(ctx->slice.group_map, 0,
		       (size_t)ctx->slice.group_map_maxlen * sizeof(uint32_t));
}

/* 8.2.2.8 (ref :224-238) */
static uint32_t mb_to_slice_group(struct orc_ctx *ctx, uint32_t mbAddr)
{
	uint32_t W = ctx->sps_derived.PicWidthInMbs, unit;
	if (ctx->sps->frame_mbs_only_flag || ctx->slice.hdr.field_pic_flag)
		unit = mbAddr;
	else if (ctx->derived.MbaffFrameFlag)
		unit = mbAddr / 2;
	else
		unit = (mbAddr / (2 * W)) * W + (mbAddr % W);
	return unit < ctx->slice.group_map_maxlen ? ctx->slice.group_map[unit]
						  : 0u;
}

/* ref :307-319 */
uint32_t orc_next_mb_addr(struct orc_ctx *ctx, uint32_t mbAddr)
{
	uint32_t group, i;
	if (ctx->pps->num_slice_groups_minus1 == 0)
		return mbAddr + 1;
	i = mbAddr + 1;
	if (i >= ctx->derived.PicSizeInMbs)
		return i; /* the loop below cannot run: see the header */
	group = mb_to_slice_group(ctx, mbAddr);
	while (i < ctx->derived.PicSizeInMbs &&
	       mb_to_slice_group(ctx, i) != group)
		i++;
	return i;
}

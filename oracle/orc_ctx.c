/**
 * oracle/orc_ctx.c -- TEST INFRASTRUCTURE (see orc.h).
 * Restatement of the reference's context layer:
 *   src/h264_ctx.c:54-124   derived variables (SPS / PPS / slice)
 *   src/h264_ctx.c:130-223  first-VCL-NAL detection (7.4.1.2.4)
 *   src/h264_ctx.c:226-398  lifecycle, SPS/PPS tables, activation
 *   src/h264_ctx.c:401-481  SEI table
 *   src/h264_ctx.c:545-574  slice header installation
 *   src/h264_ctx.c:577-671  stream summary
 *   src/h264.c:36-117       h264_get_sps_derived
 *   src/h264_types.c:311-349 SEI internal buffer fix-up
 *   src/h264_priv.h:185-193 intlog2
 */
#include "orc.h"

#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))

/* Deviation (product, kernels and oracle alike): the reference's loop shifts a
 * 32-bit value by 32 once x >= 2^31 -- undefined, and an endless loop on x86
 * (reachable from a damaged SPS/PPS: slice_group_change_cycle width,
 * src/h264_syntax.h:1365-1378).  Here the answer is the mathematical one. */
uint32_t orc_intlog2(uint32_t x)
{
	uint32_t r = 0;
	while (r < 32 && (x >> r) > 0)
		r++;
	if (r > 0 && x == (1u << (r - 1)))
		r--;
	return r;
}

int orc_get_sps_derived(const struct h264_sps *sps, struct h264_sps_derived *d)
{
	static const uint32_t sub[4][4] = {
		/* SubWidthC, SubHeightC, MbWidthC, MbHeightC */
		{0, 0, 0, 0},
		{2, 2, 8, 8},
		{2, 1, 8, 16},
		{1, 1, 16, 16},
	};
	uint32_t cat, fmo;
	if (sps == NULL || d == NULL)
		return -EINVAL;
	memset(d, 0, sizeof(*d));
	cat = sps->separate_colour_plane_flag ? 0 : sps->chroma_format_idc;
	d->ChromaArrayType = cat;
	if (cat <= 3) {
		d->SubWidthC = sub[cat][0];
		d->SubHeightC = sub[cat][1];
		d->MbWidthC = sub[cat][2];
		d->MbHeightC = sub[cat][3];
	}
	d->BitDepthLuma = 8 + sps->bit_depth_luma_minus8;
	d->QpBdOffsetLuma = 6 * sps->bit_depth_luma_minus8;
	d->BitDepthChroma = 8 + sps->bit_depth_chroma_minus8;
	d->QpBdOffsetChroma = 6 * sps->bit_depth_chroma_minus8;
	d->RawMbBits = 256 * d->BitDepthLuma +
		       2 * d->MbWidthC * d->MbHeightC * d->BitDepthChroma;
	/* out-of-range exponents: shift count taken mod 32 like the x86 code the
	 * reference compiles to, without the undefined behaviour */
	d->MaxFrameNum = 1u << ((sps->log2_max_frame_num_minus4 + 4) & 31u);
	d->MaxPicOrderCntLsb =
		1u << ((sps->log2_max_pic_order_cnt_lsb_minus4 + 4) & 31u);
	d->PicWidthInMbs = sps->pic_width_in_mbs_minus1 + 1;
	d->PicWidthInSamplesLuma = 16 * d->PicWidthInMbs;
	d->PicWidthInSamplesChroma = d->PicWidthInMbs * d->MbWidthC;
	d->PicHeightInMapUnits = sps->pic_height_in_map_units_minus1 + 1;
	d->PicSizeInMapUnits = d->PicWidthInMbs * d->PicHeightInMapUnits;
	fmo = 2 - sps->frame_mbs_only_flag;
	d->FrameHeightInMbs = fmo * d->PicHeightInMapUnits;
	if (cat == 0) {
		d->CropUnitX = 1;
		d->CropUnitY = fmo;
	} else {
		d->CropUnitX = d->SubWidthC;
		d->CropUnitY = d->SubHeightC * fmo;
	}
	d->Width = d->PicWidthInSamplesLuma -
		   d->CropUnitX * (sps->frame_crop_left_offset +
				   sps->frame_crop_right_offset);
	d->Height = 16 * d->FrameHeightInMbs -
		    d->CropUnitY * (sps->frame_crop_top_offset +
				    sps->frame_crop_bottom_offset);
	return 0;
}

/* ref src/h264_ctx.c:81-124 */
static void update_derived_slice(struct orc_ctx *ctx)
{
	const struct h264_sps *sps = ctx->sps;
	const struct h264_pps *pps = ctx->pps;
	const struct h264_slice_header *sh = &ctx->slice.hdr;
	if (sps == NULL || pps == NULL)
		return;
	ctx->derived.MbaffFrameFlag =
		sps->mb_adaptive_frame_field_flag && !sh->field_pic_flag;
	ctx->derived.PicHeightInMbs =
		ctx->sps_derived.FrameHeightInMbs / (1 + sh->field_pic_flag);
	ctx->derived.PicSizeInMbs =
		ctx->sps_derived.PicWidthInMbs * ctx->derived.PicHeightInMbs;
	ctx->derived.MaxPicNum = sh->field_pic_flag
					 ? 2 * ctx->sps_derived.MaxFrameNum
					 : ctx->sps_derived.MaxFrameNum;
	ctx->derived.CurrPicNum =
		sh->field_pic_flag ? 2 * sh->frame_num + 1 : sh->frame_num;
	ctx->derived.SliceQPLuma =
		(int32_t)((uint32_t)pps->pic_init_qp_minus26 + 26u +
			  (uint32_t)sh->slice_qp_delta); /* wraps, defined */
	ctx->derived.QSLuma = (int32_t)((uint32_t)pps->pic_init_qs_minus26 + 26u +
					(uint32_t)sh->slice_qs_delta);
	/* (the reference shifts the signed value left, src/h264_ctx.c:116-117;
	 * same result, defined) */
	ctx->derived.FilterOffsetA =
		(int32_t)((uint32_t)sh->slice_alpha_c0_offset_div2 * 2u);
	ctx->derived.FilterOffsetB =
		(int32_t)((uint32_t)sh->slice_beta_offset_div2 * 2u);
	ctx->derived.MapUnitsInSliceGroup0 =
		ORC_MIN(sh->slice_group_change_cycle *
				ctx->derived.SliceGroupChangeRate,
			ctx->sps_derived.PicSizeInMapUnits);
	if (pps->num_slice_groups_minus1 > 0)
		orc_gen_slice_group_map(ctx); /* status ignored, ref :122-123 */
}

static void update_derived_sps(struct orc_ctx *ctx)
{
	if (ctx->sps != NULL)
		orc_get_sps_derived(ctx->sps, &ctx->sps_derived);
}

static void update_derived_pps(struct orc_ctx *ctx)
{
	if (ctx->pps != NULL)
		ctx->derived.SliceGroupChangeRate =
			ctx->pps->slice_group_change_rate_minus1 + 1;
}

/* ref src/h264_ctx.c:130-223 */
static void detect_first_vcl(struct orc_ctx *ctx)
{
	const struct h264_sps *sps = ctx->sps;
	const struct h264_nalu_header *nh = &ctx->nalu.hdr;
	const struct h264_slice_header *sh = &ctx->slice.hdr;
	const struct h264_slice_header *ph = &ctx->slice.prev_slice_hdr;
	const struct h264_nalu_header *pn = &ctx->slice.prev_slice_nalu_hdr;
	int idr = nh->nal_unit_type == H264_NALU_TYPE_SLICE_IDR;
	int pidr = pn->nal_unit_type == H264_NALU_TYPE_SLICE_IDR;
	int first = 0;

	if (!ctx->nalu.is_prev_vcl && !ctx->nalu.is_prev_filler)
		first = 1;
	else if (sh->frame_num != ph->frame_num)
		first = 1;
	else if (sh->pic_parameter_set_id != ph->pic_parameter_set_id)
		first = 1;
	else if (!sps->frame_mbs_only_flag &&
		 sh->field_pic_flag != ph->field_pic_flag)
		first = 1;
	else if (!sps->frame_mbs_only_flag && sh->field_pic_flag &&
		 ph->field_pic_flag &&
		 sh->bottom_field_flag != ph->bottom_field_flag)
		first = 1;
	else if (!nh->nal_ref_idc != !pn->nal_ref_idc)
		first = 1;
	else if (sps->pic_order_cnt_type == 0 &&
		 (sh->pic_order_cnt_lsb != ph->pic_order_cnt_lsb ||
		  sh->delta_pic_order_cnt_bottom !=
			  ph->delta_pic_order_cnt_bottom))
		first = 1;
	else if (sps->pic_order_cnt_type == 1 &&
		 (sh->delta_pic_order_cnt[0] != ph->delta_pic_order_cnt[0] ||
		  sh->delta_pic_order_cnt[1] != ph->delta_pic_order_cnt[1]))
		first = 1;
	else if (idr != pidr)
		first = 1;
	else if (idr && pidr && sh->idr_pic_id != ph->idr_pic_id)
		first = 1;

	ctx->nalu.is_first_vcl = first;
	ctx->slice.prev_slice_nalu_hdr = *nh;
	ctx->slice.prev_slice_hdr = *sh;
}

int orc_ctx_new(struct orc_ctx **ret)
{
	if (ret == NULL)
		return -EINVAL;
	*ret = calloc(1, sizeof(**ret));
	return *ret ? 0 : -ENOMEM;
}

int orc_ctx_clear_sei_table(struct orc_ctx *ctx)
{
	uint32_t i;
	if (ctx == NULL)
		return -EINVAL;
	for (i = 0; i < ctx->sei_count; i++)
		free(ctx->sei_table[i].raw.buf);
	free(ctx->sei_table);
	ctx->sei_table = NULL;
	ctx->sei_count = 0;
	return 0;
}

void orc_clear_macroblock_table(struct orc_ctx *ctx)
{
	if (ctx->slice.mb_table.info != NULL)
		memset(ctx->slice.mb_table.info, 0,
		       ctx->slice.mb_table.maxlen *
			       sizeof(*ctx->slice.mb_table.info));
	ctx->slice.mb_table.len = 0;
}

/* ref src/h264_ctx.c:545-560 */
int orc_ctx_clear_slice(struct orc_ctx *ctx)
{
	if (ctx == NULL)
		return -EINVAL;
	ctx->slice.type = 0;
	memset(&ctx->slice.hdr, 0, sizeof(ctx->slice.hdr));
	ctx->slice.rawdata.partial = 0;
	ctx->slice.rawdata.partialbits = 0;
	ctx->slice.rawdata.buf = NULL;
	ctx->slice.rawdata.len = 0;
	orc_clear_macroblock_table(ctx);
	orc_clear_slice_group_map(ctx);
	memset(&ctx->_mb, 0, sizeof(ctx->_mb));
	ctx->mb = NULL;
	update_derived_slice(ctx);
	return 0;
}

/* ref src/h264_ctx.c:267-281 */
int orc_ctx_clear_nalu(struct orc_ctx *ctx)
{
	int pv, pf;
	if (ctx == NULL)
		return -EINVAL;
	pv = ctx->nalu.is_prev_vcl;
	pf = ctx->nalu.is_prev_filler;
	memset(&ctx->nalu, 0, sizeof(ctx->nalu));
	ctx->nalu.is_prev_vcl = pv;
	ctx->nalu.is_prev_filler = pf;
	memset(&ctx->aud, 0, sizeof(ctx->aud));
	orc_ctx_clear_sei_table(ctx);
	orc_ctx_clear_slice(ctx);
	return 0;
}

int orc_ctx_clear(struct orc_ctx *ctx)
{
	size_t i;
	if (ctx == NULL)
		return -EINVAL;
	orc_ctx_clear_nalu(ctx);
	for (i = 0; i < 32; i++)
		free(ctx->sps_table[i]);
	for (i = 0; i < 256; i++)
		free(ctx->pps_table[i]);
	free(ctx->slice.mb_table.info);
	free(ctx->slice.group_map);
	memset(ctx, 0, sizeof(*ctx));
	return 0;
}

int orc_ctx_destroy(struct orc_ctx *ctx)
{
	if (ctx == NULL)
		return 0;
	orc_ctx_clear(ctx);
	free(ctx);
	return 0;
}

int orc_ctx_set_nalu_header(struct orc_ctx *ctx,
			    const struct h264_nalu_header *nh)
{
	if (ctx == NULL || nh == NULL)
		return -EINVAL;
	ctx->nalu.type = nh->nal_unit_type;
	ctx->nalu.hdr = *nh;
	return 0;
}

/* ref src/h264_ctx.c:310-331 */
int orc_ctx_set_sps(struct orc_ctx *ctx, const struct h264_sps *sps)
{
	struct h264_sps **slot;
	if (ctx == NULL || sps == NULL || sps->seq_parameter_set_id >= 32)
		return -EINVAL;
	slot = &ctx->sps_table[sps->seq_parameter_set_id];
	if (*slot == NULL) {
		*slot = calloc(1, sizeof(**slot));
		if (*slot == NULL)
			return -ENOMEM;
	}
	**slot = *sps;
	ctx->sps = *slot;
	update_derived_sps(ctx);
	update_derived_slice(ctx);
	return 0;
}

/* ref src/h264_ctx.c:334-355 */
int orc_ctx_set_pps(struct orc_ctx *ctx, const struct h264_pps *pps)
{
	struct h264_pps **slot;
	if (ctx == NULL || pps == NULL || pps->pic_parameter_set_id >= 256)
		return -EINVAL;
	slot = &ctx->pps_table[pps->pic_parameter_set_id];
	if (*slot == NULL) {
		*slot = calloc(1, sizeof(**slot));
		if (*slot == NULL)
			return -ENOMEM;
	}
	**slot = *pps;
	ctx->pps = *slot;
	update_derived_pps(ctx);
	update_derived_slice(ctx);
	return 0;
}

/* ref src/h264_ctx.c:366-375 */
int orc_ctx_set_active_sps(struct orc_ctx *ctx, uint32_t id)
{
	if (ctx == NULL || id >= 32 || ctx->sps_table[id] == NULL)
		return -EINVAL;
	ctx->sps = ctx->sps_table[id];
	update_derived_sps(ctx);
	update_derived_slice(ctx);
	return 0;
}

/* ref src/h264_ctx.c:378-386 */
int orc_ctx_set_active_pps(struct orc_ctx *ctx, uint32_t id)
{
	if (ctx == NULL || id >= 256 || ctx->pps_table[id] == NULL)
		return -EINVAL;
	ctx->pps = ctx->pps_table[id];
	update_derived_pps(ctx);
	return orc_ctx_set_active_sps(ctx, ctx->pps->seq_parameter_set_id);
}

/* ref src/h264_ctx.c:563-574 */
int orc_ctx_set_slice_header(struct orc_ctx *ctx,
			     const struct h264_slice_header *sh)
{
	if (ctx == NULL || sh == NULL)
		return -EINVAL;
	orc_ctx_clear_slice(ctx);
	ctx->slice.type = H264_SLICE_TYPE(sh->slice_type);
	ctx->slice.hdr = *sh;
	update_derived_slice(ctx);
	detect_first_vcl(ctx);
	return 0;
}

/* ref src/h264_ctx.c:413-435 */
int orc_ctx_add_sei_internal(struct orc_ctx *ctx, struct h264_sei **ret)
{
	struct h264_sei *nt;
	if (ret == NULL)
		return -EINVAL;
	*ret = NULL;
	if (ctx == NULL)
		return -EINVAL;
	nt = realloc(ctx->sei_table, (ctx->sei_count + 1) * sizeof(*nt));
	if (nt == NULL)
		return -ENOMEM;
	ctx->sei_table = nt;
	*ret = &nt[ctx->sei_count++];
	memset(*ret, 0, sizeof(**ret));
	return 0;
}

/* ref src/h264_types.c:311-349 */
int orc_sei_update_internal_buf(struct h264_sei *sei)
{
	uint32_t start = 0;
	const uint8_t **buf = NULL;
	size_t *len = NULL;
	if (sei == NULL || sei->raw.buf == NULL || sei->raw.len == 0)
		return -EINVAL;
	switch (sei->type) {
	case H264_SEI_TYPE_FILLER_PAYLOAD:
		buf = &sei->filler_payload.buf;
		len = &sei->filler_payload.len;
		break;
	case H264_SEI_TYPE_USER_DATA_REGISTERED:
		start = sei->user_data_registered.country_code == 0xff ? 2 : 1;
		buf = &sei->user_data_registered.buf;
		len = &sei->user_data_registered.len;
		break;
	case H264_SEI_TYPE_USER_DATA_UNREGISTERED:
		start = 16;
		buf = &sei->user_data_unregistered.buf;
		len = &sei->user_data_unregistered.len;
		break;
	default:
		break;
	}
	if (buf != NULL) {
		if (sei->raw.len < start)
			return -EINVAL;
		*buf = sei->raw.buf + start;
		*len = sei->raw.len - start;
	}
	return 0;
}

/* ref src/h264_ctx.c:438-481 */
int orc_ctx_add_sei(struct orc_ctx *ctx, const struct h264_sei *sei)
{
	struct h264_bitstream bs;
	struct h264_sei *ns = NULL;
	int res;
	if (ctx == NULL || sei == NULL)
		return -EINVAL;
	h264_bs_init(&bs, NULL, 0, 0);
	res = orc_ctx_add_sei_internal(ctx, &ns);
	if (res < 0)
		goto fail;
	*ns = *sei;
	res = orc_write_one_sei(&bs, ctx, ns);
	if (res < 0)
		goto fail;
	if (bs.cachebits % 8 != 0 || !bs.dynamic) {
		res = -EIO;
		goto fail;
	}
	ns->raw.buf = bs.data;
	ns->raw.len = bs.off;
	bs.dynamic = 0;
	res = orc_sei_update_internal_buf(ns);
	if (res < 0)
		goto fail;
	h264_bs_clear(&bs);
	return 0;
fail:
	if (ns != NULL) {
		free(ns->raw.buf);
		ctx->sei_count--;
	}
	h264_bs_clear(&bs);
	return res;
}

/* ref src/h264_ctx.c:577-671 */
int orc_get_info_from_ps(const struct h264_sps *sps,
			 const struct h264_sps_derived *d,
			 struct h264_info *info)
{
	static const unsigned sar[17][2] = {
		{1, 1},	  {1, 1},   {12, 11}, {10, 11}, {16, 11}, {40, 33},
		{24, 11}, {20, 11}, {32, 11}, {80, 33}, {18, 11}, {15, 11},
		{64, 33}, {160, 99}, {4, 3},  {3, 2},	{2, 1},
	};
	const struct h264_vui *vui = &sps->vui;
	memset(info, 0, sizeof(*info));
	info->width = d->PicWidthInSamplesLuma;
	info->height = d->FrameHeightInMbs * 16;
	info->bit_depth_luma = d->BitDepthLuma;
	info->crop_width = info->width;
	info->crop_height = info->height;
	if (sps->frame_cropping_flag) {
		info->crop_left = sps->frame_crop_left_offset * d->CropUnitX;
		info->crop_width =
			info->width - sps->frame_crop_right_offset * d->CropUnitX;
		info->crop_top = sps->frame_crop_top_offset * d->CropUnitY;
		info->crop_height = info->height -
				    sps->frame_crop_bottom_offset * d->CropUnitY;
	}
	info->sar_width = 1;
	info->sar_height = 1;
	if (!sps->vui_parameters_present_flag)
		return 0;
	if (vui->aspect_ratio_info_present_flag) {
		if (vui->aspect_ratio_idc == H264_ASPECT_RATIO_EXTENDED_SAR) {
			info->sar_width = vui->sar_width;
			info->sar_height = vui->sar_height;
		} else if (vui->aspect_ratio_idc <= 16) {
			info->sar_width = sar[vui->aspect_ratio_idc][0];
			info->sar_height = sar[vui->aspect_ratio_idc][1];
		}
	}
	info->full_range = vui->video_full_range_flag;
	if (vui->colour_description_present_flag) {
		info->colour_description_present = 1;
		info->colour_primaries = vui->colour_primaries;
		info->transfer_characteristics = vui->transfer_characteristics;
		info->matrix_coefficients = vui->matrix_coefficients;
	} else {
		info->colour_primaries = 2;
		info->transfer_characteristics = 2;
		info->matrix_coefficients = 2;
	}
	if (vui->timing_info_present_flag) {
		info->num_units_in_tick = vui->num_units_in_tick;
		info->time_scale = vui->time_scale;
		info->framerate = info->num_units_in_tick != 0
					  ? (float)info->time_scale / 2.f /
						    info->num_units_in_tick
					  : 0.f;
		info->framerate_num = info->time_scale;
		info->framerate_den = info->num_units_in_tick;
		if (info->framerate_num % 2 == 0)
			info->framerate_num /= 2;
		else
			info->framerate_den *= 2;
	}
	if (vui->nal_hrd_parameters_present_flag) {
		info->nal_hrd_bitrate =
			(vui->nal_hrd.cpb[0].bit_rate_value_minus1 + 1)
			<< (6 + vui->nal_hrd.bit_rate_scale);
		info->nal_hrd_cpb_size =
			(vui->nal_hrd.cpb[0].cpb_size_value_minus1 + 1)
			<< (4 + vui->nal_hrd.cpb_size_scale);
	}
	if (vui->vcl_hrd_parameters_present_flag) {
		info->vcl_hrd_bitrate =
			(vui->vcl_hrd.cpb[0].bit_rate_value_minus1 + 1)
			<< (6 + vui->vcl_hrd.bit_rate_scale);
		info->vcl_hrd_cpb_size =
			(vui->vcl_hrd.cpb[0].cpb_size_value_minus1 + 1)
			<< (4 + vui->vcl_hrd.cpb_size_scale);
	}
	return 0;
}

/* ---- AVCC <-> byte stream (ref src/h264.c:184-272) --------------------- */
static int find_start_code4(const uint8_t *buf, size_t len, size_t *start)
{
	const uint8_t *p = buf;
	while (len >= 4) {
		if (p[0] == 0 && p[1] == 0 && p[2] == 0 && p[3] == 1) {
			*start = (size_t)(p - buf);
			return 0;
		}
		p++;
		len--;
	}
	return -ENOENT;
}

int orc_byte_stream_to_avcc(uint8_t *data, size_t len)
{
	size_t start, next_start, nalu_len;
	if (data == NULL || len == 0)
		return -EINVAL;
	if (find_start_code4(data, len, &start) < 0)
		return 0;
	data += start;
	len -= start;
	while (len > 4) {
		if (find_start_code4(data + 4, len - 4, &next_start) < 0)
			nalu_len = len - 4;
		else
			nalu_len = next_start;
		data[0] = (uint8_t)(nalu_len >> 24);
		data[1] = (uint8_t)(nalu_len >> 16);
		data[2] = (uint8_t)(nalu_len >> 8);
		data[3] = (uint8_t)nalu_len;
		data += 4 + nalu_len;
		len -= 4 + nalu_len;
	}
	return 0;
}

int orc_avcc_to_byte_stream(uint8_t *data, size_t len)
{
	size_t offset = 0;
	if (data == NULL || len == 0)
		return -EINVAL;
	while (offset < len) {
		uint32_t nalu_len;
		/* deviation shared with the product: the reference reads the four
		 * length bytes even when fewer are left in the buffer */
		if (len - offset < 4)
			return -EPROTO;
		nalu_len = ((uint32_t)data[offset] << 24) |
			   ((uint32_t)data[offset + 1] << 16) |
			   ((uint32_t)data[offset + 2] << 8) | data[offset + 3];
		if (nalu_len == 0)
			return -EPROTO;
		data[offset] = data[offset + 1] = data[offset + 2] = 0;
		data[offset + 3] = 1;
		offset += 4 + (size_t)nalu_len;
	}
	return 0;
}

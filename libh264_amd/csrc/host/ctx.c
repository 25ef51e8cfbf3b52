/**
 * ctx.c -- struct h264_ctx: parameter-set tables, per-NAL state, access-unit
 * detection (host, plain C).  Drop-in for the reference's src/h264_ctx.c
 * (cited per function).  Everything here is control-plane state that the GPU
 * reader replays callbacks against; none of it is on the data path.
 */
#include "h264_priv.h"

uint32_t h264_intlog2(uint32_t x)
{
	/* ceil(log2(x)), 0 for x == 0 (ref src/h264_priv.h:185-193) */
	uint32_t bits = 0;
	while (bits < 32 && (x >> bits) != 0)
		bits++;
	if (bits > 0 && (x & (x - 1)) == 0)
		bits--;
	return bits;
}

/* ---- derived variables (ref :54-124) ------------------------------------ */

static void derive_from_sps(struct h264_ctx *ctx)
{
	if (ctx->sps != NULL)
		h264_get_sps_derived(ctx->sps, &ctx->sps_derived);
}

static void derive_from_pps(struct h264_ctx *ctx)
{
	if (ctx->pps != NULL)
		ctx->derived.SliceGroupChangeRate =
			ctx->pps->slice_group_change_rate_minus1 + 1;
}

/* 7.4.3; slice-group map generation is out of scope (FMO) */
static void derive_from_slice(struct h264_ctx *ctx)
{
	const struct h264_slice_header *sh = &ctx->slice.hdr;
	const struct h264_sps_derived *sd = &ctx->sps_derived;
	uint32_t cycle;
	if (ctx->sps == NULL || ctx->pps == NULL)
		return;
	ctx->derived.MbaffFrameFlag =
		ctx->sps->mb_adaptive_frame_field_flag && !sh->field_pic_flag;
	ctx->derived.PicHeightInMbs =
		sd->FrameHeightInMbs / (1 + sh->field_pic_flag);
	ctx->derived.PicSizeInMbs = sd->PicWidthInMbs * ctx->derived.PicHeightInMbs;
	ctx->derived.MaxPicNum = sd->MaxFrameNum * (sh->field_pic_flag ? 2 : 1);
	ctx->derived.CurrPicNum =
		sh->field_pic_flag ? 2 * sh->frame_num + 1 : sh->frame_num;
	/* unsigned arithmetic throughout: a damaged stream carries any 32-bit
	 * se(v) here, and the reference's signed sums / left shifts of negative
	 * values are undefined for them (same bits where they are defined) */
	ctx->derived.SliceQPLuma =
		(int32_t)(26u + (uint32_t)ctx->pps->pic_init_qp_minus26 +
			  (uint32_t)sh->slice_qp_delta);
	ctx->derived.QSLuma =
		(int32_t)(26u + (uint32_t)ctx->pps->pic_init_qs_minus26 +
			  (uint32_t)sh->slice_qs_delta);
	ctx->derived.FilterOffsetA =
		(int32_t)((uint32_t)sh->slice_alpha_c0_offset_div2 * 2u);
	ctx->derived.FilterOffsetB =
		(int32_t)((uint32_t)sh->slice_beta_offset_div2 * 2u);
	cycle = sh->slice_group_change_cycle * ctx->derived.SliceGroupChangeRate;
	ctx->derived.MapUnitsInSliceGroup0 =
		cycle < sd->PicSizeInMapUnits ? cycle : sd->PicSizeInMapUnits;
}

/* 7.4.1.2.4: is this slice the first VCL NAL unit of a new primary coded
 * picture?  Compared against the previous slice (ref :130-223). */
static int starts_new_picture(const struct h264_ctx *ctx)
{
	const struct h264_sps *sps = ctx->sps;
	const struct h264_slice_header *cur = &ctx->slice.hdr;
	const struct h264_slice_header *prv = &ctx->slice.prev_slice_hdr;
	const struct h264_nalu_header *nh = &ctx->nalu.hdr;
	const struct h264_nalu_header *ph = &ctx->slice.prev_slice_nalu_hdr;
	int cur_idr = nh->nal_unit_type == H264_NALU_TYPE_SLICE_IDR;
	int prv_idr = ph->nal_unit_type == H264_NALU_TYPE_SLICE_IDR;

	if (!ctx->nalu.is_prev_vcl && !ctx->nalu.is_prev_filler)
		return 1;
	if (cur->frame_num != prv->frame_num ||
	    cur->pic_parameter_set_id != prv->pic_parameter_set_id)
		return 1;
	if (!sps->frame_mbs_only_flag) {
		if (cur->field_pic_flag != prv->field_pic_flag)
			return 1;
		if (cur->field_pic_flag && prv->field_pic_flag &&
		    cur->bottom_field_flag != prv->bottom_field_flag)
			return 1;
	}
	if ((nh->nal_ref_idc == 0) != (ph->nal_ref_idc == 0))
		return 1;
	if (sps->pic_order_cnt_type == 0 &&
	    (cur->pic_order_cnt_lsb != prv->pic_order_cnt_lsb ||
	     cur->delta_pic_order_cnt_bottom != prv->delta_pic_order_cnt_bottom))
		return 1;
	if (sps->pic_order_cnt_type == 1 &&
	    (cur->delta_pic_order_cnt[0] != prv->delta_pic_order_cnt[0] ||
	     cur->delta_pic_order_cnt[1] != prv->delta_pic_order_cnt[1]))
		return 1;
	if (cur_idr != prv_idr)
		return 1;
	if (cur_idr && prv_idr && cur->idr_pic_id != prv->idr_pic_id)
		return 1;
	return 0;
}

/* ---- lifecycle (ref :226-281) ------------------------------------------- */

int h264_ctx_new(struct h264_ctx **ret_obj)
{
	if (ret_obj == NULL)
		return -EINVAL;
	*ret_obj = calloc(1, sizeof(**ret_obj));
	return *ret_obj != NULL ? 0 : -ENOMEM;
}

int h264_ctx_clear_sei_table(struct h264_ctx *ctx)
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

int h264_ctx_clear_slice(struct h264_ctx *ctx)
{
	if (ctx == NULL)
		return -EINVAL;
	ctx->slice.type = 0;
	memset(&ctx->slice.hdr, 0, sizeof(ctx->slice.hdr));
	memset(&ctx->slice.rawdata, 0, sizeof(ctx->slice.rawdata));
	derive_from_slice(ctx);
	return 0;
}

int h264_ctx_clear_nalu(struct h264_ctx *ctx)
{
	int keep_vcl, keep_filler;
	if (ctx == NULL)
		return -EINVAL;
	/* the "previous NAL was ..." flags survive, they feed AU detection */
	keep_vcl = ctx->nalu.is_prev_vcl;
	keep_filler = ctx->nalu.is_prev_filler;
	memset(&ctx->nalu, 0, sizeof(ctx->nalu));
	ctx->nalu.is_prev_vcl = keep_vcl;
	ctx->nalu.is_prev_filler = keep_filler;
	memset(&ctx->aud, 0, sizeof(ctx->aud));
	h264_ctx_clear_sei_table(ctx);
	h264_ctx_clear_slice(ctx);
	return 0;
}

int h264_ctx_clear(struct h264_ctx *ctx)
{
	size_t i;
	struct h264_amd_params *wp;
	uint64_t gen;
	if (ctx == NULL)
		return -EINVAL;
	h264_ctx_clear_nalu(ctx);
	for (i = 0; i < H264_ARRAY_SIZE(ctx->sps_table); i++)
		free(ctx->sps_table[i]);
	for (i = 0; i < H264_ARRAY_SIZE(ctx->pps_table); i++)
		free(ctx->pps_table[i]);
	wp = ctx->wr_params;
	gen = ctx->ps_generation;
	memset(ctx, 0, sizeof(*ctx));
	ctx->wr_params = wp;
	/* the counter never repeats: the reader's and the writer's snapshots
	 * compare generations for equality, and a counter reset to a constant
	 * would meet an old value again after as many set_sps / set_pps calls
	 * as came before the clear -- with other parameter sets behind it */
	ctx->ps_generation = gen + 1;
	return 0;
}

int h264_ctx_destroy(struct h264_ctx *ctx)
{
	if (ctx == NULL)
		return 0;
	h264_ctx_clear(ctx);
	if (ctx->wr_params != NULL)
		h264_amd_params_destroy(ctx->wr_params);
	free(ctx);
	return 0;
}

/* ---- setters / getters (ref :284-398) ------------------------------------ */

int h264_ctx_set_nalu_header(struct h264_ctx *ctx,
			     const struct h264_nalu_header *nh)
{
	if (ctx == NULL || nh == NULL)
		return -EINVAL;
	ctx->nalu.hdr = *nh;
	ctx->nalu.type = nh->nal_unit_type;
	return 0;
}

int h264_ctx_is_nalu_unknown(struct h264_ctx *ctx)
{
	return ctx != NULL ? ctx->nalu.unknown : 0;
}

int h264_ctx_set_aud(struct h264_ctx *ctx, const struct h264_aud *aud)
{
	if (ctx == NULL || aud == NULL)
		return -EINVAL;
	ctx->aud = *aud;
	return 0;
}

int h264_ctx_set_sps(struct h264_ctx *ctx, const struct h264_sps *sps)
{
	struct h264_sps *slot;
	if (ctx == NULL || sps == NULL ||
	    sps->seq_parameter_set_id >= H264_ARRAY_SIZE(ctx->sps_table))
		return -EINVAL;
	slot = ctx->sps_table[sps->seq_parameter_set_id];
	if (slot == NULL) {
		slot = calloc(1, sizeof(*slot));
		if (slot == NULL)
			return -ENOMEM;
		ctx->sps_table[sps->seq_parameter_set_id] = slot;
	}
	*slot = *sps;
	ctx->sps = slot;
	ctx->ps_generation++;
	derive_from_sps(ctx);
	derive_from_slice(ctx);
	return 0;
}

int h264_ctx_set_pps(struct h264_ctx *ctx, const struct h264_pps *pps)
{
	struct h264_pps *slot;
	if (ctx == NULL || pps == NULL ||
	    pps->pic_parameter_set_id >= H264_ARRAY_SIZE(ctx->pps_table))
		return -EINVAL;
	slot = ctx->pps_table[pps->pic_parameter_set_id];
	if (slot == NULL) {
		slot = calloc(1, sizeof(*slot));
		if (slot == NULL)
			return -ENOMEM;
		ctx->pps_table[pps->pic_parameter_set_id] = slot;
	}
	*slot = *pps;
	ctx->pps = slot;
	ctx->ps_generation++;
	derive_from_pps(ctx);
	derive_from_slice(ctx);
	return 0;
}

int h264_ctx_set_filler(struct h264_ctx *ctx, size_t len)
{
	if (ctx == NULL)
		return -EINVAL;
	ctx->filler_len = len;
	return 0;
}

int h264_ctx_set_active_sps(struct h264_ctx *ctx, uint32_t sps_id)
{
	if (ctx == NULL || sps_id >= H264_ARRAY_SIZE(ctx->sps_table) ||
	    ctx->sps_table[sps_id] == NULL)
		return -EINVAL;
	ctx->sps = ctx->sps_table[sps_id];
	derive_from_sps(ctx);
	derive_from_slice(ctx);
	return 0;
}

int h264_ctx_set_active_pps(struct h264_ctx *ctx, uint32_t pps_id)
{
	if (ctx == NULL || pps_id >= H264_ARRAY_SIZE(ctx->pps_table) ||
	    ctx->pps_table[pps_id] == NULL)
		return -EINVAL;
	ctx->pps = ctx->pps_table[pps_id];
	derive_from_pps(ctx);
	return h264_ctx_set_active_sps(ctx, ctx->pps->seq_parameter_set_id);
}

const struct h264_sps *h264_ctx_get_sps(struct h264_ctx *ctx)
{
	return ctx != NULL ? ctx->sps : NULL;
}

const struct h264_pps *h264_ctx_get_pps(struct h264_ctx *ctx)
{
	return ctx != NULL ? ctx->pps : NULL;
}

/* ---- SEI table (ref :401-488) -------------------------------------------- */

int h264_ctx_add_sei_internal(struct h264_ctx *ctx, struct h264_sei **ret)
{
	struct h264_sei *grown;
	if (ret == NULL)
		return -EINVAL;
	*ret = NULL;
	if (ctx == NULL)
		return -EINVAL;
	grown = realloc(ctx->sei_table, (ctx->sei_count + 1) * sizeof(*grown));
	if (grown == NULL)
		return -ENOMEM;
	ctx->sei_table = grown;
	*ret = &grown[ctx->sei_count++];
	memset(*ret, 0, sizeof(**ret));
	return 0;
}

/* ref src/h264_types.c:311-349: payload pointers of the data-carrying SEI
 * types point into the raw copy */
int h264_sei_update_internal_buf(struct h264_sei *sei)
{
	size_t skip;
	if (sei == NULL || sei->raw.buf == NULL || sei->raw.len == 0)
		return -EINVAL;
	switch (sei->type) {
	case H264_SEI_TYPE_FILLER_PAYLOAD:
		sei->filler_payload.buf = sei->raw.buf;
		sei->filler_payload.len = sei->raw.len;
		break;
	case H264_SEI_TYPE_USER_DATA_REGISTERED:
		skip = sei->user_data_registered.country_code == 0xff ? 2 : 1;
		if (sei->raw.len < skip)
			return -EINVAL;
		sei->user_data_registered.buf = sei->raw.buf + skip;
		sei->user_data_registered.len = sei->raw.len - skip;
		break;
	case H264_SEI_TYPE_USER_DATA_UNREGISTERED:
		if (sei->raw.len < 16)
			return -EINVAL;
		sei->user_data_unregistered.buf = sei->raw.buf + 16;
		sei->user_data_unregistered.len = sei->raw.len - 16;
		break;
	default:
		break;
	}
	return 0;
}

int h264_ctx_add_sei(struct h264_ctx *ctx, const struct h264_sei *sei)
{
	struct h264_bitstream bs;
	struct h264_sei *slot = NULL;
	int res;
	if (ctx == NULL || sei == NULL)
		return -EINVAL;
	/* the payload is encoded WITHOUT escaping; h264_write_nalu escapes
	 * when it emits the SEI NAL unit */
	h264_bs_init(&bs, NULL, 0, 0);
	res = h264_ctx_add_sei_internal(ctx, &slot);
	if (res < 0)
		goto fail;
	*slot = *sei;
	res = h264_write_one_sei(&bs, ctx, slot);
	if (res < 0)
		goto fail;
	res = h264_bs_acquire_buf(&bs, &slot->raw.buf, &slot->raw.len);
	if (res < 0)
		goto fail;
	res = h264_sei_update_internal_buf(slot);
	if (res < 0)
		goto fail;
	h264_bs_clear(&bs);
	return 0;
fail:
	if (slot != NULL) {
		free(slot->raw.buf);
		ctx->sei_count--;
	}
	h264_bs_clear(&bs);
	return res;
}

int h264_ctx_get_sei_count(struct h264_ctx *ctx)
{
	return ctx != NULL ? (int)ctx->sei_count : -EINVAL;
}

/* D.2.2 clockTimestamp (ref :491-542) */
uint64_t h264_ctx_sei_pic_timing_to_ts(struct h264_ctx *ctx,
				       const struct h264_sei_pic_timing *sei)
{
	const struct h264_vui *vui;
	uint64_t ts, seconds;
	if (ctx == NULL || sei == NULL || ctx->sps == NULL)
		return 0;
	vui = &ctx->sps->vui;
	if (vui->time_scale == 0 || vui->num_units_in_tick == 0)
		return 0;
	seconds = ((uint64_t)sei->clk_ts[0].hours_value * 60 +
		   sei->clk_ts[0].minutes_value) * 60 +
		  sei->clk_ts[0].seconds_value;
	ts = seconds * vui->time_scale +
	     (uint64_t)sei->clk_ts[0].n_frames *
		     ((uint64_t)vui->num_units_in_tick *
		      (1 + (uint64_t)sei->clk_ts[0].nuit_field_based_flag));
	if (sei->clk_ts[0].time_offset < 0 &&
	    (uint64_t)-sei->clk_ts[0].time_offset > ts)
		return 0;
	return ts + sei->clk_ts[0].time_offset;
}

uint64_t h264_ctx_sei_pic_timing_to_us(struct h264_ctx *ctx,
				       const struct h264_sei_pic_timing *sei)
{
	uint64_t scale;
	if (ctx == NULL || sei == NULL || ctx->sps == NULL)
		return 0;
	scale = ctx->sps->vui.time_scale;
	if (scale == 0)
		return 0;
	return (h264_ctx_sei_pic_timing_to_ts(ctx, sei) * 1000000 + scale / 2) /
	       scale;
}

/* ---- slice header installation (ref :563-574) ---------------------------- */

int h264_ctx_set_slice_header(struct h264_ctx *ctx,
			      const struct h264_slice_header *sh)
{
	if (ctx == NULL || sh == NULL)
		return -EINVAL;
	h264_ctx_clear_slice(ctx);
	ctx->slice.type = H264_SLICE_TYPE(sh->slice_type);
	ctx->slice.hdr = *sh;
	derive_from_slice(ctx);
	ctx->nalu.is_first_vcl = starts_new_picture(ctx);
	ctx->slice.prev_slice_nalu_hdr = ctx->nalu.hdr;
	ctx->slice.prev_slice_hdr = ctx->slice.hdr;
	return 0;
}

/* ---- stream summary (ref :577-684) --------------------------------------- */

static const uint16_t sar_table[17][2] = {
	{1, 1},	  {1, 1},   {12, 11}, {10, 11}, {16, 11}, {40, 33},
	{24, 11}, {20, 11}, {32, 11}, {80, 33}, {18, 11}, {15, 11},
	{64, 33}, {160, 99}, {4, 3},  {3, 2},	{2, 1},
};

static void hrd_rates(const struct h264_hrd *h, uint32_t *bitrate,
		      uint32_t *cpb_size)
{
	*bitrate = (h->cpb[0].bit_rate_value_minus1 + 1) << (6 + h->bit_rate_scale);
	*cpb_size = (h->cpb[0].cpb_size_value_minus1 + 1)
		    << (4 + h->cpb_size_scale);
}

int h264_get_info_from_ps(const struct h264_sps *sps,
			  const struct h264_sps_derived *d,
			  struct h264_info *info)
{
	const struct h264_vui *vui = &sps->vui;
	memset(info, 0, sizeof(*info));
	info->width = d->PicWidthInSamplesLuma;
	info->height = 16 * d->FrameHeightInMbs;
	info->bit_depth_luma = (uint8_t)d->BitDepthLuma;
	info->crop_width = info->width;
	info->crop_height = info->height;
	info->sar_width = info->sar_height = 1;
	if (sps->frame_cropping_flag) {
		info->crop_left = d->CropUnitX * sps->frame_crop_left_offset;
		info->crop_top = d->CropUnitY * sps->frame_crop_top_offset;
		info->crop_width -= d->CropUnitX * sps->frame_crop_right_offset;
		info->crop_height -= d->CropUnitY * sps->frame_crop_bottom_offset;
	}
	if (!sps->vui_parameters_present_flag)
		return 0;
	if (vui->aspect_ratio_info_present_flag) {
		if (vui->aspect_ratio_idc == H264_ASPECT_RATIO_EXTENDED_SAR) {
			info->sar_width = vui->sar_width;
			info->sar_height = vui->sar_height;
		} else if (vui->aspect_ratio_idc < H264_ARRAY_SIZE(sar_table)) {
			info->sar_width = sar_table[vui->aspect_ratio_idc][0];
			info->sar_height = sar_table[vui->aspect_ratio_idc][1];
		}
	}
	info->full_range = vui->video_full_range_flag;
	info->colour_description_present = vui->colour_description_present_flag;
	/* 2 = unspecified */
	info->colour_primaries = info->colour_description_present
					 ? vui->colour_primaries : 2;
	info->transfer_characteristics = info->colour_description_present
						 ? vui->transfer_characteristics : 2;
	info->matrix_coefficients = info->colour_description_present
					    ? vui->matrix_coefficients : 2;
	if (vui->timing_info_present_flag) {
		info->num_units_in_tick = vui->num_units_in_tick;
		info->time_scale = vui->time_scale;
		info->framerate = info->num_units_in_tick
					  ? (float)info->time_scale / 2.f /
						    info->num_units_in_tick
					  : 0.f;
		info->framerate_num = info->time_scale;
		info->framerate_den = info->num_units_in_tick;
		if ((info->framerate_num & 1) == 0)
			info->framerate_num /= 2;
		else
			info->framerate_den *= 2;
	}
	if (vui->nal_hrd_parameters_present_flag)
		hrd_rates(&vui->nal_hrd, &info->nal_hrd_bitrate,
			  &info->nal_hrd_cpb_size);
	if (vui->vcl_hrd_parameters_present_flag)
		hrd_rates(&vui->vcl_hrd, &info->vcl_hrd_bitrate,
			  &info->vcl_hrd_cpb_size);
	return 0;
}

int h264_ctx_get_info(struct h264_ctx *ctx, struct h264_info *info)
{
	if (ctx == NULL || info == NULL)
		return -EINVAL;
	if (ctx->sps == NULL || ctx->pps == NULL)
		return -EAGAIN;
	return h264_get_info_from_ps(ctx->sps, &ctx->sps_derived, info);
}

int h264_sar_to_aspect_ratio_idc(unsigned int sar_width, unsigned int sar_height)
{
	unsigned int i;
	for (i = 1; i < H264_ARRAY_SIZE(sar_table); i++) {
		if (sar_table[i][0] == sar_width && sar_table[i][1] == sar_height)
			return (int)i;
	}
	return H264_ASPECT_RATIO_EXTENDED_SAR;
}

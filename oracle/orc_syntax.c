/**
 * oracle/orc_syntax.c -- TEST INFRASTRUCTURE (see orc.h).
 * Restatement of the reference's syntax engine, src/h264_syntax.h, for the
 * READ and WRITE instantiations (src/h264_reader.c:39-55,
 * src/h264_writer.c:30-46).  The reference describes each syntax structure
 * once and compiles it three times through macros; here each structure is one
 * function taking a direction flag `wr`, which keeps the same guarantee
 * (reader and writer walk the same fields in the same order).
 *
 *   hrd/vui            src/h264_syntax.h:36-145
 *   scaling lists      :151-257
 *   sps                :263-358
 *   pps                :364-519
 *   sei payloads       :525-971
 *   aud, filler        :977-1007
 *   rplm/pwt/drpm      :1014-1233
 *   slice_header       :1239-1387
 *   slice_layer        :1399-1428
 *   nalu               :1434-1604
 * slice_data (WRITE)   src/h264_syntax_slice_data.h:813-830
 */
#include "orc.h"

/* one field, direction chosen at run time */
#define F_U(f, n)                                                              \
	do {                                                                   \
		uint32_t _v = (uint32_t)(f);                                   \
		int _r = wr ? orc_bs_write_bits(bs, _v, (n))                   \
			    : orc_bs_read_bits(bs, &_v, (n));                  \
		if (_r < 0)                                                    \
			return _r;                                             \
		if (!wr)                                                       \
			(f) = _v;                                              \
	} while (0)

#define F_I(f, n)                                                              \
	do {                                                                   \
		int32_t _v = (int32_t)(f);                                     \
		int _r = wr ? orc_bs_write_bits(                               \
				      bs, ((uint32_t)_v) & ((1 << (n)) - 1),   \
				      (n))                                     \
			    : orc_bs_read_bits_i(bs, &_v, (n));                \
		if (_r < 0)                                                    \
			return _r;                                             \
		if (!wr)                                                       \
			(f) = _v;                                              \
	} while (0)

#define F_UE(f)                                                                \
	do {                                                                   \
		uint32_t _v = (uint32_t)(f);                                   \
		int _r = wr ? orc_bs_write_ue(bs, _v) : orc_bs_read_ue(bs, &_v); \
		if (_r < 0)                                                    \
			return _r;                                             \
		if (!wr)                                                       \
			(f) = _v;                                              \
	} while (0)

#define F_SE(f)                                                                \
	do {                                                                   \
		int32_t _v = (int32_t)(f);                                     \
		int _r = wr ? orc_bs_write_se(bs, _v) : orc_bs_read_se(bs, &_v); \
		if (_r < 0)                                                    \
			return _r;                                             \
		if (!wr)                                                       \
			(f) = _v;                                              \
	} while (0)

#define F_TRAILING()                                                           \
	do {                                                                   \
		int _r = wr ? orc_bs_write_trailing(bs)                        \
			    : orc_bs_read_trailing(bs);                        \
		if (_r < 0)                                                    \
			return _r;                                             \
	} while (0)

#define CHK(x)                                                                 \
	do {                                                                   \
		int _r = (x);                                                  \
		if (_r < 0)                                                    \
			return _r;                                             \
	} while (0)

#define NELEM(a) (sizeof(a) / sizeof((a)[0]))

/* ---- E.1.2 HRD (ref :36-63) -------------------------------------------- */
static int syn_hrd(struct h264_bitstream *bs, int wr, struct h264_hrd *h)
{
	uint32_t i;
	F_UE(h->cpb_cnt_minus1);
	F_U(h->bit_rate_scale, 4);
	F_U(h->cpb_size_scale, 4);
	/* MEMORY SAFETY, deliberate deviation: the reference tests "> ARRAY_SIZE"
	 * here and at five more places (src/h264_syntax.h:43, :545, :566, :688,
	 * PPS slice-group arrays), so a count equal to the array size writes one
	 * element past the array (cpb_cnt_minus1 = 32 is outside the standard's
	 * range 0..31 anyway).  Product and oracle reject it with -EIO. */
	if (h->cpb_cnt_minus1 >= NELEM(h->cpb))
		return -EIO;
	for (i = 0; i <= h->cpb_cnt_minus1; i++) {
		F_UE(h->cpb[i].bit_rate_value_minus1);
		F_UE(h->cpb[i].cpb_size_value_minus1);
		F_U(h->cpb[i].cbr_flag, 1);
	}
	F_U(h->initial_cpb_removal_delay_length_minus1, 5);
	F_U(h->cpb_removal_delay_length_minus1, 5);
	F_U(h->dpb_output_delay_length_minus1, 5);
	F_U(h->time_offset_length, 5);
	return 0;
}

/* ---- E.1.1 VUI (ref :69-145) ------------------------------------------- */
static int syn_vui(struct h264_bitstream *bs, int wr, struct h264_vui *v)
{
	F_U(v->aspect_ratio_info_present_flag, 1);
	if (v->aspect_ratio_info_present_flag) {
		F_U(v->aspect_ratio_idc, 8);
		if (v->aspect_ratio_idc == H264_ASPECT_RATIO_EXTENDED_SAR) {
			F_U(v->sar_width, 16);
			F_U(v->sar_height, 16);
		}
	}
	F_U(v->overscan_info_present_flag, 1);
	if (v->overscan_info_present_flag)
		F_U(v->overscan_appropriate_flag, 1);
	F_U(v->video_signal_type_present_flag, 1);
	if (v->video_signal_type_present_flag) {
		F_U(v->video_format, 3);
		F_U(v->video_full_range_flag, 1);
		F_U(v->colour_description_present_flag, 1);
		if (v->colour_description_present_flag) {
			F_U(v->colour_primaries, 8);
			F_U(v->transfer_characteristics, 8);
			F_U(v->matrix_coefficients, 8);
		}
	}
	F_U(v->chroma_loc_info_present_flag, 1);
	if (v->chroma_loc_info_present_flag) {
		F_UE(v->chroma_sample_loc_type_top_field);
		F_UE(v->chroma_sample_loc_type_bottom_field);
	}
	F_U(v->timing_info_present_flag, 1);
	if (v->timing_info_present_flag) {
		F_U(v->num_units_in_tick, 32);
		F_U(v->time_scale, 32);
		F_U(v->fixed_frame_rate_flag, 1);
	}
	F_U(v->nal_hrd_parameters_present_flag, 1);
	if (v->nal_hrd_parameters_present_flag)
		CHK(syn_hrd(bs, wr, &v->nal_hrd));
	F_U(v->vcl_hrd_parameters_present_flag, 1);
	if (v->vcl_hrd_parameters_present_flag)
		CHK(syn_hrd(bs, wr, &v->vcl_hrd));
	if (v->nal_hrd_parameters_present_flag ||
	    v->vcl_hrd_parameters_present_flag)
		F_U(v->low_delay_hrd_flag, 1);
	F_U(v->pic_struct_present_flag, 1);
	F_U(v->bitstream_restriction_flag, 1);
	if (v->bitstream_restriction_flag) {
		F_U(v->motion_vectors_over_pic_boundaries_flag, 1);
		F_UE(v->max_bytes_per_pic_denom);
		F_UE(v->max_bits_per_mb_denom);
		F_UE(v->log2_max_mv_length_horizontal);
		F_UE(v->log2_max_mv_length_vertical);
		F_UE(v->max_num_reorder_frames);
		F_UE(v->max_dec_frame_buffering);
	}
	return 0;
}

/* ---- 7.3.2.1.1.1 scaling list (ref :151-211) --------------------------- */
static int syn_scaling_list(struct h264_bitstream *bs, int wr, int32_t *list,
			    uint32_t size, int *use_default, int *optimized)
{
	int32_t last = 8, next = 8, delta = 0;
	uint32_t i;
	if (!wr) {
		for (i = 0; i < size; i++) {
			if (next != 0) {
				CHK(orc_bs_read_se(bs, &delta));
				next = (last + delta + 256) % 256;
				*use_default = (i == 0 && next == 0);
				*optimized = (next == 0);
			}
			list[i] = (next == 0) ? last : next;
			last = list[i];
		}
	} else {
		uint32_t n = 0;
		if (*optimized) {
			/* trailing run of identical values, encoded as one 0 */
			for (i = size - 1; i >= 1; i--) {
				if (list[i] == list[i - 1])
					n++;
				else
					break;
			}
			if (n == size - 1 && list[0] == last)
				n++;
		}
		for (i = 0; i < size && next != 0; i++) {
			next = i < size - n ? list[i] : 0;
			delta = (int8_t)((next - last) % 256);
			CHK(orc_bs_write_se(bs, delta));
			last = list[i];
		}
	}
	return 0;
}

/* ref :217-257 */
static int syn_scaling_matrix(struct h264_bitstream *bs, int wr,
			      struct h264_scaling_matrix *m, uint32_t size)
{
	uint32_t i;
	for (i = 0; i < size; i++) {
		F_U(m->scaling_list_present_flag[i], 1);
		if (!m->scaling_list_present_flag[i])
			continue;
		if (i < 6)
			CHK(syn_scaling_list(bs, wr, m->scaling_list_4x4[i], 16,
					     &m->use_default_4x4[i],
					     &m->_optimized_4x4[i]));
		else
			CHK(syn_scaling_list(bs, wr, m->scaling_list_8x8[i - 6],
					     64, &m->use_default_8x8[i - 6],
					     &m->_optimized_8x8[i - 6]));
	}
	return 0;
}

/* ---- 7.3.2.1 SPS (ref :263-358) ---------------------------------------- */
static int has_high_fields(uint32_t p)
{
	return p == 100 || p == 110 || p == 122 || p == 244 || p == 44 ||
	       p == 83 || p == 86 || p == 118 || p == 128 || p == 138 ||
	       p == 139 || p == 134 || p == 135;
}

static int syn_sps(struct h264_bitstream *bs, int wr, struct h264_sps *s)
{
	uint32_t i;
	F_U(s->profile_idc, 8);
	F_U(s->constraint_set0_flag, 1);
	F_U(s->constraint_set1_flag, 1);
	F_U(s->constraint_set2_flag, 1);
	F_U(s->constraint_set3_flag, 1);
	F_U(s->constraint_set4_flag, 1);
	F_U(s->constraint_set5_flag, 1);
	F_U(s->reserved_zero_2bits, 2);
	F_U(s->level_idc, 8);
	F_UE(s->seq_parameter_set_id);
	if (has_high_fields(s->profile_idc)) {
		F_UE(s->chroma_format_idc);
		if (s->chroma_format_idc == 3)
			F_U(s->separate_colour_plane_flag, 1);
		F_UE(s->bit_depth_luma_minus8);
		if (s->bit_depth_luma_minus8 > 6)
			return -EIO;
		F_UE(s->bit_depth_chroma_minus8);
		if (s->bit_depth_chroma_minus8 > 6)
			return -EIO;
		F_U(s->qpprime_y_zero_transform_bypass_flag, 1);
		F_U(s->seq_scaling_matrix_present_flag, 1);
		if (s->seq_scaling_matrix_present_flag)
			CHK(syn_scaling_matrix(
				bs, wr, &s->seq_scaling_matrix,
				s->chroma_format_idc != 3 ? 8 : 12));
	}
	F_UE(s->log2_max_frame_num_minus4);
	F_UE(s->pic_order_cnt_type);
	if (s->pic_order_cnt_type == 0) {
		F_UE(s->log2_max_pic_order_cnt_lsb_minus4);
	} else if (s->pic_order_cnt_type == 1) {
		F_U(s->delta_pic_order_always_zero_flag, 1);
		F_SE(s->offset_for_non_ref_pic);
		F_SE(s->offset_for_top_to_bottom_field);
		F_UE(s->num_ref_frames_in_pic_order_cnt_cycle);
		if (s->num_ref_frames_in_pic_order_cnt_cycle >=
		    NELEM(s->offset_for_ref_frame))
			return -EIO;
		for (i = 0; i < s->num_ref_frames_in_pic_order_cnt_cycle; i++)
			F_SE(s->offset_for_ref_frame[i]);
	}
	F_UE(s->max_num_ref_frames);
	F_U(s->gaps_in_frame_num_value_allowed_flag, 1);
	F_UE(s->pic_width_in_mbs_minus1);
	F_UE(s->pic_height_in_map_units_minus1);
	F_U(s->frame_mbs_only_flag, 1);
	if (!s->frame_mbs_only_flag)
		F_U(s->mb_adaptive_frame_field_flag, 1);
	F_U(s->direct_8x8_inference_flag, 1);
	F_U(s->frame_cropping_flag, 1);
	if (s->frame_cropping_flag) {
		F_UE(s->frame_crop_left_offset);
		F_UE(s->frame_crop_right_offset);
		F_UE(s->frame_crop_top_offset);
		F_UE(s->frame_crop_bottom_offset);
	}
	F_U(s->vui_parameters_present_flag, 1);
	if (s->vui_parameters_present_flag)
		CHK(syn_vui(bs, wr, &s->vui));
	F_TRAILING();
	return 0;
}

/* ---- 7.3.2.2 PPS (ref :364-519) ---------------------------------------- */
static int syn_pps_body(struct h264_bitstream *bs, int wr,
			const struct h264_sps *sps, struct h264_pps *p)
{
	uint32_t i, len;
	F_U(p->entropy_coding_mode_flag, 1);
	F_U(p->bottom_field_pic_order_in_frame_present_flag, 1);
	F_UE(p->num_slice_groups_minus1);
	if (p->num_slice_groups_minus1 > 0) {
		F_UE(p->slice_group_map_type);
		switch (p->slice_group_map_type) {
		case 0:
			if (p->num_slice_groups_minus1 >=
			    NELEM(p->run_length_minus1))
				return -EIO;
			for (i = 0; i <= p->num_slice_groups_minus1; i++)
				F_UE(p->run_length_minus1[i]);
			break;
		case 1:
			break;
		case 2:
			if (p->num_slice_groups_minus1 >= NELEM(p->top_left))
				return -EIO;
			for (i = 0; i < p->num_slice_groups_minus1; i++) {
				F_UE(p->top_left[i]);
				F_UE(p->bottom_right[i]);
			}
			break;
		case 3:
		case 4:
		case 5:
			F_U(p->slice_group_change_direction_flag, 1);
			F_UE(p->slice_group_change_rate_minus1);
			break;
		case 6:
			F_UE(p->pic_size_in_map_units_minus1);
			len = orc_intlog2(p->num_slice_groups_minus1 + 1);
			if (p->pic_size_in_map_units_minus1 >=
			    NELEM(p->slice_group_id))
				return -EIO;
			for (i = 0; i <= p->pic_size_in_map_units_minus1; i++)
				F_U(p->slice_group_id[i], len);
			break;
		default:
			return -EIO;
		}
	}
	F_UE(p->num_ref_idx_l0_default_active_minus1);
	F_UE(p->num_ref_idx_l1_default_active_minus1);
	F_U(p->weighted_pred_flag, 1);
	F_U(p->weighted_bipred_idc, 2);
	F_SE(p->pic_init_qp_minus26);
	F_SE(p->pic_init_qs_minus26);
	F_SE(p->chroma_qp_index_offset);
	F_U(p->deblocking_filter_control_present_flag, 1);
	F_U(p->constrained_intra_pred_flag, 1);
	F_U(p->redundant_pic_cnt_present_flag, 1);
	if (!wr && orc_bs_more_rbsp_data(bs))
		p->_more_rbsp_data_present = 1;
	if (p->_more_rbsp_data_present) {
		F_U(p->transform_8x8_mode_flag, 1);
		F_U(p->pic_scaling_matrix_present_flag, 1);
		if (p->pic_scaling_matrix_present_flag) {
			uint32_t n = 6;
			if (p->transform_8x8_mode_flag)
				n += sps->chroma_format_idc != 3 ? 2 : 6;
			CHK(syn_scaling_matrix(bs, wr, &p->pic_scaling_matrix,
					       n));
		}
		F_SE(p->second_chroma_qp_index_offset);
	}
	F_TRAILING();
	return 0;
}

static int syn_pps_with_ctx(struct h264_bitstream *bs, int wr,
			    struct orc_ctx *ctx, struct h264_pps *p)
{
	F_UE(p->pic_parameter_set_id);
	F_UE(p->seq_parameter_set_id);
	CHK(orc_ctx_set_active_sps(ctx, p->seq_parameter_set_id));
	return syn_pps_body(bs, wr, ctx->sps, p);
}

/* ---- SEI payloads (ref :525-971) --------------------------------------- */
static int syn_sei_buffering_period(struct h264_bitstream *bs, int wr,
				    struct orc_ctx *ctx,
				    struct h264_sei_buffering_period *s)
{
	const struct h264_hrd *hrd;
	uint32_t i, n;
	F_UE(s->seq_parameter_set_id);
	CHK(orc_ctx_set_active_sps(ctx, s->seq_parameter_set_id));
	if (ctx->sps->vui.nal_hrd_parameters_present_flag) {
		hrd = &ctx->sps->vui.nal_hrd;
		n = hrd->initial_cpb_removal_delay_length_minus1 + 1;
		if (hrd->cpb_cnt_minus1 >= NELEM(s->nal_hrd_cpb))
			return -EIO;
		for (i = 0; i <= hrd->cpb_cnt_minus1; i++) {
			F_U(s->nal_hrd_cpb[i].initial_cpb_removal_delay, n);
			F_U(s->nal_hrd_cpb[i].initial_cpb_removal_delay_offset,
			    n);
		}
	}
	if (ctx->sps->vui.vcl_hrd_parameters_present_flag) {
		hrd = &ctx->sps->vui.vcl_hrd;
		n = hrd->initial_cpb_removal_delay_length_minus1 + 1;
		if (hrd->cpb_cnt_minus1 >= NELEM(s->vcl_hrd_cpb))
			return -EIO;
		for (i = 0; i <= hrd->cpb_cnt_minus1; i++) {
			F_U(s->vcl_hrd_cpb[i].initial_cpb_removal_delay, n);
			F_U(s->vcl_hrd_cpb[i].initial_cpb_removal_delay_offset,
			    n);
		}
	}
	return 0;
}

static int syn_sei_pic_timing(struct h264_bitstream *bs, int wr,
			      struct orc_ctx *ctx, struct h264_sei_pic_timing *s)
{
	static const uint32_t num_clock_ts[16] = {1, 1, 1, 2, 2, 3, 3, 2,
						  3, 0, 0, 0, 0, 0, 0, 0};
	const struct h264_sps *sps = ctx->sps;
	uint32_t i, n;
	if (sps == NULL)
		return -EIO;
	if (sps->vui.nal_hrd_parameters_present_flag ||
	    sps->vui.vcl_hrd_parameters_present_flag) {
		const struct h264_hrd *hrd =
			sps->vui.nal_hrd_parameters_present_flag
				? &sps->vui.nal_hrd
				: &sps->vui.vcl_hrd;
		F_U(s->cpb_removal_delay,
		    hrd->cpb_removal_delay_length_minus1 + 1);
		F_U(s->dpb_output_delay,
		    hrd->dpb_output_delay_length_minus1 + 1);
	}
	if (!sps->vui.pic_struct_present_flag)
		return 0;
	F_U(s->pic_struct, 4);
	for (i = 0; i < num_clock_ts[s->pic_struct]; i++) {
		F_U(s->clk_ts[i].clock_timestamp_flag, 1);
		if (!s->clk_ts[i].clock_timestamp_flag)
			continue;
		F_U(s->clk_ts[i].ct_type, 2);
		F_U(s->clk_ts[i].nuit_field_based_flag, 1);
		F_U(s->clk_ts[i].counting_type, 5);
		F_U(s->clk_ts[i].full_timestamp_flag, 1);
		F_U(s->clk_ts[i].discontinuity_flag, 1);
		F_U(s->clk_ts[i].cnt_dropped_flag, 1);
		F_U(s->clk_ts[i].n_frames, 8);
		if (s->clk_ts[i].full_timestamp_flag) {
			F_U(s->clk_ts[i].seconds_value, 6);
			F_U(s->clk_ts[i].minutes_value, 6);
			F_U(s->clk_ts[i].hours_value, 5);
		} else {
			F_U(s->clk_ts[i].seconds_flag, 1);
			if (s->clk_ts[i].seconds_flag) {
				F_U(s->clk_ts[i].seconds_value, 6);
				F_U(s->clk_ts[i].minutes_flag, 1);
				if (s->clk_ts[i].minutes_flag) {
					F_U(s->clk_ts[i].minutes_value, 6);
					F_U(s->clk_ts[i].hours_flag, 1);
					if (s->clk_ts[i].hours_flag)
						F_U(s->clk_ts[i].hours_value,
						    5);
				}
			}
		}
		n = sps->vui.nal_hrd_parameters_present_flag
			    ? sps->vui.nal_hrd.time_offset_length
		    : sps->vui.vcl_hrd_parameters_present_flag
			    ? sps->vui.vcl_hrd.time_offset_length
			    : 24;
		if (n > 0)
			F_I(s->clk_ts[i].time_offset, n);
	}
	return 0;
}

static int syn_sei_pan_scan_rect(struct h264_bitstream *bs, int wr,
				 struct h264_sei_pan_scan_rect *s)
{
	uint32_t i;
	F_UE(s->pan_scan_rect_id);
	F_U(s->pan_scan_rect_cancel_flag, 1);
	if (s->pan_scan_rect_cancel_flag)
		return 0;
	F_UE(s->pan_scan_cnt_minus1);
	if (s->pan_scan_cnt_minus1 >= NELEM(s->pan_scan_rect))
		return -EIO;
	for (i = 0; i <= s->pan_scan_cnt_minus1; i++) {
		F_SE(s->pan_scan_rect[i].left_offset);
		F_SE(s->pan_scan_rect[i].right_offset);
		F_SE(s->pan_scan_rect[i].top_offset);
		F_SE(s->pan_scan_rect[i].bottom_offset);
	}
	F_UE(s->pan_scan_rect_repetition_period);
	return 0;
}

/* opaque tail of a payload (ref :731-749) */
static int syn_sei_data(struct h264_bitstream *bs, int wr, const uint8_t **buf,
			size_t *len)
{
	size_t i;
	if (!wr) {
		if (bs->cachebits % 8 != 0)
			return -EIO;
		*buf = bs->cdata + bs->off;
		*len = bs->len - bs->off;
		return 0;
	}
	if (*len != 0 && *buf == NULL)
		return -EIO;
	for (i = 0; i < *len; i++)
		CHK(orc_bs_write_bits(bs, (*buf)[i], 8));
	return 0;
}

static int syn_sei_user_data_registered(
	struct h264_bitstream *bs, int wr,
	struct h264_sei_user_data_registered *s)
{
	F_U(s->country_code, 8);
	if (s->country_code == 0xff)
		F_U(s->country_code_extension_byte, 8);
	return syn_sei_data(bs, wr, &s->buf, &s->len);
}

static int syn_sei_user_data_unregistered(
	struct h264_bitstream *bs, int wr,
	struct h264_sei_user_data_unregistered *s)
{
	uint32_t i;
	for (i = 0; i < 16; i++)
		F_U(s->uuid[i], 8);
	return syn_sei_data(bs, wr, &s->buf, &s->len);
}

static int syn_sei_recovery_point(struct h264_bitstream *bs, int wr,
				  struct h264_sei_recovery_point *s)
{
	F_UE(s->recovery_frame_cnt);
	F_U(s->exact_match_flag, 1);
	F_U(s->broken_link_flag, 1);
	F_U(s->changing_slice_group_idc, 2);
	return 0;
}

/* ref :825-863; rd may be NULL (write) */
static int syn_one_sei(struct h264_bitstream *bs, int wr, struct orc_ctx *ctx,
		       struct orc_reader *rd, struct h264_sei *sei)
{
	int res = 0;
	switch (sei->type) {
	case H264_SEI_TYPE_BUFFERING_PERIOD:
		res = syn_sei_buffering_period(bs, wr, ctx,
					       &sei->buffering_period);
		break;
	case H264_SEI_TYPE_PIC_TIMING:
		res = syn_sei_pic_timing(bs, wr, ctx, &sei->pic_timing);
		break;
	case H264_SEI_TYPE_PAN_SCAN_RECT:
		res = syn_sei_pan_scan_rect(bs, wr, &sei->pan_scan_rect);
		break;
	case H264_SEI_TYPE_FILLER_PAYLOAD:
		res = syn_sei_data(bs, wr, &sei->filler_payload.buf,
				   &sei->filler_payload.len);
		break;
	case H264_SEI_TYPE_USER_DATA_REGISTERED:
		res = syn_sei_user_data_registered(bs, wr,
						   &sei->user_data_registered);
		break;
	case H264_SEI_TYPE_USER_DATA_UNREGISTERED:
		res = syn_sei_user_data_unregistered(
			bs, wr, &sei->user_data_unregistered);
		break;
	case H264_SEI_TYPE_RECOVERY_POINT:
		res = syn_sei_recovery_point(bs, wr, &sei->recovery_point);
		break;
	default:
		return 0; /* kept raw */
	}
	if (res < 0)
		return res;
	if (rd != NULL && rd->cbs.sei_typed != NULL)
		rd->cbs.sei_typed(ctx, sei->type, sei->raw.buf, sei->raw.len,
				  sei, rd->userdata);
	if (!wr) {
		/* swallow alignment bits whatever they are */
		uint32_t bit;
		while (bs->cachebits % 8 != 0)
			CHK(orc_bs_read_bits(bs, &bit, 1));
	} else if (bs->cachebits % 8 != 0) {
		CHK(orc_bs_write_trailing(bs));
	}
	return 0;
}

int orc_write_one_sei(struct h264_bitstream *bs, struct orc_ctx *ctx,
		      const struct h264_sei *sei)
{
	return syn_one_sei(bs, 1, ctx, NULL, (struct h264_sei *)sei);
}

/* 7.3.2.3 SEI RBSP (ref :866-971) */
static int syn_sei(struct h264_bitstream *bs, int wr, struct orc_ctx *ctx,
		   struct orc_reader *rd)
{
	uint32_t i, k;
	if (!wr) {
		do {
			uint32_t type = 0, size = 0;
			struct h264_sei *sei = NULL;
			struct h264_bitstream sub;
			int res;
			CHK(orc_bs_read_ff(bs, &type));
			CHK(orc_bs_read_ff(bs, &size));
			CHK(orc_ctx_add_sei_internal(ctx, &sei));
			sei->type = type;
			sei->raw.buf = malloc(size);
			if (sei->raw.buf == NULL)
				return -ENOMEM;
			sei->raw.len = size;
			for (i = 0; i < sei->raw.len; i++) {
				uint32_t b = 0;
				CHK(orc_bs_read_bits(bs, &b, 8));
				sei->raw.buf[i] = b;
			}
			if (rd != NULL && rd->cbs.sei != NULL)
				rd->cbs.sei(ctx, sei->type, sei->raw.buf,
					    sei->raw.len, rd->userdata);
			/* payload re-parsed from the copy, no de-escaping */
			h264_bs_cinit(&sub, sei->raw.buf, sei->raw.len, 0);
			res = syn_one_sei(&sub, 0, ctx, rd, sei);
			if (res < 0)
				return res;
		} while (orc_bs_more_rbsp_data(bs));
	} else {
		if (ctx->sei_count == 0)
			return -EIO;
		for (k = 0; k < ctx->sei_count; k++) {
			const struct h264_sei *sei = &ctx->sei_table[k];
			if (sei->raw.buf == NULL || sei->raw.len == 0)
				return -EIO;
			CHK(orc_bs_write_ff(bs, sei->type));
			CHK(orc_bs_write_ff(bs, sei->raw.len));
			for (i = 0; i < sei->raw.len; i++)
				CHK(orc_bs_write_bits(bs, sei->raw.buf[i], 8));
		}
	}
	F_TRAILING();
	return 0;
}

/* ---- AUD, filler (ref :977-1007) --------------------------------------- */
static int syn_aud(struct h264_bitstream *bs, int wr, struct h264_aud *a)
{
	F_U(a->primary_pic_type, 3);
	F_TRAILING();
	return 0;
}

static int syn_filler(struct h264_bitstream *bs, int wr, size_t *len)
{
	size_t i;
	if (wr) {
		for (i = 0; i < *len; i++)
			CHK(orc_bs_write_bits(bs, 0xff, 8));
	} else {
		uint32_t b = 0;
		int res = orc_bs_next_bits(bs, &b, 8);
		*len = 0;
		while (res == 8 && b == 0xff) {
			CHK(orc_bs_read_bits(bs, &b, 8));
			*len += 1;
			res = orc_bs_next_bits(bs, &b, 8);
		}
	}
	F_TRAILING();
	return 0;
}

/* ---- slice header parts (ref :1014-1233) ------------------------------- */
static int syn_rplm_items(struct h264_bitstream *bs, int wr,
			  struct h264_rplm_item *it, uint32_t maxcount)
{
	uint32_t i = 0, val;
	do {
		if (i >= maxcount)
			return -EIO;
		F_UE(it[i].modification_of_pic_nums_idc);
		val = it[i].modification_of_pic_nums_idc;
		if (val == 0 || val == 1)
			F_UE(it[i].abs_diff_pic_num_minus1);
		else if (val == 2)
			F_UE(it[i].long_term_pic_num);
		else if (val == 4 || val == 5)
			F_UE(it[i].abs_diff_view_idx_minus1);
		i++;
	} while (val != 3);
	return 0;
}

static int syn_rplm(struct h264_bitstream *bs, int wr,
		    struct h264_slice_header *sh)
{
	int type = H264_SLICE_TYPE(sh->slice_type);
	struct h264_rplm *r = &sh->rplm;
	if (type != H264_SLICE_TYPE_I && type != H264_SLICE_TYPE_SI) {
		F_U(r->ref_pic_list_modification_flag_l0, 1);
		if (r->ref_pic_list_modification_flag_l0)
			CHK(syn_rplm_items(bs, wr, r->pic_num_l0,
					   NELEM(r->pic_num_l0)));
	}
	if (type == H264_SLICE_TYPE_B) {
		F_U(r->ref_pic_list_modification_flag_l1, 1);
		if (r->ref_pic_list_modification_flag_l1)
			CHK(syn_rplm_items(bs, wr, r->pic_num_l1,
					   NELEM(r->pic_num_l1)));
	}
	return 0;
}

static int syn_pwt_item(struct h264_bitstream *bs, int wr,
			struct h264_pwt_item *it, uint32_t cat)
{
	F_U(it->luma_weight_flag, 1);
	if (it->luma_weight_flag) {
		F_SE(it->luma_weight);
		F_SE(it->luma_offset);
	}
	if (cat != 0) {
		F_U(it->chroma_weight_flag, 1);
		if (it->chroma_weight_flag) {
			F_SE(it->chroma_weight[0]);
			F_SE(it->chroma_offset[0]);
			F_SE(it->chroma_weight[1]);
			F_SE(it->chroma_offset[1]);
		}
	}
	return 0;
}

static int syn_pwt(struct h264_bitstream *bs, int wr, struct orc_ctx *ctx,
		   struct h264_slice_header *sh)
{
	uint32_t cat = ctx->sps->separate_colour_plane_flag
			       ? 0
			       : ctx->sps->chroma_format_idc;
	struct h264_pwt *p = &sh->pwt;
	uint32_t i;
	F_UE(p->luma_log2_weight_denom);
	if (cat != 0)
		F_UE(p->chroma_log2_weight_denom);
	if (sh->num_ref_idx_l0_active_minus1 > NELEM(p->l0))
		return -EIO;
	for (i = 0; i <= sh->num_ref_idx_l0_active_minus1; i++)
		CHK(syn_pwt_item(bs, wr, &p->l0[i], cat));
	if (H264_SLICE_TYPE(sh->slice_type) != H264_SLICE_TYPE_B)
		return 0;
	if (sh->num_ref_idx_l1_active_minus1 > NELEM(p->l1))
		return -EIO;
	for (i = 0; i <= sh->num_ref_idx_l1_active_minus1; i++)
		CHK(syn_pwt_item(bs, wr, &p->l1[i], cat));
	return 0;
}

static int syn_drpm(struct h264_bitstream *bs, int wr, struct orc_ctx *ctx,
		    struct h264_slice_header *sh)
{
	struct h264_drpm *d = &sh->drpm;
	uint32_t i = 0, val;
	if (ctx->nalu.type == H264_NALU_TYPE_SLICE_IDR) {
		F_U(d->no_output_of_prior_pics_flag, 1);
		F_U(d->long_term_reference_flag, 1);
		return 0;
	}
	F_U(d->adaptive_ref_pic_marking_mode_flag, 1);
	if (!d->adaptive_ref_pic_marking_mode_flag)
		return 0;
	do {
		if (i >= NELEM(d->mm))
			return -EIO;
		F_UE(d->mm[i].memory_management_control_operation);
		val = d->mm[i].memory_management_control_operation;
		if (val == 1 || val == 3)
			F_UE(d->mm[i].difference_of_pic_nums_minus1);
		if (val == 2)
			F_UE(d->mm[i].long_term_pic_num);
		if (val == 3 || val == 6)
			F_UE(d->mm[i].long_term_frame_idx);
		if (val == 4)
			F_UE(d->mm[i].max_long_term_frame_idx_plus1);
		i++;
	} while (val != 0);
	return 0;
}

/* ---- 7.3.3 slice header (ref :1239-1387) ------------------------------- */
static int syn_slice_header(struct h264_bitstream *bs, int wr,
			    struct orc_ctx *ctx, struct h264_slice_header *sh)
{
	int idr = ctx->nalu.type == H264_NALU_TYPE_SLICE_IDR;
	int type;
	ctx->slice.hdr_len = 0;
	F_UE(sh->first_mb_in_slice);
	F_UE(sh->slice_type);
	type = H264_SLICE_TYPE(sh->slice_type);
	F_UE(sh->pic_parameter_set_id);
	CHK(orc_ctx_set_active_pps(ctx, sh->pic_parameter_set_id));
	if (!wr) {
		sh->num_ref_idx_l0_active_minus1 =
			ctx->pps->num_ref_idx_l0_default_active_minus1;
		sh->num_ref_idx_l1_active_minus1 =
			ctx->pps->num_ref_idx_l1_default_active_minus1;
	}
	if (ctx->sps->separate_colour_plane_flag)
		F_U(sh->colour_plane_id, 2);
	F_U(sh->frame_num, ctx->sps->log2_max_frame_num_minus4 + 4);
	if (!ctx->sps->frame_mbs_only_flag) {
		F_U(sh->field_pic_flag, 1);
		if (sh->field_pic_flag)
			F_U(sh->bottom_field_flag, 1);
	}
	if (idr)
		F_UE(sh->idr_pic_id);
	if (ctx->sps->pic_order_cnt_type == 0) {
		F_U(sh->pic_order_cnt_lsb,
		    ctx->sps->log2_max_pic_order_cnt_lsb_minus4 + 4);
		if (ctx->pps->bottom_field_pic_order_in_frame_present_flag &&
		    !sh->field_pic_flag)
			F_SE(sh->delta_pic_order_cnt_bottom);
	}
	if (ctx->sps->pic_order_cnt_type == 1 &&
	    !ctx->sps->delta_pic_order_always_zero_flag) {
		F_SE(sh->delta_pic_order_cnt[0]);
		if (ctx->pps->bottom_field_pic_order_in_frame_present_flag &&
		    !sh->field_pic_flag)
			F_SE(sh->delta_pic_order_cnt[1]);
	}
	if (ctx->pps->redundant_pic_cnt_present_flag)
		F_UE(sh->redundant_pic_cnt);
	if (type == H264_SLICE_TYPE_B)
		F_U(sh->direct_spatial_mv_pred_flag, 1);
	if (type == H264_SLICE_TYPE_P || type == H264_SLICE_TYPE_SP ||
	    type == H264_SLICE_TYPE_B) {
		F_U(sh->num_ref_idx_active_override_flag, 1);
		if (sh->num_ref_idx_active_override_flag) {
			F_UE(sh->num_ref_idx_l0_active_minus1);
			if (type == H264_SLICE_TYPE_B)
				F_UE(sh->num_ref_idx_l1_active_minus1);
		}
	}
	CHK(syn_rplm(bs, wr, sh));
	if ((ctx->pps->weighted_pred_flag &&
	     (type == H264_SLICE_TYPE_P || type == H264_SLICE_TYPE_SP)) ||
	    (ctx->pps->weighted_bipred_idc == 1 && type == H264_SLICE_TYPE_B))
		CHK(syn_pwt(bs, wr, ctx, sh));
	if (ctx->nalu.hdr.nal_ref_idc != 0)
		CHK(syn_drpm(bs, wr, ctx, sh));
	if (ctx->pps->entropy_coding_mode_flag && type != H264_SLICE_TYPE_I &&
	    type != H264_SLICE_TYPE_SI)
		F_UE(sh->cabac_init_idc);
	F_SE(sh->slice_qp_delta);
	if (type == H264_SLICE_TYPE_SP || type == H264_SLICE_TYPE_SI) {
		if (type == H264_SLICE_TYPE_SP)
			F_U(sh->sp_for_switch_flag, 1);
		F_SE(sh->slice_qs_delta);
	}
	if (ctx->pps->deblocking_filter_control_present_flag) {
		F_UE(sh->disable_deblocking_filter_idc);
		if (sh->disable_deblocking_filter_idc != 1) {
			F_SE(sh->slice_alpha_c0_offset_div2);
			F_SE(sh->slice_beta_offset_div2);
		}
	}
	if (ctx->pps->num_slice_groups_minus1 > 0 &&
	    ctx->pps->slice_group_map_type >= 3 &&
	    ctx->pps->slice_group_map_type <= 5) {
		uint32_t units = (ctx->sps->pic_width_in_mbs_minus1 + 1) *
				 (ctx->sps->pic_height_in_map_units_minus1 + 1);
		uint32_t n = orc_intlog2(
			units / (ctx->pps->slice_group_change_rate_minus1 + 1) +
			1);
		F_U(sh->slice_group_change_cycle, n);
	}
	/* header length in raw bits (ref :1380-1384) */
	if (wr)
		ctx->slice.hdr_len = bs->off * 8 + bs->cachebits;
	else
		ctx->slice.hdr_len = bs->off * 8 - bs->cachebits;
	return 0;
}

/* slice_data WRITE: partial byte + raw bytes
 * (ref src/h264_syntax_slice_data.h:813-830) */
static int write_slice_data(struct h264_bitstream *bs, struct orc_ctx *ctx)
{
	if (ctx->slice.rawdata.partialbits == 0 && ctx->slice.rawdata.len == 0)
		return 0;
	if (ctx->slice.rawdata.partialbits != 0)
		CHK(orc_bs_write_bits(bs, ctx->slice.rawdata.partial,
				      ctx->slice.rawdata.partialbits));
	if (bs->cachebits % 8 != 0)
		return -EIO;
	if (ctx->slice.rawdata.len != 0 && ctx->slice.rawdata.buf == NULL)
		return -EIO;
	return orc_bs_write_raw_bytes(bs, ctx->slice.rawdata.buf,
				      ctx->slice.rawdata.len);
}

/* 7.3.2.8 (ref :1399-1428) */
static int syn_slice_layer(struct h264_bitstream *bs, int wr,
			   struct orc_ctx *ctx, struct orc_reader *rd)
{
	if (!wr) {
		struct h264_slice_header sh;
		memset(&sh, 0, sizeof(sh));
		CHK(syn_slice_header(bs, 0, ctx, &sh));
		CHK(orc_ctx_set_slice_header(ctx, &sh));
		return orc_read_slice_data(bs, rd);
	}
	CHK(syn_slice_header(bs, 1, ctx, &ctx->slice.hdr));
	return write_slice_data(bs, ctx);
}

/* ---- 7.3.1 NAL header + dispatch (ref :1434-1604) ---------------------- */
static int syn_nalu_header(struct h264_bitstream *bs, int wr,
			   struct h264_nalu_header *nh)
{
	F_U(nh->forbidden_zero_bit, 1);
	if (nh->forbidden_zero_bit != 0)
		return -EIO;
	F_U(nh->nal_ref_idc, 2);
	F_U(nh->nal_unit_type, 5);
	return 0;
}

int orc_read_nalu_header(struct h264_bitstream *bs, struct h264_nalu_header *nh)
{
	return syn_nalu_header(bs, 0, nh);
}

static int syn_nalu(struct h264_bitstream *bs, int wr, struct orc_ctx *ctx,
		    struct orc_reader *rd)
{
	const uint8_t *buf = NULL;
	size_t len = 0;
	void *ud = rd ? rd->userdata : NULL;
	const struct orc_cbs *cbs = rd ? &rd->cbs : NULL;
	uint32_t t;

	if (!wr) {
		buf = bs->cdata + bs->off;
		len = bs->len;
		CHK(orc_ctx_clear_nalu(ctx));
	}
	ctx->cur_bs = bs;
	CHK(syn_nalu_header(bs, wr, &ctx->nalu.hdr));
	ctx->nalu.type = ctx->nalu.hdr.nal_unit_type;
	t = ctx->nalu.type;
	if (cbs && cbs->nalu_begin)
		cbs->nalu_begin(ctx, t, buf, len, &ctx->nalu.hdr, ud);

	switch (t) {
	case H264_NALU_TYPE_SLICE:
	case H264_NALU_TYPE_SLICE_IDR:
		CHK(syn_slice_layer(bs, wr, ctx, rd));
		if (cbs && cbs->slice)
			cbs->slice(ctx, buf, len, &ctx->slice.hdr, ud);
		break;
	case H264_NALU_TYPE_SEI:
		CHK(syn_sei(bs, wr, ctx, rd));
		break;
	case H264_NALU_TYPE_SPS: {
		struct h264_sps tmp, *sps;
		if (!wr) {
			memset(&tmp, 0, sizeof(tmp));
			tmp.chroma_format_idc = 1; /* Appendix C.9 */
			sps = &tmp;
		} else {
			sps = ctx->sps;
			if (sps == NULL)
				return -EIO;
		}
		if (ctx->nalu.hdr.nal_ref_idc == 0)
			return -EIO;
		CHK(syn_sps(bs, wr, sps));
		if (!wr)
			CHK(orc_ctx_set_sps(ctx, sps));
		if (cbs && cbs->sps)
			cbs->sps(ctx, buf, len, ctx->sps, ud);
		break;
	}
	case H264_NALU_TYPE_PPS: {
		struct h264_pps tmp, *pps;
		if (!wr) {
			memset(&tmp, 0, sizeof(tmp));
			pps = &tmp;
		} else {
			pps = ctx->pps;
			if (pps == NULL)
				return -EIO;
		}
		if (ctx->nalu.hdr.nal_ref_idc == 0)
			return -EIO;
		CHK(syn_pps_with_ctx(bs, wr, ctx, pps));
		if (!wr)
			CHK(orc_ctx_set_pps(ctx, pps));
		if (cbs && cbs->pps)
			cbs->pps(ctx, buf, len, ctx->pps, ud);
		break;
	}
	case H264_NALU_TYPE_AUD:
		if (ctx->nalu.hdr.nal_ref_idc != 0)
			return -EIO;
		CHK(syn_aud(bs, wr, &ctx->aud));
		if (cbs && cbs->aud)
			cbs->aud(ctx, buf, len, &ctx->aud, ud);
		break;
	case H264_NALU_TYPE_FILLER:
		if (ctx->nalu.hdr.nal_ref_idc != 0)
			return -EIO;
		CHK(syn_filler(bs, wr, &ctx->filler_len));
		break;
	default:
		/* data partitions 2-4 and everything else: header only */
		ctx->nalu.unknown = 1;
		break;
	}

	if (!wr) {
		/* 7.4.1.2.4 access unit change (ref :1572-1592) */
		if ((ctx->nalu.is_prev_vcl || ctx->nalu.is_prev_filler) &&
		    (t == H264_NALU_TYPE_AUD || t == H264_NALU_TYPE_SPS ||
		     t == H264_NALU_TYPE_PPS || t == H264_NALU_TYPE_SEI ||
		     (t >= 14 && t <= 18) || ctx->nalu.is_first_vcl)) {
			if (cbs && cbs->au_end)
				cbs->au_end(ctx, ud);
		}
		ctx->nalu.is_prev_vcl = (t == H264_NALU_TYPE_SLICE ||
					 t == H264_NALU_TYPE_SLICE_IDR);
		ctx->nalu.is_prev_filler = (t == H264_NALU_TYPE_FILLER);
	}
	if (cbs && cbs->nalu_end)
		cbs->nalu_end(ctx, t, buf, len, &ctx->nalu.hdr, ud);
	return 0;
}

int orc_read_nalu(struct h264_bitstream *bs, struct orc_reader *rd)
{
	return syn_nalu(bs, 0, rd->ctx, rd);
}

int orc_write_nalu(struct h264_bitstream *bs, struct orc_ctx *ctx)
{
	return syn_nalu(bs, 1, ctx, NULL);
}

/* ---- context-free helpers (ref src/h264_reader.c:183-255) -------------- */
int orc_parse_sps(const uint8_t *buf, size_t len, struct h264_sps *sps)
{
	struct h264_bitstream bs;
	struct h264_nalu_header nh;
	int res;
	if (buf == NULL || sps == NULL)
		return -EINVAL;
	memset(sps, 0, sizeof(*sps));
	sps->chroma_format_idc = 1;
	h264_bs_cinit(&bs, buf, len, 1);
	res = syn_nalu_header(&bs, 0, &nh);
	if (res < 0)
		return res;
	if (nh.nal_unit_type != H264_NALU_TYPE_SPS)
		return -EIO;
	return syn_sps(&bs, 0, sps);
}

int orc_parse_pps(const uint8_t *buf, size_t len, const struct h264_sps *sps,
		  struct h264_pps *pps)
{
	struct h264_bitstream bsv, *bs = &bsv;
	struct h264_nalu_header nh;
	const int wr = 0;
	int res;
	if (buf == NULL || sps == NULL || pps == NULL)
		return -EINVAL;
	memset(pps, 0, sizeof(*pps));
	h264_bs_cinit(bs, buf, len, 1);
	res = syn_nalu_header(bs, 0, &nh);
	if (res < 0)
		return res;
	if (nh.nal_unit_type != H264_NALU_TYPE_PPS)
		return -EIO;
	F_UE(pps->pic_parameter_set_id);
	F_UE(pps->seq_parameter_set_id);
	if (sps->seq_parameter_set_id != pps->seq_parameter_set_id)
		return -EIO;
	return syn_pps_body(bs, 0, sps, pps);
}

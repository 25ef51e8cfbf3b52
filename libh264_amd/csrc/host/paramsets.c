/**
 * paramsets.c -- non-VCL NAL unit syntax on the host: NAL header, SPS (with
 * VUI/HRD and scaling lists), PPS, SEI, AUD, filler.  Reading AND writing.
 *
 * Drop-in behaviour for the reference's syntax engine src/h264_syntax.h
 * (sections cited below).  These NAL units are a few dozen bytes each and
 * strictly sequential (each may change the parameter-set tables the next one
 * is parsed against), so they stay on the host; SURVEY.md section 2 row 3.
 *
 * Mechanism: every structure is walked once by a function that takes a cursor
 * `struct h264_syn` carrying the direction (read or write) and an error LATCH.
 * A failed field sets the latch and turns every later field into a no-op, so
 * the walkers read like the standard's syntax tables without an error branch
 * per line, and reader and writer cannot drift apart.
 */
#include "h264_priv.h"

/* ---- field primitives ------------------------------------------------------ */

static void latch(struct h264_syn *s, int res)
{
	if (res < 0 && s->err == 0)
		s->err = res;
}

/* u(n) into/from a 32-bit container (uint32_t or int) */
static void f_u(struct h264_syn *s, void *field, uint32_t n)
{
	uint32_t *p = field;
	if (s->err)
		return;
	if (s->wr) {
		latch(s, h264_bs_write_bits(s->bs, *p, n));
	} else {
		uint32_t v = 0;
		int res = h264_bs_read_bits(s->bs, &v, n);
		latch(s, res);
		if (res >= 0)
			*p = v;
	}
}

static void f_i(struct h264_syn *s, int32_t *field, uint32_t n)
{
	if (s->err)
		return;
	if (s->wr) {
		latch(s, h264_bs_write_bits_i(s->bs, *field, n));
	} else {
		int32_t v = 0;
		int res = h264_bs_read_bits_i(s->bs, &v, n);
		latch(s, res);
		if (res >= 0)
			*field = v;
	}
}

static void f_ue(struct h264_syn *s, void *field)
{
	uint32_t *p = field;
	if (s->err)
		return;
	if (s->wr) {
		latch(s, h264_bs_write_bits_ue(s->bs, *p));
	} else {
		uint32_t v = 0;
		int res = h264_bs_read_bits_ue(s->bs, &v);
		latch(s, res);
		if (res >= 0)
			*p = v;
	}
}

static void f_se(struct h264_syn *s, int32_t *field)
{
	if (s->err)
		return;
	if (s->wr) {
		latch(s, h264_bs_write_bits_se(s->bs, *field));
	} else {
		int32_t v = 0;
		int res = h264_bs_read_bits_se(s->bs, &v);
		latch(s, res);
		if (res >= 0)
			*field = v;
	}
}

static void f_u8(struct h264_syn *s, uint8_t *field)
{
	uint32_t v = *field;
	f_u(s, &v, 8);
	if (!s->wr && !s->err)
		*field = (uint8_t)v;
}

static void f_trailing(struct h264_syn *s)
{
	if (s->err)
		return;
	latch(s, s->wr ? h264_bs_write_rbsp_trailing_bits(s->bs)
		       : h264_bs_read_rbsp_trailing_bits(s->bs));
}

static void require(struct h264_syn *s, int cond)
{
	if (!cond && s->err == 0)
		s->err = -EIO;
}

/* ---- 7.3.1 NAL unit header (ref src/h264_syntax.h:1434-1443) --------------- */

int h264_syn_nalu_header(struct h264_syn *s, struct h264_nalu_header *nh)
{
	f_u(s, &nh->forbidden_zero_bit, 1);
	require(s, nh->forbidden_zero_bit == 0);
	f_u(s, &nh->nal_ref_idc, 2);
	f_u(s, &nh->nal_unit_type, 5);
	return s->err;
}

/* ---- E.1.2 HRD, E.1.1 VUI (ref :36-145) ------------------------------------ */

static void syn_hrd(struct h264_syn *s, struct h264_hrd *h)
{
	uint32_t i;
	f_ue(s, &h->cpb_cnt_minus1);
	f_u(s, &h->bit_rate_scale, 4);
	f_u(s, &h->cpb_size_scale, 4);
	/* same (off by one) acceptance test as the reference: 32 passes */
	/* strict bound: the reference accepts a count equal to the array size and
	 * writes one element past it (src/h264_syntax.h:43); -EIO here */
	require(s, h->cpb_cnt_minus1 < H264_ARRAY_SIZE(h->cpb));
	for (i = 0; !s->err && i <= h->cpb_cnt_minus1; i++) {
		f_ue(s, &h->cpb[i].bit_rate_value_minus1);
		f_ue(s, &h->cpb[i].cpb_size_value_minus1);
		f_u(s, &h->cpb[i].cbr_flag, 1);
	}
	f_u(s, &h->initial_cpb_removal_delay_length_minus1, 5);
	f_u(s, &h->cpb_removal_delay_length_minus1, 5);
	f_u(s, &h->dpb_output_delay_length_minus1, 5);
	f_u(s, &h->time_offset_length, 5);
}

static void syn_vui(struct h264_syn *s, struct h264_vui *v)
{
	f_u(s, &v->aspect_ratio_info_present_flag, 1);
	if (v->aspect_ratio_info_present_flag) {
		f_u(s, &v->aspect_ratio_idc, 8);
		if (v->aspect_ratio_idc == H264_ASPECT_RATIO_EXTENDED_SAR) {
			f_u(s, &v->sar_width, 16);
			f_u(s, &v->sar_height, 16);
		}
	}
	f_u(s, &v->overscan_info_present_flag, 1);
	if (v->overscan_info_present_flag)
		f_u(s, &v->overscan_appropriate_flag, 1);
	f_u(s, &v->video_signal_type_present_flag, 1);
	if (v->video_signal_type_present_flag) {
		f_u(s, &v->video_format, 3);
		f_u(s, &v->video_full_range_flag, 1);
		f_u(s, &v->colour_description_present_flag, 1);
		if (v->colour_description_present_flag) {
			f_u(s, &v->colour_primaries, 8);
			f_u(s, &v->transfer_characteristics, 8);
			f_u(s, &v->matrix_coefficients, 8);
		}
	}
	f_u(s, &v->chroma_loc_info_present_flag, 1);
	if (v->chroma_loc_info_present_flag) {
		f_ue(s, &v->chroma_sample_loc_type_top_field);
		f_ue(s, &v->chroma_sample_loc_type_bottom_field);
	}
	f_u(s, &v->timing_info_present_flag, 1);
	if (v->timing_info_present_flag) {
		f_u(s, &v->num_units_in_tick, 32);
		f_u(s, &v->time_scale, 32);
		f_u(s, &v->fixed_frame_rate_flag, 1);
	}
	f_u(s, &v->nal_hrd_parameters_present_flag, 1);
	if (v->nal_hrd_parameters_present_flag)
		syn_hrd(s, &v->nal_hrd);
	f_u(s, &v->vcl_hrd_parameters_present_flag, 1);
	if (v->vcl_hrd_parameters_present_flag)
		syn_hrd(s, &v->vcl_hrd);
	if (v->nal_hrd_parameters_present_flag ||
	    v->vcl_hrd_parameters_present_flag)
		f_u(s, &v->low_delay_hrd_flag, 1);
	f_u(s, &v->pic_struct_present_flag, 1);
	f_u(s, &v->bitstream_restriction_flag, 1);
	if (v->bitstream_restriction_flag) {
		f_u(s, &v->motion_vectors_over_pic_boundaries_flag, 1);
		f_ue(s, &v->max_bytes_per_pic_denom);
		f_ue(s, &v->max_bits_per_mb_denom);
		f_ue(s, &v->log2_max_mv_length_horizontal);
		f_ue(s, &v->log2_max_mv_length_vertical);
		f_ue(s, &v->max_num_reorder_frames);
		f_ue(s, &v->max_dec_frame_buffering);
	}
}

/* ---- 7.3.2.1.1.1 scaling lists (ref :151-257) ------------------------------- */

/* Reading records how the list ended (*optimized: with a zero "repeat the
 * last value" delta) so that writing reproduces the original bits. */
static void syn_scaling_list(struct h264_syn *s, int32_t *list, uint32_t size,
			     int *use_default, int *optimized)
{
	int32_t last = 8, next = 8, delta = 0;
	uint32_t i, tail = 0;
	if (!s->wr) {
		for (i = 0; !s->err && i < size; i++) {
			if (next != 0) {
				f_se(s, &delta);
				if (s->err)
					return;
				next = (last + delta + 256) % 256;
				*use_default = (i == 0 && next == 0);
				*optimized = (next == 0);
			}
			list[i] = next == 0 ? last : next;
			last = list[i];
		}
		return;
	}
	if (*optimized) {
		/* how many trailing entries merely repeat their predecessor */
		for (i = size - 1; i >= 1 && list[i] == list[i - 1]; i--)
			tail++;
		if (tail == size - 1 && list[0] == last)
			tail++; /* the whole list is the initial value 8 */
	}
	for (i = 0; !s->err && i < size && next != 0; i++) {
		next = i < size - tail ? list[i] : 0;
		delta = (int8_t)((next - last) % 256);
		f_se(s, &delta);
		last = list[i];
	}
}

static void syn_scaling_matrix(struct h264_syn *s, struct h264_scaling_matrix *m,
			       uint32_t count)
{
	uint32_t i;
	for (i = 0; !s->err && i < count; i++) {
		f_u(s, &m->scaling_list_present_flag[i], 1);
		if (!m->scaling_list_present_flag[i])
			continue;
		if (i < 6)
			syn_scaling_list(s, m->scaling_list_4x4[i], 16,
					 &m->use_default_4x4[i],
					 &m->_optimized_4x4[i]);
		else
			syn_scaling_list(s, m->scaling_list_8x8[i - 6], 64,
					 &m->use_default_8x8[i - 6],
					 &m->_optimized_8x8[i - 6]);
	}
}

/* ---- 7.3.2.1 SPS (ref :263-358) ---------------------------------------------- */

static int profile_has_chroma_info(uint32_t p)
{
	static const uint8_t list[] = {100, 110, 122, 244, 44,  83, 86,
				       118, 128, 138, 139, 134, 135};
	size_t i;
	for (i = 0; i < sizeof(list); i++)
		if (p == list[i])
			return 1;
	return 0;
}

int h264_syn_sps(struct h264_syn *s, struct h264_sps *sps)
{
	uint32_t i;
	f_u(s, &sps->profile_idc, 8);
	f_u(s, &sps->constraint_set0_flag, 1);
	f_u(s, &sps->constraint_set1_flag, 1);
	f_u(s, &sps->constraint_set2_flag, 1);
	f_u(s, &sps->constraint_set3_flag, 1);
	f_u(s, &sps->constraint_set4_flag, 1);
	f_u(s, &sps->constraint_set5_flag, 1);
	f_u(s, &sps->reserved_zero_2bits, 2);
	f_u(s, &sps->level_idc, 8);
	f_ue(s, &sps->seq_parameter_set_id);
	if (!s->err && profile_has_chroma_info(sps->profile_idc)) {
		f_ue(s, &sps->chroma_format_idc);
		if (sps->chroma_format_idc == 3)
			f_u(s, &sps->separate_colour_plane_flag, 1);
		f_ue(s, &sps->bit_depth_luma_minus8);
		require(s, sps->bit_depth_luma_minus8 <= 6);
		f_ue(s, &sps->bit_depth_chroma_minus8);
		require(s, sps->bit_depth_chroma_minus8 <= 6);
		f_u(s, &sps->qpprime_y_zero_transform_bypass_flag, 1);
		f_u(s, &sps->seq_scaling_matrix_present_flag, 1);
		if (!s->err && sps->seq_scaling_matrix_present_flag)
			syn_scaling_matrix(s, &sps->seq_scaling_matrix,
					   sps->chroma_format_idc != 3 ? 8 : 12);
	}
	f_ue(s, &sps->log2_max_frame_num_minus4);
	f_ue(s, &sps->pic_order_cnt_type);
	if (!s->err && sps->pic_order_cnt_type == 0) {
		f_ue(s, &sps->log2_max_pic_order_cnt_lsb_minus4);
	} else if (!s->err && sps->pic_order_cnt_type == 1) {
		f_u(s, &sps->delta_pic_order_always_zero_flag, 1);
		f_se(s, &sps->offset_for_non_ref_pic);
		f_se(s, &sps->offset_for_top_to_bottom_field);
		f_ue(s, &sps->num_ref_frames_in_pic_order_cnt_cycle);
		require(s, sps->num_ref_frames_in_pic_order_cnt_cycle <
				   H264_ARRAY_SIZE(sps->offset_for_ref_frame));
		for (i = 0; !s->err &&
			    i < sps->num_ref_frames_in_pic_order_cnt_cycle; i++)
			f_se(s, &sps->offset_for_ref_frame[i]);
	}
	f_ue(s, &sps->max_num_ref_frames);
	f_u(s, &sps->gaps_in_frame_num_value_allowed_flag, 1);
	f_ue(s, &sps->pic_width_in_mbs_minus1);
	f_ue(s, &sps->pic_height_in_map_units_minus1);
	f_u(s, &sps->frame_mbs_only_flag, 1);
	if (!s->err && !sps->frame_mbs_only_flag)
		f_u(s, &sps->mb_adaptive_frame_field_flag, 1);
	f_u(s, &sps->direct_8x8_inference_flag, 1);
	f_u(s, &sps->frame_cropping_flag, 1);
	if (!s->err && sps->frame_cropping_flag) {
		f_ue(s, &sps->frame_crop_left_offset);
		f_ue(s, &sps->frame_crop_right_offset);
		f_ue(s, &sps->frame_crop_top_offset);
		f_ue(s, &sps->frame_crop_bottom_offset);
	}
	f_u(s, &sps->vui_parameters_present_flag, 1);
	if (!s->err && sps->vui_parameters_present_flag)
		syn_vui(s, &sps->vui);
	f_trailing(s);
	return s->err;
}

/* ---- 7.3.2.2 PPS (ref :364-519) ------------------------------------------------ */

int h264_syn_pps_ids(struct h264_syn *s, struct h264_pps *pps)
{
	f_ue(s, &pps->pic_parameter_set_id);
	f_ue(s, &pps->seq_parameter_set_id);
	return s->err;
}

int h264_syn_pps_body(struct h264_syn *s, const struct h264_sps *sps,
		      struct h264_pps *p)
{
	uint32_t i, n;
	f_u(s, &p->entropy_coding_mode_flag, 1);
	f_u(s, &p->bottom_field_pic_order_in_frame_present_flag, 1);
	f_ue(s, &p->num_slice_groups_minus1);
	if (!s->err && p->num_slice_groups_minus1 > 0) {
		f_ue(s, &p->slice_group_map_type);
		if (s->err)
			return s->err;
		switch (p->slice_group_map_type) {
		case 0:
			require(s, p->num_slice_groups_minus1 <
					   H264_ARRAY_SIZE(p->run_length_minus1));
			for (i = 0; !s->err && i <= p->num_slice_groups_minus1; i++)
				f_ue(s, &p->run_length_minus1[i]);
			break;
		case 1:
			break;
		case 2:
			require(s, p->num_slice_groups_minus1 <
					   H264_ARRAY_SIZE(p->top_left));
			for (i = 0; !s->err && i < p->num_slice_groups_minus1; i++) {
				f_ue(s, &p->top_left[i]);
				f_ue(s, &p->bottom_right[i]);
			}
			break;
		case 3:
		case 4:
		case 5:
			f_u(s, &p->slice_group_change_direction_flag, 1);
			f_ue(s, &p->slice_group_change_rate_minus1);
			break;
		case 6:
			f_ue(s, &p->pic_size_in_map_units_minus1);
			n = h264_intlog2(p->num_slice_groups_minus1 + 1);
			require(s, p->pic_size_in_map_units_minus1 <
					   H264_ARRAY_SIZE(p->slice_group_id));
			for (i = 0; !s->err && i <= p->pic_size_in_map_units_minus1;
			     i++)
				f_u(s, &p->slice_group_id[i], n);
			break;
		default:
			require(s, 0);
			break;
		}
	}
	f_ue(s, &p->num_ref_idx_l0_default_active_minus1);
	f_ue(s, &p->num_ref_idx_l1_default_active_minus1);
	f_u(s, &p->weighted_pred_flag, 1);
	f_u(s, &p->weighted_bipred_idc, 2);
	f_se(s, &p->pic_init_qp_minus26);
	f_se(s, &p->pic_init_qs_minus26);
	f_se(s, &p->chroma_qp_index_offset);
	f_u(s, &p->deblocking_filter_control_present_flag, 1);
	f_u(s, &p->constrained_intra_pred_flag, 1);
	f_u(s, &p->redundant_pic_cnt_present_flag, 1);
	if (s->err)
		return s->err;
	/* the optional tail: remembered so that writing reproduces it */
	if (!s->wr && h264_bs_more_rbsp_data(s->bs))
		p->_more_rbsp_data_present = 1;
	if (p->_more_rbsp_data_present) {
		f_u(s, &p->transform_8x8_mode_flag, 1);
		f_u(s, &p->pic_scaling_matrix_present_flag, 1);
		if (!s->err && p->pic_scaling_matrix_present_flag) {
			n = 6;
			if (p->transform_8x8_mode_flag)
				n += sps->chroma_format_idc != 3 ? 2 : 6;
			syn_scaling_matrix(s, &p->pic_scaling_matrix, n);
		}
		f_se(s, &p->second_chroma_qp_index_offset);
	}
	f_trailing(s);
	return s->err;
}

/* ---- 7.3.2.4 AUD, 7.3.2.7 filler (ref :977-1007) ------------------------------ */

int h264_syn_aud(struct h264_syn *s, struct h264_aud *aud)
{
	f_u(s, &aud->primary_pic_type, 3);
	f_trailing(s);
	return s->err;
}

int h264_syn_filler(struct h264_syn *s, size_t *len)
{
	uint32_t byte = 0xff;
	size_t i;
	if (s->wr) {
		for (i = 0; !s->err && i < *len; i++)
			f_u(s, &byte, 8);
	} else {
		*len = 0;
		while (!s->err && h264_bs_next_bits(s->bs, &byte, 8) == 8 &&
		       byte == 0xff) {
			f_u(s, &byte, 8);
			*len += 1;
		}
	}
	f_trailing(s);
	return s->err;
}

/* ---- SEI payloads (ref :525-863) ----------------------------------------------- */

static void sei_buffering_period(struct h264_syn *s, struct h264_ctx *ctx,
				 struct h264_sei_buffering_period *b)
{
	const struct h264_hrd *hrd;
	uint32_t i, n;
	f_ue(s, &b->seq_parameter_set_id);
	if (s->err)
		return;
	latch(s, h264_ctx_set_active_sps(ctx, b->seq_parameter_set_id));
	if (s->err)
		return;
	if (ctx->sps->vui.nal_hrd_parameters_present_flag) {
		hrd = &ctx->sps->vui.nal_hrd;
		n = hrd->initial_cpb_removal_delay_length_minus1 + 1;
		require(s, hrd->cpb_cnt_minus1 < H264_ARRAY_SIZE(b->nal_hrd_cpb));
		for (i = 0; !s->err && i <= hrd->cpb_cnt_minus1; i++) {
			f_u(s, &b->nal_hrd_cpb[i].initial_cpb_removal_delay, n);
			f_u(s, &b->nal_hrd_cpb[i].initial_cpb_removal_delay_offset,
			    n);
		}
	}
	if (ctx->sps->vui.vcl_hrd_parameters_present_flag) {
		hrd = &ctx->sps->vui.vcl_hrd;
		n = hrd->initial_cpb_removal_delay_length_minus1 + 1;
		require(s, hrd->cpb_cnt_minus1 < H264_ARRAY_SIZE(b->vcl_hrd_cpb));
		for (i = 0; !s->err && i <= hrd->cpb_cnt_minus1; i++) {
			f_u(s, &b->vcl_hrd_cpb[i].initial_cpb_removal_delay, n);
			f_u(s, &b->vcl_hrd_cpb[i].initial_cpb_removal_delay_offset,
			    n);
		}
	}
}

static void sei_pic_timing(struct h264_syn *s, struct h264_ctx *ctx,
			   struct h264_sei_pic_timing *t)
{
	/* NumClockTS by pic_struct, Table D-1 */
	static const uint8_t num_clock_ts[16] = {1, 1, 1, 2, 2, 3, 3, 2, 3};
	const struct h264_vui *vui;
	const struct h264_hrd *hrd = NULL;
	uint32_t i, n;
	require(s, ctx->sps != NULL);
	if (s->err)
		return;
	vui = &ctx->sps->vui;
	if (vui->nal_hrd_parameters_present_flag)
		hrd = &vui->nal_hrd;
	else if (vui->vcl_hrd_parameters_present_flag)
		hrd = &vui->vcl_hrd;
	if (hrd != NULL) {
		f_u(s, &t->cpb_removal_delay,
		    hrd->cpb_removal_delay_length_minus1 + 1);
		f_u(s, &t->dpb_output_delay,
		    hrd->dpb_output_delay_length_minus1 + 1);
	}
	if (!vui->pic_struct_present_flag)
		return;
	f_u(s, &t->pic_struct, 4);
	if (s->err)
		return;
	n = hrd != NULL ? hrd->time_offset_length : 24;
	for (i = 0; !s->err && i < num_clock_ts[t->pic_struct & 15]; i++) {
		f_u(s, &t->clk_ts[i].clock_timestamp_flag, 1);
		if (s->err || !t->clk_ts[i].clock_timestamp_flag)
			continue;
		f_u(s, &t->clk_ts[i].ct_type, 2);
		f_u(s, &t->clk_ts[i].nuit_field_based_flag, 1);
		f_u(s, &t->clk_ts[i].counting_type, 5);
		f_u(s, &t->clk_ts[i].full_timestamp_flag, 1);
		f_u(s, &t->clk_ts[i].discontinuity_flag, 1);
		f_u(s, &t->clk_ts[i].cnt_dropped_flag, 1);
		f_u(s, &t->clk_ts[i].n_frames, 8);
		if (s->err)
			return;
		if (t->clk_ts[i].full_timestamp_flag) {
			f_u(s, &t->clk_ts[i].seconds_value, 6);
			f_u(s, &t->clk_ts[i].minutes_value, 6);
			f_u(s, &t->clk_ts[i].hours_value, 5);
		} else {
			f_u(s, &t->clk_ts[i].seconds_flag, 1);
			if (!s->err && t->clk_ts[i].seconds_flag) {
				f_u(s, &t->clk_ts[i].seconds_value, 6);
				f_u(s, &t->clk_ts[i].minutes_flag, 1);
				if (!s->err && t->clk_ts[i].minutes_flag) {
					f_u(s, &t->clk_ts[i].minutes_value, 6);
					f_u(s, &t->clk_ts[i].hours_flag, 1);
					if (!s->err && t->clk_ts[i].hours_flag)
						f_u(s, &t->clk_ts[i].hours_value,
						    5);
				}
			}
		}
		if (n > 0)
			f_i(s, &t->clk_ts[i].time_offset, n);
	}
}

static void sei_pan_scan_rect(struct h264_syn *s,
			      struct h264_sei_pan_scan_rect *r)
{
	uint32_t i;
	f_ue(s, &r->pan_scan_rect_id);
	f_u(s, &r->pan_scan_rect_cancel_flag, 1);
	if (s->err || r->pan_scan_rect_cancel_flag)
		return;
	f_ue(s, &r->pan_scan_cnt_minus1);
	require(s, r->pan_scan_cnt_minus1 < H264_ARRAY_SIZE(r->pan_scan_rect));
	for (i = 0; !s->err && i <= r->pan_scan_cnt_minus1; i++) {
		f_se(s, &r->pan_scan_rect[i].left_offset);
		f_se(s, &r->pan_scan_rect[i].right_offset);
		f_se(s, &r->pan_scan_rect[i].top_offset);
		f_se(s, &r->pan_scan_rect[i].bottom_offset);
	}
	f_ue(s, &r->pan_scan_rect_repetition_period);
}

/* opaque rest of a payload: a view into the raw copy when reading */
static void sei_blob(struct h264_syn *s, const uint8_t **buf, size_t *len)
{
	size_t i;
	if (s->err)
		return;
	if (!s->wr) {
		require(s, h264_bs_byte_aligned(s->bs));
		if (s->err)
			return;
		*buf = s->bs->cdata + s->bs->off;
		*len = s->bs->len - s->bs->off;
		return;
	}
	require(s, *len == 0 || *buf != NULL);
	for (i = 0; !s->err && i < *len; i++) {
		uint32_t b = (*buf)[i];
		f_u(s, &b, 8);
	}
}

/* one payload + its typed callback (ref H264_SEI macro,
 * src/h264_syntax_ops.h:160-171, and one_sei :794-863) */
static int one_sei(struct h264_bitstream *bs, int wr, struct h264_ctx *ctx,
		   const struct h264_ctx_cbs *cbs, void *userdata,
		   struct h264_sei *sei)
{
	struct h264_syn s = {bs, wr, 0};
	uint32_t i;
	switch (sei->type) {
	case H264_SEI_TYPE_BUFFERING_PERIOD:
		sei_buffering_period(&s, ctx, &sei->buffering_period);
		if (!s.err && cbs && cbs->sei_buffering_period)
			cbs->sei_buffering_period(ctx, sei->raw.buf, sei->raw.len,
						  &sei->buffering_period, userdata);
		break;
	case H264_SEI_TYPE_PIC_TIMING:
		sei_pic_timing(&s, ctx, &sei->pic_timing);
		if (!s.err && cbs && cbs->sei_pic_timing)
			cbs->sei_pic_timing(ctx, sei->raw.buf, sei->raw.len,
					    &sei->pic_timing, userdata);
		break;
	case H264_SEI_TYPE_PAN_SCAN_RECT:
		sei_pan_scan_rect(&s, &sei->pan_scan_rect);
		if (!s.err && cbs && cbs->sei_pan_scan_rect)
			cbs->sei_pan_scan_rect(ctx, sei->raw.buf, sei->raw.len,
					       &sei->pan_scan_rect, userdata);
		break;
	case H264_SEI_TYPE_FILLER_PAYLOAD:
		sei_blob(&s, &sei->filler_payload.buf, &sei->filler_payload.len);
		if (!s.err && cbs && cbs->sei_filler_payload)
			cbs->sei_filler_payload(ctx, sei->raw.buf, sei->raw.len,
						&sei->filler_payload, userdata);
		break;
	case H264_SEI_TYPE_USER_DATA_REGISTERED:
		f_u(&s, &sei->user_data_registered.country_code, 8);
		if (sei->user_data_registered.country_code == 0xff)
			f_u(&s,
			    &sei->user_data_registered.country_code_extension_byte,
			    8);
		sei_blob(&s, &sei->user_data_registered.buf,
			 &sei->user_data_registered.len);
		if (!s.err && cbs && cbs->sei_user_data_registered)
			cbs->sei_user_data_registered(ctx, sei->raw.buf,
						      sei->raw.len,
						      &sei->user_data_registered,
						      userdata);
		break;
	case H264_SEI_TYPE_USER_DATA_UNREGISTERED:
		for (i = 0; i < 16; i++)
			f_u8(&s, &sei->user_data_unregistered.uuid[i]);
		sei_blob(&s, &sei->user_data_unregistered.buf,
			 &sei->user_data_unregistered.len);
		if (!s.err && cbs && cbs->sei_user_data_unregistered)
			cbs->sei_user_data_unregistered(
				ctx, sei->raw.buf, sei->raw.len,
				&sei->user_data_unregistered, userdata);
		break;
	case H264_SEI_TYPE_RECOVERY_POINT:
		f_ue(&s, &sei->recovery_point.recovery_frame_cnt);
		f_u(&s, &sei->recovery_point.exact_match_flag, 1);
		f_u(&s, &sei->recovery_point.broken_link_flag, 1);
		f_u(&s, &sei->recovery_point.changing_slice_group_idc, 2);
		if (!s.err && cbs && cbs->sei_recovery_point)
			cbs->sei_recovery_point(ctx, sei->raw.buf, sei->raw.len,
						&sei->recovery_point, userdata);
		break;
	default:
		return 0; /* other payload types are kept raw */
	}
	if (s.err)
		return s.err;
	if (!wr) {
		/* whatever pads the payload to a byte is swallowed */
		uint32_t bit = 0;
		while (!s.err && !h264_bs_byte_aligned(bs))
			f_u(&s, &bit, 1);
	} else if (!h264_bs_byte_aligned(bs)) {
		f_trailing(&s);
	}
	return s.err;
}

int h264_write_one_sei(struct h264_bitstream *bs, struct h264_ctx *ctx,
		       const struct h264_sei *sei)
{
	return one_sei(bs, 1, ctx, NULL, NULL, (struct h264_sei *)sei);
}

/* 7.3.2.3 (ref :866-971): every message is copied out of the RBSP first and
 * parsed from the copy, which needs no de-escaping */
int h264_syn_sei_read(struct h264_bitstream *bs, struct h264_ctx *ctx,
		      const struct h264_ctx_cbs *cbs, void *userdata)
{
	struct h264_syn s = {bs, 0, 0};
	int res;
	do {
		uint32_t type = 0, size = 0, i, byte = 0;
		struct h264_sei *sei = NULL;
		struct h264_bitstream sub;
		res = h264_bs_read_bits_ff_coded(bs, &type);
		if (res < 0)
			return res;
		res = h264_bs_read_bits_ff_coded(bs, &size);
		if (res < 0)
			return res;
		res = h264_ctx_add_sei_internal(ctx, &sei);
		if (res < 0)
			return res;
		sei->type = type;
		sei->raw.buf = malloc(size ? size : 1);
		if (sei->raw.buf == NULL)
			return -ENOMEM;
		sei->raw.len = size;
		for (i = 0; i < size; i++) {
			f_u(&s, &byte, 8);
			if (s.err)
				return s.err;
			sei->raw.buf[i] = (uint8_t)byte;
		}
		if (cbs && cbs->sei)
			cbs->sei(ctx, sei->type, sei->raw.buf, sei->raw.len,
				 userdata);
		h264_bs_cinit(&sub, sei->raw.buf, sei->raw.len, 0);
		res = one_sei(&sub, 0, ctx, cbs, userdata, sei);
		if (res < 0)
			return res;
	} while (h264_bs_more_rbsp_data(bs));
	f_trailing(&s);
	return s.err;
}

int h264_syn_sei_write(struct h264_bitstream *bs, struct h264_ctx *ctx)
{
	struct h264_syn s = {bs, 1, 0};
	uint32_t k, i;
	if (ctx->sei_count == 0)
		return -EIO;
	for (k = 0; k < ctx->sei_count; k++) {
		const struct h264_sei *sei = &ctx->sei_table[k];
		if (sei->raw.buf == NULL || sei->raw.len == 0)
			return -EIO;
		latch(&s, h264_bs_write_bits_ff_coded(bs, sei->type));
		latch(&s, h264_bs_write_bits_ff_coded(bs, (uint32_t)sei->raw.len));
		for (i = 0; !s.err && i < sei->raw.len; i++) {
			uint32_t b = sei->raw.buf[i];
			f_u(&s, &b, 8);
		}
		if (s.err)
			return s.err;
	}
	f_trailing(&s);
	return s.err;
}

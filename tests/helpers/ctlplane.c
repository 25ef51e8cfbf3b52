/**
 * tests/helpers/ctlplane.c -- control-plane cross-check, CPU only.
 * Links BOTH the product (libh264.so, public API) and the oracle
 * (liboracle.so) and drives them with the same randomly filled structures:
 * SPS (VUI, HRD, scaling lists), PPS, AUD, filler, every SEI type the library
 * knows, stream info and the AVCC helpers.  Returns 0 or the (negative) number
 * of the first step whose outputs differ.
 */
#include <stdio.h>

#include "../../oracle/orc.h"

static uint64_t rnd(uint64_t *s)
{
	uint64_t x = *s;
	x ^= x << 13;
	x ^= x >> 7;
	x ^= x << 17;
	*s = x;
	return x;
}
#define R(n) ((uint32_t)(rnd(s) % (n)))

static void fill_hrd(struct h264_hrd *h, uint64_t *s)
{
	uint32_t i;
	h->cpb_cnt_minus1 = R(4);
	h->bit_rate_scale = R(16);
	h->cpb_size_scale = R(16);
	for (i = 0; i <= h->cpb_cnt_minus1; i++) {
		h->cpb[i].bit_rate_value_minus1 = R(100000);
		h->cpb[i].cpb_size_value_minus1 = R(100000);
		h->cpb[i].cbr_flag = R(2);
	}
	h->initial_cpb_removal_delay_length_minus1 = R(32);
	h->cpb_removal_delay_length_minus1 = R(32);
	h->dpb_output_delay_length_minus1 = R(32);
	h->time_offset_length = R(32);
}

static void fill_scaling(struct h264_scaling_matrix *m, uint64_t *s, int n)
{
	int i, j;
	for (i = 0; i < n; i++) {
		int size = i < 6 ? 16 : 64;
		int32_t *list = i < 6 ? m->scaling_list_4x4[i]
				      : m->scaling_list_8x8[i - 6];
		m->scaling_list_present_flag[i] = R(2);
		if (!m->scaling_list_present_flag[i])
			continue;
		if (R(5) == 0) {
			if (i < 6)
				m->use_default_4x4[i] = 1;
			else
				m->use_default_8x8[i - 6] = 1;
			continue;
		}
		for (j = 0; j < size; j++)
			list[j] = 1 + R(255);
	}
}

static void fill_sps(struct h264_sps *sps, uint64_t *s)
{
	struct h264_vui *v = &sps->vui;
	uint32_t i;
	static const uint32_t profiles[] = {66, 77, 88, 100, 110, 122, 244};
	memset(sps, 0, sizeof(*sps));
	sps->profile_idc = profiles[R(7)];
	sps->constraint_set1_flag = R(2);
	sps->level_idc = 10 + R(42);
	sps->seq_parameter_set_id = R(32);
	sps->chroma_format_idc = 1;
	if (sps->profile_idc >= 100) {
		sps->chroma_format_idc = R(4);
		if (sps->chroma_format_idc == 3)
			sps->separate_colour_plane_flag = R(2);
		sps->bit_depth_luma_minus8 = R(7);
		sps->bit_depth_chroma_minus8 = R(7);
		sps->qpprime_y_zero_transform_bypass_flag = R(2);
		sps->seq_scaling_matrix_present_flag = R(2);
		if (sps->seq_scaling_matrix_present_flag)
			fill_scaling(&sps->seq_scaling_matrix, s,
				     sps->chroma_format_idc != 3 ? 8 : 12);
	}
	sps->log2_max_frame_num_minus4 = R(13);
	sps->pic_order_cnt_type = R(3);
	if (sps->pic_order_cnt_type == 0) {
		sps->log2_max_pic_order_cnt_lsb_minus4 = R(13);
	} else if (sps->pic_order_cnt_type == 1) {
		sps->delta_pic_order_always_zero_flag = R(2);
		sps->offset_for_non_ref_pic = (int32_t)R(2000) - 1000;
		sps->offset_for_top_to_bottom_field = (int32_t)R(2000) - 1000;
		sps->num_ref_frames_in_pic_order_cnt_cycle = R(20);
		for (i = 0; i < sps->num_ref_frames_in_pic_order_cnt_cycle; i++)
			sps->offset_for_ref_frame[i] = (int32_t)R(2000) - 1000;
	}
	sps->max_num_ref_frames = R(17);
	sps->gaps_in_frame_num_value_allowed_flag = R(2);
	sps->pic_width_in_mbs_minus1 = R(256);
	sps->pic_height_in_map_units_minus1 = R(135);
	sps->frame_mbs_only_flag = R(4) != 0;
	if (!sps->frame_mbs_only_flag)
		sps->mb_adaptive_frame_field_flag = R(2);
	sps->direct_8x8_inference_flag = R(2);
	sps->frame_cropping_flag = R(2);
	if (sps->frame_cropping_flag) {
		sps->frame_crop_left_offset = R(4);
		sps->frame_crop_right_offset = R(4);
		sps->frame_crop_top_offset = R(4);
		sps->frame_crop_bottom_offset = R(8);
	}
	sps->vui_parameters_present_flag = R(4) != 0;
	if (!sps->vui_parameters_present_flag)
		return;
	v->aspect_ratio_info_present_flag = R(2);
	if (v->aspect_ratio_info_present_flag) {
		v->aspect_ratio_idc = R(3) == 0 ? 255 : R(17);
		if (v->aspect_ratio_idc == 255) {
			v->sar_width = 1 + R(65535);
			v->sar_height = 1 + R(65535);
		}
	}
	v->overscan_info_present_flag = R(2);
	if (v->overscan_info_present_flag)
		v->overscan_appropriate_flag = R(2);
	v->video_signal_type_present_flag = R(2);
	if (v->video_signal_type_present_flag) {
		v->video_format = R(8);
		v->video_full_range_flag = R(2);
		v->colour_description_present_flag = R(2);
		if (v->colour_description_present_flag) {
			v->colour_primaries = R(256);
			v->transfer_characteristics = R(256);
			v->matrix_coefficients = R(256);
		}
	}
	v->chroma_loc_info_present_flag = R(2);
	if (v->chroma_loc_info_present_flag) {
		v->chroma_sample_loc_type_top_field = R(6);
		v->chroma_sample_loc_type_bottom_field = R(6);
	}
	v->timing_info_present_flag = R(4) != 0;
	if (v->timing_info_present_flag) {
		v->num_units_in_tick = 1 + R(100000);
		v->time_scale = 1 + R(10000000);
		v->fixed_frame_rate_flag = R(2);
	}
	v->nal_hrd_parameters_present_flag = R(2);
	if (v->nal_hrd_parameters_present_flag)
		fill_hrd(&v->nal_hrd, s);
	v->vcl_hrd_parameters_present_flag = R(2);
	if (v->vcl_hrd_parameters_present_flag)
		fill_hrd(&v->vcl_hrd, s);
	if (v->nal_hrd_parameters_present_flag ||
	    v->vcl_hrd_parameters_present_flag)
		v->low_delay_hrd_flag = R(2);
	v->pic_struct_present_flag = R(2);
	v->bitstream_restriction_flag = R(2);
	if (v->bitstream_restriction_flag) {
		v->motion_vectors_over_pic_boundaries_flag = R(2);
		v->max_bytes_per_pic_denom = R(17);
		v->max_bits_per_mb_denom = R(17);
		v->log2_max_mv_length_horizontal = R(17);
		v->log2_max_mv_length_vertical = R(17);
		v->max_num_reorder_frames = R(17);
		v->max_dec_frame_buffering = R(17);
	}
}

static void fill_pps(struct h264_pps *pps, const struct h264_sps *sps,
		     uint64_t *s)
{
	memset(pps, 0, sizeof(*pps));
	pps->pic_parameter_set_id = R(256);
	pps->seq_parameter_set_id = sps->seq_parameter_set_id;
	pps->entropy_coding_mode_flag = R(2);
	pps->bottom_field_pic_order_in_frame_present_flag = R(2);
	pps->num_ref_idx_l0_default_active_minus1 = R(32);
	pps->num_ref_idx_l1_default_active_minus1 = R(32);
	pps->weighted_pred_flag = R(2);
	pps->weighted_bipred_idc = R(3);
	pps->pic_init_qp_minus26 = (int32_t)R(52) - 26;
	pps->pic_init_qs_minus26 = (int32_t)R(52) - 26;
	pps->chroma_qp_index_offset = (int32_t)R(25) - 12;
	pps->deblocking_filter_control_present_flag = R(2);
	pps->constrained_intra_pred_flag = R(2);
	pps->redundant_pic_cnt_present_flag = R(2);
	pps->_more_rbsp_data_present = R(2);
	if (pps->_more_rbsp_data_present) {
		pps->transform_8x8_mode_flag = R(2);
		pps->pic_scaling_matrix_present_flag = R(2);
		if (pps->pic_scaling_matrix_present_flag)
			fill_scaling(&pps->pic_scaling_matrix, s,
				     6 + (sps->chroma_format_idc != 3 ? 2 : 6) *
						 pps->transform_8x8_mode_flag);
		pps->second_chroma_qp_index_offset = (int32_t)R(25) - 12;
	}
}

static void fill_sei(struct h264_sei *sei, const struct h264_sps *sps,
		     uint64_t *s, uint8_t *blob, size_t blob_len, uint32_t kind)
{
	uint32_t i, n;
	memset(sei, 0, sizeof(*sei));
	for (i = 0; i < blob_len; i++)
		blob[i] = (uint8_t)(R(4) == 0 ? 0 : R(256));
	switch (kind) {
	case 0: {
		struct h264_sei_buffering_period *b = &sei->buffering_period;
		sei->type = H264_SEI_TYPE_BUFFERING_PERIOD;
		b->seq_parameter_set_id = sps->seq_parameter_set_id;
		if (sps->vui.nal_hrd_parameters_present_flag) {
			n = sps->vui.nal_hrd
				    .initial_cpb_removal_delay_length_minus1 +
			    1;
			for (i = 0; i <= sps->vui.nal_hrd.cpb_cnt_minus1; i++) {
				b->nal_hrd_cpb[i].initial_cpb_removal_delay =
					(uint32_t)(rnd(s) &
						   ((1ull << n) - 1));
				b->nal_hrd_cpb[i]
					.initial_cpb_removal_delay_offset =
					(uint32_t)(rnd(s) &
						   ((1ull << n) - 1));
			}
		}
		if (sps->vui.vcl_hrd_parameters_present_flag) {
			n = sps->vui.vcl_hrd
				    .initial_cpb_removal_delay_length_minus1 +
			    1;
			for (i = 0; i <= sps->vui.vcl_hrd.cpb_cnt_minus1; i++) {
				b->vcl_hrd_cpb[i].initial_cpb_removal_delay =
					(uint32_t)(rnd(s) &
						   ((1ull << n) - 1));
				b->vcl_hrd_cpb[i]
					.initial_cpb_removal_delay_offset =
					(uint32_t)(rnd(s) &
						   ((1ull << n) - 1));
			}
		}
		break;
	}
	case 1: {
		struct h264_sei_pic_timing *p = &sei->pic_timing;
		const struct h264_hrd *h =
			sps->vui.nal_hrd_parameters_present_flag
				? &sps->vui.nal_hrd
				: &sps->vui.vcl_hrd;
		int hrd = sps->vui.nal_hrd_parameters_present_flag ||
			  sps->vui.vcl_hrd_parameters_present_flag;
		static const uint32_t nclk[9] = {1, 1, 1, 2, 2, 3, 3, 2, 3};
		sei->type = H264_SEI_TYPE_PIC_TIMING;
		if (hrd) {
			n = h->cpb_removal_delay_length_minus1 + 1;
			p->cpb_removal_delay =
				(uint32_t)(rnd(s) & ((1ull << n) - 1));
			n = h->dpb_output_delay_length_minus1 + 1;
			p->dpb_output_delay =
				(uint32_t)(rnd(s) & ((1ull << n) - 1));
		}
		if (sps->vui.pic_struct_present_flag) {
			p->pic_struct = R(9);
			for (i = 0; i < nclk[p->pic_struct]; i++) {
				p->clk_ts[i].clock_timestamp_flag = R(3) != 0;
				if (!p->clk_ts[i].clock_timestamp_flag)
					continue;
				p->clk_ts[i].ct_type = R(3);
				p->clk_ts[i].nuit_field_based_flag = R(2);
				p->clk_ts[i].counting_type = R(7);
				p->clk_ts[i].full_timestamp_flag = R(2);
				p->clk_ts[i].discontinuity_flag = R(2);
				p->clk_ts[i].cnt_dropped_flag = R(2);
				p->clk_ts[i].n_frames = R(256);
				if (p->clk_ts[i].full_timestamp_flag) {
					p->clk_ts[i].seconds_value = R(60);
					p->clk_ts[i].minutes_value = R(60);
					p->clk_ts[i].hours_value = R(24);
				} else {
					p->clk_ts[i].seconds_flag = R(2);
					if (p->clk_ts[i].seconds_flag) {
						p->clk_ts[i].seconds_value = R(60);
						p->clk_ts[i].minutes_flag = R(2);
					}
					if (p->clk_ts[i].minutes_flag) {
						p->clk_ts[i].minutes_value = R(60);
						p->clk_ts[i].hours_flag = R(2);
					}
					if (p->clk_ts[i].hours_flag)
						p->clk_ts[i].hours_value = R(24);
				}
				if (hrd && h->time_offset_length > 0) {
					uint32_t tl = h->time_offset_length;
					int64_t span = 1ll << (tl - 1);
					p->clk_ts[i].time_offset = (int32_t)(
						(int64_t)(rnd(s) % (2 * span)) -
						span);
				}
			}
		}
		break;
	}
	case 2: {
		struct h264_sei_pan_scan_rect *p = &sei->pan_scan_rect;
		sei->type = H264_SEI_TYPE_PAN_SCAN_RECT;
		p->pan_scan_rect_id = R(1000);
		p->pan_scan_rect_cancel_flag = R(4) == 0;
		if (!p->pan_scan_rect_cancel_flag) {
			p->pan_scan_cnt_minus1 = R(3);
			for (i = 0; i <= p->pan_scan_cnt_minus1; i++) {
				p->pan_scan_rect[i].left_offset =
					(int32_t)R(2000) - 1000;
				p->pan_scan_rect[i].right_offset =
					(int32_t)R(2000) - 1000;
				p->pan_scan_rect[i].top_offset =
					(int32_t)R(2000) - 1000;
				p->pan_scan_rect[i].bottom_offset =
					(int32_t)R(2000) - 1000;
			}
			p->pan_scan_rect_repetition_period = R(100);
		}
		break;
	}
	case 3:
		sei->type = H264_SEI_TYPE_FILLER_PAYLOAD;
		for (i = 0; i < blob_len; i++)
			blob[i] = 0xff;
		sei->filler_payload.buf = blob;
		sei->filler_payload.len = blob_len;
		break;
	case 4:
		sei->type = H264_SEI_TYPE_USER_DATA_REGISTERED;
		sei->user_data_registered.country_code = R(3) ? R(255) : 0xff;
		if (sei->user_data_registered.country_code == 0xff)
			sei->user_data_registered.country_code_extension_byte =
				R(256);
		sei->user_data_registered.buf = blob;
		sei->user_data_registered.len = blob_len;
		break;
	case 5:
		sei->type = H264_SEI_TYPE_USER_DATA_UNREGISTERED;
		for (i = 0; i < 16; i++)
			sei->user_data_unregistered.uuid[i] = (uint8_t)R(256);
		sei->user_data_unregistered.buf = blob;
		sei->user_data_unregistered.len = blob_len;
		break;
	default:
		sei->type = H264_SEI_TYPE_RECOVERY_POINT;
		sei->recovery_point.recovery_frame_cnt = R(1000);
		sei->recovery_point.exact_match_flag = R(2);
		sei->recovery_point.broken_link_flag = R(2);
		sei->recovery_point.changing_slice_group_idc = R(3);
		break;
	}
}

/* write the NAL ctx describes with both libraries, compare */
static int both_write(struct h264_ctx *pc, struct orc_ctx *oc, uint8_t *out,
		      size_t cap, size_t *len)
{
	struct h264_bitstream a, b;
	int ra, rb, same;
	h264_bs_init(&a, NULL, 0, 1);
	h264_bs_init(&b, NULL, 0, 1);
	ra = h264_write_nalu(&a, pc);
	rb = orc_write_nalu(&b, oc);
	same = ra == rb && (ra < 0 || (a.off == b.off && a.cachebits == 0 &&
				       b.cachebits == 0 &&
				       memcmp(a.data, b.data, a.off) == 0));
	if (same && ra >= 0 && out != NULL && a.off <= cap) {
		memcpy(out, a.data, a.off);
		*len = a.off;
	}
	h264_bs_clear(&a);
	h264_bs_clear(&b);
	if (!same)
		return -1;
	return ra < 0 ? 1 : 0; /* 1: both refused */
}

static void set_hdr(struct h264_ctx *pc, struct orc_ctx *oc, uint32_t ref,
		    uint32_t type)
{
	struct h264_nalu_header nh = {0, ref, type};
	h264_ctx_clear_nalu(pc);
	orc_ctx_clear_nalu(oc);
	h264_ctx_set_nalu_header(pc, &nh);
	orc_ctx_set_nalu_header(oc, &nh);
}

#define STEP(n, cond)                                                          \
	do {                                                                   \
		if (!(cond)) {                                                 \
			res = -(n);                                            \
			goto out;                                              \
		}                                                              \
	} while (0)

int ctl_crosscheck(uint64_t seed)
{
	uint64_t state = seed * 0x9E3779B97F4A7C15ull + 1, *s = &state;
	struct h264_ctx *pc = NULL;
	struct orc_ctx *oc = NULL;
	struct h264_sps *sps = calloc(1, sizeof(*sps));
	struct h264_sps *sps2 = calloc(1, sizeof(*sps2));
	struct h264_sps *sps3 = calloc(1, sizeof(*sps3));
	struct h264_pps *pps = calloc(1, sizeof(*pps));
	struct h264_pps *pps2 = calloc(1, sizeof(*pps2));
	struct h264_pps *pps3 = calloc(1, sizeof(*pps3));
	uint8_t sps_nal[8192], pps_nal[8192], nal[65536], blob[300];
	size_t sps_len = 0, pps_len = 0, nal_len = 0;
	struct h264_sps_derived d1, d2;
	struct h264_info i1, i2;
	struct h264_aud aud;
	uint32_t k;
	int res = 0, w;

	h264_ctx_new(&pc);
	orc_ctx_new(&oc);
	fill_sps(sps, s);
	fill_pps(pps, sps, s);

	/* SPS: write, parse back, derived, info */
	set_hdr(pc, oc, 1 + R(3), H264_NALU_TYPE_SPS);
	STEP(1, h264_ctx_set_sps(pc, sps) == 0 && orc_ctx_set_sps(oc, sps) == 0);
	STEP(2, both_write(pc, oc, sps_nal, sizeof(sps_nal), &sps_len) == 0);
	STEP(3, h264_parse_sps(sps_nal, sps_len, sps2) ==
			orc_parse_sps(sps_nal, sps_len, sps3));
	STEP(4, memcmp(sps2, sps3, sizeof(*sps2)) == 0);
	STEP(5, h264_get_sps_derived(sps2, &d1) == orc_get_sps_derived(sps3, &d2));
	STEP(6, memcmp(&d1, &d2, sizeof(d1)) == 0);

	/* PPS */
	set_hdr(pc, oc, 1 + R(3), H264_NALU_TYPE_PPS);
	STEP(7, h264_ctx_set_pps(pc, pps) == 0 && orc_ctx_set_pps(oc, pps) == 0);
	STEP(8, both_write(pc, oc, pps_nal, sizeof(pps_nal), &pps_len) == 0);
	STEP(9, h264_parse_pps(pps_nal, pps_len, sps2, pps2) ==
			orc_parse_pps(pps_nal, pps_len, sps3, pps3));
	STEP(10, memcmp(pps2, pps3, sizeof(*pps2)) == 0);

	/* stream info from the two NAL units */
	memset(&i1, 0, sizeof(i1));
	memset(&i2, 0, sizeof(i2));
	w = h264_get_info(sps_nal, sps_len, pps_nal, pps_len, &i1);
	STEP(11, w == orc_get_info_from_ps(sps3, &d2, &i2) || w < 0);
	if (w == 0)
		STEP(12, memcmp(&i1, &i2, sizeof(i1)) == 0);

	/* AUD, filler */
	set_hdr(pc, oc, 0, H264_NALU_TYPE_AUD);
	aud.primary_pic_type = R(8);
	h264_ctx_set_aud(pc, &aud);
	oc->aud = aud;
	STEP(13, both_write(pc, oc, NULL, 0, &nal_len) == 0);
	set_hdr(pc, oc, 0, H264_NALU_TYPE_FILLER);
	h264_ctx_set_filler(pc, 5 + R(100));
	oc->filler_len = 0;
	{
		/* same length on both sides */
		size_t fl = 5 + R(100);
		h264_ctx_set_filler(pc, fl);
		oc->filler_len = fl;
	}
	STEP(14, both_write(pc, oc, NULL, 0, &nal_len) == 0);

	/* SEI: three random messages in one NAL unit */
	set_hdr(pc, oc, 0, H264_NALU_TYPE_SEI);
	for (k = 0; k < 3; k++) {
		struct h264_sei sei;
		int ra, rb;
		fill_sei(&sei, sps, s, blob, 1 + R(sizeof(blob) - 1), R(7));
		ra = h264_ctx_add_sei(pc, &sei);
		rb = orc_ctx_add_sei(oc, &sei);
		STEP(15, ra == rb);
	}
	STEP(16, h264_ctx_get_sei_count(pc) == (int)oc->sei_count);
	w = both_write(pc, oc, nal, sizeof(nal), &nal_len);
	STEP(17, w >= 0);
	res = 0;
out:
	h264_ctx_destroy(pc);
	orc_ctx_destroy(oc);
	free(sps);
	free(sps2);
	free(sps3);
	free(pps);
	free(pps2);
	free(pps3);
	return res;
}

/* the SEI NAL of ctl_crosscheck(seed), for feeding both readers from python */
int ctl_make_stream(uint64_t seed, uint8_t *out, size_t cap)
{
	uint64_t state = seed * 0x9E3779B97F4A7C15ull + 1, *s = &state;
	struct orc_ctx *oc = NULL;
	struct h264_sps *sps = calloc(1, sizeof(*sps));
	struct h264_pps *pps = calloc(1, sizeof(*pps));
	struct h264_bitstream bs;
	struct h264_nalu_header nh = {0, 3, H264_NALU_TYPE_SPS};
	uint8_t blob[300];
	size_t pos = 0;
	uint32_t k, t;
	int res = 0;
	orc_ctx_new(&oc);
	fill_sps(sps, s);
	fill_pps(pps, sps, s);
	for (t = 0; t < 3 && res >= 0; t++) {
		static const uint8_t sc[4] = {0, 0, 0, 1};
		orc_ctx_clear_nalu(oc);
		nh.nal_unit_type = t == 0   ? H264_NALU_TYPE_SPS
				   : t == 1 ? H264_NALU_TYPE_PPS
					    : H264_NALU_TYPE_SEI;
		nh.nal_ref_idc = t == 2 ? 0 : 3;
		orc_ctx_set_nalu_header(oc, &nh);
		if (t == 0)
			res = orc_ctx_set_sps(oc, sps);
		else if (t == 1)
			res = orc_ctx_set_pps(oc, pps);
		else {
			/* a message the writer refuses (say a filler payload
			 * that is not all 0xff) is simply left out */
			for (k = 0; k < 4 || oc->sei_count == 0; k++) {
				struct h264_sei sei;
				fill_sei(&sei, sps, s, blob,
					 1 + R(sizeof(blob) - 1), R(7));
				orc_ctx_add_sei(oc, &sei);
			}
		}
		if (res < 0)
			break;
		h264_bs_init(&bs, NULL, 0, 1);
		res = orc_write_nalu(&bs, oc);
		if (res >= 0 && pos + 4 + bs.off <= cap) {
			memcpy(out + pos, sc, 4);
			memcpy(out + pos + 4, bs.data, bs.off);
			pos += 4 + bs.off;
		} else if (res >= 0) {
			res = -ENOBUFS;
		}
		h264_bs_clear(&bs);
	}
	orc_ctx_destroy(oc);
	free(sps);
	free(pps);
	return res < 0 ? res : (int)pos;
}

/* AVCC <-> Annex-B in place, 4-byte codes: product only (the oracle does not
 * restate these); the round trip must give the input back. */
int ctl_avcc_roundtrip(uint8_t *data, size_t len)
{
	uint8_t *copy = malloc(len ? len : 1);
	int res;
	memcpy(copy, data, len);
	res = h264_byte_stream_to_avcc(data, len);
	if (res == 0)
		res = h264_avcc_to_byte_stream(data, len);
	if (res == 0 && memcmp(copy, data, len) != 0)
		res = -1000;
	free(copy);
	return res;
}

/**
 * dump.c -- per-NAL JSON dump (include/h264/h264_dump.h), drop-in for the
 * reference's src/h264_dump.c + the DUMP instantiation of its syntax walk
 * (src/h264_syntax.h, src/h264_syntax_slice_data.h).
 *
 * What the reference does: it walks the syntax of the NAL unit the context
 * holds a third time (after READ and WRITE); every field becomes
 * `field("<expression text>", value)`, and the JSON key is cut out of that
 * text by extract_key() (ref src/h264_dump.c:154-191): what follows the last
 * '.' and the last '>', up to the last '['.  That is the name of the struct
 * member, which is public ABI -- so this file walks the same fields in the
 * same order under the same conditions and names them by member.  Structure
 * (objects, arrays, array items) follows the reference's H264_BEGIN_ / H264_END_
 * marks one for one, including json-c's habits that show in the text: a
 * struct closed inside an ARRAY is dropped (h264_dump_json_end_struct only
 * adds to objects), a key added twice keeps its first position and its last
 * value, and the text is json-c's default "spaced" form.
 *
 * Slice data (H264_DUMP_FLAGS_SLICE_DATA): the reference re-reads the raw slice
 * data with its CPU parser (src/h264_syntax_slice_data.h:832-857); here the
 * slice NAL unit is put together again by the writer, decoded by the
 * slice-decode kernel at the full-syntax tier, and the macroblock records are
 * printed with the reference's keys -- "mb_addr" = frame_num * 10000 + mbAddr
 * (:619), coefficient keys "Luma(blk,idx)" (:228-236).  MBAFF frames and
 * slice groups are decoded on the device like everything else (macroblock
 * addresses from the map the launch built); CABAC slice data is not dumped,
 * like the reference's parse (src/h264_syntax_slice_data.h:715-717).
 *
 * The reference's own output is not available in this image (json-c absent):
 * the TEXT of the dump is unpinned; tests walk it the way the reference's
 * tests/check_stream.py does.
 */
#include "h264_priv.h"

#include <stdio.h>

/* ---- a small JSON tree with json-c's object semantics ----------------------- */

enum jtype { J_INT, J_OBJ, J_ARR };

struct jnode {
	enum jtype type;
	char *key; /* when the parent is an object */
	int64_t val;
	struct jnode **kids;
	size_t n, cap;
};

#define DUMP_MAX_STACK 16

struct h264_dump {
	struct h264_dump_cfg cfg;
	uint32_t flags;
	struct jnode *stack[DUMP_MAX_STACK];
	uint32_t depth;
	int err;
	char *text;
	size_t text_len, text_cap;
};

static void jfree(struct jnode *n)
{
	size_t i;
	if (n == NULL)
		return;
	for (i = 0; i < n->n; i++)
		jfree(n->kids[i]);
	free(n->kids);
	free(n->key);
	free(n);
}

static struct jnode *jnew(enum jtype t, int64_t v)
{
	struct jnode *n = calloc(1, sizeof(*n));
	if (n != NULL) {
		n->type = t;
		n->val = v;
	}
	return n;
}

static int jappend(struct jnode *parent, struct jnode *kid)
{
	if (parent->n == parent->cap) {
		size_t nc = parent->cap ? parent->cap * 2 : 8;
		struct jnode **nk = realloc(parent->kids, nc * sizeof(*nk));
		if (nk == NULL)
			return -ENOMEM;
		parent->kids = nk;
		parent->cap = nc;
	}
	parent->kids[parent->n++] = kid;
	return 0;
}

/* json_object_object_add: an existing key keeps its place, takes the value */
static int jobj_add(struct jnode *obj, const char *key, struct jnode *kid)
{
	size_t i;
	for (i = 0; i < obj->n; i++) {
		if (strcmp(obj->kids[i]->key, key) == 0) {
			kid->key = obj->kids[i]->key;
			obj->kids[i]->key = NULL;
			jfree(obj->kids[i]);
			obj->kids[i] = kid;
			return 0;
		}
	}
	kid->key = strdup(key);
	if (kid->key == NULL)
		return -ENOMEM;
	return jappend(obj, kid);
}

static struct jnode *cur(struct h264_dump *d)
{
	return d->depth ? d->stack[d->depth - 1] : NULL;
}

static void latch(struct h264_dump *d, int res)
{
	if (res < 0 && d->err == 0)
		d->err = res;
}

static void push(struct h264_dump *d, enum jtype t)
{
	struct jnode *n;
	if (d->err)
		return;
	if (d->depth >= DUMP_MAX_STACK) {
		d->err = -EPROTO;
		return;
	}
	n = jnew(t, 0);
	if (n == NULL) {
		d->err = -ENOMEM;
		return;
	}
	d->stack[d->depth++] = n;
}

static struct jnode *pop(struct h264_dump *d)
{
	if (d->err || d->depth < 2) {
		if (!d->err)
			d->err = -EPROTO;
		return NULL;
	}
	return d->stack[--d->depth];
}

/* the seven operations of the reference's dump backend (ref :108-211) */
static void begin_struct(struct h264_dump *d)
{
	push(d, J_OBJ);
}

static void end_struct(struct h264_dump *d, const char *name)
{
	struct jnode *n = pop(d);
	if (n == NULL)
		return;
	if (cur(d)->type == J_OBJ)
		latch(d, jobj_add(cur(d), name, n));
	else
		jfree(n); /* the reference adds a struct to objects only */
}

static void begin_array(struct h264_dump *d)
{
	push(d, J_ARR);
}

static void end_array(struct h264_dump *d, const char *name)
{
	struct jnode *n = pop(d);
	if (n == NULL)
		return;
	if (cur(d)->type == J_OBJ)
		latch(d, jobj_add(cur(d), name, n));
	else
		latch(d, jappend(cur(d), n));
}

static void begin_item(struct h264_dump *d)
{
	push(d, J_OBJ);
}

static void end_item(struct h264_dump *d)
{
	struct jnode *n = pop(d);
	if (n == NULL)
		return;
	if (cur(d)->type == J_ARR)
		latch(d, jappend(cur(d), n));
	else
		jfree(n);
}

static void field(struct h264_dump *d, const char *key, int64_t val)
{
	struct jnode *n;
	if (d->err)
		return;
	n = jnew(J_INT, val);
	if (n == NULL) {
		d->err = -ENOMEM;
		return;
	}
	if (cur(d)->type == J_ARR)
		latch(d, jappend(cur(d), n));
	else
		latch(d, jobj_add(cur(d), key, n));
}

/* F(sps, profile_idc): the member name is the key extract_key() would cut out
 * of the reference's "sps->profile_idc" */
#define F(p, m) field(d, #m, (int64_t)(p)->m)
#define FV(name, v) field(d, name, (int64_t)(v))

/* ---- text (json-c JSON_C_TO_STRING_SPACED) --------------------------------- */

static void out(struct h264_dump *d, const char *s, size_t n)
{
	if (d->err)
		return;
	if (d->text_len + n + 1 > d->text_cap) {
		size_t nc = d->text_cap ? d->text_cap * 2 : 4096;
		char *nt;
		while (nc < d->text_len + n + 1)
			nc *= 2;
		nt = realloc(d->text, nc);
		if (nt == NULL) {
			d->err = -ENOMEM;
			return;
		}
		d->text = nt;
		d->text_cap = nc;
	}
	memcpy(d->text + d->text_len, s, n);
	d->text_len += n;
	d->text[d->text_len] = '\0';
}

static void outs(struct h264_dump *d, const char *s)
{
	out(d, s, strlen(s));
}

static void render(struct h264_dump *d, const struct jnode *n)
{
	size_t i;
	char num[32];
	switch (n->type) {
	case J_INT:
		snprintf(num, sizeof(num), "%lld", (long long)n->val);
		outs(d, num);
		break;
	case J_OBJ:
		outs(d, "{ ");
		for (i = 0; i < n->n; i++) {
			if (i)
				outs(d, ", ");
			outs(d, "\"");
			outs(d, n->kids[i]->key); /* member names: nothing to escape
						   * but the coefficient keys'
						   * "(,)" which are plain */
			outs(d, "\": ");
			render(d, n->kids[i]);
		}
		outs(d, n->n ? " }" : "}");
		break;
	case J_ARR:
		outs(d, "[ ");
		for (i = 0; i < n->n; i++) {
			if (i)
				outs(d, ", ");
			render(d, n->kids[i]);
		}
		outs(d, n->n ? " ]" : "]");
		break;
	}
}

/* ---- non-VCL syntax (ref src/h264_syntax.h:36-1007) ------------------------- */

static void dump_hrd(struct h264_dump *d, const struct h264_hrd *hrd)
{
	uint32_t i;
	F(hrd, cpb_cnt_minus1);
	F(hrd, bit_rate_scale);
	F(hrd, cpb_size_scale);
	begin_array(d);
	for (i = 0; i <= hrd->cpb_cnt_minus1 && i < H264_ARRAY_SIZE(hrd->cpb); i++) {
		begin_item(d);
		F(&hrd->cpb[i], bit_rate_value_minus1);
		F(&hrd->cpb[i], cpb_size_value_minus1);
		F(&hrd->cpb[i], cbr_flag);
		end_item(d);
	}
	end_array(d, "cpb");
	F(hrd, initial_cpb_removal_delay_length_minus1);
	F(hrd, cpb_removal_delay_length_minus1);
	F(hrd, dpb_output_delay_length_minus1);
	F(hrd, time_offset_length);
}

static void dump_vui(struct h264_dump *d, const struct h264_vui *vui)
{
	F(vui, aspect_ratio_info_present_flag);
	if (vui->aspect_ratio_info_present_flag) {
		F(vui, aspect_ratio_idc);
		if (vui->aspect_ratio_idc == H264_ASPECT_RATIO_EXTENDED_SAR) {
			F(vui, sar_width);
			F(vui, sar_height);
		}
	}
	F(vui, overscan_info_present_flag);
	if (vui->overscan_info_present_flag)
		F(vui, overscan_appropriate_flag);
	F(vui, video_signal_type_present_flag);
	if (vui->video_signal_type_present_flag) {
		F(vui, video_format);
		F(vui, video_full_range_flag);
		F(vui, colour_description_present_flag);
		if (vui->colour_description_present_flag) {
			F(vui, colour_primaries);
			F(vui, transfer_characteristics);
			F(vui, matrix_coefficients);
		}
	}
	F(vui, chroma_loc_info_present_flag);
	if (vui->chroma_loc_info_present_flag) {
		F(vui, chroma_sample_loc_type_top_field);
		F(vui, chroma_sample_loc_type_bottom_field);
	}
	F(vui, timing_info_present_flag);
	if (vui->timing_info_present_flag) {
		F(vui, num_units_in_tick);
		F(vui, time_scale);
		F(vui, fixed_frame_rate_flag);
	}
	F(vui, nal_hrd_parameters_present_flag);
	if (vui->nal_hrd_parameters_present_flag) {
		begin_struct(d);
		dump_hrd(d, &vui->nal_hrd);
		end_struct(d, "nal_hrd");
	}
	F(vui, vcl_hrd_parameters_present_flag);
	if (vui->vcl_hrd_parameters_present_flag) {
		begin_struct(d);
		dump_hrd(d, &vui->vcl_hrd);
		end_struct(d, "vcl_hrd");
	}
	if (vui->nal_hrd_parameters_present_flag ||
	    vui->vcl_hrd_parameters_present_flag)
		F(vui, low_delay_hrd_flag);
	F(vui, pic_struct_present_flag);
	F(vui, bitstream_restriction_flag);
	if (vui->bitstream_restriction_flag) {
		F(vui, motion_vectors_over_pic_boundaries_flag);
		F(vui, max_bytes_per_pic_denom);
		F(vui, max_bits_per_mb_denom);
		F(vui, log2_max_mv_length_horizontal);
		F(vui, log2_max_mv_length_vertical);
		F(vui, max_num_reorder_frames);
		F(vui, max_dec_frame_buffering);
	}
}

/* ref :210-257: per list an array item {scaling_list_present_flag,
 * scaling_list_4x4 / _8x8: [values]} */
static void dump_scaling_matrix(struct h264_dump *d,
				const struct h264_scaling_matrix *m, uint32_t size)
{
	uint32_t i, k;
	for (i = 0; i < size && i < 12; i++) {
		begin_item(d);
		FV("scaling_list_present_flag", m->scaling_list_present_flag[i]);
		if (m->scaling_list_present_flag[i]) {
			begin_array(d);
			if (i < 6) {
				for (k = 0; k < 16; k++)
					FV("scaling_list", m->scaling_list_4x4[i][k]);
				end_array(d, "scaling_list_4x4");
			} else {
				for (k = 0; k < 64; k++)
					FV("scaling_list",
					   m->scaling_list_8x8[i - 6][k]);
				end_array(d, "scaling_list_8x8");
			}
		}
		end_item(d);
	}
}

static int profile_has_chroma_info(uint32_t p)
{
	return p == 100 || p == 110 || p == 122 || p == 244 || p == 44 ||
	       p == 83 || p == 86 || p == 118 || p == 128 || p == 138 ||
	       p == 139 || p == 134 || p == 135;
}

static void dump_sps(struct h264_dump *d, const struct h264_sps *sps)
{
	uint32_t i;
	F(sps, profile_idc);
	F(sps, constraint_set0_flag);
	F(sps, constraint_set1_flag);
	F(sps, constraint_set2_flag);
	F(sps, constraint_set3_flag);
	F(sps, constraint_set4_flag);
	F(sps, constraint_set5_flag);
	F(sps, reserved_zero_2bits);
	F(sps, level_idc);
	F(sps, seq_parameter_set_id);
	if (profile_has_chroma_info(sps->profile_idc)) {
		F(sps, chroma_format_idc);
		if (sps->chroma_format_idc == 3)
			F(sps, separate_colour_plane_flag);
		F(sps, bit_depth_luma_minus8);
		F(sps, bit_depth_chroma_minus8);
		F(sps, qpprime_y_zero_transform_bypass_flag);
		F(sps, seq_scaling_matrix_present_flag);
		if (sps->seq_scaling_matrix_present_flag) {
			begin_array(d);
			dump_scaling_matrix(d, &sps->seq_scaling_matrix,
					    sps->chroma_format_idc != 3 ? 8 : 12);
			end_array(d, "seq_scaling_matrix");
		}
	}
	F(sps, log2_max_frame_num_minus4);
	F(sps, pic_order_cnt_type);
	if (sps->pic_order_cnt_type == 0) {
		F(sps, log2_max_pic_order_cnt_lsb_minus4);
	} else if (sps->pic_order_cnt_type == 1) {
		F(sps, delta_pic_order_always_zero_flag);
		F(sps, offset_for_non_ref_pic);
		F(sps, offset_for_top_to_bottom_field);
		F(sps, num_ref_frames_in_pic_order_cnt_cycle);
		begin_array(d);
		for (i = 0; i < sps->num_ref_frames_in_pic_order_cnt_cycle &&
			    i < H264_ARRAY_SIZE(sps->offset_for_ref_frame);
		     i++)
			FV("offset_for_ref_frame", sps->offset_for_ref_frame[i]);
		end_array(d, "offset_for_ref_frame");
	}
	F(sps, max_num_ref_frames);
	F(sps, gaps_in_frame_num_value_allowed_flag);
	F(sps, pic_width_in_mbs_minus1);
	F(sps, pic_height_in_map_units_minus1);
	F(sps, frame_mbs_only_flag);
	if (!sps->frame_mbs_only_flag)
		F(sps, mb_adaptive_frame_field_flag);
	F(sps, direct_8x8_inference_flag);
	F(sps, frame_cropping_flag);
	if (sps->frame_cropping_flag) {
		F(sps, frame_crop_left_offset);
		F(sps, frame_crop_right_offset);
		F(sps, frame_crop_top_offset);
		F(sps, frame_crop_bottom_offset);
	}
	F(sps, vui_parameters_present_flag);
	if (sps->vui_parameters_present_flag) {
		begin_struct(d);
		dump_vui(d, &sps->vui);
		end_struct(d, "vui");
	}
}

static int dump_pps(struct h264_dump *d, struct h264_ctx *ctx,
		    const struct h264_pps *pps)
{
	const struct h264_sps *sps;
	uint32_t i, n;
	F(pps, pic_parameter_set_id);
	F(pps, seq_parameter_set_id);
	if (h264_ctx_set_active_sps(ctx, pps->seq_parameter_set_id) < 0)
		return -EINVAL;
	sps = ctx->sps;
	F(pps, entropy_coding_mode_flag);
	F(pps, bottom_field_pic_order_in_frame_present_flag);
	F(pps, num_slice_groups_minus1);
	if (pps->num_slice_groups_minus1 > 0) {
		F(pps, slice_group_map_type);
		switch (pps->slice_group_map_type) {
		case 0:
			if (pps->num_slice_groups_minus1 >=
			    H264_ARRAY_SIZE(pps->run_length_minus1))
				return -EIO;
			begin_array(d);
			for (i = 0; i <= pps->num_slice_groups_minus1; i++)
				FV("run_length_minus1", pps->run_length_minus1[i]);
			end_array(d, "run_length_minus1");
			break;
		case 1:
			break;
		case 2:
			if (pps->num_slice_groups_minus1 >=
			    H264_ARRAY_SIZE(pps->top_left))
				return -EIO;
			begin_array(d);
			for (i = 0; i < pps->num_slice_groups_minus1; i++) {
				begin_item(d);
				FV("top_left", pps->top_left[i]);
				FV("bottom_right", pps->bottom_right[i]);
				end_item(d);
			}
			end_array(d, "pos");
			break;
		case 3:
		case 4:
		case 5:
			F(pps, slice_group_change_direction_flag);
			F(pps, slice_group_change_rate_minus1);
			break;
		case 6:
			F(pps, pic_size_in_map_units_minus1);
			if (pps->pic_size_in_map_units_minus1 >=
			    H264_ARRAY_SIZE(pps->slice_group_id))
				return -EIO;
			begin_array(d);
			for (i = 0; i <= pps->pic_size_in_map_units_minus1; i++)
				FV("slice_group_id", pps->slice_group_id[i]);
			end_array(d, "slice_group_id");
			break;
		default:
			return -EIO;
		}
	}
	F(pps, num_ref_idx_l0_default_active_minus1);
	F(pps, num_ref_idx_l1_default_active_minus1);
	F(pps, weighted_pred_flag);
	F(pps, weighted_bipred_idc);
	F(pps, pic_init_qp_minus26);
	F(pps, pic_init_qs_minus26);
	F(pps, chroma_qp_index_offset);
	F(pps, deblocking_filter_control_present_flag);
	F(pps, constrained_intra_pred_flag);
	F(pps, redundant_pic_cnt_present_flag);
	if (pps->_more_rbsp_data_present) {
		F(pps, transform_8x8_mode_flag);
		F(pps, pic_scaling_matrix_present_flag);
		if (pps->pic_scaling_matrix_present_flag) {
			n = 6;
			if (pps->transform_8x8_mode_flag)
				n += (sps->chroma_format_idc != 3) ? 2 : 6;
			begin_array(d);
			dump_scaling_matrix(d, &pps->pic_scaling_matrix, n);
			end_array(d, "pic_scaling_matrix");
		}
		F(pps, second_chroma_qp_index_offset);
	}
	return 0;
}

static void dump_sei_data(struct h264_dump *d, const uint8_t *buf, size_t len)
{
	size_t i;
	begin_array(d);
	for (i = 0; i < len && buf != NULL; i++)
		FV("buf", buf[i]);
	end_array(d, "data");
}

static int dump_one_sei(struct h264_dump *d, struct h264_ctx *ctx,
			const struct h264_sei *sei)
{
	static const uint32_t num_clock_ts[16] = {1, 1, 1, 2, 2, 3, 3, 2,
						  3, 0, 0, 0, 0, 0, 0, 0};
	uint32_t i;
	switch (sei->type) {
	case H264_SEI_TYPE_BUFFERING_PERIOD: {
		const struct h264_sei_buffering_period *s = &sei->buffering_period;
		const struct h264_hrd *hrd;
		F(s, seq_parameter_set_id);
		if (h264_ctx_set_active_sps(ctx, s->seq_parameter_set_id) < 0)
			return -EINVAL;
		if (ctx->sps->vui.nal_hrd_parameters_present_flag) {
			hrd = &ctx->sps->vui.nal_hrd;
			if (hrd->cpb_cnt_minus1 >= H264_ARRAY_SIZE(s->nal_hrd_cpb))
				return -EIO;
			begin_array(d);
			for (i = 0; i <= hrd->cpb_cnt_minus1; i++) {
				begin_item(d);
				F(&s->nal_hrd_cpb[i], initial_cpb_removal_delay);
				F(&s->nal_hrd_cpb[i],
				  initial_cpb_removal_delay_offset);
				end_item(d);
			}
			end_array(d, "nal_hrd_cpb");
		}
		if (ctx->sps->vui.vcl_hrd_parameters_present_flag) {
			hrd = &ctx->sps->vui.vcl_hrd;
			if (hrd->cpb_cnt_minus1 >= H264_ARRAY_SIZE(s->vcl_hrd_cpb))
				return -EIO;
			begin_array(d);
			for (i = 0; i <= hrd->cpb_cnt_minus1; i++) {
				begin_item(d);
				F(&s->vcl_hrd_cpb[i], initial_cpb_removal_delay);
				F(&s->vcl_hrd_cpb[i],
				  initial_cpb_removal_delay_offset);
				end_item(d);
			}
			end_array(d, "vcl_hrd_cpb");
		}
		break;
	}
	case H264_SEI_TYPE_PIC_TIMING: {
		const struct h264_sei_pic_timing *s = &sei->pic_timing;
		const struct h264_sps *sps = ctx->sps;
		if (sps == NULL)
			return -EIO;
		if (sps->vui.nal_hrd_parameters_present_flag ||
		    sps->vui.vcl_hrd_parameters_present_flag) {
			F(s, cpb_removal_delay);
			F(s, dpb_output_delay);
		}
		if (sps->vui.pic_struct_present_flag) {
			F(s, pic_struct);
			begin_array(d);
			for (i = 0; i < num_clock_ts[s->pic_struct & 15] && i < 3;
			     i++) {
				uint32_t n;
				begin_item(d);
				F(&s->clk_ts[i], clock_timestamp_flag);
				if (!s->clk_ts[i].clock_timestamp_flag)
					goto next_ts;
				F(&s->clk_ts[i], ct_type);
				F(&s->clk_ts[i], nuit_field_based_flag);
				F(&s->clk_ts[i], counting_type);
				F(&s->clk_ts[i], full_timestamp_flag);
				F(&s->clk_ts[i], discontinuity_flag);
				F(&s->clk_ts[i], cnt_dropped_flag);
				F(&s->clk_ts[i], n_frames);
				if (s->clk_ts[i].full_timestamp_flag) {
					F(&s->clk_ts[i], seconds_value);
					F(&s->clk_ts[i], minutes_value);
					F(&s->clk_ts[i], hours_value);
				} else {
					F(&s->clk_ts[i], seconds_flag);
					if (!s->clk_ts[i].seconds_flag)
						goto time_offset;
					F(&s->clk_ts[i], seconds_value);
					F(&s->clk_ts[i], minutes_flag);
					if (!s->clk_ts[i].minutes_flag)
						goto time_offset;
					F(&s->clk_ts[i], minutes_value);
					F(&s->clk_ts[i], hours_flag);
					if (!s->clk_ts[i].hours_flag)
						goto time_offset;
					F(&s->clk_ts[i], hours_value);
				}
time_offset:
				n = sps->vui.nal_hrd_parameters_present_flag
					    ? sps->vui.nal_hrd.time_offset_length
				    : sps->vui.vcl_hrd_parameters_present_flag
					    ? sps->vui.vcl_hrd.time_offset_length
					    : 24;
				if (n > 0)
					F(&s->clk_ts[i], time_offset);
next_ts:
				end_item(d);
			}
			end_array(d, "clk_ts");
		}
		break;
	}
	case H264_SEI_TYPE_PAN_SCAN_RECT: {
		const struct h264_sei_pan_scan_rect *s = &sei->pan_scan_rect;
		F(s, pan_scan_rect_id);
		F(s, pan_scan_rect_cancel_flag);
		if (!s->pan_scan_rect_cancel_flag) {
			F(s, pan_scan_cnt_minus1);
			if (s->pan_scan_cnt_minus1 >=
			    H264_ARRAY_SIZE(s->pan_scan_rect))
				return -EIO;
			begin_array(d);
			for (i = 0; i <= s->pan_scan_cnt_minus1; i++) {
				begin_item(d);
				F(&s->pan_scan_rect[i], left_offset);
				F(&s->pan_scan_rect[i], right_offset);
				F(&s->pan_scan_rect[i], top_offset);
				F(&s->pan_scan_rect[i], bottom_offset);
				end_item(d);
			}
			end_array(d, "pan_scan_rect");
			F(s, pan_scan_rect_repetition_period);
		}
		break;
	}
	case H264_SEI_TYPE_FILLER_PAYLOAD:
		dump_sei_data(d, sei->filler_payload.buf, sei->filler_payload.len);
		break;
	case H264_SEI_TYPE_USER_DATA_REGISTERED: {
		const struct h264_sei_user_data_registered *s =
			&sei->user_data_registered;
		F(s, country_code);
		if (s->country_code == 0xff)
			F(s, country_code_extension_byte);
		dump_sei_data(d, s->buf, s->len);
		break;
	}
	case H264_SEI_TYPE_USER_DATA_UNREGISTERED: {
		const struct h264_sei_user_data_unregistered *s =
			&sei->user_data_unregistered;
		begin_array(d);
		for (i = 0; i < 16; i++)
			FV("uuid", s->uuid[i]);
		end_array(d, "uuid");
		dump_sei_data(d, s->buf, s->len);
		break;
	}
	case H264_SEI_TYPE_RECOVERY_POINT: {
		const struct h264_sei_recovery_point *s = &sei->recovery_point;
		F(s, recovery_frame_cnt);
		F(s, exact_match_flag);
		F(s, broken_link_flag);
		F(s, changing_slice_group_idc);
		break;
	}
	default:
		break; /* kept raw by the reader, nothing to show (ref :844-846) */
	}
	return 0;
}

/* ---- slice header (ref src/h264_syntax.h:1014-1387) ------------------------- */

static void dump_rplm_items(struct h264_dump *d, const struct h264_rplm_item *it)
{
	uint32_t i = 0, val;
	do {
		begin_item(d);
		val = it[i].modification_of_pic_nums_idc;
		F(&it[i], modification_of_pic_nums_idc);
		if (val == 0 || val == 1)
			F(&it[i], abs_diff_pic_num_minus1);
		else if (val == 2)
			F(&it[i], long_term_pic_num);
		else if (val == 4 || val == 5)
			F(&it[i], abs_diff_view_idx_minus1);
		end_item(d);
		i++;
	} while (val != 3 && i < 32);
}

static void dump_pwt_item(struct h264_dump *d, const struct h264_pwt_item *it,
			  uint32_t cat)
{
	F(it, luma_weight_flag);
	if (it->luma_weight_flag) {
		F(it, luma_weight);
		F(it, luma_offset);
	}
	if (cat != 0) {
		F(it, chroma_weight_flag);
		if (it->chroma_weight_flag) {
			begin_array(d);
			begin_item(d);
			FV("chroma_weight", it->chroma_weight[0]);
			FV("chroma_offset", it->chroma_offset[0]);
			end_item(d);
			begin_item(d);
			FV("chroma_weight", it->chroma_weight[1]);
			FV("chroma_offset", it->chroma_offset[1]);
			end_item(d);
			end_array(d, "chroma");
		}
	}
}

static int dump_slice_header(struct h264_dump *d, struct h264_ctx *ctx,
			     const struct h264_slice_header *sh)
{
	const struct h264_sps *sps;
	const struct h264_pps *pps;
	uint32_t type = sh->slice_type % 5u, i, cat;
	int idr = ctx->nalu.type == H264_NALU_TYPE_SLICE_IDR;

	F(sh, first_mb_in_slice);
	F(sh, slice_type);
	F(sh, pic_parameter_set_id);
	if (h264_ctx_set_active_pps(ctx, sh->pic_parameter_set_id) < 0)
		return -EINVAL;
	sps = ctx->sps;
	pps = ctx->pps;
	cat = sps->separate_colour_plane_flag ? 0u : sps->chroma_format_idc;
	if (sps->separate_colour_plane_flag)
		F(sh, colour_plane_id);
	F(sh, frame_num);
	if (!sps->frame_mbs_only_flag) {
		F(sh, field_pic_flag);
		if (sh->field_pic_flag)
			F(sh, bottom_field_flag);
	}
	if (idr)
		F(sh, idr_pic_id);
	if (sps->pic_order_cnt_type == 0) {
		F(sh, pic_order_cnt_lsb);
		if (pps->bottom_field_pic_order_in_frame_present_flag &&
		    !sh->field_pic_flag)
			F(sh, delta_pic_order_cnt_bottom);
	}
	if (sps->pic_order_cnt_type == 1 &&
	    !sps->delta_pic_order_always_zero_flag) {
		begin_array(d);
		FV("delta_pic_order_cnt", sh->delta_pic_order_cnt[0]);
		if (pps->bottom_field_pic_order_in_frame_present_flag &&
		    !sh->field_pic_flag)
			FV("delta_pic_order_cnt", sh->delta_pic_order_cnt[1]);
		end_array(d, "delta_pic_order_cnt");
	}
	if (pps->redundant_pic_cnt_present_flag)
		F(sh, redundant_pic_cnt);
	if (type == H264_SLICE_TYPE_B)
		F(sh, direct_spatial_mv_pred_flag);
	if (type == H264_SLICE_TYPE_P || type == H264_SLICE_TYPE_SP ||
	    type == H264_SLICE_TYPE_B) {
		F(sh, num_ref_idx_active_override_flag);
		if (sh->num_ref_idx_active_override_flag) {
			F(sh, num_ref_idx_l0_active_minus1);
			if (type == H264_SLICE_TYPE_B)
				F(sh, num_ref_idx_l1_active_minus1);
		}
	}

	begin_struct(d);
	if (type != H264_SLICE_TYPE_I && type != H264_SLICE_TYPE_SI) {
		F(&sh->rplm, ref_pic_list_modification_flag_l0);
		if (sh->rplm.ref_pic_list_modification_flag_l0) {
			begin_array(d);
			dump_rplm_items(d, sh->rplm.pic_num_l0);
			end_array(d, "pic_num_l0");
		}
	}
	if (type == H264_SLICE_TYPE_B) {
		F(&sh->rplm, ref_pic_list_modification_flag_l1);
		if (sh->rplm.ref_pic_list_modification_flag_l1) {
			begin_array(d);
			dump_rplm_items(d, sh->rplm.pic_num_l1);
			end_array(d, "pic_num_l1");
		}
	}
	end_struct(d, "rplm");

	if ((pps->weighted_pred_flag &&
	     (type == H264_SLICE_TYPE_P || type == H264_SLICE_TYPE_SP)) ||
	    (pps->weighted_bipred_idc == 1 && type == H264_SLICE_TYPE_B)) {
		begin_struct(d);
		F(&sh->pwt, luma_log2_weight_denom);
		if (cat != 0)
			F(&sh->pwt, chroma_log2_weight_denom);
		if (sh->num_ref_idx_l0_active_minus1 >= 32)
			return -EIO;
		begin_array(d);
		for (i = 0; i <= sh->num_ref_idx_l0_active_minus1; i++) {
			begin_item(d);
			dump_pwt_item(d, &sh->pwt.l0[i], cat);
			end_item(d);
		}
		end_array(d, "l0");
		if (type == H264_SLICE_TYPE_B) {
			if (sh->num_ref_idx_l1_active_minus1 >= 32)
				return -EIO;
			begin_array(d);
			for (i = 0; i <= sh->num_ref_idx_l1_active_minus1; i++) {
				begin_item(d);
				dump_pwt_item(d, &sh->pwt.l1[i], cat);
				end_item(d);
			}
			end_array(d, "l1");
		}
		end_struct(d, "pwt");
	}

	if (ctx->nalu.hdr.nal_ref_idc != 0) {
		begin_struct(d);
		if (idr) {
			F(&sh->drpm, no_output_of_prior_pics_flag);
			F(&sh->drpm, long_term_reference_flag);
		} else {
			F(&sh->drpm, adaptive_ref_pic_marking_mode_flag);
			if (sh->drpm.adaptive_ref_pic_marking_mode_flag) {
				uint32_t val;
				begin_array(d);
				i = 0;
				do {
					const struct h264_drpm_item *m =
						&sh->drpm.mm[i];
					begin_item(d);
					val = m->memory_management_control_operation;
					F(m, memory_management_control_operation);
					if (val == 1 || val == 3)
						F(m, difference_of_pic_nums_minus1);
					if (val == 2)
						F(m, long_term_pic_num);
					if (val == 3 || val == 6)
						F(m, long_term_frame_idx);
					if (val == 4)
						F(m, max_long_term_frame_idx_plus1);
					end_item(d);
					i++;
				} while (val != 0 && i < 64);
				end_array(d, "mm");
			}
		}
		end_struct(d, "drpm");
	}

	if (pps->entropy_coding_mode_flag && type != H264_SLICE_TYPE_I &&
	    type != H264_SLICE_TYPE_SI)
		F(sh, cabac_init_idc);
	F(sh, slice_qp_delta);
	if (type == H264_SLICE_TYPE_SP || type == H264_SLICE_TYPE_SI) {
		if (type == H264_SLICE_TYPE_SP)
			F(sh, sp_for_switch_flag);
		F(sh, slice_qs_delta);
	}
	if (pps->deblocking_filter_control_present_flag) {
		F(sh, disable_deblocking_filter_idc);
		if (sh->disable_deblocking_filter_idc != 1) {
			F(sh, slice_alpha_c0_offset_div2);
			F(sh, slice_beta_offset_div2);
		}
	}
	if (pps->num_slice_groups_minus1 > 0 && pps->slice_group_map_type >= 3 &&
	    pps->slice_group_map_type <= 5)
		F(sh, slice_group_change_cycle);
	return 0;
}

/* ---- slice data from the kernel's full-syntax records ------------------------
 * (ref src/h264_syntax_slice_data.h:103-862 in its DUMP instantiation) */

/* Table 7-13 / 7-14: prediction modes of the partitions of a P / B macroblock
 * type as coded; 0 L0, 1 L1, 2 Bi, 3 Direct, 4 none (sub-macroblock types) */
enum { PL0, PL1, PBI, PDIRECT, PSUB };

static void mb_parts(uint32_t slice_type, uint32_t raw, uint32_t *nparts,
		     uint32_t pm[2])
{
	static const uint8_t b_pm0[9] = {PL0, PL1, PL0, PL1, PL0, PL1, PBI, PBI, PBI};
	static const uint8_t b_pm1[9] = {PL0, PL1, PL1, PL0, PBI, PBI, PL0, PL1, PBI};
	*nparts = 1;
	pm[0] = pm[1] = PL0;
	if (slice_type == H264_SLICE_TYPE_P || slice_type == H264_SLICE_TYPE_SP) {
		*nparts = raw == 0 ? 1u : raw <= 2 ? 2u : 4u;
		if (*nparts == 4)
			pm[0] = pm[1] = PSUB;
		return;
	}
	if (raw == 0) {
		pm[0] = PDIRECT;
	} else if (raw <= 3) {
		pm[0] = raw - 1u;
	} else if (raw <= 21) {
		uint32_t k = (raw - 4u) >> 1;
		*nparts = 2;
		pm[0] = b_pm0[k];
		pm[1] = b_pm1[k];
	} else {
		*nparts = 4;
		pm[0] = pm[1] = PSUB;
	}
}

/* Table 7-17 / 7-18 */
static void sub_parts(int is_b, uint32_t st, uint32_t *n, uint32_t *pm)
{
	if (!is_b) {
		*pm = PL0;
		*n = st == 0 ? 1u : st == 3 ? 4u : 2u;
		return;
	}
	*pm = st == 0 ? (uint32_t)PDIRECT
	      : st <= 3 ? st - 1u
	      : st <= 9 ? (st - 4u) >> 1
			: st - 10u;
	*n = st == 0 ? 4u : st <= 3 ? 1u : st <= 9 ? 2u : 4u;
}

static void dump_block(struct h264_dump *d, const char *name, uint32_t blk,
		       const int16_t *lv, uint32_t n)
{
	uint32_t i;
	char key[32];
	for (i = 0; i < n; i++) {
		if (lv[i] == 0)
			continue;
		snprintf(key, sizeof(key), "%s(%u,%u)", name, blk, i);
		field(d, key, lv[i]);
	}
}

static void dump_residual(struct h264_dump *d, const uint8_t *rec, uint32_t cat,
			  int i16, uint32_t cbp)
{
	static const char *const n4[3] = {"Luma", "Cb", "Cr"};
	static const char *const ndc[3] = {"Luma16DC", "Cb16DC", "Cr16DC"};
	static const char *const nac[3] = {"Luma16AC", "Cb16AC", "Cr16AC"};
	const int16_t *co = (const int16_t *)(rec + H264_AMD_T2_HEAD);
	uint32_t c, b, ncomp = cat == 3 ? 3u : 1u;
	for (c = 0; c < ncomp; c++) {
		const int16_t *base = co + 256 * c;
		if (i16)
			dump_block(d, ndc[c], 0, base, 16);
		for (b = 0; b < 16; b++) {
			if (!(cbp & (1u << (b >> 2))))
				continue;
			if (i16)
				dump_block(d, nac[c], b, base + 16 + 15 * b, 15);
			else
				dump_block(d, n4[c], b, base + 16 * b, 16);
		}
	}
	if (cat == 1 || cat == 2) {
		uint32_t n = cat == 1 ? 4u : 8u, cbc = cbp >> 4;
		if (cbc & 3u) {
			dump_block(d, "CbDC", 0, co + 256, n);
			dump_block(d, "CrDC", 0, co + 256 + n, n);
		}
		if (cbc & 2u) {
			for (c = 0; c < 2; c++)
				for (b = 0; b < n; b++)
					dump_block(d, c ? "CrAC" : "CbAC", b,
						   co + 256 + 2 * n +
							   15 * (n * c + b),
						   15);
		}
	}
}

static void dump_mvd(struct h264_dump *d, const uint8_t *rec, uint32_t list,
		     uint32_t nparts, const uint32_t *use, const uint32_t *nsub)
{
	const int16_t *mv = (const int16_t *)(rec + 128 + 64 * list);
	uint32_t p, s;
	begin_array(d);
	for (p = 0; p < nparts; p++) {
		begin_array(d);
		if (use[p]) {
			for (s = 0; s < nsub[p]; s++) {
				begin_array(d);
				FV("mvd", mv[8 * p + 2 * s + 0]);
				FV("mvd", mv[8 * p + 2 * s + 1]);
				end_array(d, "mvd");
			}
		}
		end_array(d, "mvd");
	}
	end_array(d, list ? "mvd_l1" : "mvd_l0");
}

static void dump_macroblock(struct h264_dump *d, const struct h264_ctx *ctx,
			    const uint8_t *rec, uint32_t mb_addr, int field_flag)
{
	const struct h264_amd_mb_summary *s =
		(const struct h264_amd_mb_summary *)rec;
	const struct h264_slice_header *sh = &ctx->slice.hdr;
	const uint32_t cat = ctx->sps_derived.ChromaArrayType;
	const uint32_t stype = sh->slice_type % 5u;
	const int is_b = stype == H264_SLICE_TYPE_B;
	const int t8 = ctx->pps->transform_8x8_mode_flag;
	uint32_t i, p;

	begin_item(d);
	FV("mbAddr", mb_addr);
	FV("MbaffFrameFlag", ctx->derived.MbaffFrameFlag);
	/* read with the top macroblock of a pair, or with the bottom one behind
	 * a skipped top (ref src/h264_syntax_slice_data.h:755-761) */
	if (field_flag >= 0)
		FV("mb_field_decoding_flag", field_flag);
	FV("mb_addr", (uint64_t)sh->frame_num * 10000u + mb_addr);
	FV("mb_type", s->raw_mb_type);

	if (s->mb_type == H264_MB_TYPE_I_PCM) {
		const uint16_t *pc = (const uint16_t *)(rec + H264_AMD_T2_HEAD);
		uint32_t nc = ctx->sps_derived.MbWidthC * ctx->sps_derived.MbHeightC;
		begin_array(d);
		for (i = 0; i < 256; i++)
			FV("pcm_sample_luma", pc[i]);
		end_array(d, "pcm_sample_luma");
		begin_array(d);
		for (p = 0; p < 2; p++) {
			begin_array(d);
			for (i = 0; i < nc; i++)
				FV("pcm_sample_chroma", pc[256 + nc * p + i]);
			end_array(d, "pcm_sample_chroma");
		}
		end_array(d, "pcm_sample_chroma");
		end_item(d);
		return;
	}

	{
		const int intra_nxn = s->mb_type == H264_MB_TYPE_I_NxN ||
				      s->mb_type == H264_MB_TYPE_SI;
		const int i16 = s->mb_type == H264_MB_TYPE_I_16x16;
		const int intra = intra_nxn || i16;
		uint32_t nparts = 1, pm[2] = {PL0, PL0};
		if (!intra)
			mb_parts(stype, s->raw_mb_type, &nparts, pm);

		if (!intra && nparts == 4) {
			/* sub_mb_pred (:422-503) */
			uint32_t spm[4], sn[4], use[4];
			begin_array(d);
			for (p = 0; p < 4; p++) {
				FV("sub_mb_type", s->pred[p]);
				sub_parts(is_b, s->pred[p], &sn[p], &spm[p]);
			}
			end_array(d, "sub_mb_type");
			if (sh->num_ref_idx_l0_active_minus1 > 0 &&
			    s->mb_type != H264_MB_TYPE_P_8x8ref0) {
				begin_array(d);
				for (p = 0; p < 4; p++)
					FV("ref_idx_l0",
					   spm[p] != PDIRECT && spm[p] != PL1
						   ? rec[64 + p]
						   : 0);
				end_array(d, "ref_idx_l0");
			}
			if (sh->num_ref_idx_l1_active_minus1 > 0) {
				begin_array(d);
				for (p = 0; p < 4; p++)
					FV("ref_idx_l1",
					   spm[p] != PDIRECT && spm[p] != PL0
						   ? rec[68 + p]
						   : 0);
				end_array(d, "ref_idx_l1");
			}
			for (p = 0; p < 4; p++)
				use[p] = spm[p] != PDIRECT && spm[p] != PL1;
			dump_mvd(d, rec, 0, 4, use, sn);
			for (p = 0; p < 4; p++)
				use[p] = spm[p] != PDIRECT && spm[p] != PL0;
			dump_mvd(d, rec, 1, 4, use, sn);
		} else {
			if (t8 && s->mb_type == H264_MB_TYPE_I_NxN)
				FV("transform_size_8x8_flag",
				   !!(s->flags & H264_AMD_MB_TRANSFORM_8X8));
			/* mb_pred (:506-601) */
			if (intra) {
				if (!i16) {
					int i8 = (s->flags &
						  H264_AMD_MB_TRANSFORM_8X8) != 0;
					uint32_t n = i8 ? 4u : 16u;
					begin_array(d);
					for (i = 0; i < n; i++) {
						uint32_t nib =
							(s->pred[i >> 1] >>
							 (4 * (i & 1))) & 15u;
						FV("pred_mode", (int64_t)nib - 1);
					}
					end_array(d, i8 ? "intra8x8_pred_mode"
							: "intra4x4_pred_mode");
				}
				if (cat == 1 || cat == 2)
					FV("intra_chroma_pred_mode",
					   s->intra_chroma_pred_mode);
			} else if (pm[0] != PDIRECT) {
				uint32_t use[4] = {0, 0, 0, 0};
				uint32_t one[4] = {1, 1, 1, 1};
				if (sh->num_ref_idx_l0_active_minus1 > 0) {
					begin_array(d);
					for (p = 0; p < nparts; p++)
						FV("ref_idx_l0",
						   pm[p ? 1 : 0] != PL1
							   ? rec[64 + p]
							   : 0);
					end_array(d, "ref_idx_l0");
				}
				if (sh->num_ref_idx_l1_active_minus1 > 0) {
					begin_array(d);
					for (p = 0; p < nparts; p++)
						FV("ref_idx_l1",
						   pm[p ? 1 : 0] != PL0
							   ? rec[68 + p]
							   : 0);
					end_array(d, "ref_idx_l1");
				}
				for (p = 0; p < nparts; p++)
					use[p] = pm[p ? 1 : 0] != PL1;
				dump_mvd(d, rec, 0, nparts, use, one);
				for (p = 0; p < nparts; p++)
					use[p] = pm[p ? 1 : 0] != PL0;
				dump_mvd(d, rec, 1, nparts, use, one);
			}
		}
		if (!i16) {
			FV("coded_block_pattern", s->cbp);
			/* the flag is coded behind the pattern for everything but
			 * I_NxN (:675-682); it can only be set when it was coded */
			if (t8 && s->mb_type != H264_MB_TYPE_I_NxN &&
			    (s->cbp & 15u) &&
			    (s->flags & H264_AMD_MB_TRANSFORM_8X8))
				FV("transform_size_8x8_flag", 1);
		}
		if ((s->cbp & 15u) || (s->cbp >> 4) || i16) {
			FV("mb_qp_delta", s->mb_qp_delta);
			begin_struct(d);
			dump_residual(d, rec, cat, i16, s->cbp);
			end_struct(d, "residual");
		}
	}
	end_item(d);
}

static int dump_slice_data(struct h264_dump *d, struct h264_ctx *ctx)
{
	struct h264_amd_engine *eng;
	struct h264_amd_params *params;
	struct h264_amd_decode_args a;
	struct h264_amd_slice_desc desc;
	struct h264_amd_slice_result r;
	struct h264_bitstream bs;
	void *d_nal = NULL, *d_desc = NULL, *d_res = NULL, *d_hdr = NULL,
	     *d_mb = NULL;
	uint8_t *recs = NULL, *group_map = NULL;
	size_t cap[5] = {0, 0, 0, 0, 0};
	uint32_t epoch = 0, stride, mbcap, m, run = 0, cur;
	const uint32_t pic_mbs = ctx->derived.PicSizeInMbs;
	int res, had_run = 0;

	if (ctx->pps == NULL || ctx->pps->entropy_coding_mode_flag)
		return 0; /* CABAC not supported for parsing (ref :715-717) */
	res = h264_host_engine(&eng);
	if (res < 0)
		return res;
	/* the slice NAL unit again: header bits, the partial byte, raw data */
	h264_bs_init(&bs, NULL, 0, 1);
	bs.dynamic = 1;
	res = h264_write_nalu(&bs, ctx);
	if (res < 0 || bs.off > UINT32_MAX) {
		h264_bs_clear(&bs);
		return res < 0 ? res : -E2BIG;
	}
	res = h264_host_ctx_params(ctx, &params, &epoch);
	stride = h264_amd_t2_stride(3);
	mbcap = h264_amd_params_max_mbs(params);
	if (mbcap == 0)
		mbcap = 1;
	h264_host_lock();
	for (;;) {
		if (res == 0)
			res = h264_host_grow(&d_nal, &cap[0], bs.off + 64);
		if (res == 0)
			res = h264_amd_upload(eng, d_nal, bs.data, bs.off);
		desc.offset = 0;
		desc.size = (uint32_t)bs.off;
		desc.epoch = epoch;
		if (res == 0)
			res = h264_host_grow(&d_desc, &cap[1], sizeof(desc));
		if (res == 0)
			res = h264_amd_upload(eng, d_desc, &desc, sizeof(desc));
		if (res == 0)
			res = h264_host_grow(&d_res, &cap[2], sizeof(r));
		if (res == 0)
			res = h264_host_grow(&d_hdr, &cap[3],
					     sizeof(struct h264_slice_header));
		if (res == 0)
			res = h264_host_grow(&d_mb, &cap[4],
					     (size_t)mbcap * stride);
		if (res < 0)
			break;
		memset(&a, 0, sizeof(a));
		a.d_buf = d_nal;
		a.d_descs = d_desc;
		a.n = 1;
		a.reader_flags = H264_READER_FLAGS_SLICE_DATA;
		a.tier = H264_AMD_TIER_T2;
		a.mb_capacity = mbcap;
		a.d_results = d_res;
		a.d_headers = d_hdr;
		a.d_mb_out = d_mb;
		a.mb_stride = stride;
		a.d_buf_len = bs.off + 64;
		res = h264_amd_decode_slices(eng, params, &a);
		if (res == 0)
			res = h264_amd_download(eng, &r, d_res, sizeof(r));
		if (res < 0 || r.status != H264_AMD_E_MB_OVERFLOW ||
		    mbcap >= H264_AMD_MAX_SLICE_MBS)
			break;
		mbcap = mbcap > H264_AMD_MAX_SLICE_MBS / 8 ? H264_AMD_MAX_SLICE_MBS
							   : mbcap * 8;
	}
	if (res == 0 && r.mb_count > 0) {
		recs = malloc((size_t)r.mb_count * stride);
		res = recs == NULL ? -ENOMEM
				   : h264_amd_download(eng, recs, d_mb,
						       (size_t)r.mb_count * stride);
	}
	if (res == 0 && ctx->pps->num_slice_groups_minus1 > 0) {
		/* the map the kernel walked (ref src/h264_fmo.c:307-319) */
		group_map = malloc(pic_mbs ? pic_mbs : 1);
		res = group_map == NULL
			      ? -ENOMEM
			      : h264_amd_slice_group_map(eng, 0, group_map, pic_mbs);
	}
	h264_amd_free(eng, d_nal);
	h264_amd_free(eng, d_desc);
	h264_amd_free(eng, d_res);
	h264_amd_free(eng, d_hdr);
	h264_amd_free(eng, d_mb);
	h264_host_unlock();
	h264_bs_clear(&bs);
	if (res < 0) {
		free(recs);
		free(group_map);
		return res;
	}

	cur = r.first_mb_in_slice * (1u + (uint32_t)ctx->derived.MbaffFrameFlag);
	begin_struct(d);
	begin_array(d);
	/* skipped macroblocks show as their run, coded ones in full; the
	 * records are in slice order */
	for (m = 0; m <= r.mb_count; m++) {
		const uint8_t *rec = m < r.mb_count ? recs + (size_t)m * stride
						    : NULL;
		int skipped = rec != NULL &&
			      (((const struct h264_amd_mb_summary *)rec)->flags &
			       H264_AMD_MB_SKIPPED);
		int inter_slice = ctx->slice.hdr.slice_type % 5u !=
					  H264_SLICE_TYPE_I &&
				  ctx->slice.hdr.slice_type % 5u !=
					  H264_SLICE_TYPE_SI;
		uint32_t addr = cur;
		int field_flag = -1;
		if (group_map != NULL && cur < pic_mbs) {
			uint32_t i = cur + 1;
			while (i < pic_mbs && group_map[i] != group_map[cur])
				i++;
			cur = i;
		} else {
			cur++;
		}
		if (skipped) {
			run++;
			continue;
		}
		/* a run is coded in front of every coded macroblock of a P / B
		 * slice (0 when none is skipped), and a last one may end the
		 * slice */
		had_run = run > 0;
		if (inter_slice && (rec != NULL || run > 0)) {
			begin_item(d);
			FV("mb_skip_run", run);
			end_item(d);
			run = 0;
		}
		if (rec != NULL && ctx->derived.MbaffFrameFlag &&
		    (addr % 2 == 0 || (inter_slice && had_run)))
			field_flag = (((const struct h264_amd_mb_summary *)rec)->flags &
				      H264_AMD_MB_FIELD) != 0;
		if (rec != NULL)
			dump_macroblock(d, ctx, rec, addr, field_flag);
	}
	end_array(d, "mb");
	end_struct(d, "slice_data");
	free(recs);
	free(group_map);
	return r.status < 0 ? r.status : 0;
}

/* ---- the public entry points (ref src/h264_dump.c:214-316) ------------------ */

static void tree_clear(struct h264_dump *d)
{
	uint32_t i;
	/* open containers are only linked from the stack */
	for (i = 0; i < d->depth; i++)
		jfree(d->stack[i]);
	d->depth = 0;
	d->err = 0;
	d->text_len = 0;
}

int h264_dump_new(const struct h264_dump_cfg *cfg, struct h264_dump **ret_obj)
{
	struct h264_dump *d;
	if (ret_obj == NULL)
		return -EINVAL;
	*ret_obj = NULL;
	if (cfg == NULL)
		return -EINVAL;
	d = calloc(1, sizeof(*d));
	if (d == NULL)
		return -ENOMEM;
	d->cfg = *cfg;
	d->stack[0] = jnew(J_OBJ, 0);
	if (d->stack[0] == NULL) {
		free(d);
		return -ENOMEM;
	}
	d->depth = 1;
	*ret_obj = d;
	return 0;
}

int h264_dump_destroy(struct h264_dump *dump)
{
	if (dump == NULL)
		return 0;
	tree_clear(dump);
	free(dump->text);
	free(dump);
	return 0;
}

int h264_dump_clear(struct h264_dump *dump)
{
	if (dump == NULL)
		return -EINVAL;
	tree_clear(dump);
	dump->stack[0] = jnew(J_OBJ, 0);
	if (dump->stack[0] == NULL)
		return -ENOMEM;
	dump->depth = 1;
	return 0;
}

int h264_dump_get_json_object(struct h264_dump *dump, struct json_object **jobj)
{
	if (dump == NULL || jobj == NULL)
		return -EINVAL;
	*jobj = NULL;
	return -ENOSYS; /* no json-c in this build: h264_dump_get_json_str */
}

int h264_dump_get_json_str(struct h264_dump *dump, const char **str)
{
	if (dump == NULL || str == NULL)
		return -EINVAL;
	if (dump->cfg.type != H264_DUMP_TYPE_JSON || dump->depth == 0)
		return -EINVAL;
	dump->text_len = 0;
	/* jcurrent: the innermost open container (the root after a complete
	 * h264_dump_nalu) */
	render(dump, dump->stack[dump->depth - 1]);
	if (dump->err < 0)
		return dump->err;
	*str = dump->text;
	return 0;
}

int h264_dump_nalu(struct h264_dump *dump, struct h264_ctx *ctx, uint32_t flags)
{
	struct h264_dump *d = dump;
	int res;
	if (dump == NULL || ctx == NULL)
		return -EINVAL;
	dump->flags = flags;
	res = h264_dump_clear(dump);
	if (res < 0)
		return res;

	begin_struct(d);
	F(&ctx->nalu.hdr, forbidden_zero_bit);
	if (ctx->nalu.hdr.forbidden_zero_bit != 0) {
		end_struct(d, "nalu_header");
		return -EIO;
	}
	F(&ctx->nalu.hdr, nal_ref_idc);
	F(&ctx->nalu.hdr, nal_unit_type);
	end_struct(d, "nalu_header");
	ctx->nalu.type = ctx->nalu.hdr.nal_unit_type;

	res = 0;
	switch (ctx->nalu.type) {
	case H264_NALU_TYPE_SLICE:
	case H264_NALU_TYPE_SLICE_IDR:
		begin_struct(d);
		begin_struct(d);
		res = dump_slice_header(d, ctx, &ctx->slice.hdr);
		end_struct(d, "slice_header");
		if (res == 0 && (flags & H264_DUMP_FLAGS_SLICE_DATA) &&
		    (ctx->slice.rawdata.partialbits != 0 ||
		     ctx->slice.rawdata.len != 0)) {
			if (ctx->slice.rawdata.len != 0 &&
			    ctx->slice.rawdata.buf == NULL)
				res = -EIO;
			else
				res = dump_slice_data(d, ctx);
		}
		end_struct(d, "slice");
		break;
	case H264_NALU_TYPE_SEI: {
		uint32_t i;
		begin_array(d);
		if (ctx->sei_count == 0)
			res = -EIO;
		for (i = 0; res == 0 && i < ctx->sei_count; i++) {
			const struct h264_sei *sei = &ctx->sei_table[i];
			begin_item(d);
			FV("payload_type", sei->type);
			FV("payload_size", sei->raw.len);
			res = dump_one_sei(d, ctx, sei);
			end_item(d);
		}
		end_array(d, "sei");
		break;
	}
	case H264_NALU_TYPE_SPS:
		if (ctx->sps == NULL || ctx->nalu.hdr.nal_ref_idc == 0)
			return -EIO;
		begin_struct(d);
		dump_sps(d, ctx->sps);
		end_struct(d, "sps");
		break;
	case H264_NALU_TYPE_PPS:
		if (ctx->pps == NULL || ctx->nalu.hdr.nal_ref_idc == 0)
			return -EIO;
		begin_struct(d);
		res = dump_pps(d, ctx, ctx->pps);
		end_struct(d, "pps");
		break;
	case H264_NALU_TYPE_AUD:
		if (ctx->nalu.hdr.nal_ref_idc != 0)
			return -EIO;
		begin_struct(d);
		F(&ctx->aud, primary_pic_type);
		end_struct(d, "aud");
		break;
	case H264_NALU_TYPE_FILLER:
		if (ctx->nalu.hdr.nal_ref_idc != 0)
			return -EIO;
		begin_struct(d);
		end_struct(d, "filler");
		break;
	default:
		ctx->nalu.unknown = 1;
		break;
	}
	if (res == 0 && d->err < 0)
		res = d->err;
	return res;
}

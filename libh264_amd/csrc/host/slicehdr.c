/**
 * slicehdr.c -- slice HEADER parse on the host, for calls that never look at
 * slice data.
 *
 * h264_reader_parse_nalu(flags = 0) on one slice NAL unit is the pattern an
 * application uses per frame; a header is 4 - 40 bytes, and
 * sending it through an upload, a kernel launch and a download costs far more
 * than reading it here.  Slices of a CABAC picture take the same way whatever
 * the flags say: the reference does not parse their data either
 * (src/h264_syntax_slice_data.h:715-717).
 *
 * This is the host-side twin of sd_slice_header() in kernels/slice_decode.hip
 * (the walk of ref src/h264_syntax.h:1014-1387 incl. rplm / pwt / drpm) and
 * fills the very records the kernel would: a struct h264_amd_slice_result and
 * the struct h264_slice_header image, which reader.c then replays exactly like
 * a kernel result.  The batched entry points (h264_reader_parse, SLICE_DATA)
 * keep going through the GPU; tests/test_gpu_dropin.py holds both ways against
 * the oracle's event log and against each other.
 */
#include "h264_priv.h"

#define RD(call)                                                               \
	do {                                                                   \
		if ((call) < 0)                                                \
			return -EIO;                                           \
	} while (0)

static int hdr_u(struct h264_bitstream *bs, uint32_t n, uint32_t *v)
{
	/* any width, as the reference's reader behaves under a damaged SPS
	 * (include/h264/h264_bitstream.h:192-218): 0 bits read nothing, more
	 * than 32 keep the last 32 */
	*v = 0;
	if (n == 0)
		return 0;
	return h264_bs_read_bits(bs, v, n);
}

static int rplm_items(struct h264_bitstream *bs, struct h264_rplm_item *it)
{
	uint32_t i = 0, val;
	do {
		if (i >= 32)
			return -EIO;
		RD(h264_bs_read_bits_ue(bs, &val));
		it[i].modification_of_pic_nums_idc = val;
		if (val == 0 || val == 1 || val == 2 || val == 4 || val == 5)
			RD(h264_bs_read_bits_ue(bs, &it[i].abs_diff_pic_num_minus1));
		i++;
	} while (val != 3);
	return 0;
}

static int pwt_list(struct h264_bitstream *bs, struct h264_pwt_item *l,
		    uint32_t count, uint32_t cat)
{
	uint32_t i, f;
	for (i = 0; i < count; i++) {
		RD(h264_bs_read_bits(bs, &f, 1));
		l[i].luma_weight_flag = (int)f;
		if (f) {
			RD(h264_bs_read_bits_se(bs, &l[i].luma_weight));
			RD(h264_bs_read_bits_se(bs, &l[i].luma_offset));
		}
		if (cat != 0) {
			RD(h264_bs_read_bits(bs, &f, 1));
			l[i].chroma_weight_flag = (int)f;
			if (f) {
				RD(h264_bs_read_bits_se(bs, &l[i].chroma_weight[0]));
				RD(h264_bs_read_bits_se(bs, &l[i].chroma_offset[0]));
				RD(h264_bs_read_bits_se(bs, &l[i].chroma_weight[1]));
				RD(h264_bs_read_bits_se(bs, &l[i].chroma_offset[1]));
			}
		}
	}
	return 0;
}

static int header_body(struct h264_bitstream *bs, const struct h264_ctx *ctx,
		       uint32_t nal_ref_idc, uint32_t nal_unit_type,
		       struct h264_slice_header *sh,
		       struct h264_amd_slice_result *r)
{
	const struct h264_sps *sps;
	const struct h264_pps *pps;
	uint32_t v, type, nref0, nref1, cat;
	int field_pic = 0;

	RD(h264_bs_read_bits_ue(bs, &sh->first_mb_in_slice));
	RD(h264_bs_read_bits_ue(bs, &sh->slice_type));
	type = sh->slice_type % 5u;
	RD(h264_bs_read_bits_ue(bs, &sh->pic_parameter_set_id));
	/* h264_ctx_set_active_pps -> set_active_sps (ref src/h264_ctx.c:366-386) */
	if (sh->pic_parameter_set_id >= 256)
		return -EINVAL;
	pps = ctx->pps_table[sh->pic_parameter_set_id];
	if (pps == NULL || pps->seq_parameter_set_id >= 32)
		return -EINVAL;
	sps = ctx->sps_table[pps->seq_parameter_set_id];
	if (sps == NULL)
		return -EINVAL;
	cat = sps->separate_colour_plane_flag ? 0u : sps->chroma_format_idc;
	nref0 = pps->num_ref_idx_l0_default_active_minus1;
	nref1 = pps->num_ref_idx_l1_default_active_minus1;

	if (sps->separate_colour_plane_flag)
		RD(h264_bs_read_bits(bs, &sh->colour_plane_id, 2));
	RD(hdr_u(bs, sps->log2_max_frame_num_minus4 + 4u, &sh->frame_num));
	if (!sps->frame_mbs_only_flag) {
		RD(h264_bs_read_bits(bs, &v, 1));
		sh->field_pic_flag = field_pic = (int)v;
		if (field_pic) {
			RD(h264_bs_read_bits(bs, &v, 1));
			sh->bottom_field_flag = (int)v;
		}
	}
	if (nal_unit_type == H264_NALU_TYPE_SLICE_IDR)
		RD(h264_bs_read_bits_ue(bs, &sh->idr_pic_id));
	if (sps->pic_order_cnt_type == 0) {
		RD(hdr_u(bs, sps->log2_max_pic_order_cnt_lsb_minus4 + 4u,
			 &sh->pic_order_cnt_lsb));
		if (pps->bottom_field_pic_order_in_frame_present_flag && !field_pic)
			RD(h264_bs_read_bits_se(bs, &sh->delta_pic_order_cnt_bottom));
	}
	if (sps->pic_order_cnt_type == 1 &&
	    !sps->delta_pic_order_always_zero_flag) {
		RD(h264_bs_read_bits_se(bs, &sh->delta_pic_order_cnt[0]));
		if (pps->bottom_field_pic_order_in_frame_present_flag && !field_pic)
			RD(h264_bs_read_bits_se(bs, &sh->delta_pic_order_cnt[1]));
	}
	if (pps->redundant_pic_cnt_present_flag)
		RD(h264_bs_read_bits_ue(bs, &sh->redundant_pic_cnt));
	if (type == H264_SLICE_TYPE_B) {
		RD(h264_bs_read_bits(bs, &v, 1));
		sh->direct_spatial_mv_pred_flag = (int)v;
	}
	if (type == H264_SLICE_TYPE_P || type == H264_SLICE_TYPE_SP ||
	    type == H264_SLICE_TYPE_B) {
		RD(h264_bs_read_bits(bs, &v, 1));
		sh->num_ref_idx_active_override_flag = (int)v;
		if (v) {
			RD(h264_bs_read_bits_ue(bs, &nref0));
			if (type == H264_SLICE_TYPE_B)
				RD(h264_bs_read_bits_ue(bs, &nref1));
		}
	}
	sh->num_ref_idx_l0_active_minus1 = nref0;
	sh->num_ref_idx_l1_active_minus1 = nref1;

	/* ref_pic_list_modification */
	if (type != H264_SLICE_TYPE_I && type != H264_SLICE_TYPE_SI) {
		RD(h264_bs_read_bits(bs, &v, 1));
		sh->rplm.ref_pic_list_modification_flag_l0 = (int)v;
		if (v) {
			int res = rplm_items(bs, sh->rplm.pic_num_l0);
			if (res < 0)
				return res;
		}
	}
	if (type == H264_SLICE_TYPE_B) {
		RD(h264_bs_read_bits(bs, &v, 1));
		sh->rplm.ref_pic_list_modification_flag_l1 = (int)v;
		if (v) {
			int res = rplm_items(bs, sh->rplm.pic_num_l1);
			if (res < 0)
				return res;
		}
	}
	/* pred_weight_table */
	if ((pps->weighted_pred_flag &&
	     (type == H264_SLICE_TYPE_P || type == H264_SLICE_TYPE_SP)) ||
	    (pps->weighted_bipred_idc == 1 && type == H264_SLICE_TYPE_B)) {
		int res;
		RD(h264_bs_read_bits_ue(bs, &sh->pwt.luma_log2_weight_denom));
		if (cat != 0)
			RD(h264_bs_read_bits_ue(bs,
						&sh->pwt.chroma_log2_weight_denom));
		if (nref0 > 32)
			return -EIO;
		res = pwt_list(bs, sh->pwt.l0, nref0 + 1, cat);
		if (res < 0)
			return res;
		if (type == H264_SLICE_TYPE_B) {
			if (nref1 > 32)
				return -EIO;
			res = pwt_list(bs, sh->pwt.l1, nref1 + 1, cat);
			if (res < 0)
				return res;
		}
	}
	/* dec_ref_pic_marking */
	if (nal_ref_idc != 0) {
		if (nal_unit_type == H264_NALU_TYPE_SLICE_IDR) {
			RD(h264_bs_read_bits(bs, &v, 1));
			sh->drpm.no_output_of_prior_pics_flag = (int)v;
			RD(h264_bs_read_bits(bs, &v, 1));
			sh->drpm.long_term_reference_flag = (int)v;
		} else {
			RD(h264_bs_read_bits(bs, &v, 1));
			sh->drpm.adaptive_ref_pic_marking_mode_flag = (int)v;
			if (v) {
				uint32_t i = 0, op;
				do {
					struct h264_drpm_item *m;
					if (i >= 64)
						return -EIO;
					m = &sh->drpm.mm[i];
					RD(h264_bs_read_bits_ue(bs, &op));
					m->memory_management_control_operation = op;
					if (op == 1 || op == 3)
						RD(h264_bs_read_bits_ue(
							bs,
							&m->difference_of_pic_nums_minus1));
					if (op == 2)
						RD(h264_bs_read_bits_ue(
							bs, &m->long_term_pic_num));
					if (op == 3 || op == 6)
						RD(h264_bs_read_bits_ue(
							bs, &m->long_term_frame_idx));
					if (op == 4)
						RD(h264_bs_read_bits_ue(
							bs,
							&m->max_long_term_frame_idx_plus1));
					i++;
				} while (op != 0);
			}
		}
	}
	if (pps->entropy_coding_mode_flag && type != H264_SLICE_TYPE_I &&
	    type != H264_SLICE_TYPE_SI)
		RD(h264_bs_read_bits_ue(bs, &sh->cabac_init_idc));
	RD(h264_bs_read_bits_se(bs, &sh->slice_qp_delta));
	if (type == H264_SLICE_TYPE_SP || type == H264_SLICE_TYPE_SI) {
		if (type == H264_SLICE_TYPE_SP) {
			RD(h264_bs_read_bits(bs, &v, 1));
			sh->sp_for_switch_flag = (int)v;
		}
		RD(h264_bs_read_bits_se(bs, &sh->slice_qs_delta));
	}
	if (pps->deblocking_filter_control_present_flag) {
		RD(h264_bs_read_bits_ue(bs, &sh->disable_deblocking_filter_idc));
		if (sh->disable_deblocking_filter_idc != 1) {
			RD(h264_bs_read_bits_se(bs, &sh->slice_alpha_c0_offset_div2));
			RD(h264_bs_read_bits_se(bs, &sh->slice_beta_offset_div2));
		}
	}
	if (pps->num_slice_groups_minus1 > 0 && pps->slice_group_map_type >= 3 &&
	    pps->slice_group_map_type <= 5) {
		uint32_t units = (sps->pic_width_in_mbs_minus1 + 1u) *
				 (sps->pic_height_in_map_units_minus1 + 1u);
		uint32_t n = h264_intlog2(
			units / (pps->slice_group_change_rate_minus1 + 1u) + 1u);
		RD(hdr_u(bs, n, &sh->slice_group_change_cycle));
	}

	r->slice_type = (int32_t)type;
	r->first_mb_in_slice = sh->first_mb_in_slice;
	/* the reference sums in int; unsigned here, same bits */
	r->slice_qp = (int32_t)((uint32_t)pps->pic_init_qp_minus26 + 26u +
				(uint32_t)sh->slice_qp_delta);
	r->chroma_array_type = (uint16_t)cat;
	r->pic_width_in_mbs = sps->pic_width_in_mbs_minus1 + 1u;
	r->flags = H264_AMD_SR_HEADER_OK |
		   (pps->entropy_coding_mode_flag ? H264_AMD_SR_CABAC : 0u);
	return 0;
}

/* 1 when the host may answer for this slice NAL unit: header-only request, or
 * a slice whose PPS says CABAC (the reference stops behind the header there).
 * The PPS is looked up the way the header parse will; anything doubtful goes
 * to the device path, which answers every case. */
int h264_host_slice_is_header_only(const struct h264_ctx *ctx, uint32_t flags,
				   const uint8_t *nal, size_t len)
{
	struct h264_bitstream bs;
	uint32_t v[3];
	const struct h264_pps *pps;
	if (!(flags & H264_READER_FLAGS_SLICE_DATA))
		return 1;
	if (len < 2)
		return 0;
	h264_bs_cinit(&bs, nal + 1, len - 1, 1);
	if (h264_bs_read_bits_ue(&bs, &v[0]) < 0 ||
	    h264_bs_read_bits_ue(&bs, &v[1]) < 0 ||
	    h264_bs_read_bits_ue(&bs, &v[2]) < 0 || v[2] >= 256)
		return 0;
	pps = ctx->pps_table[v[2]];
	return pps != NULL && pps->entropy_coding_mode_flag != 0;
}

int h264_host_slice_header(const struct h264_ctx *ctx, const uint8_t *nal,
			   size_t len, struct h264_amd_slice_result *r,
			   struct h264_slice_header *sh)
{
	struct h264_bitstream bs;
	uint32_t h = 0;
	int res;

	memset(r, 0, sizeof(*r));
	memset(sh, 0, sizeof(*sh));
	r->slice_type = -1;
	if (len == 0 || len > UINT32_MAX) {
		r->status = -EIO;
		return 0;
	}
	h264_bs_cinit(&bs, nal, len, 1);
	if (h264_bs_read_bits(&bs, &h, 8) < 0 || (h & 0x80u)) {
		r->status = -EIO;
		r->nal_ref_idc = (h >> 5) & 3u;
		r->nal_unit_type = h & 31u;
		return 0;
	}
	r->nal_ref_idc = (h >> 5) & 3u;
	r->nal_unit_type = h & 31u;
	res = header_body(&bs, ctx, r->nal_ref_idc, r->nal_unit_type, sh, r);
	if (res < 0) {
		r->status = res;
		r->flags = 0;
		r->slice_type = -1;
		r->first_mb_in_slice = 0;
		r->slice_qp = 0;
		r->chroma_array_type = 0;
		r->pic_width_in_mbs = 0;
		return 0;
	}
	/* ctx->slice.hdr_len and rawdata (ref src/h264_syntax.h:1383,
	 * src/h264_syntax_slice_data.h:803-806): raw bytes, escapes included */
	r->hdr_len_bits = (uint32_t)(bs.off * 8u - bs.cachebits);
	r->rawdata_off = (uint32_t)bs.off;
	r->rawdata_len = (uint32_t)(len - bs.off);
	r->partial = bs.cache;
	r->partialbits = bs.cachebits;
	return 0;
}

/**
 * writer.c -- h264_write_nalu and friends, drop-in for the reference's
 * src/h264_writer.c:30-370 (WRITE instantiation of src/h264_syntax.h).
 *
 * Slice NAL units -- header re-encoding, raw slice-data copy, the synthetic
 * grey-I / skipped-P slices in CAVLC and CABAC -- are serialised by the MI355X
 * kernels (h264_amd_write_slices / h264_amd_encode_synth_slices) as a batch of
 * one; applications with many slices call those entry points directly with the
 * whole batch.  Parameter sets, SEI, AUD and filler are a few bytes of
 * control-plane syntax and are written here on the host.  There is no CPU
 * slice writer: without a GPU the slice entry points return -ENODEV.
 */
#include "h264_priv.h"

#define WR_HDR_ROOM 8192u /* generous bound for NAL + slice header bytes */

/* device scratch of the writer entry points, guarded by h264_host_lock() */
static struct {
	void *src, *hdr, *res, *desc, *out, *meta;
	size_t cap_src, cap_hdr, cap_res, cap_desc, cap_out, cap_meta;
	uint8_t *h_out;
	size_t hcap_out;
} g_wr;

/* result of one batch-of-one launch */
struct wr_out {
	const uint8_t *bytes; /* whole bytes */
	size_t len;
	uint32_t hdr_bits; /* header end, bits from the NAL's first byte */
	uint32_t pend_bits; /* 0..7 bits waiting behind `len` bytes ... */
	uint8_t pend; /* ... left aligned in this byte */
};

static uint32_t prev_zeros(const struct h264_bitstream *bs)
{
	uint32_t z = 0;
	if (bs->off >= 1 && bs->data[bs->off - 1] == 0) {
		z = 1;
		if (bs->off >= 2 && bs->data[bs->off - 2] == 0)
			z = 2;
	}
	return z;
}

static uint32_t wflags_of(const struct h264_bitstream *bs)
{
	if (!bs->emulation_prevention)
		return H264_AMD_W_NO_ESCAPE;
	return H264_AMD_W_PREV_ZEROS(prev_zeros(bs));
}

/* meta block on the device: [0] out offset (u64), [2] size, [3] hdr bits */
static int wr_collect(struct h264_amd_engine *eng, size_t out_cap,
		      struct wr_out *o)
{
	uint32_t meta[4];
	int32_t size;
	int res = h264_amd_download(eng, meta, g_wr.meta, sizeof(meta));
	if (res < 0)
		return res;
	size = (int32_t)meta[2];
	if (size < 0)
		return size;
	if ((size_t)size + 1 > out_cap)
		return -EIO;
	if (g_wr.hcap_out < out_cap) {
		free(g_wr.h_out);
		g_wr.h_out = malloc(out_cap);
		g_wr.hcap_out = g_wr.h_out != NULL ? out_cap : 0;
		if (g_wr.h_out == NULL)
			return -ENOMEM;
	}
	o->hdr_bits = meta[3];
	o->len = (size_t)size;
	/* one byte more: the pending bits of a header-only NAL */
	res = h264_amd_download(eng, g_wr.h_out, g_wr.out, o->len + 1);
	if (res < 0)
		return res;
	o->bytes = g_wr.h_out;
	o->pend_bits = 0;
	o->pend = 0;
	return 0;
}

static int wr_prepare(struct h264_amd_engine *eng, size_t out_cap)
{
	uint64_t zero[2] = {0, 0};
	int res = h264_host_grow(&g_wr.out, &g_wr.cap_out, out_cap + 64);
	if (res == 0)
		res = h264_host_grow(&g_wr.meta, &g_wr.cap_meta, sizeof(zero));
	if (res == 0)
		res = h264_host_grow(&g_wr.hdr, &g_wr.cap_hdr,
				     sizeof(struct h264_slice_header));
	if (res == 0)
		res = h264_host_grow(&g_wr.desc, &g_wr.cap_desc, 64);
	if (res == 0)
		res = h264_amd_upload(eng, g_wr.meta, zero, sizeof(zero));
	return res;
}

/* NAL header + slice header (+ partial byte + raw slice data) of the slice
 * described by ctx, through h264_amd_write_slices() */
static int wr_slice(struct h264_ctx *ctx, uint32_t wflags, size_t out_limit,
		    int with_data, struct wr_out *o)
{
	struct h264_amd_engine *eng;
	struct h264_amd_params *params;
	struct h264_amd_slice_desc desc;
	struct h264_amd_slice_result sr;
	struct h264_amd_write_args a;
	size_t raw_len = with_data ? ctx->slice.rawdata.len : 0;
	size_t out_cap;
	uint32_t epoch = 0;
	int res = h264_host_engine(&eng);
	if (res < 0)
		return res;
	if (raw_len != 0 && ctx->slice.rawdata.buf == NULL)
		return -EIO;
	if (raw_len > UINT32_MAX - 2 * WR_HDR_ROOM)
		return -E2BIG;
	out_cap = out_limit ? out_limit : raw_len + WR_HDR_ROOM;
	res = h264_host_ctx_params(ctx, &params, &epoch);
	if (res < 0)
		return res;

	memset(&desc, 0, sizeof(desc));
	desc.size = (uint32_t)raw_len;
	desc.epoch = epoch;
	memset(&sr, 0, sizeof(sr));
	sr.flags = H264_AMD_SR_HEADER_OK;
	sr.nal_ref_idc = ctx->nalu.hdr.nal_ref_idc;
	sr.nal_unit_type = ctx->nalu.hdr.nal_unit_type;
	sr.rawdata_len = (uint32_t)raw_len;
	if (with_data) {
		sr.partial = ctx->slice.rawdata.partial;
		sr.partialbits = ctx->slice.rawdata.partialbits;
	}

	res = wr_prepare(eng, out_cap);
	if (res == 0)
		res = h264_host_grow(&g_wr.res, &g_wr.cap_res, sizeof(sr));
	if (res == 0)
		res = h264_host_grow(&g_wr.src, &g_wr.cap_src, raw_len + 64);
	if (res == 0 && raw_len)
		res = h264_amd_upload(eng, g_wr.src, ctx->slice.rawdata.buf,
				      raw_len);
	if (res == 0)
		res = h264_amd_upload(eng, g_wr.hdr, &ctx->slice.hdr,
				      sizeof(ctx->slice.hdr));
	if (res == 0)
		res = h264_amd_upload(eng, g_wr.desc, &desc, sizeof(desc));
	if (res == 0)
		res = h264_amd_upload(eng, g_wr.res, &sr, sizeof(sr));
	if (res < 0)
		return res;

	memset(&a, 0, sizeof(a));
	a.d_src = g_wr.src;
	a.d_descs = g_wr.desc;
	a.d_results = g_wr.res;
	a.d_headers = g_wr.hdr;
	a.n = 1;
	a.wflags = wflags;
	a.d_out = g_wr.out;
	a.d_out_offsets = g_wr.meta;
	a.d_out_sizes = (uint32_t *)g_wr.meta + 2;
	a.d_out_hdr_bits = (uint32_t *)g_wr.meta + 3;
	a.out_capacity = (uint32_t)out_cap;
	res = h264_amd_write_slices(eng, params, &a);
	if (res < 0)
		return res;
	res = wr_collect(eng, out_cap + 64, o);
	if (res < 0)
		return res;
	if (o->len * 8 < o->hdr_bits) {
		/* nothing followed the header and it ends inside a byte */
		o->pend_bits = o->hdr_bits & 7u;
		o->pend = o->bytes[o->len];
	}
	return 0;
}

/* the bytes of `o` go behind what the caller's bitstream already holds */
static int wr_append(struct h264_bitstream *bs, const struct wr_out *o)
{
	int res = h264_bs_write_raw_bytes(bs, o->bytes, o->len);
	if (res < 0)
		return res;
	if (o->pend_bits) {
		res = h264_bs_write_bits(bs, o->pend >> (8 - o->pend_bits),
					 o->pend_bits);
		if (res < 0)
			return res;
	}
	return 0;
}

static int write_slice_nalu(struct h264_bitstream *bs, struct h264_ctx *ctx)
{
	struct wr_out o;
	size_t base;
	int res;
	if (!h264_bs_byte_aligned(bs))
		return -EIO;
	/* writing a slice header activates its PPS, as reading one does
	 * (ref src/h264_syntax.h:1256-1262) */
	res = h264_ctx_set_active_pps(ctx, ctx->slice.hdr.pic_parameter_set_id);
	if (res < 0)
		return res;
	base = bs->off;
	h264_host_lock();
	res = wr_slice(ctx, wflags_of(bs), 0, 1, &o);
	if (res == 0) {
		/* ref src/h264_syntax.h:1380-1381: position in the CALLER's
		 * bitstream, whatever it already contained */
		ctx->slice.hdr_len = base * 8 + o.hdr_bits;
		res = wr_append(bs, &o);
	}
	h264_host_unlock();
	return res;
}

/* ---- host side: non-VCL NAL units ------------------------------------------ */

static int write_pps(struct h264_syn *s, struct h264_ctx *ctx)
{
	struct h264_pps *pps = ctx->pps;
	if (pps == NULL)
		return -EIO;
	if (h264_syn_pps_ids(s, pps) < 0)
		return s->err;
	/* as when reading, the SPS the PPS names becomes the active one
	 * (ref src/h264_syntax.h:505-512) */
	if (h264_ctx_set_active_sps(ctx, pps->seq_parameter_set_id) < 0)
		return -EINVAL;
	return h264_syn_pps_body(s, ctx->sps, pps);
}

int h264_write_nalu(struct h264_bitstream *bs, struct h264_ctx *ctx)
{
	struct h264_syn s = {bs, 1, 0};
	uint32_t t;
	if (bs == NULL || ctx == NULL)
		return -EINVAL;
	t = ctx->nalu.hdr.nal_unit_type;
	ctx->nalu.type = t;
	if (t == H264_NALU_TYPE_SLICE || t == H264_NALU_TYPE_SLICE_IDR) {
		if (ctx->nalu.hdr.forbidden_zero_bit != 0)
			return -EIO;
		return write_slice_nalu(bs, ctx);
	}
	if (h264_syn_nalu_header(&s, &ctx->nalu.hdr) < 0)
		return s.err;
	switch (t) {
	case H264_NALU_TYPE_SEI:
		return h264_syn_sei_write(bs, ctx);
	case H264_NALU_TYPE_SPS:
		if (ctx->sps == NULL || ctx->nalu.hdr.nal_ref_idc == 0)
			return -EIO;
		return h264_syn_sps(&s, ctx->sps);
	case H264_NALU_TYPE_PPS:
		if (ctx->pps == NULL || ctx->nalu.hdr.nal_ref_idc == 0)
			return -EIO;
		return write_pps(&s, ctx);
	case H264_NALU_TYPE_AUD:
		if (ctx->nalu.hdr.nal_ref_idc != 0)
			return -EIO;
		return h264_syn_aud(&s, &ctx->aud);
	case H264_NALU_TYPE_FILLER:
		if (ctx->nalu.hdr.nal_ref_idc != 0)
			return -EIO;
		return h264_syn_filler(&s, &ctx->filler_len);
	default:
		ctx->nalu.unknown = 1;
		return 0;
	}
}

/* ---- synthetic slices (ref src/h264_writer.c:49-309) ------------------------ */

static int write_synth(struct h264_bitstream *bs, struct h264_ctx *ctx,
		       uint32_t kind, uint32_t mb_count)
{
	struct h264_amd_engine *eng;
	struct h264_amd_params *params;
	struct h264_amd_synth_desc desc;
	struct h264_amd_synth_args a;
	struct wr_out o;
	size_t out_cap, base;
	uint32_t epoch = 0, t;
	int res;

	if (bs == NULL || ctx == NULL || mb_count == 0)
		return -EINVAL;
	res = h264_ctx_set_active_pps(ctx, ctx->slice.hdr.pic_parameter_set_id);
	if (res < 0)
		return res;
	t = ctx->nalu.hdr.nal_unit_type;
	if (t != H264_NALU_TYPE_SLICE && t != H264_NALU_TYPE_SLICE_IDR) {
		/* the reference emits whatever NAL ctx describes and appends
		 * macroblocks to it; only slices make sense here */
		return -EINVAL;
	}
	if (ctx->slice.rawdata.len != 0 || ctx->slice.rawdata.partialbits != 0)
		return -EINVAL; /* ctx still carries a parsed slice's data */
	if (!h264_bs_byte_aligned(bs) || ctx->nalu.hdr.forbidden_zero_bit)
		return -EIO;
	res = h264_host_engine(&eng);
	if (res < 0)
		return res;
	if (mb_count > (1u << 28))
		return -E2BIG;
	out_cap = (size_t)2 * mb_count + WR_HDR_ROOM;
	base = bs->off;

	h264_host_lock();
	res = h264_host_ctx_params(ctx, &params, &epoch);
	if (res == 0)
		res = wr_prepare(eng, out_cap);
	memset(&desc, 0, sizeof(desc));
	desc.kind = kind;
	desc.epoch = epoch;
	desc.mb_count = mb_count;
	desc.nal_ref_idc = ctx->nalu.hdr.nal_ref_idc;
	desc.nal_unit_type = t;
	if (res == 0)
		res = h264_amd_upload(eng, g_wr.desc, &desc, sizeof(desc));
	if (res == 0)
		res = h264_amd_upload(eng, g_wr.hdr, &ctx->slice.hdr,
				      sizeof(ctx->slice.hdr));
	if (res == 0) {
		memset(&a, 0, sizeof(a));
		a.d_descs = g_wr.desc;
		a.d_headers = g_wr.hdr;
		a.n = 1;
		a.out_capacity = (uint32_t)out_cap;
		a.d_out = g_wr.out;
		a.d_out_offsets = g_wr.meta;
		a.d_out_sizes = (int32_t *)g_wr.meta + 2;
		a.d_out_hdr_bits = (uint32_t *)g_wr.meta + 3;
		a.wflags = wflags_of(bs);
		res = h264_amd_encode_synth_slices(eng, params, &a);
	}
	if (res == 0)
		res = wr_collect(eng, out_cap + 64, &o);
	if (res == 0) {
		ctx->slice.type = H264_SLICE_TYPE(ctx->slice.hdr.slice_type);
		ctx->slice.hdr_len = base * 8 + o.hdr_bits;
		res = wr_append(bs, &o);
	}
	h264_host_unlock();
	return res;
}

int h264_write_grey_i_slice(struct h264_bitstream *bs, struct h264_ctx *ctx,
			    uint32_t mb_count)
{
	return write_synth(bs, ctx, H264_AMD_SYNTH_GREY_I, mb_count);
}

int h264_write_skipped_p_slice(struct h264_bitstream *bs, struct h264_ctx *ctx,
			       uint32_t mb_count)
{
	return write_synth(bs, ctx, H264_AMD_SYNTH_SKIP_P, mb_count);
}

/* ---- in-place header patch (ref src/h264_writer.c:312-370) ------------------ */

int h264_rewrite_slice_header(struct h264_bitstream *bs, struct h264_ctx *ctx,
			      const struct h264_slice_header *sh)
{
	struct wr_out o;
	size_t saved_len;
	int res;
	if (bs == NULL || ctx == NULL || sh == NULL)
		return -EINVAL;
	ctx->slice.saved_hdr = ctx->slice.hdr;
	saved_len = ctx->slice.hdr_len;
	res = h264_ctx_set_slice_header(ctx, sh);
	if (res < 0)
		return res;
	/* the new header is serialised into a 64-byte scratch NAL of its own:
	 * no preceding bytes, escaping on */
	h264_host_lock();
	res = wr_slice(ctx, 0, 64, 0, &o);
	if (res == 0) {
		ctx->slice.hdr_len = o.hdr_bits;
		if (ctx->slice.hdr_len != saved_len)
			res = -EPROTO;
		else if (o.len + (o.pend_bits ? 1 : 0) > bs->len)
			res = -EIO;
	}
	if (res < 0) {
		ctx->slice.hdr = ctx->slice.saved_hdr;
		ctx->slice.hdr_len = saved_len;
	} else {
		memcpy(bs->data, o.bytes, o.len);
		if (o.pend_bits) {
			uint8_t keep = (uint8_t)((1u << (8 - o.pend_bits)) - 1);
			bs->data[o.len] = (uint8_t)((o.pend & ~keep) |
						    (bs->data[o.len] & keep));
		}
	}
	h264_host_unlock();
	return res;
}

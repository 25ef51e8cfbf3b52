/**
 * oracle/orc_api.c -- TEST INFRASTRUCTURE (see orc.h).
 * Reader object (ref src/h264_reader.c:58-162) and a recorder that turns a
 * parse into the canonical event log of evlog.h plus the engine's record
 * formats (include/h264_amd.h), for byte-wise comparison with the GPU path.
 * The recorder also performs the reference tool's per-NAL write-back check
 * (ref tools/h264_dump.c:229-262).
 */
#include "orc.h"
#include "evlog.h"

/* ---- reader (ref src/h264_reader.c) ------------------------------------ */

int orc_reader_new(const struct orc_cbs *cbs, void *userdata,
		   struct orc_reader **ret)
{
	struct orc_reader *rd;
	int res;
	if (ret == NULL)
		return -EINVAL;
	*ret = NULL;
	if (cbs == NULL)
		return -EINVAL;
	rd = calloc(1, sizeof(*rd));
	if (rd == NULL)
		return -ENOMEM;
	rd->cbs = *cbs;
	rd->userdata = userdata;
	res = orc_ctx_new(&rd->ctx);
	if (res < 0) {
		free(rd);
		return res;
	}
	*ret = rd;
	return 0;
}

int orc_reader_destroy(struct orc_reader *rd)
{
	if (rd == NULL)
		return 0;
	orc_ctx_destroy(rd->ctx);
	free(rd);
	return 0;
}

int orc_reader_parse_nalu(struct orc_reader *rd, uint32_t flags,
			  const uint8_t *buf, size_t len)
{
	struct h264_bitstream bs;
	if (rd == NULL || buf == NULL)
		return -EINVAL;
	rd->stop = 0;
	rd->flags = flags;
	h264_bs_cinit(&bs, buf, len, 1);
	bs.priv = rd;
	return orc_read_nalu(&bs, rd);
}

int orc_reader_parse(struct orc_reader *rd, uint32_t flags, const uint8_t *buf,
		     size_t len, size_t *off)
{
	size_t start = 0, end = 0;
	int res;
	if (rd == NULL || buf == NULL || off == NULL)
		return -EINVAL;
	rd->stop = 0;
	*off = 0;
	while (*off < len && !rd->stop) {
		res = orc_find_nalu(buf + *off, len - *off, &start, &end);
		if (res < 0 && res != -EAGAIN)
			break;
		/* per-NAL errors are swallowed (Appendix C.2) */
		orc_reader_parse_nalu(rd, flags, buf + *off + start,
				      end - start);
		*off += end;
	}
	return 0;
}

/* ---- recorder ----------------------------------------------------------- */

#define ORC_REC_WANT_T2 0x1 /* keep a T2 record per macroblock */
#define ORC_REC_WANT_REWRITE 0x2 /* h264-dump style write-back check */
#define ORC_REC_T2_CAP 65536u

struct orc_rewrite {
	uint64_t off; /* into rewrites blob */
	uint32_t len;
	int32_t res; /* orc_write_nalu result */
	uint32_t nal_index;
	int32_t match; /* 1 identical (trailing zero bytes tolerated) */
};

struct orc_rec {
	struct orc_reader *rd;
	const uint8_t *base;
	uint32_t flags, want;
	uint32_t nal_index;
	uint32_t cur_mbs;
	uint64_t cur_sum;
	uint32_t slice_index;
	struct evvec events, sps, pps, headers, results, mbrecs, mboffs;
	struct evvec rewrites, rewrite_idx, nals;
	/* h264_reader_stop() from the nalu_end callback of the n-th NAL unit of
	 * a parse (0: never) */
	uint32_t stop_after, ends;
};

static void cb_au_end(struct orc_ctx *ctx, void *ud)
{
	struct orc_rec *r = ud;
	(void)ctx;
	ev_push(&r->events, EV_AU_END, 0, 0, 0);
}

static void cb_nalu_begin(struct orc_ctx *ctx, uint32_t type,
			  const uint8_t *buf, size_t len,
			  const struct h264_nalu_header *nh, void *ud)
{
	struct orc_rec *r = ud;
	(void)ctx;
	ev_push(&r->events, EV_NALU_BEGIN, type | (nh->nal_ref_idc << 8),
		(uint64_t)(buf - r->base), len);
}

static void cb_nalu_end(struct orc_ctx *ctx, uint32_t type, const uint8_t *buf,
			size_t len, const struct h264_nalu_header *nh, void *ud)
{
	struct orc_rec *r = ud;
	ev_push(&r->events, EV_NALU_END, type | (nh->nal_ref_idc << 8),
		(uint64_t)(buf - r->base), len);
	if (r->stop_after && ++r->ends == r->stop_after)
		r->rd->stop = 1; /* h264_reader_stop (ref src/h264_reader.c:107-114) */
	if ((r->want & ORC_REC_WANT_REWRITE) && !ctx->nalu.unknown) {
		struct h264_bitstream bs;
		struct orc_rewrite w;
		size_t i;
		h264_bs_init(&bs, NULL, 0, 1);
		memset(&w, 0, sizeof(w));
		w.res = orc_write_nalu(&bs, ctx);
		w.nal_index = r->nal_index;
		w.off = r->rewrites.len;
		w.len = (uint32_t)bs.off;
		if (w.res >= 0) {
			if (bs.off == len) {
				w.match = memcmp(bs.data, buf, len) == 0;
			} else if (bs.off < len) {
				w.match = memcmp(bs.data, buf, bs.off) == 0;
				for (i = bs.off; w.match && i < len; i++)
					w.match = buf[i] == 0;
			}
		}
		if (bs.off)
			evvec_push(&r->rewrites, bs.data, bs.off);
		evvec_push(&r->rewrite_idx, &w, sizeof(w));
		h264_bs_clear(&bs);
	}
}

static void cb_slice(struct orc_ctx *ctx, const uint8_t *buf, size_t len,
		     const struct h264_slice_header *sh, void *ud)
{
	struct orc_rec *r = ud;
	uint32_t idx = (uint32_t)(r->headers.len / sizeof(*sh));
	(void)ctx;
	evvec_push(&r->headers, sh, sizeof(*sh));
	ev_push(&r->events, EV_SLICE, idx, (uint64_t)(buf - r->base), len);
}

static void cb_sd_begin(struct orc_ctx *ctx, const struct h264_slice_header *sh,
			void *ud)
{
	struct orc_rec *r = ud;
	(void)ctx;
	ev_push(&r->events, EV_SLICE_DATA_BEGIN, sh->first_mb_in_slice, 0, 0);
}

static void cb_sd_end(struct orc_ctx *ctx, const struct h264_slice_header *sh,
		      uint32_t mb_count, void *ud)
{
	struct orc_rec *r = ud;
	(void)ctx;
	(void)sh;
	ev_push(&r->events, EV_SLICE_DATA_END, mb_count, 0, 0);
}

static void cb_sd_mb(struct orc_ctx *ctx, const struct h264_slice_header *sh,
		     uint32_t mb_addr, uint32_t mb_type, void *ud)
{
	struct orc_rec *r = ud;
	uint32_t stride = h264_amd_t2_stride(ctx->sps_derived.ChromaArrayType);
	uint8_t tmp[H264_AMD_T2_HEAD + 2 * 768];
	const uint16_t *w = (const uint16_t *)tmp;
	uint32_t i;
	(void)sh;
	ev_push(&r->events, EV_SLICE_DATA_MB, mb_addr, mb_type, 0);
	orc_mb_to_record(ctx, tmp, stride);
	for (i = 0; i < stride / 4; i++) {
		uint32_t dword = (uint32_t)w[2 * i] | ((uint32_t)w[2 * i + 1] << 16);
		if (dword != 0)
			r->cur_sum = h264_amd_checksum_add(
				r->cur_sum,
				h264_amd_checksum_term(r->cur_mbs * 512u + i,
						       dword));
	}
	/* full records only for the first ORC_REC_T2_CAP macroblocks of a slice:
	 * a damaged skip run would otherwise cost a KiB per phantom macroblock */
	if ((r->want & ORC_REC_WANT_T2) && r->cur_mbs < ORC_REC_T2_CAP)
		evvec_push(&r->mbrecs, tmp, stride);
	r->cur_mbs++;
}

static void cb_sps(struct orc_ctx *ctx, const uint8_t *buf, size_t len,
		   const struct h264_sps *sps, void *ud)
{
	struct orc_rec *r = ud;
	uint32_t idx = (uint32_t)(r->sps.len / sizeof(*sps));
	(void)ctx;
	(void)buf;
	(void)len;
	evvec_push(&r->sps, sps, sizeof(*sps));
	ev_push(&r->events, EV_SPS, idx, 0, 0);
}

static void cb_pps(struct orc_ctx *ctx, const uint8_t *buf, size_t len,
		   const struct h264_pps *pps, void *ud)
{
	struct orc_rec *r = ud;
	uint32_t idx = (uint32_t)(r->pps.len / sizeof(*pps));
	(void)ctx;
	(void)buf;
	(void)len;
	evvec_push(&r->pps, pps, sizeof(*pps));
	ev_push(&r->events, EV_PPS, idx, 0, 0);
}

static void cb_aud(struct orc_ctx *ctx, const uint8_t *buf, size_t len,
		   const struct h264_aud *aud, void *ud)
{
	struct orc_rec *r = ud;
	(void)ctx;
	(void)buf;
	(void)len;
	ev_push(&r->events, EV_AUD, aud->primary_pic_type, 0, 0);
}

static void cb_sei(struct orc_ctx *ctx, uint32_t type, const uint8_t *buf,
		   size_t len, void *ud)
{
	struct orc_rec *r = ud;
	(void)ctx;
	ev_push(&r->events, EV_SEI, type, ev_fnv1a(buf, len, 0), len);
}

static void cb_sei_typed(struct orc_ctx *ctx, uint32_t type, const uint8_t *buf,
			 size_t len, const struct h264_sei *sei, void *ud)
{
	struct orc_rec *r = ud;
	(void)ctx;
	(void)len;
	/* all union members start at the same address */
	ev_push(&r->events, EV_SEI_TYPED, type,
		ev_sei_typed_hash(type, &sei->buffering_period, buf), 0);
}

struct orc_rec *orc_rec_new(uint32_t reader_flags, uint32_t want)
{
	struct orc_cbs cbs;
	struct orc_rec *r = calloc(1, sizeof(*r));
	if (r == NULL)
		return NULL;
	memset(&cbs, 0, sizeof(cbs));
	cbs.au_end = cb_au_end;
	cbs.nalu_begin = cb_nalu_begin;
	cbs.nalu_end = cb_nalu_end;
	cbs.slice = cb_slice;
	cbs.slice_data_begin = cb_sd_begin;
	cbs.slice_data_end = cb_sd_end;
	cbs.slice_data_mb = cb_sd_mb;
	cbs.sps = cb_sps;
	cbs.pps = cb_pps;
	cbs.aud = cb_aud;
	cbs.sei = cb_sei;
	cbs.sei_typed = cb_sei_typed;
	if (orc_reader_new(&cbs, r, &r->rd) < 0) {
		free(r);
		return NULL;
	}
	r->flags = reader_flags;
	r->want = want;
	return r;
}

void orc_rec_free(struct orc_rec *r)
{
	if (r == NULL)
		return;
	orc_reader_destroy(r->rd);
	evvec_free(&r->events);
	evvec_free(&r->sps);
	evvec_free(&r->pps);
	evvec_free(&r->headers);
	evvec_free(&r->results);
	evvec_free(&r->mbrecs);
	evvec_free(&r->mboffs);
	evvec_free(&r->rewrites);
	evvec_free(&r->rewrite_idx);
	evvec_free(&r->nals);
	free(r);
}

/* Parse one NAL unit (no start code).  `base` anchors the offsets reported in
 * the event log.  Appends a slice result for nal_unit_type 1/5. */
int orc_rec_parse_nalu(struct orc_rec *r, const uint8_t *base,
		       const uint8_t *buf, size_t len)
{
	struct orc_reader *rd = r->rd;
	struct orc_ctx *ctx = rd->ctx;
	struct h264_bitstream bs;
	uint64_t mboff = r->mbrecs.len;
	int res;

	r->base = base;
	r->cur_mbs = 0;
	r->cur_sum = 0;
	rd->stop = 0;
	rd->flags = r->flags;
	h264_bs_cinit(&bs, buf, len, 1);
	bs.priv = rd;
	res = orc_read_nalu(&bs, rd);
	ev_push(&r->events, EV_PARSE_RESULT, r->nal_index,
		(uint64_t)(int64_t)res, 0);

	if (len > 0 && (buf[0] & 0x80) == 0 &&
	    ((buf[0] & 0x1f) == H264_NALU_TYPE_SLICE ||
	     (buf[0] & 0x1f) == H264_NALU_TYPE_SLICE_IDR)) {
		struct h264_amd_slice_result sr;
		int hdr_ok = ctx->slice.rawdata.buf != NULL;
		memset(&sr, 0, sizeof(sr));
		sr.status = res;
		sr.nal_ref_idc = ctx->nalu.hdr.nal_ref_idc;
		sr.nal_unit_type = ctx->nalu.hdr.nal_unit_type;
		sr.slice_type = -1;
		if (hdr_ok) {
			int cabac = ctx->pps->entropy_coding_mode_flag;
			sr.flags |= H264_AMD_SR_HEADER_OK;
			if (cabac)
				sr.flags |= H264_AMD_SR_CABAC;
			if ((r->flags & H264_READER_FLAGS_SLICE_DATA) &&
			    (!cabac ||
			     (r->flags & H264_AMD_READER_FLAGS_CABAC)) &&
			    res != H264_AMD_E_UNSUPPORTED)
				sr.flags |= H264_AMD_SR_DATA_PARSED;
			sr.slice_type = ctx->slice.type;
			sr.first_mb_in_slice = ctx->slice.hdr.first_mb_in_slice;
			sr.slice_qp = ctx->derived.SliceQPLuma;
			sr.hdr_len_bits = (uint32_t)ctx->slice.hdr_len;
			sr.rawdata_off =
				(uint32_t)(ctx->slice.rawdata.buf - buf);
			sr.rawdata_len = (uint32_t)ctx->slice.rawdata.len;
			sr.partial = ctx->slice.rawdata.partial;
			sr.partialbits = ctx->slice.rawdata.partialbits;
			/* chroma_format_idc is not range-checked by the
			 * reference: reported saturating, as the kernels keep it */
			sr.chroma_array_type = (uint16_t)(
				ctx->sps_derived.ChromaArrayType < 255
					? ctx->sps_derived.ChromaArrayType
					: 255);
			sr.pic_width_in_mbs = ctx->sps_derived.PicWidthInMbs;
			if (res == 0)
				sr.end_bit =
					(uint32_t)orc_bs_rbsp_bits_consumed(&bs);
		}
		sr.mb_count = r->cur_mbs;
		sr.checksum = r->cur_sum;
		evvec_push(&r->results, &sr, sizeof(sr));
		evvec_push(&r->mboffs, &mboff, sizeof(mboff));
		r->slice_index++;
	}
	r->nal_index++;
	return res;
}

/* Annex-B stream, NAL loop of the reference (src/h264_reader.c:133-140).
 * Also records the NAL table as struct h264_amd_nal. */
int orc_rec_parse_stream(struct orc_rec *r, const uint8_t *buf, size_t len,
			 size_t *off)
{
	size_t start = 0, end = 0;
	int res;
	if (r == NULL || buf == NULL || off == NULL)
		return -EINVAL;
	*off = 0;
	r->rd->stop = 0;
	r->ends = 0;
	/* (the reference's loop looks at the stop flag where it looks at the
	 * offset, src/h264_reader.c:133) */
	while (*off < len && !r->rd->stop) {
		struct h264_amd_nal n;
		res = orc_find_nalu(buf + *off, len - *off, &start, &end);
		if (res < 0 && res != -EAGAIN)
			break;
		n.start = *off + start;
		n.end = *off + end;
		evvec_push(&r->nals, &n, sizeof(n));
		orc_rec_parse_nalu(r, buf, buf + *off + start, end - start);
		*off += end;
	}
	return 0;
}

#define ACCESSOR(name, field, type)                                            \
	const type *orc_rec_##name(struct orc_rec *r, size_t *count)           \
	{                                                                      \
		*count = r->field.len / sizeof(type);                          \
		return (const type *)r->field.p;                               \
	}

ACCESSOR(events, events, struct ev)
ACCESSOR(sps, sps, struct h264_sps)
ACCESSOR(pps, pps, struct h264_pps)
ACCESSOR(headers, headers, struct h264_slice_header)
ACCESSOR(results, results, struct h264_amd_slice_result)
ACCESSOR(mbrecs, mbrecs, uint8_t)
ACCESSOR(mboffs, mboffs, uint64_t)
ACCESSOR(rewrites, rewrites, uint8_t)
ACCESSOR(rewrite_idx, rewrite_idx, struct orc_rewrite)
ACCESSOR(nals, nals, struct h264_amd_nal)

void orc_rec_stop_after(struct orc_rec *r, uint32_t n)
{
	r->stop_after = n;
}

struct orc_ctx *orc_rec_ctx(struct orc_rec *r)
{
	return r->rd->ctx;
}

/* ---- small entry points for table / KAT tests --------------------------- */

size_t orc_sizeof(const char *what)
{
#define S(n, t)                                                                \
	if (strcmp(what, n) == 0)                                              \
		return sizeof(t)
	S("h264_bitstream", struct h264_bitstream);
	S("h264_nalu_header", struct h264_nalu_header);
	S("h264_sps", struct h264_sps);
	S("h264_pps", struct h264_pps);
	S("h264_slice_header", struct h264_slice_header);
	S("h264_sei", struct h264_sei);
	S("h264_sps_derived", struct h264_sps_derived);
	S("h264_info", struct h264_info);
	S("h264_ctx_cbs", struct h264_ctx_cbs);
	S("h264_amd_slice_result", struct h264_amd_slice_result);
	S("h264_amd_mb_summary", struct h264_amd_mb_summary);
#undef S
	return 0;
}

/* Serialise one NAL from explicit parts: the building block the test-side
 * stream builders use (parameter sets, synthetic slices).  kind: 0 write_nalu,
 * 1 grey-I slice, 2 skipped-P slice.  Returns bytes written or -errno. */
int orc_build_nalu(struct orc_ctx *ctx, const struct h264_nalu_header *nh,
		   const struct h264_sps *sps, const struct h264_pps *pps,
		   const struct h264_slice_header *sh, int kind,
		   uint32_t mb_count, uint8_t *out, size_t cap)
{
	struct h264_bitstream bs;
	int res;
	if (ctx == NULL || nh == NULL || out == NULL)
		return -EINVAL;
	res = orc_ctx_clear_nalu(ctx);
	if (res < 0)
		return res;
	if (sps != NULL && (res = orc_ctx_set_sps(ctx, sps)) < 0)
		return res;
	if (pps != NULL && (res = orc_ctx_set_pps(ctx, pps)) < 0)
		return res;
	orc_ctx_set_nalu_header(ctx, nh);
	if (sh != NULL) {
		/* activation as the reader would do it */
		res = orc_ctx_set_active_pps(ctx, sh->pic_parameter_set_id);
		if (res < 0)
			return res;
		res = orc_ctx_set_slice_header(ctx, sh);
		if (res < 0)
			return res;
	}
	h264_bs_init(&bs, NULL, 0, 1);
	if (kind == 1)
		res = orc_write_grey_i_slice(&bs, ctx, mb_count);
	else if (kind == 2)
		res = orc_write_skipped_p_slice(&bs, ctx, mb_count);
	else
		res = orc_write_nalu(&bs, ctx);
	if (res >= 0) {
		if (bs.cachebits != 0)
			res = -EIO;
		else if (bs.off > cap)
			res = -ENOBUFS;
		else {
			memcpy(out, bs.data, bs.off);
			res = (int)bs.off;
		}
	}
	h264_bs_clear(&bs);
	return res;
}

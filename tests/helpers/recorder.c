/**
 * tests/helpers/recorder.c -- drives the PRODUCT (libh264.so, public API only)
 * and turns what its callbacks deliver into the canonical event log of
 * oracle/evlog.h, so that tests can compare it byte for byte with the log the
 * oracle's recorder (oracle/orc_api.c) produces for the same input.
 * With want & 2 every parsed NAL unit is written back from nalu_end with
 * h264_write_nalu(), like the reference's dump tool does (tools/h264_dump.c);
 * with want & 4 every NAL unit is also dumped from nalu_end with
 * h264_dump_nalu(H264_DUMP_FLAGS_SLICE_DATA), one JSON text per line, again
 * like that tool (tools/h264_dump.c:195-227).
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <h264/h264.h>

#include "../../oracle/evlog.h"

#define REC_WANT_REWRITE 0x2
#define REC_WANT_DUMP 0x4

struct rec_rewrite {
	uint64_t off;
	uint32_t len;
	int32_t res;
	uint32_t nal_index;
	int32_t match;
};

struct rec {
	struct h264_reader *rd;
	const uint8_t *base;
	uint32_t flags, want;
	uint32_t nal_index;
	struct evvec events, sps, pps, headers, rewrites, rewrite_idx, dumps;
	struct h264_dump *dump;
	/* h264_reader_stop() from the nalu_end callback of the n-th NAL unit of
	 * a parse (0: never) */
	uint32_t stop_after, ends;
};

static void cb_au_end(struct h264_ctx *ctx, void *ud)
{
	struct rec *r = ud;
	(void)ctx;
	ev_push(&r->events, EV_AU_END, 0, 0, 0);
}

static void cb_nalu_begin(struct h264_ctx *ctx, enum h264_nalu_type type,
			  const uint8_t *buf, size_t len,
			  const struct h264_nalu_header *nh, void *ud)
{
	struct rec *r = ud;
	(void)ctx;
	ev_push(&r->events, EV_NALU_BEGIN, type | (nh->nal_ref_idc << 8),
		(uint64_t)(buf - r->base), len);
}

static void cb_nalu_end(struct h264_ctx *ctx, enum h264_nalu_type type,
			const uint8_t *buf, size_t len,
			const struct h264_nalu_header *nh, void *ud)
{
	struct rec *r = ud;
	ev_push(&r->events, EV_NALU_END, type | (nh->nal_ref_idc << 8),
		(uint64_t)(buf - r->base), len);
	if (r->stop_after && ++r->ends == r->stop_after)
		h264_reader_stop(r->rd);
	if (r->want & REC_WANT_DUMP) {
		const char *str = NULL;
		char num[32];
		int res;
		if (r->dump == NULL) {
			struct h264_dump_cfg cfg = {H264_DUMP_TYPE_JSON};
			if (h264_dump_new(&cfg, &r->dump) < 0)
				abort();
		}
		res = h264_dump_nalu(r->dump, ctx, H264_DUMP_FLAGS_SLICE_DATA);
		if (res >= 0)
			res = h264_dump_get_json_str(r->dump, &str);
		if (res >= 0) {
			evvec_push(&r->dumps, str, strlen(str));
		} else {
			int n = snprintf(num, sizeof(num), "ERR %d", res);
			evvec_push(&r->dumps, num, (size_t)n);
		}
		evvec_push(&r->dumps, "\n", 1);
	}
	if ((r->want & REC_WANT_REWRITE) && !h264_ctx_is_nalu_unknown(ctx)) {
		struct h264_bitstream bs;
		struct rec_rewrite w;
		size_t i;
		h264_bs_init(&bs, NULL, 0, 1);
		memset(&w, 0, sizeof(w));
		w.res = h264_write_nalu(&bs, ctx);
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

static void cb_slice(struct h264_ctx *ctx, const uint8_t *buf, size_t len,
		     const struct h264_slice_header *sh, void *ud)
{
	struct rec *r = ud;
	uint32_t idx = (uint32_t)(r->headers.len / sizeof(*sh));
	(void)ctx;
	evvec_push(&r->headers, sh, sizeof(*sh));
	ev_push(&r->events, EV_SLICE, idx, (uint64_t)(buf - r->base), len);
}

static void cb_sd_begin(struct h264_ctx *ctx, const struct h264_slice_header *sh,
			void *ud)
{
	struct rec *r = ud;
	(void)ctx;
	ev_push(&r->events, EV_SLICE_DATA_BEGIN, sh->first_mb_in_slice, 0, 0);
}

static void cb_sd_end(struct h264_ctx *ctx, const struct h264_slice_header *sh,
		      uint32_t mb_count, void *ud)
{
	struct rec *r = ud;
	(void)ctx;
	(void)sh;
	ev_push(&r->events, EV_SLICE_DATA_END, mb_count, 0, 0);
}

static void cb_sd_mb(struct h264_ctx *ctx, const struct h264_slice_header *sh,
		     uint32_t mb_addr, enum h264_mb_type mb_type, void *ud)
{
	struct rec *r = ud;
	(void)ctx;
	(void)sh;
	ev_push(&r->events, EV_SLICE_DATA_MB, mb_addr, mb_type, 0);
}

static void cb_sps(struct h264_ctx *ctx, const uint8_t *buf, size_t len,
		   const struct h264_sps *sps, void *ud)
{
	struct rec *r = ud;
	uint32_t idx = (uint32_t)(r->sps.len / sizeof(*sps));
	(void)ctx;
	(void)buf;
	(void)len;
	evvec_push(&r->sps, sps, sizeof(*sps));
	ev_push(&r->events, EV_SPS, idx, 0, 0);
}

static void cb_pps(struct h264_ctx *ctx, const uint8_t *buf, size_t len,
		   const struct h264_pps *pps, void *ud)
{
	struct rec *r = ud;
	uint32_t idx = (uint32_t)(r->pps.len / sizeof(*pps));
	(void)ctx;
	(void)buf;
	(void)len;
	evvec_push(&r->pps, pps, sizeof(*pps));
	ev_push(&r->events, EV_PPS, idx, 0, 0);
}

static void cb_aud(struct h264_ctx *ctx, const uint8_t *buf, size_t len,
		   const struct h264_aud *aud, void *ud)
{
	struct rec *r = ud;
	(void)ctx;
	(void)buf;
	(void)len;
	ev_push(&r->events, EV_AUD, aud->primary_pic_type, 0, 0);
}

static void cb_sei(struct h264_ctx *ctx, enum h264_sei_type type,
		   const uint8_t *buf, size_t len, void *ud)
{
	struct rec *r = ud;
	(void)ctx;
	ev_push(&r->events, EV_SEI, type, ev_fnv1a(buf, len, 0), len);
}

#define TYPED(name, stype, code)                                               \
	static void cb_##name(struct h264_ctx *ctx, const uint8_t *buf,        \
			      size_t len, const struct stype *sei, void *ud)   \
	{                                                                      \
		struct rec *r = ud;                                            \
		(void)ctx;                                                     \
		(void)len;                                                     \
		ev_push(&r->events, EV_SEI_TYPED, code,                        \
			ev_sei_typed_hash(code, sei, buf), 0);                 \
	}

TYPED(bp, h264_sei_buffering_period, H264_SEI_TYPE_BUFFERING_PERIOD)
TYPED(pt, h264_sei_pic_timing, H264_SEI_TYPE_PIC_TIMING)
TYPED(psr, h264_sei_pan_scan_rect, H264_SEI_TYPE_PAN_SCAN_RECT)
TYPED(fp, h264_sei_filler_payload, H264_SEI_TYPE_FILLER_PAYLOAD)
TYPED(udr, h264_sei_user_data_registered, H264_SEI_TYPE_USER_DATA_REGISTERED)
TYPED(udu, h264_sei_user_data_unregistered,
      H264_SEI_TYPE_USER_DATA_UNREGISTERED)
TYPED(rp, h264_sei_recovery_point, H264_SEI_TYPE_RECOVERY_POINT)

struct rec *rec_new(uint32_t reader_flags, uint32_t want)
{
	struct h264_ctx_cbs cbs;
	struct rec *r = calloc(1, sizeof(*r));
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
	cbs.sei_buffering_period = cb_bp;
	cbs.sei_pic_timing = cb_pt;
	cbs.sei_pan_scan_rect = cb_psr;
	cbs.sei_filler_payload = cb_fp;
	cbs.sei_user_data_registered = cb_udr;
	cbs.sei_user_data_unregistered = cb_udu;
	cbs.sei_recovery_point = cb_rp;
	if (h264_reader_new(&cbs, r, &r->rd) < 0) {
		free(r);
		return NULL;
	}
	r->flags = reader_flags;
	r->want = want;
	return r;
}

void rec_free(struct rec *r)
{
	if (r == NULL)
		return;
	h264_reader_destroy(r->rd);
	evvec_free(&r->events);
	evvec_free(&r->sps);
	evvec_free(&r->pps);
	evvec_free(&r->headers);
	evvec_free(&r->rewrites);
	evvec_free(&r->rewrite_idx);
	evvec_free(&r->dumps);
	h264_dump_destroy(r->dump);
	free(r);
}

/* one NAL unit (no start code); the status lands in the log like the
 * oracle's EV_PARSE_RESULT */
int rec_parse_nalu(struct rec *r, const uint8_t *base, const uint8_t *buf,
		   size_t len)
{
	uint32_t idx = r->nal_index;
	int res;
	r->base = base;
	res = h264_reader_parse_nalu(r->rd, r->flags, buf, len);
	ev_push(&r->events, EV_PARSE_RESULT, idx, (uint64_t)(int64_t)res, 0);
	/* a NAL that failed never reached nalu_end */
	r->nal_index = idx + 1;
	return res;
}

int rec_parse_stream(struct rec *r, const uint8_t *buf, size_t len, size_t *off)
{
	r->base = buf;
	r->ends = 0;
	return h264_reader_parse(r->rd, r->flags, buf, len, off);
}

void rec_stop_after(struct rec *r, uint32_t n)
{
	r->stop_after = n;
}

struct h264_ctx *rec_ctx(struct rec *r)
{
	return h264_reader_get_ctx(r->rd);
}

#define ACCESSOR(name, field, type)                                            \
	const type *rec_##name(struct rec *r, size_t *count)                   \
	{                                                                      \
		*count = r->field.len / sizeof(type);                          \
		return (const type *)r->field.p;                               \
	}

ACCESSOR(events, events, struct ev)
ACCESSOR(sps, sps, struct h264_sps)
ACCESSOR(pps, pps, struct h264_pps)
ACCESSOR(headers, headers, struct h264_slice_header)
ACCESSOR(rewrites, rewrites, uint8_t)
ACCESSOR(rewrite_idx, rewrite_idx, struct rec_rewrite)
ACCESSOR(dumps, dumps, uint8_t)

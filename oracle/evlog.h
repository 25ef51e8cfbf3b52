/**
 * oracle/evlog.h -- TEST INFRASTRUCTURE.
 * Canonical event log: what a libh264-style reader told its callbacks, in
 * order, in a form that can be compared byte for byte between the oracle
 * (oracle/orc_api.c) and the product driven through its public C API
 * (tests/helpers/recorder.c).
 */
#ifndef EVLOG_H
#define EVLOG_H

#include <h264/h264.h>

enum ev_kind {
	EV_AU_END = 1,
	EV_NALU_BEGIN = 2, /* a = type | ref_idc << 8, b = offset, c = len */
	EV_NALU_END = 3, /* same */
	EV_SLICE = 4, /* a = index into headers[], b = offset, c = len */
	EV_SLICE_DATA_BEGIN = 5, /* a = first_mb_in_slice */
	EV_SLICE_DATA_END = 6, /* a = mb_count */
	EV_SLICE_DATA_MB = 7, /* a = mb_addr, b = mb_type */
	EV_SPS = 8, /* a = index into sps[] */
	EV_PPS = 9, /* a = index into pps[] */
	EV_AUD = 10, /* a = primary_pic_type */
	EV_SEI = 11, /* a = type, b = fnv1a(payload), c = len */
	EV_SEI_TYPED = 12, /* a = type, b = fnv1a(canonical struct) */
	EV_PARSE_RESULT = 13, /* a = nal index, b = (uint64)(int64)status */
};

struct ev {
	uint32_t kind;
	uint32_t a;
	uint64_t b;
	uint64_t c;
};

/* growable byte vector */
struct evvec {
	uint8_t *p;
	size_t len, cap;
};

static inline void *evvec_push(struct evvec *v, const void *src, size_t n)
{
	void *dst;
	if (v->len + n > v->cap) {
		size_t nc = v->cap ? v->cap * 2 : 4096;
		uint8_t *np;
		while (nc < v->len + n)
			nc *= 2;
		np = (uint8_t *)realloc(v->p, nc);
		if (np == NULL)
			abort();
		v->p = np;
		v->cap = nc;
	}
	dst = v->p + v->len;
	if (src != NULL)
		memcpy(dst, src, n);
	else
		memset(dst, 0, n);
	v->len += n;
	return dst;
}

static inline void evvec_free(struct evvec *v)
{
	free(v->p);
	v->p = NULL;
	v->len = v->cap = 0;
}

static inline uint64_t ev_fnv1a(const void *data, size_t n, uint64_t h)
{
	const uint8_t *p = (const uint8_t *)data;
	size_t i;
	if (h == 0)
		h = 0xcbf29ce484222325ull;
	for (i = 0; i < n; i++) {
		h ^= p[i];
		h *= 0x100000001b3ull;
	}
	return h;
}

static inline void ev_push(struct evvec *v, uint32_t kind, uint32_t a,
			   uint64_t b, uint64_t c)
{
	struct ev e;
	e.kind = kind;
	e.a = a;
	e.b = b;
	e.c = c;
	evvec_push(v, &e, sizeof(e));
}

/* Hash of a typed SEI payload struct with pointers replaced by their offset
 * into the raw payload, so that two implementations with different heaps
 * compare equal.  `specific` points at the union member for `type`. */
static inline uint64_t ev_sei_typed_hash(uint32_t type, const void *specific,
					 const uint8_t *rawbuf)
{
	uint64_t h = 0;
	uint64_t v[2];
	switch (type) {
	case H264_SEI_TYPE_BUFFERING_PERIOD:
		return ev_fnv1a(specific,
				sizeof(struct h264_sei_buffering_period), 0);
	case H264_SEI_TYPE_PIC_TIMING:
		return ev_fnv1a(specific, sizeof(struct h264_sei_pic_timing), 0);
	case H264_SEI_TYPE_PAN_SCAN_RECT:
		return ev_fnv1a(specific, sizeof(struct h264_sei_pan_scan_rect),
				0);
	case H264_SEI_TYPE_RECOVERY_POINT:
		return ev_fnv1a(specific, sizeof(struct h264_sei_recovery_point),
				0);
	case H264_SEI_TYPE_FILLER_PAYLOAD: {
		const struct h264_sei_filler_payload *s =
			(const struct h264_sei_filler_payload *)specific;
		v[0] = (uint64_t)(s->buf - rawbuf);
		v[1] = s->len;
		return ev_fnv1a(v, sizeof(v), 0);
	}
	case H264_SEI_TYPE_USER_DATA_REGISTERED: {
		const struct h264_sei_user_data_registered *s =
			(const struct h264_sei_user_data_registered *)specific;
		h = ev_fnv1a(&s->country_code, 4, 0);
		h = ev_fnv1a(&s->country_code_extension_byte, 4, h);
		v[0] = (uint64_t)(s->buf - rawbuf);
		v[1] = s->len;
		return ev_fnv1a(v, sizeof(v), h);
	}
	case H264_SEI_TYPE_USER_DATA_UNREGISTERED: {
		const struct h264_sei_user_data_unregistered *s =
			(const struct h264_sei_user_data_unregistered *)specific;
		h = ev_fnv1a(s->uuid, 16, 0);
		v[0] = (uint64_t)(s->buf - rawbuf);
		v[1] = s->len;
		return ev_fnv1a(v, sizeof(v), h);
	}
	default:
		return 0;
	}
}

#endif /* EVLOG_H */

/**
 * libh264_amd -- bit-level reader/writer cursor of the libh264 C API.
 *
 * Drop-in for the reference's include/h264/h264_bitstream.h: the 48-byte
 * struct layout (ref :31-60) is frozen because callers allocate it and the
 * static-inline accessors below are compiled into THEIR objects.  The inline
 * helpers restate the reference semantics (ref :117-317):
 *   - the cursor keeps ONE partially consumed/filled byte in `cache`;
 *   - on read, an emulation-prevention byte (0x03 after two raw 0x00 bytes)
 *     is skipped when the byte after it is fetched, tested on RAW bytes;
 *   - a 0x03 that is the last byte of the buffer is an error (-EIO).
 *
 * This per-bit host cursor exists for the control plane (parameter sets, SEI,
 * callers poking at a NAL by hand).  Slice NAL units never go through it: they
 * are parsed by the wavefront bit reader in csrc/kernels/bitreader.hip.h.
 */
#ifndef _H264_BITSTREAM_H_
#define _H264_BITSTREAM_H_

struct h264_bitstream {
	union {
		const uint8_t *cdata; /* read side */
		uint8_t *data; /* write side */
	};
	size_t len; /* buffer length (capacity when writing) */
	size_t off; /* next raw byte to fetch / to store */
	uint8_t cache; /* partially read / written byte */
	uint8_t cachebits; /* read: unread bits in cache; write: filled bits */
	int emulation_prevention;
	int dynamic; /* writer owns and grows `data` */
	void *priv;
};

/* ---- out-of-line API (ref :63-114) --------------------------------------- */

/* First NAL unit of an Annex-B buffer: *start = first payload byte, *end = one
 * past the last.  -ENOENT: no start code; -EAGAIN: NAL runs to end of buffer
 * (start/end still valid).  Single-shot host helper; the batched equivalent is
 * h264_amd_scan() in <h264_amd.h>. */
H264_API int
h264_find_nalu(const uint8_t *buf, size_t len, size_t *start, size_t *end);

H264_API int h264_bs_write_bits(struct h264_bitstream *bs, uint64_t v, uint32_t n);
H264_API int h264_bs_read_bits_ue(struct h264_bitstream *bs, uint32_t *v);
H264_API int h264_bs_write_bits_ue(struct h264_bitstream *bs, uint32_t v);
H264_API int h264_bs_read_bits_ff_coded(struct h264_bitstream *bs, uint32_t *v);
H264_API int h264_bs_write_bits_ff_coded(struct h264_bitstream *bs, uint32_t v);
H264_API int h264_bs_more_rbsp_data(const struct h264_bitstream *bs);
H264_API int
h264_bs_next_bits(const struct h264_bitstream *bs, uint32_t *v, uint32_t n);
H264_API int h264_bs_read_rbsp_trailing_bits(struct h264_bitstream *bs);
H264_API int h264_bs_write_rbsp_trailing_bits(struct h264_bitstream *bs);
H264_API int
h264_bs_read_raw_bytes(struct h264_bitstream *bs, uint8_t *buf, size_t len);
H264_API int h264_bs_write_raw_bytes(struct h264_bitstream *bs,
				     const uint8_t *buf,
				     size_t len);
H264_API int
h264_bs_acquire_buf(struct h264_bitstream *bs, uint8_t **buf, size_t *len);

/* ---- inline accessors (ref :117-317) -------------------------------------- */

static inline void h264_bs_cinit(struct h264_bitstream *bs,
				 const uint8_t *buf,
				 size_t len,
				 int emulation_prevention)
{
	memset(bs, 0, sizeof(*bs));
	bs->cdata = buf;
	bs->len = len;
	bs->emulation_prevention = emulation_prevention;
}

/* buf == NULL && len == 0 selects a growable writer */
static inline void h264_bs_init(struct h264_bitstream *bs,
				uint8_t *buf,
				size_t len,
				int emulation_prevention)
{
	memset(bs, 0, sizeof(*bs));
	bs->data = buf;
	bs->len = len;
	bs->emulation_prevention = emulation_prevention;
	bs->dynamic = (buf == NULL && len == 0) ? 1 : 0;
}

static inline void h264_bs_clear(struct h264_bitstream *bs)
{
	if (bs->dynamic)
		free(bs->data);
	memset(bs, 0, sizeof(*bs));
}

static inline int h264_bs_byte_aligned(const struct h264_bitstream *bs)
{
	return (bs->cachebits & 7) == 0;
}

static inline int h264_bs_eos(const struct h264_bitstream *bs)
{
	return bs->cachebits == 0 && bs->off >= bs->len;
}

static inline size_t h264_bs_rem_raw_bits(const struct h264_bitstream *bs)
{
	return 8 * (bs->len - bs->off) + bs->cachebits;
}

/* Load the next RBSP byte into the cache, hopping over an escape byte. */
static inline int h264_bs_fetch(struct h264_bitstream *bs)
{
	const uint8_t *p = bs->cdata;
	size_t o = bs->off;
	/* o < len first: the reference looks at p[o] even at the end of the buffer
	 * (one byte past it); the outcome is -EIO either way */
	int escaped = bs->emulation_prevention && o >= 2 && o < bs->len &&
		      p[o] == 0x03 && p[o - 1] == 0x00 && p[o - 2] == 0x00;
	if (escaped) {
		if (o + 1 >= bs->len)
			return -EIO;
		o++;
	} else if (o >= bs->len) {
		return -EIO;
	}
	bs->cache = p[o];
	bs->cachebits = 8;
	bs->off = o + 1;
	return 0;
}

/* Read n (<= 32) bits MSB first; returns n or -EIO when data runs out. */
static inline int
h264_bs_read_bits(struct h264_bitstream *bs, uint32_t *v, uint32_t n)
{
	uint32_t acc = 0;
	uint32_t left = n;
	*v = 0;
	while (left != 0) {
		uint32_t take;
		if (bs->cachebits == 0 && h264_bs_fetch(bs) < 0)
			return -EIO;
		take = left < bs->cachebits ? left : bs->cachebits;
		bs->cachebits -= take;
		acc = (acc << take) |
		      ((bs->cache >> bs->cachebits) & ((1u << take) - 1));
		*v = acc;
		left -= take;
	}
	return (int)n;
}

static inline int
h264_bs_read_bits_u(struct h264_bitstream *bs, uint32_t *v, uint32_t n)
{
	return h264_bs_read_bits(bs, v, n);
}

static inline int
h264_bs_write_bits_u(struct h264_bitstream *bs, uint32_t v, uint32_t n)
{
	return h264_bs_write_bits(bs, v, n);
}

/* two's complement n-bit field */
static inline int
h264_bs_read_bits_i(struct h264_bitstream *bs, int32_t *v, uint32_t n)
{
	uint32_t u = 0;
	int res = h264_bs_read_bits(bs, &u, n);
	if (res < 0)
		return res;
	if (u & (1u << (n - 1)))
		u |= ((uint32_t)-1) << n;
	*v = (int32_t)u;
	return res;
}

static inline int
h264_bs_write_bits_i(struct h264_bitstream *bs, int32_t v, uint32_t n)
{
	return h264_bs_write_bits_u(bs, ((uint32_t)v) & ((1 << n) - 1), n);
}

/* se(v): k odd -> +(k+1)/2, k even -> -(k/2)  (9.1.1) */
static inline int h264_bs_read_bits_se(struct h264_bitstream *bs, int32_t *v)
{
	uint32_t k = 0;
	int res = h264_bs_read_bits_ue(bs, &k);
	if (res < 0)
		return res;
	*v = (k & 1) ? ((int32_t)k + 1) / 2 : -((int32_t)k + 1) / 2;
	return res;
}

static inline int h264_bs_write_bits_se(struct h264_bitstream *bs, int32_t v)
{
	/* unsigned arithmetic: the same bits as the reference's signed products,
	 * defined for |v| >= 2^30 as well */
	uint32_t k = v > 0 ? 2u * (uint32_t)v - 1u : 0u - 2u * (uint32_t)v;
	return h264_bs_write_bits_ue(bs, k);
}

/* te(v) with range m: one inverted bit when m == 1, else ue(v) */
static inline int
h264_bs_read_bits_te(struct h264_bitstream *bs, uint32_t *v, uint32_t m)
{
	int res;
	if (m != 1)
		return h264_bs_read_bits_ue(bs, v);
	res = h264_bs_read_bits(bs, v, 1);
	*v = !*v;
	return res;
}

static inline int
h264_bs_write_bits_te(struct h264_bitstream *bs, uint32_t v, uint32_t m)
{
	if (m != 1)
		return h264_bs_write_bits_ue(bs, v);
	return h264_bs_write_bits(bs, !v, 1);
}

#endif /* !_H264_BITSTREAM_H_ */

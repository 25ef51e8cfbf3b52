/**
 * bitstream.c -- out-of-line half of <h264/h264_bitstream.h> (host, plain C).
 *
 * Drop-in for the functions the reference defines in src/h264_bitstream.c
 * (cited per function).  These byte-granular helpers serve the control plane
 * (parameter sets, SEI, callers assembling NAL units by hand); the batched
 * equivalents for whole buffers run on the GPU (h264_amd_scan,
 * csrc/kernels/wave_bits.hip.h).
 */
#include "h264_priv.h"

/* grow a dynamic writer in 256-byte steps (ref :28-51) */
static int bs_reserve(struct h264_bitstream *bs, size_t need)
{
	uint8_t *p;
	if (need <= bs->len)
		return 0;
	if (!bs->dynamic)
		return -EIO;
	need = (need + 255) & ~(size_t)255;
	p = realloc(bs->data, need);
	if (p == NULL)
		return -ENOMEM;
	bs->data = p;
	bs->len = need;
	return 0;
}

/* A byte is complete: store it, preceded by 0x03 when the last two STORED
 * bytes are zero and it is <= 3 (ref h264_bs_flush :54-81). */
static int bs_store_cache(struct h264_bitstream *bs)
{
	size_t o = bs->off;
	int escape = bs->emulation_prevention && o >= 2 && bs->cache <= 3 &&
		     bs->data[o - 1] == 0 && bs->data[o - 2] == 0;
	int res = bs_reserve(bs, o + 1 + (escape ? 1 : 0));
	if (res < 0)
		return res;
	if (escape)
		bs->data[o++] = 0x03;
	bs->data[o++] = bs->cache;
	bs->off = o;
	bs->cache = 0;
	bs->cachebits = 0;
	return 0;
}

/* ref :211-239 */
int h264_bs_write_bits(struct h264_bitstream *bs, uint64_t v, uint32_t n)
{
	int written = 0;
	if (n > 64)
		return -EINVAL;
	while (n != 0) {
		uint32_t room = 8u - bs->cachebits;
		uint32_t take = room < n ? room : n;
		uint32_t part = (uint32_t)(v >> (n - take)) & ((1u << take) - 1u);
		bs->cache |= (uint8_t)(part << (room - take));
		bs->cachebits += take;
		n -= take;
		written += (int)take;
		if (bs->cachebits == 8 && bs_store_cache(bs) < 0)
			return -EIO;
	}
	return written;
}

/* Exp-Golomb, zeros counted bit by bit like the reference (:190-208); the
 * wavefront version uses one count-leading-zeros instead */
int h264_bs_read_bits_ue(struct h264_bitstream *bs, uint32_t *v)
{
	uint32_t zeros = 0, bit = 0, tail = 0;
	for (;;) {
		if (h264_bs_read_bits(bs, &bit, 1) < 0)
			return -EIO;
		if (bit)
			break;
		zeros++;
	}
	/* 32 or more zeros: not representable (and "1 << 32" is undefined in the
	 * reference); an error, as in the kernels */
	if (zeros >= 32)
		return -EIO;
	if (zeros && h264_bs_read_bits(bs, &tail, zeros) < 0)
		return -EIO;
	*v = (1u << zeros) - 1u + tail;
	return (int)(2 * zeros + 1);
}

/* ref :245-287: value + 1 written on 2 * bitlength - 1 bits */
int h264_bs_write_bits_ue(struct h264_bitstream *bs, uint32_t v)
{
	uint32_t x, n = 0, t;
	if (v == 0)
		return h264_bs_write_bits(bs, 1, 1);
	x = v + 1;
	for (t = x; t; t >>= 1)
		n++;
	if (n == 0)
		n = 1; /* v == 0xffffffff wraps, as in the reference */
	return h264_bs_write_bits(bs, x, 2 * n - 1);
}

/* SEI-style value: sum of bytes, continued while a byte is 0xff (ref :290-319) */
int h264_bs_read_bits_ff_coded(struct h264_bitstream *bs, uint32_t *v)
{
	uint32_t byte = 0;
	int total = 0, res;
	*v = 0;
	do {
		res = h264_bs_read_bits(bs, &byte, 8);
		if (res < 0)
			return res;
		total += res;
		*v += byte;
	} while (byte == 0xff);
	return total;
}

int h264_bs_write_bits_ff_coded(struct h264_bitstream *bs, uint32_t v)
{
	uint32_t byte;
	int total = 0, res;
	do {
		byte = v > 0xff ? 0xff : v;
		res = h264_bs_write_bits(bs, byte, 8);
		if (res < 0)
			return res;
		total += res;
		v -= byte;
	} while (byte == 0xff);
	return total;
}

/* 7.2 more_rbsp_data() with the reference's tolerance for ONE trailing zero
 * byte, tested on the raw byte (ref :325-355) */
int h264_bs_more_rbsp_data(const struct h264_bitstream *bs)
{
	struct h264_bitstream probe = *bs;
	uint32_t bit = 0;
	if (h264_bs_read_bits(&probe, &bit, 1) < 0)
		return 0;
	if (!bit)
		return 1;
	while (!h264_bs_byte_aligned(&probe)) {
		if (h264_bs_read_bits(&probe, &bit, 1) < 0)
			return 0;
		if (bit)
			return 1;
	}
	if (h264_bs_eos(&probe))
		return 0;
	return probe.off + 1 < probe.len || probe.cdata[probe.off] != 0;
}

/* ref :358-364 */
int h264_bs_next_bits(const struct h264_bitstream *bs, uint32_t *v, uint32_t n)
{
	struct h264_bitstream probe = *bs;
	return h264_bs_read_bits(&probe, v, n);
}

/* ref :370-392 */
int h264_bs_read_rbsp_trailing_bits(struct h264_bitstream *bs)
{
	uint32_t bit = 0;
	int res = h264_bs_read_bits(bs, &bit, 1);
	if (res < 0)
		return res;
	if (bit != 1)
		return -EIO;
	while (!h264_bs_byte_aligned(bs)) {
		res = h264_bs_read_bits(bs, &bit, 1);
		if (res < 0)
			return res;
		if (bit != 0)
			return -EIO;
	}
	return 0;
}

/* ref :398-415 */
int h264_bs_write_rbsp_trailing_bits(struct h264_bitstream *bs)
{
	int res = h264_bs_write_bits(bs, 1, 1);
	while (res >= 0 && !h264_bs_byte_aligned(bs))
		res = h264_bs_write_bits(bs, 0, 1);
	return res < 0 ? res : 0;
}

/* ref :418-426 */
int h264_bs_read_raw_bytes(struct h264_bitstream *bs, uint8_t *buf, size_t len)
{
	if (!h264_bs_byte_aligned(bs) || bs->len - bs->off != len)
		return -EIO;
	memcpy(buf, bs->cdata + bs->off, len);
	bs->off += len;
	return 0;
}

/* ref :428-440: no escaping, the bytes are taken as already escaped */
int h264_bs_write_raw_bytes(struct h264_bitstream *bs, const uint8_t *buf,
			    size_t len)
{
	int res;
	if (!h264_bs_byte_aligned(bs))
		return -EIO;
	res = bs_reserve(bs, bs->off + len);
	if (res < 0)
		return res;
	if (len)
		memcpy(bs->data + bs->off, buf, len);
	bs->off += len;
	return 0;
}

/* ref :443-451 */
int h264_bs_acquire_buf(struct h264_bitstream *bs, uint8_t **buf, size_t *len)
{
	if (!h264_bs_byte_aligned(bs) || !bs->dynamic)
		return -EIO;
	*buf = bs->data;
	*len = bs->off;
	bs->dynamic = 0;
	return 0;
}

/* ---- Annex-B, single shot (ref :87-184) --------------------------------- */

/* index of the first 00 00 0x (x <= maxlast) triple at or after `from`, or
 * len when there is none; three bytes must remain */
static size_t find_triple(const uint8_t *b, size_t len, size_t from,
			  uint8_t minlast, uint8_t maxlast)
{
	size_t i;
	for (i = from; i + 3 <= len; i++) {
		if (b[i] == 0 && b[i + 1] == 0 && b[i + 2] >= minlast &&
		    b[i + 2] <= maxlast)
			return i;
	}
	return len;
}

int h264_find_nalu(const uint8_t *buf, size_t len, size_t *start, size_t *end)
{
	size_t s, e;
	if (buf == NULL || start == NULL || end == NULL)
		return -EINVAL;
	/* a 4-byte start code is a 3-byte one behind a zero: same payload
	 * start, so only the 00 00 01 triple matters here */
	s = find_triple(buf, len, 0, 1, 1);
	if (s == len)
		return -ENOENT;
	*start = s + 3;
	e = find_triple(buf, len, *start, 0, 1);
	*end = e;
	return e == len ? -EAGAIN : 0;
}

/**
 * oracle/orc_bits.c -- TEST INFRASTRUCTURE (see orc.h).
 * Bit-serial restatement of the reference's bitstream layer:
 *   include/h264/h264_bitstream.h:168-304 (fetch, read_bits, se, te)
 *   src/h264_bitstream.c:28-451 (capacity, flush, find_nalu, ue, write_bits,
 *   ff-coded, more_rbsp_data, next_bits, trailing bits, raw bytes).
 * Deliberately kept one-bit/one-byte at a time like the reference: this file
 * defines behaviour, the wavefront reader in csrc/kernels defines speed.
 */
#include "orc.h"

/* ref include/h264/h264_bitstream.h:168-190 */
int orc_bs_fetch(struct h264_bitstream *bs)
{
	/* off < len first: the reference reads cdata[off] even at the end of the
	 * buffer (one byte past it); the outcome is -EIO either way */
	if (bs->emulation_prevention && bs->off >= 2 && bs->off < bs->len &&
	    bs->cdata[bs->off - 2] == 0 && bs->cdata[bs->off - 1] == 0 &&
	    bs->cdata[bs->off] == 3) {
		/* escape byte: deliver the byte behind it; a trailing 03 is
		 * an error */
		if (bs->off + 1 >= bs->len)
			return -EIO;
		bs->cache = bs->cdata[bs->off + 1];
		bs->off += 2;
	} else if (bs->off < bs->len) {
		bs->cache = bs->cdata[bs->off];
		bs->off += 1;
	} else {
		return -EIO;
	}
	bs->cachebits = 8;
	return 0;
}

/* ref include/h264/h264_bitstream.h:193-218 */
int orc_bs_read_bits(struct h264_bitstream *bs, uint32_t *v, uint32_t n)
{
	int got = 0;
	*v = 0;
	while (n > 0) {
		uint32_t k;
		if (bs->cachebits == 0 && orc_bs_fetch(bs) < 0)
			return -EIO;
		k = n < bs->cachebits ? n : bs->cachebits;
		*v = (*v << k) |
		     ((bs->cache >> (bs->cachebits - k)) & ((1u << k) - 1));
		bs->cachebits -= k;
		n -= k;
		got += k;
	}
	return got;
}

/* ref include/h264/h264_bitstream.h:235-251 */
int orc_bs_read_bits_i(struct h264_bitstream *bs, int32_t *v, uint32_t n)
{
	uint32_t u = 0;
	int res = orc_bs_read_bits(bs, &u, n);
	if (res >= 0) {
		if (u & (1u << (n - 1)))
			*v = (int32_t)(u | (((uint32_t)-1) << n));
		else
			*v = (int32_t)u;
	}
	return res;
}

/* ref src/h264_bitstream.c:190-208: zeros counted one bit per read */
int orc_bs_read_ue(struct h264_bitstream *bs, uint32_t *v)
{
	int lz = -1;
	uint32_t bit = 0;
	for (bit = 0; !bit; lz++) {
		if (orc_bs_read_bits(bs, &bit, 1) < 0)
			return -EIO;
	}
	bit = 0;
	if (lz) {
		if (orc_bs_read_bits(bs, &bit, lz) < 0)
			return -EIO;
	}
	/* 32 leading zeros: the value does not fit 32 bits and the reference's
	 * "1 << 32" is undefined; an error here as in the kernels */
	if (lz >= 32)
		return -EIO;
	*v = (1u << lz) - 1 + bit;
	return lz * 2 + 1;
}

/* ref include/h264/h264_bitstream.h:264-275 */
int orc_bs_read_se(struct h264_bitstream *bs, int32_t *v)
{
	uint32_t k = 0;
	int res = orc_bs_read_ue(bs, &k);
	if (res >= 0)
		*v = (k & 1) ? (((int32_t)k + 1) / 2) : (-((int32_t)k + 1) / 2);
	return res;
}

/* ref include/h264/h264_bitstream.h:293-304 */
int orc_bs_read_te(struct h264_bitstream *bs, uint32_t *v, uint32_t m)
{
	int res;
	if (m == 1) {
		res = orc_bs_read_bits(bs, v, 1);
		*v = !*v;
	} else {
		res = orc_bs_read_ue(bs, v);
	}
	return res;
}

/* ref src/h264_bitstream.c:290-303 */
int orc_bs_read_ff(struct h264_bitstream *bs, uint32_t *v)
{
	int res, n = 0;
	uint32_t b = 0;
	*v = 0;
	do {
		res = orc_bs_read_bits(bs, &b, 8);
		if (res < 0)
			return res;
		n += res;
		*v += b;
	} while (b == 0xff);
	return n;
}

/* ref src/h264_bitstream.c:325-355 */
int orc_bs_more_rbsp_data(const struct h264_bitstream *bs)
{
	struct h264_bitstream t = *bs;
	uint32_t bit = 0;
	if (orc_bs_read_bits(&t, &bit, 1) < 0)
		return 0;
	if (bit != 1)
		return 1;
	while (t.cachebits % 8 != 0) {
		if (orc_bs_read_bits(&t, &bit, 1) < 0)
			return 0;
		if (bit != 0)
			return 1;
	}
	if (t.off >= t.len && t.cachebits == 0)
		return 0;
	/* one trailing zero byte is tolerated; RAW byte peeked (C.6) */
	return t.off + 1 < t.len || t.cdata[t.off] != 0x00;
}

/* ref src/h264_bitstream.c:358-364 */
int orc_bs_next_bits(const struct h264_bitstream *bs, uint32_t *v, uint32_t n)
{
	struct h264_bitstream t = *bs;
	return orc_bs_read_bits(&t, v, n);
}

/* ref src/h264_bitstream.c:370-392 */
int orc_bs_read_trailing(struct h264_bitstream *bs)
{
	uint32_t bit = 0;
	int res = orc_bs_read_bits(bs, &bit, 1);
	if (res < 0)
		return res;
	if (bit != 1)
		return -EIO;
	while (bs->cachebits % 8 != 0) {
		res = orc_bs_read_bits(bs, &bit, 1);
		if (res < 0)
			return res;
		if (bit != 0)
			return -EIO;
	}
	return 0;
}

uint64_t orc_bs_rbsp_bits_consumed(const struct h264_bitstream *bs)
{
	/* escape bytes among the raw bytes fetched so far */
	uint64_t epb = 0;
	size_t i;
	if (bs->emulation_prevention) {
		for (i = 2; i < bs->off; i++) {
			if (bs->cdata[i] == 3 && bs->cdata[i - 1] == 0 &&
			    bs->cdata[i - 2] == 0)
				epb++;
		}
	}
	return 8 * ((uint64_t)bs->off - epb) - bs->cachebits;
}

/* ---- Annex-B delimiting (ref src/h264_bitstream.c:87-184) -------------- */

static int find_start_code(const uint8_t *buf, size_t len, size_t *start,
			   size_t *end)
{
	size_t i = 0;
	while (i + 3 <= len) {
		if (buf[i] == 0) {
			if (len - i >= 4 && buf[i + 1] == 0 &&
			    buf[i + 2] == 0 && buf[i + 3] == 1) {
				*start = i;
				*end = i + 4;
				return 0;
			}
			if (buf[i + 1] == 0 && buf[i + 2] == 1) {
				*start = i;
				*end = i + 3;
				return 0;
			}
		}
		i++;
	}
	return -ENOENT;
}

static int find_end_code(const uint8_t *buf, size_t len, size_t *end)
{
	size_t i = 0;
	while (i + 3 <= len) {
		if (buf[i] == 0 && buf[i + 1] == 0 &&
		    (buf[i + 2] == 0 || buf[i + 2] == 1)) {
			*end = i;
			return 0;
		}
		i++;
	}
	return -ENOENT;
}

int orc_find_nalu(const uint8_t *buf, size_t len, size_t *start, size_t *end)
{
	size_t sc_begin = 0, sc_end = 0, ec = 0;
	int res;
	if (buf == NULL || start == NULL || end == NULL)
		return -EINVAL;
	res = find_start_code(buf, len, &sc_begin, &sc_end);
	if (res < 0)
		return res;
	*start = sc_end;
	res = find_end_code(buf + *start, len - *start, &ec);
	if (res < 0) {
		*end = len;
		return -EAGAIN;
	}
	*end = *start + ec;
	return 0;
}

/* ---- writer (ref src/h264_bitstream.c:28-81,211-287,306-319,398-440) ---- */

static int ensure_capacity(struct h264_bitstream *bs, size_t cap)
{
	uint8_t *nb;
	if (cap <= bs->len)
		return 0;
	if (!bs->dynamic)
		return -EIO;
	cap = (cap + 255) & ~(size_t)255;
	nb = realloc(bs->data, cap);
	if (nb == NULL)
		return -ENOMEM;
	bs->data = nb;
	bs->len = cap;
	return 0;
}

/* a completed byte leaves the cache; 0x03 is inserted when the two previous
 * OUTPUT bytes are zero and the new byte is <= 3 */
static int flush_byte(struct h264_bitstream *bs)
{
	int res;
	if (bs->emulation_prevention && bs->off >= 2 &&
	    bs->data[bs->off - 2] == 0 && bs->data[bs->off - 1] == 0 &&
	    bs->cache <= 3) {
		res = ensure_capacity(bs, bs->off + 2);
		if (res < 0)
			return res;
		bs->data[bs->off++] = 3;
		bs->data[bs->off++] = bs->cache;
	} else {
		res = ensure_capacity(bs, bs->off + 1);
		if (res < 0)
			return res;
		bs->data[bs->off++] = bs->cache;
	}
	bs->cache = 0;
	bs->cachebits = 0;
	return 0;
}

int orc_bs_write_bits(struct h264_bitstream *bs, uint64_t v, uint32_t n)
{
	int res = 0;
	if (n > 64)
		return -EINVAL;
	while (n > 0) {
		uint32_t k = 8 - bs->cachebits;
		uint32_t part;
		if (k >= n)
			k = n;
		part = (uint32_t)(v >> (n - k)) & ((1u << k) - 1);
		bs->cache |= part << (8 - bs->cachebits - k);
		n -= k;
		bs->cachebits += k;
		res += k;
		if (bs->cachebits == 8 && flush_byte(bs) < 0)
			return -EIO;
	}
	return res;
}

int orc_bs_write_ue(struct h264_bitstream *bs, uint32_t v)
{
	uint32_t n = 0, t;
	if (v == 0)
		return orc_bs_write_bits(bs, 1, 1);
	v++;
	/* n = number of significant bits of v (the reference uses a 256-entry
	 * log table, src/h264_bitstream.c:248-285) */
	for (t = v; t != 0; t >>= 1)
		n++;
	if (n == 0)
		n = 1; /* v wrapped to 0: the reference's table[0] is 1 */
	return orc_bs_write_bits(bs, v, 2 * n - 1);
}

int orc_bs_write_se(struct h264_bitstream *bs, int32_t v)
{
	/* the reference multiplies the signed value (include/h264/h264_bitstream.h
	 * :277-290): same bits, but defined for |v| >= 2^30 too */
	if (v <= 0)
		return orc_bs_write_ue(bs, 0u - 2u * (uint32_t)v);
	return orc_bs_write_ue(bs, 2u * (uint32_t)v - 1u);
}

int orc_bs_write_te(struct h264_bitstream *bs, uint32_t v, uint32_t m)
{
	if (m == 1)
		return orc_bs_write_bits(bs, !v, 1);
	return orc_bs_write_ue(bs, v);
}

int orc_bs_write_ff(struct h264_bitstream *bs, uint32_t v)
{
	int res, n = 0;
	uint32_t b;
	do {
		b = v > 0xff ? 0xff : v;
		res = orc_bs_write_bits(bs, b, 8);
		if (res < 0)
			return res;
		n += res;
		v -= b;
	} while (b == 0xff);
	return n;
}

int orc_bs_write_trailing(struct h264_bitstream *bs)
{
	int res = orc_bs_write_bits(bs, 1, 1);
	if (res < 0)
		return res;
	while (bs->cachebits % 8 != 0) {
		res = orc_bs_write_bits(bs, 0, 1);
		if (res < 0)
			return res;
	}
	return 0;
}

int orc_bs_write_raw_bytes(struct h264_bitstream *bs, const uint8_t *buf,
			   size_t len)
{
	int res;
	if (bs->cachebits % 8 != 0)
		return -EIO;
	res = ensure_capacity(bs, bs->off + len);
	if (res < 0)
		return res;
	if (len)
		memcpy(bs->data + bs->off, buf, len);
	bs->off += len;
	return 0;
}

/* Same contract as ref_bs_read_script() of ref_harness/ref_bits.c, on the
 * oracle's reader: lets tests/test_oracle_pins.py compare long read sequences
 * with the reference's inline reader. */
int orc_bs_read_script(const uint8_t *buf, size_t len, int ep,
		       const uint8_t *counts, size_t n, uint32_t *out,
		       uint64_t *pos)
{
	struct h264_bitstream bs;
	size_t i;
	h264_bs_cinit(&bs, buf, len, ep);
	for (i = 0; i < n; i++) {
		uint32_t v = 0;
		int res = orc_bs_read_bits(&bs, &v, counts[i]);
		out[i] = res < 0 ? 0xFFFFFFFFu : v;
		pos[i] = ((uint64_t)bs.off << 8) | bs.cachebits;
		if (res < 0)
			return (int)i;
	}
	return (int)n;
}

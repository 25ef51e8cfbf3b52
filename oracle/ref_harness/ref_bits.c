/**
 * oracle/ref_harness/ref_bits.c -- TEST INFRASTRUCTURE.
 * Thin exports around the static-inline bit reader of the REFERENCE's own
 * public header, compiled from where it lies (/root/reference/include) by
 * oracle/Makefile into oracle/_ref/libref_bits.so.  This is the only part of
 * the reference that builds in this image without a stand-in for a missing
 * dependency (everything in src/ needs <ulog.h>), so it is the only part the
 * oracle can be pinned against by execution.  No reference source is copied:
 * this file only calls h264_bs_cinit / h264_bs_fetch / h264_bs_read_bits /
 * h264_bs_read_bits_i / h264_bs_byte_aligned / h264_bs_eos /
 * h264_bs_rem_raw_bits (include/h264/h264_bitstream.h:117-251).
 */
#include <h264/h264.h>

void ref_bs_cinit(struct h264_bitstream *bs, const uint8_t *buf, size_t len,
		  int ep)
{
	h264_bs_cinit(bs, buf, len, ep);
}

int ref_bs_fetch(struct h264_bitstream *bs)
{
	return h264_bs_fetch(bs);
}

int ref_bs_read_bits(struct h264_bitstream *bs, uint32_t *v, uint32_t n)
{
	return h264_bs_read_bits(bs, v, n);
}

int ref_bs_read_bits_i(struct h264_bitstream *bs, int32_t *v, uint32_t n)
{
	return h264_bs_read_bits_i(bs, v, n);
}

int ref_bs_byte_aligned(const struct h264_bitstream *bs)
{
	return h264_bs_byte_aligned(bs);
}

int ref_bs_eos(const struct h264_bitstream *bs)
{
	return h264_bs_eos(bs);
}

size_t ref_bs_rem_raw_bits(const struct h264_bitstream *bs)
{
	return h264_bs_rem_raw_bits(bs);
}

size_t ref_sizeof_bitstream(void)
{
	return sizeof(struct h264_bitstream);
}

/* Read a script of bit counts from one cursor; out[i] = value or
 * 0xFFFFFFFF on error, pos[i] = (off, cachebits) after the read.  Lets a test
 * compare long read sequences without one ctypes call per read. */
int ref_bs_read_script(const uint8_t *buf, size_t len, int ep,
		       const uint8_t *counts, size_t n, uint32_t *out,
		       uint64_t *pos)
{
	struct h264_bitstream bs;
	size_t i;
	h264_bs_cinit(&bs, buf, len, ep);
	for (i = 0; i < n; i++) {
		uint32_t v = 0;
		int res = h264_bs_read_bits(&bs, &v, counts[i]);
		out[i] = res < 0 ? 0xFFFFFFFFu : v;
		pos[i] = ((uint64_t)bs.off << 8) | bs.cachebits;
		if (res < 0)
			return (int)i;
	}
	return (int)n;
}

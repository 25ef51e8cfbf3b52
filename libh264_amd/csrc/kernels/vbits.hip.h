/**
 * vbits.hip.h -- the slice-data view of the bit window: the cursor kept on the
 * VECTOR unit in wave-uniform registers.
 *
 * Why.  The syntax walk of slice_decode.hip is one serial dependency chain per
 * wavefront and was bound by the compute unit's scalar port (tools/probe/
 * issue_probe.hip, profiles/r02_decode_cost_model.md): one scalar instruction
 * per ~4.6 cycles and SIMD, a v_readlane with a scalar lane index ~9 cycles,
 * most vector instructions 2.8 - 4.8 cycles on a port that stood idle.  A peek
 * through WaveBits costs three scalar instructions and two such v_readlane;
 * here it is ONE v_lshlrev_b64 on a 64-bit piece of the window that every lane
 * holds a copy of, a skip is one v_add, and what is decoded from the bits
 * (prefix lengths, code values, suffix lengths) stays in vector registers with
 * the same value in every lane.  Only what steers control flow or addresses a
 * lane crosses to the scalar side (v_readfirstlane / ballot).
 *
 *   W   64 bits of the window, identical in all lanes: high dword = window
 *       dword [sd], low dword = [sd + 1]
 *   vo  bit offset of the cursor inside W, identical in all lanes
 *   sd  scalar: index of W's first dword in WaveBits::wreg
 *
 * vb_norm() brings vo below 32 (at least 33 valid bits at the cursor).  A peek
 * at offset vo sees 64 - vo stream bits followed by zeros, so a syntax element
 * of at most L bits may be decoded while vo <= 64 - L: callers normalise before
 * every element unless the bits consumed since the last vb_norm() are bounded
 * (the bound is written at each such place).
 *
 * Same observable semantics as the reference's reader
 * (include/h264/h264_bitstream.h:168-218, src/h264_bitstream.c:190-208); the
 * ingest, de-escaping and the window slide stay in wave_bits.hip.h.
 */
#ifndef H264_AMD_VBITS_HIP_H
#define H264_AMD_VBITS_HIP_H

#include "wave_bits.hip.h"

struct VBits {
	uint64_t W; /* per lane, all lanes equal */
	uint32_t vo; /* per lane, all lanes equal */
	uint32_t sd; /* scalar */
};

/* keep a wave-uniform value in a vector register: what is computed from it then
 * stays on the vector unit (the optimiser would move it to the scalar one) */
#define VB_VREG(x) asm volatile("" : "+v"(x))

/* a condition that is the same in every lane, computed on the vector unit, as
 * a branch condition */
__device__ __forceinline__ bool vb_true(bool c)
{
	return __builtin_amdgcn_ballot_w64(c) != 0ull;
}

__device__ __forceinline__ void vb_load(VBits &b, const WaveBits &s)
{
	b.sd = s.bitpos >> 5;
	uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)s.wreg, (int)b.sd);
	uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)s.wnext, (int)b.sd);
	uint64_t w = ((uint64_t)hi << 32) | lo;
	VB_VREG(w);
	b.W = w;
	uint32_t o = s.bitpos & 31u;
	VB_VREG(o);
	b.vo = o;
}

/* hand the position back to the scalar cursor (window slide, end-of-data
 * tests, raw-byte positions) */
__device__ __forceinline__ void vb_store(const VBits &b, WaveBits &s)
{
	s.bitpos = 32u * b.sd +
		   (uint32_t)__builtin_amdgcn_readfirstlane((int)b.vo);
}

/* vo < 64 on entry */
__device__ __forceinline__ void vb_norm(VBits &b, const WaveBits &s)
{
	/* one time in three or four; said to be unlikely so that the refill sits
	 * out of line and the usual way is the branch not taken (19 cycles of a
	 * lone wave against 31-34 taken) */
	if (__builtin_expect(vb_true(b.vo >= 32u), 0)) {
		b.sd += 1u;
		uint32_t nx = (uint32_t)__builtin_amdgcn_readlane((int)s.wnext,
								  (int)b.sd);
		VB_VREG(nx);
		b.W = (b.W << 32) | nx;
		b.vo -= 32u;
	}
}

/* the window slide of wb_sync() for a cursor kept here */
__device__ __forceinline__ void vb_sync(VBits &b, WaveBits &s, uint32_t lane)
{
	if (__builtin_expect(b.sd >= WB_SLIDE_BITS / 32u, 0)) {
		vb_store(b, s);
		wb_sync(s, lane);
		vb_load(b, s);
	}
}

/* next 32 bits, MSB first, in every lane */
__device__ __forceinline__ uint32_t vb_peek(const VBits &b)
{
	uint32_t r = (uint32_t)((b.W << b.vo) >> 32);
	/* opaque: tests of the result stay 32-bit compares instead of being
	 * folded into 64-bit compares of the shifted pair */
	asm volatile("" : "+v"(r));
	return r;
}

/* the same as a scalar */
__device__ __forceinline__ uint32_t vb_peek_s(const VBits &b)
{
	uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)vb_peek(b));
	asm volatile("" : "+s"(r)); /* see wb_peek() */
	return r;
}

__device__ __forceinline__ void vb_skip(VBits &b, uint32_t n)
{
	b.vo += n;
}

/* RBSP bits consumed since the start of the NAL */
__device__ __forceinline__ uint32_t vb_pos(const VBits &b, const WaveBits &s)
{
	return s.base * 8u + 32u * b.sd +
	       (uint32_t)__builtin_amdgcn_readfirstlane((int)b.vo);
}

/* u(n), 1 <= n <= 32, scalar result; normalises first */
__device__ __forceinline__ uint32_t vb_u(VBits &b, const WaveBits &s, uint32_t n)
{
	vb_norm(b, s);
	uint32_t v = vb_peek_s(b);
	vb_skip(b, n);
	return v >> (32u - n);
}

/* ue(v), scalar result (see wb_ue) */
__device__ __forceinline__ uint32_t vb_ue(VBits &b, const WaveBits &s, bool &ok)
{
	vb_norm(b, s);
	uint32_t v = vb_peek_s(b);
	/* (the usual way falls through: a branch not taken is 12-15 cycles
	 * cheaper for a wave alone on its SIMD than one taken) */
	if (__builtin_expect(v >= 0x10000u, 1)) {
		uint32_t lz = (uint32_t)__builtin_clz(v);
		vb_skip(b, 2u * lz + 1u);
		return (v >> (31u - 2u * lz)) - 1u;
	}
	if (v == 0) {
		ok = false;
		return 0;
	}
	uint32_t lz = (uint32_t)__builtin_clz(v);
	vb_skip(b, lz);
	vb_norm(b, s);
	uint32_t u = vb_peek_s(b);
	vb_skip(b, lz + 1u);
	return (u >> (31u - lz)) - 1u;
}

__device__ __forceinline__ int32_t vb_se(VBits &b, const WaveBits &s, bool &ok)
{
	uint32_t k = vb_ue(b, s, ok);
	int32_t t = (int32_t)(k + 1u);
	return (k & 1u) ? t / 2 : (int32_t)(0u - (uint32_t)t) / 2;
}

__device__ __forceinline__ uint32_t vb_te(VBits &b, const WaveBits &s, uint32_t m,
					  bool &ok)
{
	if (m == 1)
		return vb_u(b, s, 1) ^ 1u;
	return vb_ue(b, s, ok);
}

#endif /* H264_AMD_VBITS_HIP_H */

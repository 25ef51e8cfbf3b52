/**
 * wave_bits.hip.h -- one-NAL-per-wavefront RBSP bit reader for gfx950.
 *
 * GPU counterpart of the reference's byte-at-a-time cursor
 * (include/h264/h264_bitstream.h:168-218: 8-bit cache, one escape test per
 * fetched byte; src/h264_bitstream.c:190-208: Exp-Golomb one bit per call).
 * Same observable semantics, different mechanics:
 *
 *   HBM --(64 lanes x 4 B, coalesced, prefetched one chunk ahead)--> registers
 *       --(ballot / mbcnt compaction drops 00 00 [03])--> LDS ring of RBSP
 *       --(one ds_read_b32 per lane)--> 256-byte window held in a VGPR
 *       --(v_readlane -> SGPR)--> 32-bit peeks, all parse state wave-uniform
 *
 * Every value returned by peek()/ue()/... is wave-uniform (it comes out of
 * v_readlane), so the compiler keeps the whole syntax walk on the scalar unit:
 * s_flbit for leading zeros, s_lshl/s_lshr for field extraction, s_cbranch for
 * control flow.  The 64 lanes only work together when the window slides.
 *
 * The slide (ingest + compaction + window reload) is ONE out-of-line function
 * so that the many places a parser may need it cost a call, not a copy: its
 * rarely changing state lives in a 64-byte LDS block, only the hot cursor
 * (base, bitpos, bytes ingested) stays in scalar registers.
 *
 * Escape rule (evaluated on RAW bytes): raw[i] is
 * dropped iff i >= 2 && raw[i-2] == 0 && raw[i-1] == 0 && raw[i] == 3.
 */
#ifndef H264_AMD_WAVE_BITS_HIP_H
#define H264_AMD_WAVE_BITS_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define WB_RING_BYTES 1024u /* LDS ring of de-escaped bytes, per wave */
#define WB_STATE_BYTES 64u /* LDS block of slow-path state, per wave */
#define WB_CHUNK 256u /* raw bytes ingested per step */
#define WB_SLIDE_BITS 1024u /* window slides in 128-byte steps */

typedef unsigned long long wb_mask_t;

/* slow-path state, LDS resident (uint32 words) */
enum {
	WBS_ABASE_LO,
	WBS_ABASE_HI,
	WBS_MIS,
	WBS_LEN,
	WBS_NCHUNKS,
	WBS_CARRY, /* last raw dword of the previous chunk */
	WBS_ZERO_HI, /* ring is zero padded up to this RBSP byte index */
	WBS_HINT_CHUNK, /* 4 most recent chunks ... */
	WBS_HINT_CLEAN = WBS_HINT_CHUNK + 4, /* ... RBSP bytes before each */
	WBS_WORDS = WBS_HINT_CLEAN + 4,
};

struct WaveBits {
	/* ---- wave-uniform, register resident ---- */
	const uint8_t *nal; /* first byte of the NAL unit */
	uint32_t len; /* raw NAL length */
	uint32_t nchunks; /* raw chunks covering the NAL */
	uint32_t chunk; /* chunks ingested so far */
	uint32_t clean_total; /* RBSP bytes in the ring so far */
	uint32_t base; /* RBSP byte index of the window start */
	uint32_t bitpos; /* bit position relative to base */
	/* ---- per lane ---- */
	uint32_t wreg; /* lane i: RBSP bytes base+4i.. as a big-endian word */
	uint32_t wnext; /* lane i: wreg of lane i + 1 (one index serves a peek) */
	uint32_t pre; /* raw dword of the next chunk (prefetch) */
	/* ---- LDS ---- */
	uint8_t *ring;
	uint32_t *state;
};

__device__ __forceinline__ uint32_t wb_lane()
{
	return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

__device__ __forceinline__ uint32_t wb_uniform(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

/* vec with lane `idx` replaced by the uniform value `val` (v_cmp + v_cndmask;
 * this clang has no v_writelane builtin and an asm one would need its own
 * hazard nops) */
__device__ __forceinline__ uint32_t wb_setlane(uint32_t vec, uint32_t idx,
					       uint32_t val, uint32_t lane)
{
	return lane == idx ? val : vec;
}

/* number of set bits of m below this lane */
__device__ __forceinline__ uint32_t wb_rank(wb_mask_t m)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
					 __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

/* ---- slow path (out of line) --------------------------------------------- */

struct WbSlowCtx {
	const uint8_t *abase; /* 4-byte aligned address <= NAL start */
	uint32_t mis, len, nchunks;
};

__device__ __forceinline__ void wb_slow_load(WbSlowCtx &c, const uint32_t *st)
{
	uint64_t a = ((uint64_t)wb_uniform(st[WBS_ABASE_HI]) << 32) |
		     wb_uniform(st[WBS_ABASE_LO]);
	c.abase = reinterpret_cast<const uint8_t *>(a);
	c.mis = wb_uniform(st[WBS_MIS]);
	c.len = wb_uniform(st[WBS_LEN]);
	c.nchunks = wb_uniform(st[WBS_NCHUNKS]);
}

/* raw dword `lane` of chunk c, zero where it lies wholly outside the NAL */
__device__ __forceinline__ uint32_t wb_load_chunk(const WbSlowCtx &s, uint32_t c,
						  uint32_t lane)
{
	uint32_t off = c * WB_CHUNK + 4u * lane; /* from abase */
	uint32_t v = 0;
	if (off < s.mis + s.len)
		v = *reinterpret_cast<const uint32_t *>(s.abase + off);
	return v;
}

/* Classify the 4 bytes of this lane's dword of chunk c: bit j of the result
 * is set if byte j belongs to the RBSP (inside the NAL, not an escape byte).
 * prev = raw dword preceding w in address order. */
__device__ __forceinline__ uint32_t wb_classify(const WbSlowCtx &s, uint32_t c,
						uint32_t lane, uint32_t w,
						uint32_t prev)
{
	int32_t r0 = (int32_t)(c * WB_CHUNK + 4u * lane) - (int32_t)s.mis;
	uint32_t keep = 0;
	/* bytes in address order: p2 p3 | b0 b1 b2 b3 */
	uint32_t b[6];
	b[0] = (prev >> 16) & 0xff;
	b[1] = prev >> 24;
	b[2] = w & 0xff;
	b[3] = (w >> 8) & 0xff;
	b[4] = (w >> 16) & 0xff;
	b[5] = w >> 24;
#pragma unroll
	for (int j = 0; j < 4; j++) {
		int32_t r = r0 + j; /* raw index inside the NAL */
		bool inside = r >= 0 && r < (int32_t)s.len;
		bool esc = r >= 2 && b[j] == 0 && b[j + 1] == 0 && b[j + 2] == 3;
		if (inside && !esc)
			keep |= 1u << j;
	}
	return keep;
}

struct WbFill {
	uint32_t wreg, pre, clean_total, chunk;
};

/* Make the ring hold the window at `base` and ~512 bytes beyond it (or the
 * whole rest of the NAL), pad with zeros behind the end of the data, and
 * return the window.  The only vector-heavy code of the reader. */
__device__ __noinline__ WbFill wb_fill_slow(uint32_t *st, uint8_t *ring,
					    uint32_t pre, uint32_t base,
					    uint32_t clean_total, uint32_t chunk)
{
	const uint32_t lane = wb_lane();
	WbSlowCtx s;
	wb_slow_load(s, st);
	base = wb_uniform(base);
	clean_total = wb_uniform(clean_total);
	chunk = wb_uniform(chunk);
	uint32_t carry = wb_uniform(st[WBS_CARRY]);

	while (chunk < s.nchunks && clean_total < base + 512u) {
		uint32_t w = pre;
		/* prefetch the chunk after this one: latency hides under parsing */
		pre = wb_load_chunk(s, chunk + 1, lane);
		uint32_t up = (uint32_t)__shfl_up((int)w, 1);
		uint32_t prev = lane == 0 ? carry : up;
		uint32_t keep = wb_classify(s, chunk, lane, w, prev);
		wb_mask_t m0 = __ballot(keep & 1u), m1 = __ballot(keep & 2u);
		wb_mask_t m2 = __ballot(keep & 4u), m3 = __ballot(keep & 8u);
		uint32_t dst = clean_total + wb_rank(m0) + wb_rank(m1) +
			       wb_rank(m2) + wb_rank(m3);
#pragma unroll
		for (int j = 0; j < 4; j++) {
			if (keep & (1u << j)) {
				ring[dst & (WB_RING_BYTES - 1)] = (uint8_t)(w >> (8 * j));
				dst++;
			}
		}
		/* hints for wb_raw_of_clean(): (chunk, RBSP bytes before it) */
		if (lane < 3) {
			st[WBS_HINT_CHUNK + lane] = st[WBS_HINT_CHUNK + lane + 1];
			st[WBS_HINT_CLEAN + lane] = st[WBS_HINT_CLEAN + lane + 1];
		}
		if (lane == 0) {
			st[WBS_HINT_CHUNK + 3] = chunk;
			st[WBS_HINT_CLEAN + 3] = clean_total;
		}
		clean_total += (uint32_t)(__popcll(m0) + __popcll(m1) +
					  __popcll(m2) + __popcll(m3));
		carry = (uint32_t)__builtin_amdgcn_readlane((int)w, 63);
		chunk++;
	}
	if (lane == 0)
		st[WBS_CARRY] = carry;
	if (chunk >= s.nchunks) {
		/* end of NAL: reads past it must see zeros.  Pad from the end
		 * of the data up to base + 768, which never wraps onto live
		 * bytes because the ring holds [base, base + 1024). */
		uint32_t zero_hi = wb_uniform(st[WBS_ZERO_HI]);
		uint32_t lo = clean_total > zero_hi ? clean_total : zero_hi;
		uint32_t hi = base + 768u;
		if (lo < hi) {
			for (uint32_t t = lo + lane; t < hi; t += 64u)
				ring[t & (WB_RING_BYTES - 1)] = 0;
			if (lane == 0)
				st[WBS_ZERO_HI] = hi;
		}
	}
	/* same-wave LDS traffic is executed in program order: the byte stores
	 * above are visible to this read without a barrier */
	uint32_t o = (base + 4u * lane) & (WB_RING_BYTES - 1);
	uint32_t v = *reinterpret_cast<const uint32_t *>(ring + o);
	WbFill r;
	r.wreg = __builtin_bswap32(v);
	r.pre = pre;
	r.clean_total = clean_total;
	r.chunk = chunk;
	return r;
}

/* Raw index (inside the NAL) of RBSP byte k.  Rescans raw chunks from memory,
 * starting at the most recent ingest hint not beyond k; used a handful of
 * times per slice (header end, tail tests). */
__device__ __noinline__ uint32_t wb_raw_of_clean_slow(const uint32_t *st,
						      uint32_t k, uint32_t chunks_in)
{
	const uint32_t lane = wb_lane();
	WbSlowCtx s;
	wb_slow_load(s, st);
	k = wb_uniform(k);
	chunks_in = wb_uniform(chunks_in);
	uint32_t c = 0, before = 0;
	for (int i = 0; i < 4; i++) {
		uint32_t hc = wb_uniform(st[WBS_HINT_CHUNK + i]);
		uint32_t hb = wb_uniform(st[WBS_HINT_CLEAN + i]);
		if (hb <= k && hc >= c && hc < chunks_in) {
			c = hc;
			before = hb;
		}
	}
	for (; c < s.nchunks; c++) {
		uint32_t w = wb_load_chunk(s, c, lane);
		uint32_t off = c * WB_CHUNK + 4u * lane;
		uint32_t prev = 0xffffffffu;
		if (off >= 4u)
			prev = *reinterpret_cast<const uint32_t *>(s.abase + off - 4u);
		uint32_t keep = wb_classify(s, c, lane, w, prev);
		wb_mask_t m0 = __ballot(keep & 1u), m1 = __ballot(keep & 2u);
		wb_mask_t m2 = __ballot(keep & 4u), m3 = __ballot(keep & 8u);
		uint32_t total = (uint32_t)(__popcll(m0) + __popcll(m1) +
					    __popcll(m2) + __popcll(m3));
		if (k < before + total) {
			uint32_t ord = before + wb_rank(m0) + wb_rank(m1) +
				       wb_rank(m2) + wb_rank(m3);
			uint32_t hit = 0xffffffffu;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				if (keep & (1u << j)) {
					if (ord == k)
						hit = off + j - s.mis;
					ord++;
				}
			}
			wb_mask_t hm = __ballot(hit != 0xffffffffu);
			int src = hm ? (int)__builtin_ctzll(hm) : 0;
			return (uint32_t)__builtin_amdgcn_readlane((int)hit, src);
		}
		before += total;
	}
	return s.len; /* k is past the end */
}

/* ---- fast path ------------------------------------------------------------ */

__device__ __forceinline__ void wb_refill(WaveBits &s)
{
	WbFill r = wb_fill_slow(s.state, s.ring, s.pre, s.base, s.clean_total,
				s.chunk);
	s.wreg = r.wreg;
	s.wnext = (uint32_t)__shfl_down((int)r.wreg, 1);
	s.pre = r.pre;
	s.clean_total = wb_uniform(r.clean_total);
	s.chunk = wb_uniform(r.chunk);
}

__device__ __forceinline__ void wb_open(WaveBits &s, const uint8_t *nal,
					uint32_t len, uint8_t *ring,
					uint32_t *state, uint32_t lane)
{
	uintptr_t a = reinterpret_cast<uintptr_t>(nal);
	uint32_t mis = (uint32_t)(a & 3u);
	uintptr_t ab = a - mis;
	s.nal = nal;
	s.len = len;
	s.nchunks = (mis + len + WB_CHUNK - 1) / WB_CHUNK;
	s.chunk = 0;
	s.clean_total = 0;
	s.base = 0;
	s.bitpos = 0;
	s.ring = ring;
	s.state = state;
	/* branch-free (see slice_decode.hip): the 16-word state block is
	 * written by every lane, lane k storing word k & 15 */
	{
		uint32_t w = lane & 15u, v = 0;
		v = w == WBS_ABASE_LO ? (uint32_t)ab : v;
		v = w == WBS_ABASE_HI ? (uint32_t)((uint64_t)ab >> 32) : v;
		v = w == WBS_MIS ? mis : v;
		v = w == WBS_LEN ? len : v;
		v = w == WBS_NCHUNKS ? s.nchunks : v;
		v = w == WBS_CARRY ? 0xffffffffu : v;
		state[w] = v;
	}
	/* first chunk prefetch; lanes past the end re-read dword 0 and drop it */
	{
		uint32_t off = 4u * lane;
		bool in = off < mis + len;
		uint32_t v = *reinterpret_cast<const uint32_t *>(
			reinterpret_cast<const uint8_t *>(ab) + (in ? off : 0u));
		s.pre = in ? v : 0u;
	}
	wb_refill(s);
}

__device__ __forceinline__ bool wb_all_in(const WaveBits &s)
{
	return s.chunk >= s.nchunks;
}

/* Slide the window when the cursor has left its first half.  The window is
 * 2048 bits and no syntax structure parsed between two wb_sync() calls is
 * longer than ~960 bits (a residual block: 16 + 16 * 48 + 9 + 15 * 11), so
 * the peeks in between never look past it.  Callers place wb_sync() at the
 * top of their loops; it is the only place where vector work happens. */
__device__ __forceinline__ void wb_sync(WaveBits &s, uint32_t lane)
{
	(void)lane;
	if (s.bitpos >= WB_SLIDE_BITS) {
		s.base += WB_SLIDE_BITS / 8;
		s.bitpos -= WB_SLIDE_BITS;
		wb_refill(s);
	}
}

/* RBSP bits consumed since the start of the NAL */
__device__ __forceinline__ uint32_t wb_pos(const WaveBits &s)
{
	return s.base * 8u + s.bitpos;
}

/* total RBSP bits, valid once everything is ingested */
__device__ __forceinline__ uint32_t wb_total_bits(const WaveBits &s)
{
	return s.clean_total * 8u;
}

/* true if the cursor ran past the end of the NAL (lazy end-of-data test: the
 * ring is zero padded, so a parse that crossed the end is caught at the next
 * check instead of at every read; both ways the slice ends with -EIO) */
__device__ __forceinline__ bool wb_overrun(const WaveBits &s)
{
	return wb_all_in(s) && wb_pos(s) > wb_total_bits(s);
}

/* next 32 bits, MSB first */
__device__ __forceinline__ uint32_t wb_peek(WaveBits &s, uint32_t lane)
{
	(void)lane;
	uint32_t d = s.bitpos >> 5, sh = s.bitpos & 31u;
	uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)s.wreg, (int)d);
	uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)s.wnext, (int)d);
	uint64_t v = ((uint64_t)hi << 32) | lo;
	uint32_t r = (uint32_t)((v << sh) >> 32);
	/* opaque to the optimiser: comparisons of the result must stay 32-bit
	 * scalar compares instead of being folded into 64-bit compares of the
	 * unshifted pair, which only the vector ALU has */
	asm volatile("" : "+s"(r));
	return r;
}

__device__ __forceinline__ void wb_skip(WaveBits &s, uint32_t n)
{
	s.bitpos += n;
}

/* u(n), 1 <= n <= 32 */
__device__ __forceinline__ uint32_t wb_u(WaveBits &s, uint32_t lane, uint32_t n)
{
	uint32_t v = wb_peek(s, lane);
	wb_skip(s, n);
	return v >> (32u - n);
}

/* u(n) for ANY n, as the reference's reader behaves when a damaged parameter
 * set hands it a width outside 1..32 (include/h264/h264_bitstream.h:192-218):
 * n == 0 reads nothing and yields 0; n > 32 consumes all n bits and keeps the
 * last 32.  ok is cleared when the NAL cannot hold that many bits (the
 * reference runs out of data: -EIO); shorter overruns are caught lazily like
 * everywhere else.  Cold: only field widths taken from an SPS come here. */
__device__ __forceinline__ uint32_t wb_u_any(WaveBits &s, uint32_t lane,
					     uint32_t n, bool &ok)
{
	if (n == 0)
		return 0;
	if (n > 32) {
		uint32_t extra = n - 32u;
		if ((uint64_t)extra > (uint64_t)s.len * 8u) {
			ok = false;
			return 0;
		}
		while (extra) {
			uint32_t k = extra < 512u ? extra : 512u;
			wb_skip(s, k);
			wb_sync(s, lane);
			extra -= k;
		}
		n = 32;
	}
	return wb_u(s, lane, n);
}

__device__ __forceinline__ uint32_t wb_u1(WaveBits &s, uint32_t lane)
{
	return wb_u(s, lane, 1);
}

/* ue(v) (9.1).  ok is cleared when the code has 32 or more leading zeros,
 * which no value representable by the reference's reader produces. */
__device__ __forceinline__ uint32_t wb_ue(WaveBits &s, uint32_t lane, bool &ok)
{
	uint32_t v = wb_peek(s, lane);
	if (v >= 0x10000u) {
		/* at most 15 leading zeros: the whole code is in the peek */
		uint32_t lz = (uint32_t)__builtin_clz(v);
		wb_skip(s, 2 * lz + 1);
		return (v >> (31u - 2u * lz)) - 1u;
	}
	if (v == 0) {
		ok = false;
		return 0;
	}
	uint32_t lz = (uint32_t)__builtin_clz(v);
	wb_skip(s, lz);
	uint32_t u = wb_peek(s, lane);
	wb_skip(s, lz + 1);
	return (u >> (31u - lz)) - 1u;
}

/* se(v): odd k -> +(k+1)/2, even k -> -(k/2) */
__device__ __forceinline__ int32_t wb_se(WaveBits &s, uint32_t lane, bool &ok)
{
	uint32_t k = wb_ue(s, lane, ok);
	/* the reference's expression, with wrapping int32 arithmetic */
	int32_t t = (int32_t)(k + 1u);
	return (k & 1u) ? t / 2 : (int32_t)(0u - (uint32_t)t) / 2;
}

/* te(v) with range m (ref include/h264/h264_bitstream.h:293-304) */
__device__ __forceinline__ uint32_t wb_te(WaveBits &s, uint32_t lane, uint32_t m,
					  bool &ok)
{
	if (m == 1)
		return wb_u1(s, lane) ^ 1u;
	return wb_ue(s, lane, ok);
}

__device__ __forceinline__ uint32_t wb_raw_of_clean(const WaveBits &s, uint32_t k,
						    uint32_t lane)
{
	(void)lane;
	return wb_uniform(wb_raw_of_clean_slow(s.state, k, s.chunk));
}

/* h264_bs_more_rbsp_data at the cursor (ref src/h264_bitstream.c:325-355):
 * false when no bit is left, or when what is left is the stop bit, zero
 * alignment and then nothing but at most ONE raw zero byte. */
__device__ __forceinline__ bool wb_more_rbsp_data(WaveBits &s, uint32_t lane)
{
	uint32_t p = wb_pos(s);
	/* two or more RBSP bytes behind the current one: data, whatever the bits
	 * at the cursor look like (both exits below would say so) */
	if (s.clean_total >= (p >> 3) + 3u)
		return true;
	uint32_t v = wb_peek(s, lane);
	if (wb_all_in(s) && p >= wb_total_bits(s))
		return false;
	uint32_t rem = 8u - (p & 7u); /* bits left in this byte */
	if ((v >> (32u - rem)) != (1u << (rem - 1u)))
		return true;
	uint32_t k = p >> 3; /* RBSP byte holding the candidate stop bit */
	if (s.clean_total >= k + 3u)
		return true; /* two more RBSP bytes follow: data */
	/* here every raw byte is in (the fill keeps 512 bytes ahead otherwise) */
	uint32_t raw = wb_raw_of_clean(s, k, lane);
	uint32_t left = s.len - (raw + 1u);
	if (left == 0)
		return false;
	if (left >= 2)
		return true;
	return wb_uniform(s.nal[s.len - 1u]) != 0;
}

#endif /* H264_AMD_WAVE_BITS_HIP_H */

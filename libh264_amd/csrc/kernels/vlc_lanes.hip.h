/**
 * vlc_lanes.hip.h -- CAVLC prefix-code tables resident in the lanes of a wave.
 *
 * The reference decodes every CAVLC code word one bit at a time against
 * [numbits][rawcode] tables (src/h264_slice_data.c:33-72, tables :81-831).
 * Here each code word of a table lives in one LANE as (mask, code): all 64
 * candidates are tested against the next 32 bits in one v_and + v_cmp, the
 * ballot is reduced with s_ff1 and the winner's length/value come back through
 * v_readlane.  No LDS or memory access on the decode path, and the result is
 * wave-uniform (SGPR) so the syntax walk stays on the scalar unit.
 *
 * Row layout of struct dev_vlc_tables (64 lanes per row; built on the host by
 * vlc_build_tables() in engine.hip from ../common/h264_cavlc_tables.inc):
 *   coeff_token   rows 3c+0 mask, 3c+1 code, 3c+2 value (TrailingOnes<<5 |
 *                 TotalCoeff) for c = 0,1,2 (0<=nC<2, 2<=nC<4, 4<=nC<8);
 *                 rows 9..11: chroma DC, 4:2:0 in lanes 0..13, 4:2:2 in 16..45
 *   total_zeros   4x4 blocks, tzVlcIndex t has 17-t code words whose lane
 *                 order is the value: rows 12,13 t=1..4; 14,15 t=5..9;
 *                 16,17 t=10..15, tables packed back to back
 *   total_zeros   chroma DC: rows 18,19; 2x2 (t=1..3, 5-t words) from lane 0,
 *                 2x4 (t=1..7, 9-t words) from lane 16
 *   run_before    rows 20,21; zerosLeft class z=1..6 has z+1 words, class 7
 *                 (zerosLeft > 6) has 15, packed back to back
 * mask/code are left aligned in 32 bits; unused lanes hold mask 0, code 1.
 */
#ifndef H264_AMD_VLC_LANES_HIP_H
#define H264_AMD_VLC_LANES_HIP_H

#include "dev_params.h"
#include "wave_bits.hip.h"

#define VLC_ROWS 22

struct VlcLanes {
	uint32_t r[VLC_ROWS]; /* one VGPR per row */
};

__device__ __forceinline__ void vlc_load(VlcLanes &t, const dev_vlc_tables *src,
					 uint32_t lane)
{
#pragma unroll
	for (int i = 0; i < 20; i++)
		t.r[i] = src->e[i][lane];
	/* run_before, zerosLeft 1..6: entry = len << 4 | run for each value of
	 * the next 3 bits, 8 entries packed into the 64-bit word of lane
	 * zerosLeft (Table 9-10):
	 *  1: 1 0            2: 1 01 00          3: 11 10 01 00
	 *  4: 11 10 01 001 000                   5: 11 10 011 010 001 000
	 *  6: 11 000 001 011 010 101 100 (runs 0..6) */
	uint64_t tb = lane == 1   ? 0x1010101011111111ull
		      : lane == 2 ? 0x1010101021212222ull
		      : lane == 3 ? 0x2020212122222323ull
		      : lane == 4 ? 0x2020212122223334ull
		      : lane == 5 ? 0x2020212132333435ull
		      : lane == 6 ? 0x2020353633343231ull
				  : 0ull;
	t.r[20] = (uint32_t)tb;
	t.r[21] = (uint32_t)(tb >> 32);
}

/* A table row as an opaque vector value: without this the optimiser merges
 * "cond ? r[i] : r[j]" into r[cond ? i : j], a dynamic index that forces the
 * whole row array out of registers into scratch memory. */
__device__ __forceinline__ uint32_t vlc_row(uint32_t x)
{
	asm("" : "+v"(x));
	return x;
}

/* Decoders below never branch on a malformed code word: they return a "bad"
 * flag (0 / 1) for the caller to OR into its error word and hand back benign
 * values, so that the parse loop carries plain 32-bit integers instead of
 * booleans (which the backend would keep as 64-bit lane masks). */

/* first lane inside `group` whose code word prefixes `bits`; -1 if none
 * (s_ff1_i32_b64 of zero) */
__device__ __forceinline__ int vlc_find(uint32_t mask, uint32_t code,
					uint32_t bits, wb_mask_t group)
{
	wb_mask_t m = __ballot((bits & mask) == code) & group;
	return __builtin_ffsll((long long)m) - 1;
}

__device__ __forceinline__ uint32_t vlc_len(uint32_t mask_row, int lane_hit)
{
	return (uint32_t)__builtin_popcount(
		(uint32_t)__builtin_amdgcn_readlane((int)mask_row, lane_hit));
}

__device__ __forceinline__ wb_mask_t vlc_group(uint32_t base, uint32_t n)
{
	return ((1ull << n) - 1ull) << base;
}

/* coeff_token for the tables selected by nC.  chroma_dc: 0 no, 1 4:2:0
 * (nC = -1), 2 4:2:2 (nC = -2).  Returns the bad flag. */
__device__ __forceinline__ uint32_t vlc_coeff_token(const VlcLanes &t, WaveBits &s,
						    uint32_t lane, uint32_t nC,
						    uint32_t chroma_dc, uint32_t &t1,
						    uint32_t &tc)
{
	uint32_t bits = wb_peek(s, lane);
	if (chroma_dc == 0 && nC >= 8) {
		/* 6-bit fixed length code: xxxxyy, TotalCoeff = xxxx + 1,
		 * TrailingOnes = yy; 0000 11 means no coefficient */
		uint32_t c = bits >> 26;
		uint32_t none = c == 3 ? 1u : 0u;
		wb_skip(s, 6);
		t1 = none ? 0u : c & 3u;
		tc = none ? 0u : (c >> 2) + 1u;
		/* 0000 10 and 0001 11 are not code words */
		uint32_t bad = t1 > tc ? 1u : 0u;
		tc = bad ? 0u : tc;
		return bad;
	}
	uint32_t mrow, crow, vrow;
	wb_mask_t g;
	if (chroma_dc) {
		mrow = vlc_row(t.r[9]);
		crow = vlc_row(t.r[10]);
		vrow = vlc_row(t.r[11]);
		g = chroma_dc == 1 ? vlc_group(0, 14) : vlc_group(16, 30);
	} else {
		mrow = nC < 2 ? vlc_row(t.r[0]) : nC < 4 ? vlc_row(t.r[3]) : vlc_row(t.r[6]);
		crow = nC < 2 ? vlc_row(t.r[1]) : nC < 4 ? vlc_row(t.r[4]) : vlc_row(t.r[7]);
		vrow = nC < 2 ? vlc_row(t.r[2]) : nC < 4 ? vlc_row(t.r[5]) : vlc_row(t.r[8]);
		g = vlc_group(0, 62);
	}
	int hit = vlc_find(mrow, crow, bits, g);
	uint32_t bad = (uint32_t)hit >> 31;
	uint32_t len = vlc_len(mrow, hit & 63);
	uint32_t val = (uint32_t)__builtin_amdgcn_readlane((int)vrow, hit & 63);
	wb_skip(s, bad ? 0u : len);
	t1 = bad ? 0u : (val >> 5) & 3u;
	tc = bad ? 0u : val & 31u;
	return bad;
}

/* total_zeros; maxnum selects the table family (4: chroma DC 2x2, 8: chroma
 * DC 2x4, otherwise 4x4 with tzVlcIndex = tc).  Caller guarantees
 * 1 <= tc < maxnum.  Returns the bad flag. */
__device__ __forceinline__ uint32_t vlc_total_zeros(const VlcLanes &t, WaveBits &s,
						    uint32_t lane, uint32_t tc,
						    uint32_t maxnum, uint32_t &tz)
{
	uint32_t bits = wb_peek(s, lane);
	uint32_t base, n, bad = 0, mrow, crow;
	if (maxnum == 4) {
		bad = tc > 3 ? 1u : 0u;
		base = 5u * (tc - 1u) - tc * (tc - 1u) / 2u;
		n = 5u - tc;
		mrow = vlc_row(t.r[18]);
		crow = vlc_row(t.r[19]);
	} else if (maxnum == 8) {
		bad = tc > 7 ? 1u : 0u;
		base = 16u + 9u * (tc - 1u) - tc * (tc - 1u) / 2u;
		n = 9u - tc;
		mrow = vlc_row(t.r[18]);
		crow = vlc_row(t.r[19]);
	} else {
		/* cum(t) = lanes used by tables 1..t-1 */
		uint32_t cum = 17u * (tc - 1u) - tc * (tc - 1u) / 2u;
		bad = tc > 15 ? 1u : 0u;
		n = 17u - tc;
		base = tc <= 4 ? cum : tc <= 9 ? cum - 58u : cum - 108u;
		mrow = tc <= 4 ? vlc_row(t.r[12]) : tc <= 9 ? vlc_row(t.r[14]) : vlc_row(t.r[16]);
		crow = tc <= 4 ? vlc_row(t.r[13]) : tc <= 9 ? vlc_row(t.r[15]) : vlc_row(t.r[17]);
	}
	base = bad ? 0u : base;
	n = bad ? 1u : n;
	int hit = vlc_find(mrow, crow, bits, vlc_group(base, n));
	bad |= (uint32_t)hit >> 31;
	uint32_t len = vlc_len(mrow, hit & 63);
	wb_skip(s, bad ? 0u : len);
	tz = bad ? 0u : (uint32_t)hit - base;
	return bad;
}

/* run_before with zeros_left >= 1.  Table 9-10 in closed form: for
 * zerosLeft <= 6 the code words have at most 3 bits and sit in a 3-bit indexed
 * table of (len << 4 | run) bytes packed into one 64-bit constant per class;
 * for zerosLeft > 6 the code is 3 bits (111 .. 001 -> 0 .. 6) or a unary tail
 * (0001 -> 7, 00001 -> 8, ...). */
__device__ __forceinline__ uint32_t vlc_run_before(const VlcLanes &t, WaveBits &s,
						   uint32_t lane,
						   uint32_t zeros_left,
						   uint32_t &run)
{
	uint32_t bits = wb_peek(s, lane);
	uint32_t top3 = bits >> 29;
	if (zeros_left > 6) {
		uint32_t lz = (uint32_t)__builtin_clz(bits | 0x00100000u);
		uint32_t bad = (bits >> 21) == 0 ? 1u : 0u; /* > 11 bits */
		run = top3 ? 7u - top3 : lz + 4u;
		wb_skip(s, bad ? 0u : (top3 ? 3u : lz + 1u));
		run = bad ? 0u : run;
		return bad;
	}
	/* lane z of rows 20/21 holds the table of class z (vlc_load) */
	uint64_t tb = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(
			       (int)t.r[21], (int)zeros_left)
		       << 32) |
		      (uint32_t)__builtin_amdgcn_readlane((int)t.r[20],
							  (int)zeros_left);
	uint32_t e = (uint32_t)(tb >> (8u * top3)) & 0xffu;
	wb_skip(s, e >> 4);
	run = e & 15u;
	return 0;
}

#endif /* H264_AMD_VLC_LANES_HIP_H */

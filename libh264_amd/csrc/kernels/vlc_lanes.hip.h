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
	for (int i = 0; i < VLC_ROWS; i++)
		t.r[i] = src->e[i][lane];
}

/* first lane inside `group` whose code word prefixes `bits`; -1 if none */
__device__ __forceinline__ int vlc_find(uint32_t mask, uint32_t code,
					uint32_t bits, wb_mask_t group)
{
	wb_mask_t m = __ballot((bits & mask) == code) & group;
	return m ? (int)__builtin_ctzll(m) : -1;
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

/* coeff_token for the tables selected by nC; returns false on an invalid code.
 * chroma_dc: 0 no, 1 4:2:0 (nC = -1), 2 4:2:2 (nC = -2). */
__device__ __forceinline__ bool vlc_coeff_token(const VlcLanes &t, WaveBits &s,
						uint32_t lane, uint32_t nC,
						uint32_t chroma_dc, uint32_t &t1,
						uint32_t &tc)
{
	uint32_t bits = wb_peek(s, lane);
	int hit;
	uint32_t len, val;
	if (chroma_dc) {
		wb_mask_t g = chroma_dc == 1 ? vlc_group(0, 14) : vlc_group(16, 30);
		hit = vlc_find(t.r[9], t.r[10], bits, g);
		if (hit < 0)
			return false;
		len = vlc_len(t.r[9], hit);
		val = (uint32_t)__builtin_amdgcn_readlane((int)t.r[11], hit);
	} else if (nC >= 8) {
		/* 6-bit fixed length code: xxxxyy, TotalCoeff = xxxx + 1,
		 * TrailingOnes = yy; 0000 11 means no coefficient */
		uint32_t c = bits >> 26;
		wb_skip(s, 6);
		if (c == 3) {
			t1 = 0;
			tc = 0;
			return true;
		}
		t1 = c & 3u;
		tc = (c >> 2) + 1u;
		return t1 <= tc; /* 0000 10 and 0001 11 are not code words */
	} else {
		const wb_mask_t g = vlc_group(0, 62);
		if (nC < 2) {
			hit = vlc_find(t.r[0], t.r[1], bits, g);
			if (hit < 0)
				return false;
			len = vlc_len(t.r[0], hit);
			val = (uint32_t)__builtin_amdgcn_readlane((int)t.r[2], hit);
		} else if (nC < 4) {
			hit = vlc_find(t.r[3], t.r[4], bits, g);
			if (hit < 0)
				return false;
			len = vlc_len(t.r[3], hit);
			val = (uint32_t)__builtin_amdgcn_readlane((int)t.r[5], hit);
		} else {
			hit = vlc_find(t.r[6], t.r[7], bits, g);
			if (hit < 0)
				return false;
			len = vlc_len(t.r[6], hit);
			val = (uint32_t)__builtin_amdgcn_readlane((int)t.r[8], hit);
		}
	}
	wb_skip(s, len);
	t1 = (val >> 5) & 3u;
	tc = val & 31u;
	return true;
}

/* total_zeros; maxnum selects the table family (4: chroma DC 2x2, 8: chroma
 * DC 2x4, otherwise 4x4 with tzVlcIndex = tc).  Caller guarantees
 * 1 <= tc < maxnum. */
__device__ __forceinline__ bool vlc_total_zeros(const VlcLanes &t, WaveBits &s,
						uint32_t lane, uint32_t tc,
						uint32_t maxnum, uint32_t &tz)
{
	uint32_t bits = wb_peek(s, lane);
	uint32_t base, n, len;
	int hit;
	if (maxnum == 4) {
		if (tc > 3)
			return false;
		base = 5u * (tc - 1u) - tc * (tc - 1u) / 2u;
		n = 5u - tc;
		hit = vlc_find(t.r[18], t.r[19], bits, vlc_group(base, n));
		if (hit < 0)
			return false;
		len = vlc_len(t.r[18], hit);
	} else if (maxnum == 8) {
		if (tc > 7)
			return false;
		base = 16u + 9u * (tc - 1u) - tc * (tc - 1u) / 2u;
		n = 9u - tc;
		hit = vlc_find(t.r[18], t.r[19], bits, vlc_group(base, n));
		if (hit < 0)
			return false;
		len = vlc_len(t.r[18], hit);
	} else {
		/* cum(t) = lanes used by tables 1..t-1 */
		uint32_t cum = 17u * (tc - 1u) - tc * (tc - 1u) / 2u;
		if (tc > 15)
			return false;
		n = 17u - tc;
		if (tc <= 4) {
			base = cum;
			hit = vlc_find(t.r[12], t.r[13], bits, vlc_group(base, n));
			if (hit < 0)
				return false;
			len = vlc_len(t.r[12], hit);
		} else if (tc <= 9) {
			base = cum - 58u;
			hit = vlc_find(t.r[14], t.r[15], bits, vlc_group(base, n));
			if (hit < 0)
				return false;
			len = vlc_len(t.r[14], hit);
		} else {
			base = cum - 108u;
			hit = vlc_find(t.r[16], t.r[17], bits, vlc_group(base, n));
			if (hit < 0)
				return false;
			len = vlc_len(t.r[16], hit);
		}
	}
	wb_skip(s, len);
	tz = (uint32_t)hit - base;
	return true;
}

/* run_before with zeros_left >= 1 */
__device__ __forceinline__ bool vlc_run_before(const VlcLanes &t, WaveBits &s,
					       uint32_t lane, uint32_t zeros_left,
					       uint32_t &run)
{
	uint32_t bits = wb_peek(s, lane);
	uint32_t z = zeros_left < 7u ? zeros_left : 7u;
	uint32_t base = (z - 1u) * (z + 2u) / 2u;
	uint32_t n = z < 7u ? z + 1u : 15u;
	int hit = vlc_find(t.r[20], t.r[21], bits, vlc_group(base, n));
	if (hit < 0)
		return false;
	wb_skip(s, vlc_len(t.r[20], hit));
	run = (uint32_t)hit - base;
	return true;
}

#endif /* H264_AMD_VLC_LANES_HIP_H */

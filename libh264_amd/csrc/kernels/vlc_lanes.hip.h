/**
 * vlc_lanes.hip.h -- CAVLC prefix codes matched by all lanes of a wave at once.
 *
 * The reference decodes every CAVLC code word one bit at a time against
 * [numbits][rawcode] tables (src/h264_slice_data.c:33-72, tables :81-831).
 * Here a table is a row of (mask, cw) entries in LDS, one entry per LANE: every
 * lane tests its code word against the next 32 bits with one xor + and + cmp,
 * the ballot is reduced with s_ff1 and the winner's payload comes back through
 * v_readlane.  The result is wave-uniform (SGPR), so the syntax walk stays on
 * the scalar unit.  Which table applies (nC class, tzVlcIndex) only changes the
 * LDS address of the row: no table-selection branches or selects.
 *
 * struct dev_vlc_tables (built by vlc_build_tables() in engine.hip from
 * ../common/h264_cavlc_tables.inc), copied to LDS once per workgroup:
 *   ct[class][64]  coeff_token (Table 9-5); class 0: 0 <= nC < 2, 1: 2 <= nC
 *                  < 4, 2: 4 <= nC < 8, 3: 8 <= nC (the 6-bit FLC as a table),
 *                  4: chroma DC 4:2:0 (nC = -1), 5: chroma DC 4:2:2 (nC = -2);
 *                  payload = len | TrailingOnes << 5 | TotalCoeff << 7
 *   tz[k][16]      total_zeros (Tables 9-7 .. 9-9), entry index = value;
 *                  k = tzVlcIndex - 1 for 4x4 blocks, 14 + tzVlcIndex for
 *                  chroma DC 2x2, 17 + tzVlcIndex for chroma DC 2x4;
 *                  payload = len
 * coeff_token rows end in catch-all entries (payload DEV_VLC_BAD): something
 * always matches.  total_zeros entries without a code word repeat entry 0 of
 * their table.
 * run_before (Table 9-10) is decoded in closed form, see vlc_run_before().
 */
#ifndef H264_AMD_VLC_LANES_HIP_H
#define H264_AMD_VLC_LANES_HIP_H

#include "dev_params.h"
#include "wave_bits.hip.h"

struct VlcTabs {
	const uint8_t *ct; /* LDS, uniform */
	const uint8_t *tz; /* LDS, uniform */
	uint32_t off64; /* per lane: 8 * lane */
	uint32_t off16; /* per lane: 8 * (lane & 15) */
	uint32_t rb_lo, rb_hi; /* per lane: run_before table of class `lane` */
};

/* workgroup-wide copy of the tables into LDS; the caller synchronises */
__device__ __forceinline__ void vlc_stage_tables(uint8_t *lds_tabs,
						 const dev_vlc_tables *src,
						 uint32_t tid, uint32_t nthreads)
{
	const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
	uint32_t *d = reinterpret_cast<uint32_t *>(lds_tabs);
	for (uint32_t i = tid; i < DEV_VLC_LDS_BYTES / 4u; i += nthreads)
		d[i] = s[i];
}

__device__ __forceinline__ void vlc_open(VlcTabs &t, const uint8_t *lds_tabs,
					 uint32_t lane)
{
	t.ct = lds_tabs;
	t.tz = lds_tabs + DEV_VLC_CT_CLASSES * 64u * 8u;
	t.off64 = 8u * lane;
	t.off16 = 8u * (lane & 15u);
	/* run_before, zerosLeft 1..6: entry = len << 4 | (zerosLeft - run), i.e.
	 * what is left behind the run, for each value of the next 3 bits, 8
	 * entries packed into the 64-bit word of lane zerosLeft (Table 9-10):
	 *  1: 1 0            2: 1 01 00          3: 11 10 01 00
	 *  4: 11 10 01 001 000                   5: 11 10 011 010 001 000
	 *  6: 11 000 001 011 010 101 100 (runs 0..6) */
	uint64_t tb = lane == 1   ? 0x1111111110101010ull
		      : lane == 2 ? 0x1212121221212020ull
		      : lane == 3 ? 0x2323222221212020ull
		      : lane == 4 ? 0x2424232322223130ull
		      : lane == 5 ? 0x2525242433323130ull
		      : lane == 6 ? 0x2626313033323435ull
				  : 0ull;
	t.rb_lo = (uint32_t)tb;
	t.rb_hi = (uint32_t)(tb >> 32);
}

/* Decoders never branch on a malformed code word: they return a "bad" flag
 * (0 / 1) for the caller to OR into its error word and hand back benign values,
 * so that the parse loop carries plain 32-bit integers instead of booleans
 * (which the backend would keep as 64-bit lane masks). */

/* lowest entry of the row at `row` (+ this lane's offset) whose code word
 * prefixes `bits`: returns its cw, or bad = 1 */
__device__ __forceinline__ uint32_t vlc_match(const uint8_t *row, uint32_t off,
					      uint32_t bits, uint32_t &bad,
					      int &hit)
{
	const uint2 e = *reinterpret_cast<const uint2 *>(row + off);
	wb_mask_t m = __ballot(((bits ^ e.y) & e.x) == 0u);
	/* s_ff1 of zero is -1: "no entry matches" needs no compare */
	asm("s_ff1_i32_b64 %0, %1" : "=s"(hit) : "s"(m));
	bad = (uint32_t)hit >> 31;
	return (uint32_t)__builtin_amdgcn_readlane((int)e.y, hit & 63);
}

/* the same for a row that ends in catch-all entries: something always matches */
__device__ __forceinline__ uint32_t vlc_match_total(const uint8_t *row,
						    uint32_t off, uint32_t bits)
{
	const uint2 e = *reinterpret_cast<const uint2 *>(row + off);
	wb_mask_t m = __ballot(((bits ^ e.y) & e.x) == 0u);
	/* m != 0: the catch-all entries */
	return (uint32_t)__builtin_amdgcn_readlane((int)e.y, __builtin_ctzll(m));
}

/* class of coeff_token table for a wave-uniform nC = 0..8+: 0 0 1 1 2 2 2 2 3,
 * two bits each of 0x3AA50, as a scalar bit field extract with the operand
 * 2 c | 2 << 16 (three scalar operations; shift, shift and mask are four) */
__device__ __forceinline__ uint32_t vlc_ct_class(uint32_t nC)
{
	uint32_t c = nC < 8u ? nC : 8u;
	uint32_t cls;
	asm("s_lshl1_add_u32 %0, %1, 0x20000\n\ts_bfe_u32 %0, 0x3aa50, %0"
	    : "=&s"(cls)
	    : "s"(c));
	return cls;
}

/* total_zeros; maxnum selects the table family (4: chroma DC 2x2, 8: chroma
 * DC 2x4, otherwise 4x4 with tzVlcIndex = tc).  Caller guarantees
 * 1 <= tc < maxnum.  Returns the bad flag. */
__device__ __forceinline__ uint32_t vlc_total_zeros(const VlcTabs &t, WaveBits &s,
						    uint32_t lane, uint32_t tc,
						    uint32_t maxnum, uint32_t &tz)
{
	uint32_t bits = wb_peek(s, lane);
	/* 1 <= tc < maxnum and maxnum in {4, 8, 15, 16} (the callers' block
	 * constants) keep k inside the 25 tables: tc - 1 <= 14, 14 + tc <= 17,
	 * 17 + tc <= 24 */
	uint32_t k = maxnum == 4 ? 14u + tc : maxnum == 8 ? 17u + tc : tc - 1u;
	uint32_t bad;
	int hit;
	uint32_t cw = vlc_match(t.tz + 128u * k, t.off16, bits, bad, hit);
	uint32_t okm = bad - 1u;
	wb_skip(s, cw & 31u & okm);
	tz = (uint32_t)hit & okm;
	return bad;
}

/* run_before with zeros_left >= 1.  Table 9-10 in closed form: for
 * zerosLeft <= 6 the code words have at most 3 bits and sit in a 3-bit indexed
 * table of (len << 4 | run) bytes packed into one 64-bit constant per class;
 * for zerosLeft > 6 the code is 3 bits (111 .. 001 -> 0 .. 6) or a unary tail
 * (0001 -> 7, 00001 -> 8, ...). */
__device__ __forceinline__ uint32_t vlc_run_before(const VlcTabs &t, WaveBits &s,
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
	uint64_t tb = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(
			       (int)t.rb_hi, (int)zeros_left)
		       << 32) |
		      (uint32_t)__builtin_amdgcn_readlane((int)t.rb_lo,
							  (int)zeros_left);
	uint32_t e = (uint32_t)(tb >> (8u * top3)) & 0xffu;
	wb_skip(s, e >> 4);
	run = zeros_left - (e & 15u);
	return 0;
}

#endif /* H264_AMD_VLC_LANES_HIP_H */

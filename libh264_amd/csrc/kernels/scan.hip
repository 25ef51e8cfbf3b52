/**
 * scan.hip -- Annex-B NAL unit delimiting on gfx950, one pass over the bytes.
 *
 * Replaces the byte-at-a-time loops of h264_find_start_code /
 * h264_find_end_code / h264_find_nalu (ref src/h264_bitstream.c:87-184) and the
 * NAL loop of h264_reader_parse (ref src/h264_reader.c:133-140) for a whole
 * buffer.  Data-parallel formulation, with
 *   Z(i) = b[i] == 0 && b[i+1] == 0,
 *   E(i) = Z(i) && b[i+2] <= 1      "end code"   (00 00 00 or 00 00 01)
 *   S(i) = Z(i) && b[i+2] == 1      "start code" (every S is also an E)
 * every S(i) opens a NAL unit whose payload starts at i + 3, and the NAL ends
 * at the first E position >= its start (there is none in (i, i+3)), or at the
 * end of the buffer.  A 4-byte start code is S(i+1) preceded by a zero, which
 * is itself an E: it only ends the previous NAL one byte earlier, exactly as
 * the reference does.
 *
 * Kernel structure (HBM bound; every input byte is read once):
 *   k_scan_mark   each WAVE walks one 16 KiB segment in 1 KiB steps (64 lanes
 *                 x 16 B, four steps in flight), builds E/S bit masks with
 *                 byte-parallel arithmetic, and -- because at most one NAL is
 *                 open at any time -- resolves starts and ends with a scalar
 *                 state machine that only runs on the (rare) steps containing
 *                 a pattern.  NAL records go to a per-launch list, a NAL still
 *                 open at the end of the segment is left unresolved.
 *   k_scan_place  exclusive scan of the per-segment counts, resolves open NALs
 *                 with the first E of the following segments and scatters the
 *                 records to their final, stream-ordered position.
 */
#ifndef H264_AMD_SCAN_HIP
#define H264_AMD_SCAN_HIP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <h264_amd.h>

#define SCAN_SEG_BYTES 16384u /* per wave */
#define SCAN_WAVES_PER_BLOCK 4u
#define SCAN_NONE 0xffffffffffffffffull

struct scan_rec {
	uint64_t start, end; /* end == SCAN_NONE: open at the end of the segment */
	uint32_t seg, local; /* segment and ordinal inside it */
};

struct scan_scratch {
	scan_rec *d_recs;
	size_t cap_recs;
	uint32_t *d_seg_count; /* NALs opened per segment */
	uint64_t *d_seg_first_e; /* first E position per segment or SCAN_NONE */
	uint32_t *d_seg_base;
	size_t cap_segs;
	uint32_t *d_counter; /* [0] records appended, [1] total NALs */
};

/* 0x80 in every byte of x that is zero (exact, no cross-byte carries) */
__device__ __forceinline__ uint32_t scan_zero_bytes(uint32_t x)
{
	uint32_t t = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
	return ~(t | x | 0x7f7f7f7fu);
}

/* gather the four 0x80 flags of f into bits 0..3 */
__device__ __forceinline__ uint32_t scan_movemask(uint32_t f)
{
	return (((f >> 7) * 0x00204081u) >> 21) & 0xfu;
}

__device__ __forceinline__ uint4 scan_load16(const uint8_t *buf, uint64_t off,
					     uint64_t len)
{
	if (off + 16 <= len)
		return *reinterpret_cast<const uint4 *>(buf + off);
	/* ragged tail: bytes past the end read as 0xff (never part of a code) */
	uint32_t w[4];
#pragma unroll
	for (int k = 0; k < 4; k++) {
		uint32_t v = 0xffffffffu;
		for (int j = 0; j < 4; j++) {
			uint64_t p = off + 4u * k + j;
			if (p < len)
				v = (v & ~(0xffu << (8 * j))) |
				    ((uint32_t)buf[p] << (8 * j));
		}
		w[k] = v;
	}
	return make_uint4(w[0], w[1], w[2], w[3]);
}

struct scan_mark_args {
	const uint8_t *buf;
	uint64_t len;
	uint32_t nsegs;
	scan_rec *recs;
	uint32_t cap_recs;
	uint32_t *seg_count;
	uint64_t *seg_first_e;
	uint32_t *counter;
};

extern "C" __global__ void __launch_bounds__(64 * SCAN_WAVES_PER_BLOCK)
h264_amd_k_scan_mark(scan_mark_args a)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t seg = (uint32_t)__builtin_amdgcn_readfirstlane(
		(int)(blockIdx.x * SCAN_WAVES_PER_BLOCK + (threadIdx.x >> 6)));
	if (seg >= a.nsegs)
		return;
	const uint64_t lo = (uint64_t)seg * SCAN_SEG_BYTES;
	uint64_t hi = lo + SCAN_SEG_BYTES;
	if (hi > a.len)
		hi = a.len;

	/* the two bytes before the segment (0xff before the buffer) */
	uint32_t carry = 0xffffu; /* bits 0-7: byte lo-2, 8-15: byte lo-1 */
	if (lo >= 2)
		carry = (uint32_t)a.buf[lo - 2] | ((uint32_t)a.buf[lo - 1] << 8);
	carry = (uint32_t)__builtin_amdgcn_readfirstlane((int)carry);

	uint64_t pending = SCAN_NONE; /* payload start of the open NAL */
	uint64_t first_e = SCAN_NONE;
	uint32_t opened = 0; /* NALs opened in this segment */

	for (uint64_t pos = lo; pos < hi; pos += 4096u) {
		uint4 q[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			uint64_t off = pos + 1024u * u + 16u * lane;
			q[u] = off < hi ? scan_load16(a.buf, off, a.len)
					: make_uint4(~0u, ~0u, ~0u, ~0u);
		}
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const uint64_t cbase = pos + 1024u * u;
			if (cbase >= hi)
				break;
			const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
			uint32_t z16 = 0, le16 = 0;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				z16 |= scan_movemask(scan_zero_bytes(w[k])) << (4 * k);
				le16 |= scan_movemask(scan_zero_bytes(w[k] & 0xfefefefeu))
					<< (4 * k);
			}
			/* zero flags of the two bytes before this lane's 16 */
			uint32_t zup = (uint32_t)__shfl_up((int)z16, 1) >> 14;
			uint32_t zc = ((carry & 0xffu) == 0 ? 1u : 0u) |
				      ((carry >> 8) == 0 ? 2u : 0u);
			uint32_t zprev = lane == 0 ? zc : zup;
			uint32_t z18 = (z16 << 2) | zprev; /* bit j+2 = byte j */
			/* code ENDING at byte j: bytes j-2, j-1 zero, byte j <= 1 */
			uint32_t e16 = z18 & (z18 >> 1) & le16;
			uint32_t s16 = e16 & ~z16;
			/* next step's carry: last two bytes of lane 63 */
			carry = (uint32_t)__builtin_amdgcn_readlane((int)(w[3] >> 16), 63);

			unsigned long long em = __ballot(e16 != 0);
			while (em) {
				int l = __builtin_ctzll(em);
				em &= em - 1;
				uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)e16, l);
				uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)s16, l);
				while (e) {
					uint32_t j = (uint32_t)__builtin_ctz(e);
					e &= e - 1;
					/* position of the code's first byte */
					uint64_t p = cbase + 16u * (uint32_t)l + j - 2u;
					if (first_e == SCAN_NONE)
						first_e = p;
					if (pending != SCAN_NONE) {
						if (lane == 0) {
							uint32_t slot = atomicAdd(&a.counter[0], 1u);
							if (slot < a.cap_recs) {
								scan_rec r;
								r.start = pending;
								r.end = p;
								r.seg = seg;
								r.local = opened - 1u;
								a.recs[slot] = r;
							}
						}
						pending = SCAN_NONE;
					}
					if ((s >> j) & 1u) {
						pending = p + 3u;
						opened++;
					}
				}
			}
		}
	}
	if (lane == 0) {
		if (pending != SCAN_NONE) {
			uint32_t slot = atomicAdd(&a.counter[0], 1u);
			if (slot < a.cap_recs) {
				scan_rec r;
				r.start = pending;
				r.end = SCAN_NONE;
				r.seg = seg;
				r.local = opened - 1u;
				a.recs[slot] = r;
			}
		}
		a.seg_count[seg] = opened;
		a.seg_first_e[seg] = first_e;
	}
}

/* Single block: exclusive scan of seg_count into seg_base, total into
 * counter[1]. */
extern "C" __global__ void __launch_bounds__(1024)
h264_amd_k_scan_prefix(const uint32_t *seg_count, uint32_t *seg_base,
		       uint32_t nsegs, uint32_t *counter)
{
	__shared__ uint32_t part[1024];
	const uint32_t t = threadIdx.x;
	const uint32_t per = (nsegs + 1023u) / 1024u;
	uint32_t b = t * per, sum = 0;
	for (uint32_t i = b; i < b + per && i < nsegs; i++)
		sum += seg_count[i];
	part[t] = sum;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) {
		uint32_t v = t >= d ? part[t - d] : 0;
		__syncthreads();
		part[t] += v;
		__syncthreads();
	}
	uint32_t run = part[t] - sum;
	for (uint32_t i = b; i < b + per && i < nsegs; i++) {
		seg_base[i] = run;
		run += seg_count[i];
	}
	if (t == 1023)
		counter[1] = part[1023];
}

/* One thread per record: resolve an open end with the first E of a following
 * segment and scatter to the stream-ordered slot. */
extern "C" __global__ void __launch_bounds__(256)
h264_amd_k_scan_place(const scan_rec *recs, uint32_t nrecs,
		      const uint32_t *seg_base, const uint64_t *seg_first_e,
		      uint32_t nsegs, uint64_t len, struct h264_amd_nal *out,
		      uint32_t cap)
{
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= nrecs)
		return;
	scan_rec r = recs[i];
	uint64_t end = r.end;
	if (end == SCAN_NONE) {
		end = len;
		for (uint32_t s = r.seg + 1; s < nsegs; s++) {
			uint64_t f = seg_first_e[s];
			if (f != SCAN_NONE) {
				end = f;
				break;
			}
		}
	}
	uint32_t idx = seg_base[r.seg] + r.local;
	if (idx < cap) {
		out[idx].start = r.start;
		out[idx].end = end;
	}
}

static inline void scan_scratch_free(scan_scratch *s)
{
	hipFree(s->d_recs);
	hipFree(s->d_seg_count);
	hipFree(s->d_seg_first_e);
	hipFree(s->d_seg_base);
	hipFree(s->d_counter);
	memset(s, 0, sizeof(*s));
}

#endif /* H264_AMD_SCAN_HIP */

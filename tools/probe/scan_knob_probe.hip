// Scan-mark kernel with pieces switched off, to find what keeps it below the
// pure-read rate (hbm_read_probe).  Generated from kernels/scan.hip by
// tools/probe/make_scan_probe.py; knobs: 1 no carry load, 2 no tail stores,
// 4 no pattern work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#define SCAN_BLOCK_BYTES 4096u
#define SCAN_WAVES_PER_BLOCK 4u
#define SCAN_NONE 0xffffffffffffffffull
struct scan_rec { uint64_t start, end; uint32_t seg, local; };
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
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
	uint32_t seg_bytes; /* multiple of SCAN_BLOCK_BYTES */
	scan_rec *recs;
	uint32_t cap_recs;
	uint32_t *seg_count;
	uint64_t *seg_first_e;
	uint32_t *counter;
};

/* wave-uniform walk state of one segment */
struct scan_state {
	uint32_t carry; /* bits 0-7: byte before the previous one, 8-15: previous */
	uint32_t opened; /* NALs opened in this segment */
	uint64_t pending; /* payload start of the open NAL */
	uint64_t first_e;
};

/* value of the lane below (lane 0: `first`): one DPP move, no LDS crossbar */
__device__ __forceinline__ uint32_t scan_from_lane_below(uint32_t v, uint32_t first,
							 uint32_t lane)
{
	uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */,
							    0xf, 0xf, true);
	return lane == 0 ? first : up;
}

/* One 1 KiB step: lane l holds bytes cbase + 16 l .. + 15 in q.
 *
 * Entropy-coded data holds no two consecutive zero bytes except at NAL
 * boundaries (that is what emulation prevention is for), and every code needs
 * such a pair.  The step therefore first asks, in the 0x80-flag-per-byte
 * domain, whether ANY lane sees a zero pair (or the step inherits one): about
 * 30 vector instructions per 16 bytes.  Only then are the exact E / S masks
 * built and walked by the scalar state machine. */
__device__ __forceinline__ void scan_step(const scan_mark_args &a, scan_state &st,
					  uint4 q, uint64_t cbase, uint32_t seg,
					  uint32_t lane)
{
	const uint32_t w[4] = {q.x, q.y, q.z, q.w};
	uint32_t z[4];
#pragma unroll
	for (int k = 0; k < 4; k++)
		z[k] = scan_zero_bytes(w[k]);
	/* zero flags of the four bytes before this lane's 16 (bits 23, 31 used) */
	const uint32_t zc = ((st.carry & 0xffu) == 0 ? 0x00800000u : 0u) |
			    ((st.carry >> 8) == 0 ? 0x80000000u : 0u);
	const uint32_t zb = scan_from_lane_below(z[3], zc, lane);
	/* flag at byte j: bytes j - 1 and j are both zero */
	uint32_t pair = z[0] & __builtin_amdgcn_alignbit(z[0], zb, 24);
	pair |= z[1] & __builtin_amdgcn_alignbit(z[1], z[0], 24);
	pair |= z[2] & __builtin_amdgcn_alignbit(z[2], z[1], 24);
	pair |= z[3] & __builtin_amdgcn_alignbit(z[3], z[2], 24);
	const uint32_t carry_in = st.carry;
	/* next step's carry: last two bytes of lane 63 */
	st.carry = (uint32_t)__builtin_amdgcn_readlane((int)(w[3] >> 16), 63);
	if (__ballot(pair != 0) == 0 && carry_in != 0)
		return;

	uint32_t z16 = 0, le16 = 0;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		z16 |= scan_movemask(z[k]) << (4 * k);
		le16 |= scan_movemask(scan_zero_bytes(w[k] & 0xfefefefeu)) << (4 * k);
	}
	uint32_t zprev = ((zb >> 23) & 1u) | ((zb >> 30) & 2u);
	uint32_t z18 = (z16 << 2) | zprev; /* bit j+2 = byte j */
	/* code ENDING at byte j: bytes j-2, j-1 zero, byte j <= 1 */
	uint32_t e16 = z18 & (z18 >> 1) & le16;
	uint32_t s16 = e16 & ~z16;

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
			if (st.first_e == SCAN_NONE)
				st.first_e = p;
			if (st.pending != SCAN_NONE) {
				if (lane == 0) {
					uint32_t slot = atomicAdd(&a.counter[0], 1u);
					if (slot < a.cap_recs) {
						scan_rec r;
						r.start = st.pending;
						r.end = p;
						r.seg = seg;
						r.local = st.opened - 1u;
						a.recs[slot] = r;
					}
				}
				st.pending = SCAN_NONE;
			}
			if ((s >> j) & 1u) {
				st.pending = p + 3u;
				st.opened++;
			}
		}
	}
}

template <int KNOB>
__global__ void __launch_bounds__(64 * SCAN_WAVES_PER_BLOCK)
k_scan_mark_probe(scan_mark_args a)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t seg = (uint32_t)__builtin_amdgcn_readfirstlane(
		(int)(blockIdx.x * SCAN_WAVES_PER_BLOCK + (threadIdx.x >> 6)));
	if (seg >= a.nsegs)
		return;
	const uint64_t lo = (uint64_t)seg * a.seg_bytes;
	uint64_t hi = lo + a.seg_bytes;
	if (hi > a.len)
		hi = a.len;
	/* blocks of 4 KiB that lie wholly inside the buffer take plain 16-byte
	 * loads, software-pipelined one block ahead; only the buffer's last,
	 * ragged block goes through the careful loader */
	const uint64_t full_hi = lo + ((hi - lo) & ~(uint64_t)(SCAN_BLOCK_BYTES - 1u));

	uint4 q[4], nq[4];
	if (lo < full_hi) {
#pragma unroll
		for (int u = 0; u < 4; u++)
			q[u] = *reinterpret_cast<const uint4 *>(a.buf + lo + 1024u * u +
								16u * lane);
	}
	scan_state st;
	/* the two bytes before the segment (0xff before the buffer) */
	st.carry = 0xffffu;
	if (lo >= 2 && !(KNOB & 1))
		st.carry = (uint32_t)a.buf[lo - 2] | ((uint32_t)a.buf[lo - 1] << 8);
	st.carry = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.carry);
	st.pending = SCAN_NONE;
	st.first_e = SCAN_NONE;
	st.opened = 0;

	for (uint64_t pos = lo; pos < full_hi; pos += SCAN_BLOCK_BYTES) {
		const uint64_t nx = pos + SCAN_BLOCK_BYTES;
		if (nx < full_hi) {
#pragma unroll
			for (int u = 0; u < 4; u++)
				nq[u] = *reinterpret_cast<const uint4 *>(
					a.buf + nx + 1024u * u + 16u * lane);
		}
#pragma unroll
		for (int u = 0; u < 4; u++) {
			if (KNOB & 4)
				st.opened ^= q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
			else
				scan_step(a, st, q[u], pos + 1024u * u, seg, lane);
		}
#pragma unroll
		for (int u = 0; u < 4; u++)
			q[u] = nq[u];
	}
	for (uint64_t cbase = full_hi; cbase < hi; cbase += 1024u) {
		uint64_t off = cbase + 16u * lane;
		uint4 t = off < hi ? scan_load16(a.buf, off, a.len)
				   : make_uint4(~0u, ~0u, ~0u, ~0u);
		scan_step(a, st, t, cbase, seg, lane);
	}
	if (lane == 0) {
		if (st.pending != SCAN_NONE) {
			uint32_t slot = atomicAdd(&a.counter[0], 1u);
			if (slot < a.cap_recs) {
				scan_rec r;
				r.start = st.pending;
				r.end = SCAN_NONE;
				r.seg = seg;
				r.local = st.opened - 1u;
				a.recs[slot] = r;
			}
		}
		if (!(KNOB & 2) || st.opened == 0x12345678u) {
			a.seg_count[seg] = st.opened;
			a.seg_first_e[seg] = st.first_e;
		}
	}
}


template <typename F> static double timeit(F f, int reps = 10)
{
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	std::vector<float> ms;
	for (int r = 0; r < reps + 2; r++) {
		CHECK(hipEventRecord(e0)); f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
		float t; CHECK(hipEventElapsedTime(&t, e0, e1)); if (r >= 2) ms.push_back(t);
	}
	std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2];
}
template <int K> static void run(scan_mark_args a, size_t bytes)
{
	uint32_t blocks = (a.nsegs + 3) / 4;
	double ms = timeit([&] { CHECK(hipMemsetAsync(a.counter, 0, 8)); k_scan_mark_probe<K><<<blocks, 256>>>(a); });
	printf("{\"probe\": \"scan knob %d seg %u\", \"ms\": %.4f, \"GBps\": %.1f}\n", K, a.seg_bytes, ms, bytes / ms / 1e6);
	fflush(stdout);
}
int main(int argc, char **argv)
{
	size_t mib = argc > 1 ? atol(argv[1]) : 1024, bytes = mib << 20;
	uint8_t *d; CHECK(hipMalloc(&d, bytes + 64)); CHECK(hipMemset(d, 0x5a, bytes + 64));
	for (uint32_t seg : {16384u, 8192u}) {
		scan_mark_args a; memset(&a, 0, sizeof(a));
		a.buf = d; a.len = bytes; a.seg_bytes = seg; a.nsegs = (uint32_t)((bytes + seg - 1) / seg); a.cap_recs = 1 << 16;
		CHECK(hipMalloc(&a.recs, sizeof(scan_rec) << 16)); CHECK(hipMalloc(&a.seg_count, 4 * (size_t)a.nsegs));
		CHECK(hipMalloc(&a.seg_first_e, 8 * (size_t)a.nsegs)); CHECK(hipMalloc(&a.counter, 8));
		run<0>(a, bytes); run<1>(a, bytes); run<2>(a, bytes); run<3>(a, bytes); run<4>(a, bytes); run<7>(a, bytes);
	}
	return 0;
}

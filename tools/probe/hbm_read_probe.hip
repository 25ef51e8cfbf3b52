// Calibration: what does a pure streaming READ reach on this box, and how does
// the work-to-wave mapping change it?  Build: hipcc --offload-arch=gfx950 -O3
//   hbm_read_probe.hip -o hbm_read_probe ; run: ./hbm_read_probe [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t fold(u32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// A: grid-stride, every wave-instruction of the whole grid reads adjacent KiB
template <int NT>
__global__ void __launch_bounds__(256) k_gridstride(const u32x4 *p, size_t n16, uint32_t *out)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
	uint32_t acc = 0;
	for (; i + 3 * stride < n16; i += 4 * stride) {
		u32x4 a, b, c, d;
		if (NT) {
			a = __builtin_nontemporal_load(p + i); b = __builtin_nontemporal_load(p + i + stride);
			c = __builtin_nontemporal_load(p + i + 2 * stride); d = __builtin_nontemporal_load(p + i + 3 * stride);
		} else {
			a = p[i]; b = p[i + stride]; c = p[i + 2 * stride]; d = p[i + 3 * stride];
		}
		acc ^= fold(a) ^ fold(b) ^ fold(c) ^ fold(d);
	}
	for (; i < n16; i += stride) acc ^= fold(p[i]);
	if (acc == 0x12345678u) out[0] = acc;
}

// B: one contiguous segment per wave (the scan kernel's mapping), 4 KiB in flight
__global__ void __launch_bounds__(256) k_segment(const u32x4 *p, size_t n16, uint32_t seg16, uint32_t *out)
{
	size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	uint32_t lane = threadIdx.x & 63;
	size_t lo = wave * seg16, hi = lo + seg16;
	if (hi > n16) hi = n16;
	uint32_t acc = 0;
	for (size_t i = lo; i + 256 <= hi; i += 256) {
		u32x4 a = p[i + lane], b = p[i + 64 + lane], c = p[i + 128 + lane], d = p[i + 192 + lane];
		acc ^= fold(a) ^ fold(b) ^ fold(c) ^ fold(d);
	}
	if (acc == 0x12345678u) out[0] = acc;
}

// C: persistent waves, block-interleaved: wave w reads 4 KiB blocks w, w+W, ...
__global__ void __launch_bounds__(256) k_interleave(const u32x4 *p, size_t n16, uint32_t *out)
{
	size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), W = (size_t)gridDim.x * 4;
	uint32_t lane = threadIdx.x & 63;
	uint32_t acc = 0;
	for (size_t i = wave * 256; i + 256 <= n16; i += W * 256) {
		u32x4 a = p[i + lane], b = p[i + 64 + lane], c = p[i + 128 + lane], d = p[i + 192 + lane];
		acc ^= fold(a) ^ fold(b) ^ fold(c) ^ fold(d);
	}
	if (acc == 0x12345678u) out[0] = acc;
}

// D: plain copy, 16 B per lane, each wave its own contiguous segment (what the
// slice write-back does per slice) / grid-stride
__global__ void __launch_bounds__(256) k_copy_segment(const u32x4 *p, u32x4 *q, size_t n16, uint32_t seg16)
{
	size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	uint32_t lane = threadIdx.x & 63;
	size_t lo = wave * seg16, hi = lo + seg16;
	if (hi > n16) hi = n16;
	for (size_t i = lo; i + 256 <= hi; i += 256) {
		u32x4 a = p[i + lane], b = p[i + 64 + lane], c = p[i + 128 + lane], d = p[i + 192 + lane];
		q[i + lane] = a; q[i + 64 + lane] = b; q[i + 128 + lane] = c; q[i + 192 + lane] = d;
	}
}
__global__ void __launch_bounds__(256) k_copy_gridstride(const u32x4 *p, u32x4 *q, size_t n16)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
	for (; i + 3 * stride < n16; i += 4 * stride) {
		u32x4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
		q[i] = a; q[i + stride] = b; q[i + 2 * stride] = c; q[i + 3 * stride] = d;
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

int main(int argc, char **argv)
{
	size_t mib = argc > 1 ? atol(argv[1]) : 1024, bytes = mib << 20, n16 = bytes / 16;
	u32x4 *d; uint32_t *out;
	CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&out, 4)); CHECK(hipMemset(d, 0x5a, bytes));
	auto rep = [&](const char *name, double ms) {
		printf("{\"probe\": \"%s\", \"mib\": %zu, \"ms\": %.4f, \"GBps\": %.1f}\n", name, mib, ms, bytes / ms / 1e6);
		fflush(stdout);
	};
	for (int blocks : {2048, 4096, 8192})
		rep(("gridstride b" + std::to_string(blocks)).c_str(), timeit([&] { k_gridstride<0><<<blocks, 256>>>(d, n16, out); }));
	rep("gridstride nt b2048", timeit([&] { k_gridstride<1><<<2048, 256>>>(d, n16, out); }));
	for (uint32_t seg : {4096u, 16384u, 131072u}) {
		uint32_t seg16 = seg / 16; size_t waves = (n16 + seg16 - 1) / seg16;
		rep(("segment " + std::to_string(seg)).c_str(), timeit([&] { k_segment<<<(waves + 3) / 4, 256>>>(d, n16, seg16, out); }));
	}
	for (int blocks : {1024, 2048, 4096})
		rep(("interleave b" + std::to_string(blocks)).c_str(), timeit([&] { k_interleave<<<blocks, 256>>>(d, n16, out); }));
	{
		u32x4 *q; CHECK(hipMalloc(&q, bytes));
		auto rep2 = [&](const char *name, double ms) {
			printf("{\"probe\": \"%s\", \"mib\": %zu, \"ms\": %.4f, \"GBps_read_plus_write\": %.1f}\n", name, mib, ms, 2.0 * bytes / ms / 1e6);
			fflush(stdout);
		};
		for (uint32_t seg : {16384u, 163840u}) {
			uint32_t seg16 = seg / 16; size_t waves = (n16 + seg16 - 1) / seg16;
			rep2(("copy segment " + std::to_string(seg)).c_str(), timeit([&] { k_copy_segment<<<(waves + 3) / 4, 256>>>(d, q, n16, seg16); }));
		}
		for (int blocks : {2048, 8192})
			rep2(("copy gridstride b" + std::to_string(blocks)).c_str(), timeit([&] { k_copy_gridstride<<<blocks, 256>>>(d, q, n16); }));
	}
	return 0;
}

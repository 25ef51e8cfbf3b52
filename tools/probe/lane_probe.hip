// does v_readlane / v_writelane mask the lane select to 6 bits on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
extern "C" __global__ void k(unsigned *out, unsigned sel)
{
	unsigned v = threadIdx.x * 3u + 1u;
	unsigned s = __builtin_amdgcn_readfirstlane(sel);
	unsigned r = __builtin_amdgcn_readlane(v, s);
	unsigned w = v;
	asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(w) : "s"(r + 1000u), "s"(s) : "m0");
	out[threadIdx.x] = w;
	if (threadIdx.x == 0)
		out[64] = r;
}
int main()
{
	unsigned *d, h[65];
	hipMalloc(&d, sizeof(h));
	for (unsigned sel : {5u, 69u, 0x145u}) {
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sel);
		hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
		printf("sel %#x: readlane -> %u (lane %u); writelane hit lane:", sel, h[64], (h[64] - 1) / 3);
		for (int i = 0; i < 64; i++)
			if (h[i] != i * 3u + 1u)
				printf(" %d=%u", i, h[i]);
		printf("\n");
	}
	return 0;
}

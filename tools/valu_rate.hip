// Micro-benchmark: issue rate of the integer VALU ops the matcher lives on (v_xor/v_and/v_bcnt/v_add3) and of FP64 add/mul on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
	uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x9e3779b9u, c = a + 12345u, d = b * 7u;
	uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
	double f0 = a * 1e-9, f1 = b * 1e-9, f2 = 1.0000001, f3 = 0.9999999;
	for (int i = 0; i < iters; ++i) {
		if (MODE == 0) {   // 4 independent chains of (xor, and, bcnt-accumulate): 12 int ops
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				acc0 += __popc((a ^ (c + u)) & b); acc1 += __popc((b ^ (d + u)) & c); acc2 += __popc((c ^ (a + u)) & d); acc3 += __popc((d ^ (b + u)) & a);
			}
			a += acc0; b += acc1;
		} else {           // FP64 mul + add chains (no FMA contraction: compiled with -ffp-contract=off)
#pragma unroll
			for (int u = 0; u < 8; ++u) { f0 = f0 * f2 + f1; f1 = f1 * f3 + f0; }
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc0 + acc1 + acc2 + acc3 + (uint32_t)(f0 + f1);
}
int main() {
	uint32_t* d; hipMalloc(&d, 4096 * 256 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int mode = 0; mode < 2; ++mode) {
		const int iters = 4000, blocks = 4096;
		for (int rep = 0; rep < 2; ++rep) {
			hipEventRecord(e0);
			if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
			else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
			hipEventRecord(e1); hipEventSynchronize(e1);
		}
		float ms; hipEventElapsedTime(&ms, e0, e1);
		// lane-ops per iteration (approx static count): mode 0: 8*4*(add,xor,and,bcnt)=128 + 2; mode 1: 8*4 = 32 FP64 ops
		const double ops = (mode == 0 ? 130.0 : 32.0) * iters * blocks * 256.0;
		printf("mode %d: %.3f ms  -> %.2f T lane-ops/s  (= %.1f lanes/clk/SIMD at 2.4 GHz over 1024 SIMDs)\n", mode, ms, ops / ms / 1e9, ops / (ms * 1e-3) / 2.4e9 / 1024);
	}
	return 0;
}

// Micro-benchmark (round 6): issue rate of single VALU instructions on gfx950, eight independent chains per lane so that latency does not limit the rate.
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate2.hip -o tools/valu_rate2 && tools/valu_rate2
// Prints cycles per wave-instruction and SIMD (4.0 = full rate: 16 lanes per clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
	double d[8]; float f[8]; uint32_t u[8]; unsigned long long q[8];
	const double dm = 1.0000001, da = 1e-9;
	const float fm = 1.0001f;
	for (int i = 0; i < 8; ++i) { d[i] = threadIdx.x * 1e-3 + i; f[i] = threadIdx.x * 1e-3f + i; u[i] = threadIdx.x * 2654435761u + i; q[i] = u[i]; }
	uint32_t s1 = 48 + (iters & 1), s2 = 3;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			if (MODE == 0) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(da));
				REP8(X)
#undef X
			} else if (MODE == 1) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
				REP8(X)
#undef X
			} else if (MODE == 2) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
				REP8(X)
#undef X
			} else if (MODE == 3) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fm));
				REP8(X)
#undef X
			} else if (MODE == 4) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(s2));
				REP8(X)
#undef X
			} else if (MODE == 5) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(s1));
				REP8(X)
#undef X
			} else if (MODE == 6) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(u[i]), "v"(s1) : "vcc");
				REP8(X)
#undef X
			} else if (MODE == 7) {
#define X(i) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(u[i]) : "v"(s2));
				REP8(X)
#undef X
			} else if (MODE == 8) {
#define X(i) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(u[i]) : "v"(s2));
				REP8(X)
#undef X
			} else if (MODE == 9) {
#define X(i) asm volatile("v_cvt_f32_f64_e32 %0, %1" : "=v"(f[i]) : "v"(d[i]));
				REP8(X)
#undef X
			} else if (MODE == 10) {
#define X(i) asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
				REP8(X)
#undef X
			} else if (MODE == 11) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
				REP8(X)
#undef X
			} else if (MODE == 12) {
#define X(i) asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(u[i]) : "v"(s1));
				REP8(X)
#undef X
			} else if (MODE == 13) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
				REP8(X)
#undef X
			} else if (MODE == 14) {
#define X(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s1) : "v"(u[i]));
				REP8(X)
#undef X
			} else if (MODE == 15) {
#define X(i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(da));
				REP8(X)
#undef X
			} else if (MODE == 16) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(dm));
				REP8(X)
#undef X
			}
		}
	}
	uint32_t acc = s1;
	for (int i = 0; i < 8; ++i) acc += (uint32_t)d[i] + (uint32_t)f[i] + u[i] + (uint32_t)q[i];
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}
static const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_fma_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_u32_u24", "v_add_u32", "v_cvt_f32_f64",
                              "v_cvt_f64_f32", "v_lshl_add_u64", "v_min_u32", "v_mov_b32_dpp", "v_readlane_b32", "v_fmac_f64", "v_pk_fma_f32"};
template <int M> void run(uint32_t* d) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 2000, blocks = 2048;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
	float ms = 0;
	for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
	const double winst = 32.0 * iters * blocks * 4.0;   // wave-instructions
	const double cyc = ms * 1e-3 * 2.4e9 * 1024.0 / winst;
	printf("%-16s %.3f ms  %.2f cycles per wave-instruction and SIMD (at 2.4 GHz)\n", names[M], ms, cyc);
}
int main() {
	uint32_t* d; hipMalloc(&d, 2048 * 256 * 4);
	run<8>(d); run<3>(d); run<16>(d); run<0>(d); run<15>(d); run<1>(d); run<2>(d); run<4>(d); run<5>(d); run<6>(d); run<7>(d); run<9>(d); run<10>(d); run<11>(d); run<12>(d); run<13>(d); run<14>(d);
	return 0;
}

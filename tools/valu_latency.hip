// Micro-benchmark: cycles per wave-instruction on gfx950 for dependent chains vs independent streams, at 1..4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_latency.hip -o tools/valu_latency && tools/valu_latency
// One 256-thread block = 4 waves = one wave per SIMD of a CU; 256*W blocks -> W waves per SIMD (the dispatcher spreads blocks evenly).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
	double f[8];
	uint32_t u[8];
	for (int i = 0; i < 8; ++i) { f[i] = 1.0 + threadIdx.x * 1e-9 + i; u[i] = threadIdx.x * 2654435761u + i; }
	const double c = 1.0000001;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int r = 0; r < 16; ++r) {
			if (MODE == 0) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[0]) : "v"(c)); }                                   // dependent FP64 add
			if (MODE == 1) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f[0]) : "v"(c)); }                               // dependent FP64 fma
			if (MODE == 2) {                                                                                                  // 8 independent FP64 adds
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[j]) : "v"(c));
			}
			if (MODE == 3) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[0]) : "v"(u[1])); }                                 // dependent int add
			if (MODE == 4) {                                                                                                  // 8 independent int adds
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 5) {                                                                                                  // 8 independent v_mov
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 6) {                                                                                                  // 8 independent bcnt
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 7) { asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(u[0]) : "v"(u[1])); }                            // dependent bcnt
			if (MODE == 9) {                                                                                                  // 8 independent v_bitop3 ((a ^ b) & c)
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x60" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
			}
			if (MODE == 10) {                                                                                                 // 8 independent v_xor
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 11) {                                                                                                 // 8 independent v_lshl_or
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 12) {                                                                                                 // 8 independent v_rndne_f64
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_rndne_f64 %0, %0" : "+v"(f[j]));
			}
			if (MODE == 13) {                                                                                                 // 8 independent v_rsq_f64
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_rsq_f64 %0, %0" : "+v"(f[j]));
			}
			if (MODE == 14) {                                                                                                 // 8 independent v_cvt_i32_f64
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[j]) : "v"(f[j]));
			}
			if (MODE == 15) {                                                                                                 // 8 independent v_and_or
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
			}
			if (MODE == 16) {                                                                                                 // 8 independent v_mul_lo_u32
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 17) {                                                                                                 // 8 independent v_mul_u32_u24
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 18) {                                                                                                 // 8 independent v_mad_u32_u24
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
			}
			if (MODE == 19) {                                                                                                 // 8 independent v_pk_min_i16
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 20) {                                                                                                 // 8 independent v_min_i32
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_min_i32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 21) {                                                                                                 // 8 independent v_mad_u64_u32 (64-bit address arithmetic)
#pragma unroll
				for (int j = 0; j < 4; ++j) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(f[j]) : "v"(f[(j + 1) & 3]));
			}
			if (MODE == 22) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
			}
			if (MODE == 23) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
			}
			if (MODE == 24) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
			}
			if (MODE == 25) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 26) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
			}
			if (MODE == 27) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 28) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
			}
			if (MODE == 29) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(u[j]));
			}
			if (MODE == 30) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(u[j]));
			}
			if (MODE == 31) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 32) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 33) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 34) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 35) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 36) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(u[(j + 1) & 7]) : "vcc");
			}
			if (MODE == 37) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_max_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 38) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
			}
			if (MODE == 39) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
			}
			if (MODE == 40) {
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
			}
			if (MODE == 8) {                                                                                                  // 8 independent FP64 mul
#pragma unroll
				for (int j = 0; j < 8; ++j) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[j]) : "v"(c));
			}
		}
	}
	double s = 0;
	for (int i = 0; i < 8; ++i) s += f[i] + u[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int per, double* d) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 2000;
	for (int W = 1; W <= 4; ++W) {
		float best = 1e9f;
		for (int rep = 0; rep < 3; ++rep) {
			hipEventRecord(e0);
			hipLaunchKernelGGL(k<MODE>, dim3(256 * W), dim3(256), 0, 0, d, iters);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			best = ms < best ? ms : best;
		}
		const double instr = (double)iters * 16 * per;               // wave-instructions per wave
		const double cyc = best * 1e-3 * 2.4e9;                      // at 2.4 GHz
		printf("%-28s W=%d  %.3f ms  %.2f cycles per wave-instruction per wave, %.2f per SIMD issue slot\n", name, W, best, cyc / instr, cyc / instr / W);
	}
}

int main() {
	double* d; hipMalloc(&d, 1024 * 256 * 8);
	run<0>("dependent v_add_f64", 1, d);
	run<1>("dependent v_fma_f64", 1, d);
	run<2>("8 independent v_add_f64", 8, d);
	run<8>("8 independent v_mul_f64", 8, d);
	run<3>("dependent v_add_u32", 1, d);
	run<4>("8 independent v_add_u32", 8, d);
	run<5>("8 independent v_mov_b32", 8, d);
	run<7>("dependent v_bcnt_u32_b32", 1, d);
	run<6>("8 independent v_bcnt_u32_b32", 8, d);
	run<9>("8 independent v_bitop3_b32", 8, d);
	run<10>("8 independent v_xor_b32", 8, d);
	run<11>("8 independent v_lshl_or_b32", 8, d);
	run<15>("8 independent v_and_or_b32", 8, d);
	run<12>("8 independent v_rndne_f64", 8, d);
	run<13>("8 independent v_rsq_f64", 8, d);
	run<14>("8 independent v_cvt_i32_f64", 8, d);
	run<16>("8 independent v_mul_lo_u32", 8, d);
	run<17>("8 independent v_mul_u32_u24", 8, d);
	run<18>("8 independent v_mad_u32_u24", 8, d);
	run<19>("8 independent v_pk_min_i16", 8, d);
	run<20>("8 independent v_min_i32", 8, d);
	run<21>("4 independent v_lshl_add_u64", 4, d);
	run<22>("8 independent v_dot4_u32_u8", 8, d);
	run<23>("8 independent v_dot2_u32_u16", 8, d);
	run<24>("8 independent v_perm_b32", 8, d);
	run<25>("8 independent v_mul_hi_u32_u24", 8, d);
	run<26>("8 independent v_sad_u8", 8, d);
	run<27>("8 independent v_alignbyte_b32", 8, d);
	run<28>("8 independent v_add3_u32", 8, d);
	run<29>("8 independent v_lshrrev_b32", 8, d);
	run<30>("8 independent v_bfe_u32", 8, d);
	run<31>("8 independent v_add_u32_sdwa", 8, d);
	run<32>("8 independent v_sub_u32", 8, d);
	run<33>("8 independent v_mul_hi_u32", 8, d);
	run<34>("8 independent v_and_b32", 8, d);
	run<35>("8 independent v_pk_add_u16", 8, d);
	run<36>("8 independent v_cmp_gt_u32+v_cndmask", 16, d);
	run<37>("8 independent v_max_u32", 8, d);
	run<38>("8 independent v_max3_u32", 8, d);
	run<39>("8 independent v_pk_max_u16", 8, d);
	run<40>("8 independent v_med3_i32", 8, d);
	return 0;
}

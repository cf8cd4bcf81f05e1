// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands holding only 0.0 (nibble 0) and 1.0 (nibble 2): does
//   D[i][j] = sum_k A[i][k] * B[k][j]
// come out as the exact bit-vector dot product under  A: lane l = row (l & 31), k-half (l >> 5), 32 nibbles in 4 dwords;  B: lane l = column (l & 31),
// same k mapping;  C/D: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)?   Build: hipcc --offload-arch=gfx950 -O2 -o mfma_fp4_probe mfma_fp4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const uint32_t* a, const uint32_t* b, float* d, int scale) {
	const int lane = threadIdx.x;
	v8i va = {0, 0, 0, 0, 0, 0, 0, 0}, vb = {0, 0, 0, 0, 0, 0, 0, 0};
	for (int i = 0; i < 4; ++i) { va[i] = (int)a[lane * 4 + i]; vb[i] = (int)b[lane * 4 + i]; }
	v16f c = {0};
	if (scale == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 4, 4, 0, 0, 0, 0);
	else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
	c = scale == 0 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 4, 4, 0, 0, 0, 0) : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // accumulate twice
	for (int r = 0; r < 16; ++r) d[lane * 16 + r] = c[r];
}

int main() {
	std::vector<uint8_t> A(32 * 64), B(64 * 32);
	uint64_t st = 12345;
	auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
	for (auto& v : A) v = rnd() & 1;
	for (auto& v : B) v = (rnd() % 3) == 0;   // asymmetric density
	std::vector<uint32_t> ha(64 * 4, 0), hb(64 * 4, 0);
	for (int l = 0; l < 64; ++l)
		for (int e = 0; e < 32; ++e) {
			const int kk = 32 * (l >> 5) + e;
			if (A[(l & 31) * 64 + kk]) ha[l * 4 + e / 8] |= 2u << (4 * (e % 8));
			if (B[kk * 32 + (l & 31)]) hb[l * 4 + e / 8] |= 2u << (4 * (e % 8));
		}
	uint32_t *da, *db; float* dd;
	hipMalloc(&da, ha.size() * 4); hipMalloc(&db, hb.size() * 4); hipMalloc(&dd, 64 * 16 * 4);
	hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
	for (int scale = 0; scale < 2; ++scale) {
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd, scale);
		std::vector<float> hd(64 * 16);
		hipMemcpy(hd.data(), dd, hd.size() * 4, hipMemcpyDeviceToHost);
		int bad = 0;
		for (int l = 0; l < 64; ++l)
			for (int r = 0; r < 16; ++r) {
				const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
				int ref = 0;
				for (int kk = 0; kk < 64; ++kk) ref += A[row * 64 + kk] * B[kk * 32 + col];
				if (hd[l * 16 + r] != 2.0f * ref) { if (bad < 5) printf("scale %d lane %d reg %d: got %g want %d\n", scale, l, r, hd[l * 16 + r], 2 * ref); ++bad; }
			}
		printf("scale-arg variant %d: %d mismatches of 1024\n", scale, bad);
	}
	return 0;
}

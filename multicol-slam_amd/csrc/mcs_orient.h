// mcs_orient.h — E5: IC_Angle (src/mdBRIEFextractorOct.cpp:221-248) + cv::fastAtan2, shared by the oct-tree kernel (which computes the
// orientation of the keys it has just selected, while the other (image, level) workgroups are still in their passes) and the descriptor kernels (which read it).
#pragma once
#include "mcs_common.h"

namespace mcs {

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
	const float K = (float)(180 / 3.1415926535897932384626433832795);
	const float p1 = 0.9997878412794807f * K, p3 = -0.3258083974640975f * K, p5 = 0.1555786518463281f * K,
	            p7 = -0.04432655554792128f * K;
	const float eps = (float)2.2204460492503131e-16;
	float ax = fabsf(x), ay = fabsf(y);
	float a, c, c2;
	if (ax >= ay) {
		c = ay / (ax + eps);
		c2 = c * c;
		a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
	} else {
		c = ax / (ay + eps);
		c2 = c * c;
		a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
	}
	if (x < 0) a = 180.f - a;
	if (y < 0) a = 360.f - a;
	return a;
}

// The orientation of `count` selected keys of one (image, level) by a whole workgroup: 16 lanes per key (lane j of a group owns disc rows j - 16, j and —
// lane 0 — 16: nine unaligned dwords per row), int32 moments reduced inside the group (exact, order-free), then the float polynomial of cv::fastAtan2.
// sel[k] = x | y << 12 | score << 24 relative to kMinBorder; out[k] = angle in degrees.  umax = half-width of disc row |v| (PyrDesc.umax).
__device__ __forceinline__ void orient_selected(const uint8_t* raw, int rstride, const int* umax, const uint32_t* sel, int count, float* out) {
	const int tid = threadIdx.x, j = tid & 15, groups = blockDim.x >> 4;
	// the disc rows of ONE key owned by this lane: 27 unaligned dwords, all requested before the first is used
	auto moments = [&](int k, int& m10, int& m01) {
		m10 = 0; m01 = 0;
		if (k >= count) return;
		const uint32_t rec = sel[k];
		const int col = (int)(rec & 0xFFF) + kMinBorder, row = (int)((rec >> 12) & 0xFFF) + kMinBorder;
		uint32_t w[3][9];
#pragma unroll
		for (int t = 0; t < 3; ++t) {
			const int r = min(j + 16 * t, 2 * kHalfPatch);   // disc row index 0..32 (the surplus lanes of the third trip re-read row 32, unused)
			const uint8_t* rp = raw + (size_t)(row + r - kHalfPatch) * rstride + (col - kHalfPatch);
#pragma unroll
			for (int q = 0; q < 9; ++q) __builtin_memcpy(&w[t][q], rp + 4 * q, 4);
		}
#pragma unroll
		for (int t = 0; t < 3; ++t) {
			const int r = j + 16 * t;
			if (r <= 2 * kHalfPatch) {
				const int v = r - kHalfPatch;
				const int um = umax[v < 0 ? -v : v];
				int rowSum = 0, rowMom = 0;
#pragma unroll
				for (int jj = 0; jj <= 2 * kHalfPatch; ++jj) {
					const int u = jj - kHalfPatch;
					int val = (int)((w[t][jj >> 2] >> (8 * (jj & 3))) & 0xffu);
					val = (u >= -um && u <= um) ? val : 0;
					rowSum += val;
					rowMom += u * val;
				}
				m10 += rowMom;
				m01 += v * rowSum;
			}
		}
	};
	for (int k0 = 0; k0 < count; k0 += 2 * groups) {   // two keys per 16-lane group and trip: their loads overlap
		const int ka = k0 + (tid >> 4), kb = ka + groups;
		int a10, a01, b10, b01;
		moments(ka, a10, a01);
		moments(kb, b10, b01);
#pragma unroll
		for (int o = 8; o > 0; o >>= 1) {   // within the 16-lane group (exact integer sums: order-free)
			a10 += __shfl_xor(a10, o); a01 += __shfl_xor(a01, o);
			b10 += __shfl_xor(b10, o); b01 += __shfl_xor(b01, o);
		}
		if (j == 0) {
			if (ka < count) out[ka] = fast_atan2_deg((float)a01, (float)a10);
			if (kb < count) out[kb] = fast_atan2_deg((float)b01, (float)b10);
		}
	}
}

}  // namespace mcs

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

// Per disc row |v| the orientation needs  sum(val)  and  sum(u * val)  over the columns |u| <= umax[|v|]: as byte dot products against two words per dword of the
// row — mask bytes 1 inside the disc / 0 outside, and weight bytes (u + 16) inside / 0 outside (u * val = (u + 16) * val - 16 * val) — v_dot4_u32_u8 does four
// pixels per instruction.  Table in LDS, built once per workgroup: [|v| = 0 .. 16][10 mask words | 10 weight words] (words 9: past the disc, zero).
constexpr int kOrientTabRow = 20, kOrientTabWords = (kHalfPatch + 1) * kOrientTabRow;
__device__ __forceinline__ void orient_table(const int* umax, uint32_t* tab) {   // all threads of the workgroup; the caller synchronises
	for (int i = threadIdx.x; i < (kHalfPatch + 1) * 10; i += blockDim.x) {
		const int a = i / 10, q = i - 10 * a, um = umax[a];
		uint32_t mask = 0, wt = 0;
#pragma unroll
		for (int e = 0; e < 4; ++e) {
			const int jj = 4 * q + e, u = jj - kHalfPatch;
			if (jj <= 2 * kHalfPatch && u >= -um && u <= um) { mask |= 1u << (8 * e); wt |= (uint32_t)jj << (8 * e); }
		}
		tab[a * kOrientTabRow + q] = mask;
		tab[a * kOrientTabRow + 10 + q] = wt;
	}
}

// The orientation of `count` selected keys of one (image, level) by a whole workgroup: 16 lanes per key.  Round 4: the lanes of a group read CONTIGUOUS bytes —
// lane j = 5 sub + c takes the 8 bytes 8c .. 8c + 7 of disc row 3 i + sub in step i = 0 .. 10 (11 steps x 3 rows = the 33 rows; lane 15 idles), so a wave's load
// touches a dozen cache lines where the row-per-lane form (27 dwords of three whole rows per lane) touched 64: the tail of k_octree was bound by those line
// requests (55 of the kernel's 157 us; the byte dot products alone changed nothing).  int32 moments reduced inside the group (exact, order-free), then the float
// polynomial of cv::fastAtan2.  sel[k] = x | y << 12 | score << 24 relative to kMinBorder; out[k] = angle in degrees.  tab = orient_table's words (LDS).
// (The 4 bytes a row's fifth chunk reads past the disc's 36 lie inside the level's pitch or are the next row's first bytes: a key sits >= 22 px inside.)
__device__ __forceinline__ void orient_selected(const uint8_t* raw, int rstride, const uint32_t* tab, const uint32_t* sel, int count, float* out) {
	const int tid = threadIdx.x, j = tid & 15, groups = blockDim.x >> 4;
	const int sub = j / 5, c = j - 5 * sub;   // j = 15: sub = 3, no row
	auto moments = [&](int k, int& m10, int& m01) {
		m10 = 0; m01 = 0;
		if (k >= count || sub > 2) return;
		const uint32_t rec = sel[k];
		const int col = (int)(rec & 0xFFF) + kMinBorder, row = (int)((rec >> 12) & 0xFFF) + kMinBorder;
		const uint8_t* p0 = raw + (size_t)(row - kHalfPatch + sub) * rstride + (col - kHalfPatch + 8 * c);
		uint2 w[11];
#pragma unroll
		for (int i = 0; i < 11; ++i) __builtin_memcpy(&w[i], p0 + (size_t)(3 * i) * rstride, 8);   // all requested before the first is used
#pragma unroll
		for (int i = 0; i < 11; ++i) {
			const int v = 3 * i + sub - kHalfPatch;
			const uint32_t* tw = tab + (v < 0 ? -v : v) * kOrientTabRow + 2 * c;
			const uint2 mk = *reinterpret_cast<const uint2*>(tw), wt = *reinterpret_cast<const uint2*>(tw + 10);
			const uint32_t rowSum = __builtin_amdgcn_udot4(w[i].y, mk.y, __builtin_amdgcn_udot4(w[i].x, mk.x, 0u, false), false);
			const uint32_t rowW = __builtin_amdgcn_udot4(w[i].y, wt.y, __builtin_amdgcn_udot4(w[i].x, wt.x, 0u, false), false);
			m10 += (int)rowW - kHalfPatch * (int)rowSum;   // this chunk's share of sum(u * val)
			m01 += v * (int)rowSum;
		}
	};
#ifndef MCS_ORIENT_KEYS
#define MCS_ORIENT_KEYS 3   // keys per 16-lane group and trip: their loads overlap (11 eight-byte loads per key and lane in flight)
#endif
	constexpr int KPT = MCS_ORIENT_KEYS;
	for (int k0 = 0; k0 < count; k0 += KPT * groups) {
		int m10[KPT], m01[KPT];
#pragma unroll
		for (int q = 0; q < KPT; ++q) moments(k0 + (tid >> 4) + q * groups, m10[q], m01[q]);
#pragma unroll
		for (int o = 8; o > 0; o >>= 1)   // within the 16-lane group (exact integer sums: order-free)
#pragma unroll
			for (int q = 0; q < KPT; ++q) { m10[q] += __shfl_xor(m10[q], o); m01[q] += __shfl_xor(m01[q], o); }
		if (j == 0) {
#pragma unroll
			for (int q = 0; q < KPT; ++q) {
				const int k = k0 + (tid >> 4) + q * groups;
				if (k < count) out[k] = fast_atan2_deg((float)m01[q], (float)m10[q]);
			}
		}
	}
}

}  // namespace mcs

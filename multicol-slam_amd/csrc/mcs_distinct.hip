// mcs_distinct.hip — cMapPoint::ComputeDistinctiveDescriptors (src/cMapPoint.cpp:294-382, median() include/misc.h:95-104), SURVEY §8f row 3,
// for a batch of map points: one 64-thread workgroup per map point.
//   distances   N <= 128: the upper triangle is computed once (lanes stride over the pairs) into an LDS uint16 matrix; larger N
//               (not seen in practice: a map point has one observation per keyframe camera) recompute pairs from global memory.
//   median      row i's median over j > i is sorted[(N-1-i)/2]; distances are integers in [0, 8*dim], so each lane finds it for its
//               rows by bisection on the value: the smallest v with #{d <= v} >= (N-1-i)/2 + 1 (10 passes over the row).
//   argmin      smallest (median, i) over i < N-1 (strict '<' in the reference keeps the first row), N <= 2 -> 0.
#include "mcs_common.h"

namespace mcs {

constexpr int kDistinctLds = 128;

template <int DW, bool MASKED>
__device__ __forceinline__ int pair_distance(const uint8_t* desc, const uint8_t* mask, int stride, size_t a, size_t b) {
	const uint32_t* x = reinterpret_cast<const uint32_t*>(desc + a * stride);
	const uint32_t* y = reinterpret_cast<const uint32_t*>(desc + b * stride);
	int acc = 0;
	if (MASKED) {
		const uint32_t* mx = reinterpret_cast<const uint32_t*>(mask + a * stride);
		const uint32_t* my = reinterpret_cast<const uint32_t*>(mask + b * stride);
#pragma unroll
		for (int w = 0; w < DW; ++w) { const uint32_t v = x[w] ^ y[w]; acc += __popc(v & mx[w]); acc += __popc(v & my[w]); }
		return acc >> 1;
	}
#pragma unroll
	for (int w = 0; w < DW; ++w) acc += __popc(x[w] ^ y[w]);
	return acc;
}

template <int DW, bool MASKED>
__global__ __launch_bounds__(64) void k_distinct(DistinctArgs a) {
	__shared__ unsigned short dm[kDistinctLds * kDistinctLds];
	__shared__ unsigned int best;
	const int mp = blockIdx.x, lane = threadIdx.x;
	const int lo = a.offsets[mp], N = a.offsets[mp + 1] - lo;
	if (N <= 2) { if (lane == 0) a.bestIdx[mp] = N <= 0 ? -1 : 0; return; }
	const bool inLds = N <= kDistinctLds;
	if (lane == 0) best = 0xFFFFFFFFu;
	if (inLds) {
		const int pairs = N * N;
		for (int t = lane; t < pairs; t += 64) {
			const int i = t / N, j = t - i * N;
			if (j > i) dm[i * N + j] = (unsigned short)pair_distance<DW, MASKED>(a.desc, a.mask, a.stride, (size_t)lo + i, (size_t)lo + j);
		}
	}
	__syncthreads();
	for (int i = lane; i < N - 1; i += 64) {
		const int cnt = N - 1 - i, need = cnt / 2 + 1;
		int vlo = 0, vhi = 32 * DW;   // distances lie in [0, 8*dim]
		while (vlo < vhi) {
			const int mid = (vlo + vhi) >> 1;
			int c = 0;
			for (int j = i + 1; j < N; ++j) {
				const int d = inLds ? (int)dm[i * N + j] : pair_distance<DW, MASKED>(a.desc, a.mask, a.stride, (size_t)lo + i, (size_t)lo + j);
				c += d <= mid;
			}
			if (c >= need) vhi = mid; else vlo = mid + 1;
		}
		atomicMin(&best, ((unsigned)vlo << 16) | (unsigned)i);   // N - 1 <= 65535 rows
	}
	__syncthreads();
	if (lane == 0) a.bestIdx[mp] = (int)(best & 0xFFFFu);
}

template <int DW>
static void launch_dw(const DistinctArgs& a, hipStream_t s) {
	if (a.mask) hipLaunchKernelGGL((k_distinct<DW, true>), dim3(a.npoints), dim3(64), 0, s, a);
	else hipLaunchKernelGGL((k_distinct<DW, false>), dim3(a.npoints), dim3(64), 0, s, a);
}

void launch_distinct(const DistinctArgs& a, hipStream_t s) {
	if (a.npoints <= 0) return;
	if (a.dim == 16) launch_dw<4>(a, s);
	else if (a.dim == 32) launch_dw<8>(a, s);
	else launch_dw<16>(a, s);
}

}  // namespace mcs

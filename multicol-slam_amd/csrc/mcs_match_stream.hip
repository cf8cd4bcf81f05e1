// mcs_match_stream.hip — scalar-streamed variant of the top-K Hamming kernel (same outputs as k_match_partial).
// The train row is the same for every lane of a wave (lane = query), so it is fetched through the SCALAR path
// (s_load_dwordx8 from the constant cache) straight into SGPRs that feed v_xor/v_and/v_bcnt as scalar operands: no LDS
// staging, no workgroup barriers, no LDS-broadcast traffic competing with the VALU.  Eligibility comes from a per-row
// int32 flag (camera group or -1) prepared by k_build_tflag.  Reference arithmetic: src/cORBmatcher.cpp:2438-2474.
#include "mcs_common.h"

namespace mcs {

__global__ void k_build_tflag(const uint8_t* tvalid, const int* tgroup, size_t rows, int* out) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < rows) {
		const bool ok = tvalid ? tvalid[i] != 0 : true;
		out[i] = ok ? (tgroup ? tgroup[i] : 0) : -1;
	}
}

void launch_build_tflag(const uint8_t* tvalid, const int* tgroup, size_t rows, int* out, hipStream_t s) {
	hipLaunchKernelGGL(k_build_tflag, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, tvalid, tgroup, rows, out);
}

template <int K, int DW, bool MASKED>
__global__ __launch_bounds__(256) void k_match_stream(MatchArgs a, const int* __restrict__ tflag32) {
	constexpr int CB = 16;
	__shared__ uint32_t cand[CB * 256];
	const int tid = threadIdx.x;
	const int set = blockIdx.z, split = blockIdx.y;
	const int qi = blockIdx.x * 256 + tid;
	const size_t qrow0 = (size_t)set * a.qpitch, trow0 = (size_t)set * a.tpitch;
	bool qok = qi < a.nq;
	if (qok && a.qvalid) qok = a.qvalid[qrow0 + qi] != 0;
	uint32_t q[DW], qm[DW];
	int qg = 0;
#pragma unroll
	for (int w = 0; w < DW; ++w) { q[w] = 0; qm[w] = 0; }
	if (qok) {
		const uint32_t* qp = reinterpret_cast<const uint32_t*>(a.qd + (qrow0 + qi) * a.qstride);
#pragma unroll
		for (int w = 0; w < DW; ++w) q[w] = qp[w];
		if (MASKED) {
			const uint32_t* mp = reinterpret_cast<const uint32_t*>(a.qm + (qrow0 + qi) * a.qstride);
#pragma unroll
			for (int w = 0; w < DW; ++w) qm[w] = mp[w];
		}
		if (a.qgroup) qg = a.qgroup[qrow0 + qi];
	}
	uint32_t best[K];
#pragma unroll
	for (int p = 0; p < K; ++p) best[p] = 0xFFFFFFFFu;
	int countLe = 0, cnt = 0;
	const bool useGroup = a.qgroup != nullptr && a.tgroup != nullptr;
	auto flush = [&]() {
		int m = cnt;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
		for (int e = 0; e < m; ++e) {
			uint32_t key = e < cnt ? cand[e * 256 + tid] : 0xFFFFFFFFu;
			if (__any(key < best[K - 1])) {
#pragma unroll
				for (int p = 0; p < K; ++p) { const uint32_t lo = min(best[p], key), hi = max(best[p], key); best[p] = lo; key = hi; }
			}
		}
		cnt = 0;
	};

	const int per = (a.nt + a.splits - 1) / a.splits;
	const int t0 = split * per, t1 = min(a.nt, t0 + per);
	const uint8_t* __restrict__ tdp = a.td;
	const uint8_t* __restrict__ tmp = a.tm;
	for (int j = t0; j < t1; j += 4) {
		uint32_t key[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int jj = min(j + u, t1 - 1);                        // wave-uniform row index
			const uint32_t* tp = reinterpret_cast<const uint32_t*>(tdp + (trow0 + jj) * a.tstride);
			const uint32_t* mp = MASKED ? reinterpret_cast<const uint32_t*>(tmp + (trow0 + jj) * a.tstride) : tp;
			const int g = tflag32 ? tflag32[trow0 + jj] : 0;
			int acc = 0;
#pragma unroll
			for (int w = 0; w < DW; ++w) {
				const uint32_t x = q[w] ^ tp[w];
				if (MASKED) { acc += __popc(x & qm[w]); acc += __popc(x & mp[w]); }
				else acc += __popc(x);
			}
			const int dist = MASKED ? acc >> 1 : acc;
			const bool ok = (j + u < t1) && g >= 0 && (!useGroup || g == qg) && qok;
			countLe += (ok && dist <= a.countThresh) ? 1 : 0;
			key[u] = (ok && dist <= a.maxDist) ? (((uint32_t)dist << 20) | (uint32_t)jj) : 0xFFFFFFFFu;
		}
#pragma unroll
		for (int u = 0; u < 4; ++u)
			if (key[u] < best[K - 1]) { cand[cnt * 256 + tid] = key[u]; ++cnt; }
		if (__any(cnt > CB - 4)) flush();
	}
	flush();
	if (qi < a.nq) {
		uint32_t* dst = a.splits == 1 ? a.keys + (size_t)set * K * a.nq : a.partial + ((size_t)set * a.splits + split) * K * a.nq;
#pragma unroll
		for (int p = 0; p < K; ++p) dst[(size_t)p * a.nq + qi] = best[p];
		if (a.splits == 1) a.outCount[(size_t)set * a.nq + qi] = countLe;
		else a.partialCount[((size_t)set * a.splits + split) * a.nq + qi] = countLe;
	}
}

template <int K, int DW>
static void launch_sd(const MatchArgs& a, const int* tflag32, hipStream_t s) {
	dim3 grid((a.nq + 255) / 256, a.splits, a.nsets);
	if (a.qm && a.tm) hipLaunchKernelGGL((k_match_stream<K, DW, true>), grid, dim3(256), 0, s, a, tflag32);
	else hipLaunchKernelGGL((k_match_stream<K, DW, false>), grid, dim3(256), 0, s, a, tflag32);
}

template <int K>
static void launch_s(const MatchArgs& a, const int* tflag32, hipStream_t s) {
	if (a.dim == 16) launch_sd<K, 4>(a, tflag32, s);
	else if (a.dim == 32) launch_sd<K, 8>(a, tflag32, s);
	else launch_sd<K, 16>(a, tflag32, s);
}

// first stage only (partial / final packed lists); the merge + unpack stages of mcs_match.hip follow unchanged
void launch_match_stream(const MatchArgs& a, const int* tflag32, hipStream_t s) {
	switch (a.K) {
		case 1: launch_s<1>(a, tflag32, s); break;
		case 2: launch_s<2>(a, tflag32, s); break;
		case 4: launch_s<4>(a, tflag32, s); break;
		case 8: launch_s<8>(a, tflag32, s); break;
		case 16: launch_s<16>(a, tflag32, s); break;
		default: launch_s<32>(a, tflag32, s); break;
	}
}

}  // namespace mcs

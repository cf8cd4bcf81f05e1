// mcs_match.hip — M1/M2 + the brute-force inner loops of M4/M4'/M5: K nearest train descriptors per query by
// (distance, train index), eligibility = valid flag and (optionally) equal camera group.
// Reference: DescriptorDistance64 / DescriptorDistance64Masked src/cORBmatcher.cpp:2438-2474 (xor + popcount, the masked
// form sums popcnt((a^b)&ma) + popcnt((a^b)&mb) over all words and halves the TOTAL once); loops :907-948 (SearchByBoW
// KF,KF), :231-262 (SearchByBoW KF,F), :1037-1066 (SearchForTriangulationRaw).  The strict-'<' best/second tracking of
// the reference means "ties -> lowest train index", which is the (distance<<20 | index) ordering used here.
//
// Integer VALU-bound (descriptors are re-used nt times): one query per lane held in registers, 256 train rows (+masks,
// +eligibility) staged per step in LDS and read back as wave-uniform broadcasts (v_xor + v_and + v_bcnt accumulate).
// The train range is split over blockIdx.y so that a single 3000x3000 pair still fills 256 CUs; a second tiny kernel
// merges the per-split sorted lists.  Sets (keyframes) are blockIdx.z: one launch sweeps a whole keyframe database.
// Lists are kept as packed keys in a [set][K][query] layout: lane = query, so every list store/load is coalesced.
#include "mcs_common.h"

#include <cstdlib>

namespace mcs {

constexpr int MT = 256;   // train rows per LDS step

// Raw popcount total of one (query, train row) pair: sum_w popc((q^t)&qm) + popc((q^t)&tm) (masked; the reference halves this total
// ONCE, src/cORBmatcher.cpp:2452-2474) or sum_w popc(q^t).  v_bcnt_u32_b32 d, a, b = popcount(a) + b, so the running total rides on
// the popcounts: the chain is written as asm because the compiler otherwise emits bcnt(x, 0) plus a tree of v_add3, and is seeded
// with the literal 0 (no v_mov).  t / tm point into LDS at 16-byte aligned rows and are read as ds_read_b128 broadcasts.
template <int DW, bool MASKED>
__device__ __forceinline__ uint32_t pair_total(const uint32_t* q, const uint32_t* qm, const uint4* t, const uint4* tm) {
	uint32_t acc = 0;
#pragma unroll
	for (int w4 = 0; w4 < DW / 4; ++w4) {
		const uint4 tv = t[w4];
		const uint32_t tw[4] = {tv.x, tv.y, tv.z, tv.w};
		uint32_t mw[4] = {0, 0, 0, 0};
		if (MASKED) { const uint4 mv = tm[w4]; mw[0] = mv.x; mw[1] = mv.y; mw[2] = mv.z; mw[3] = mv.w; }
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int w = 4 * w4 + k;
			const uint32_t x = q[w] ^ tw[k];
			if (MASKED) {
				const uint32_t xa = x & qm[w], xb = x & mw[k];
				if (w == 0) asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(acc) : "v"(xa));
				else asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(xa));
				asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(xb));
			} else {
				if (w == 0) asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(acc) : "v"(x));
				else asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
			}
		}
	}
	return acc;
}

// Two train rows at once, their accumulate chains interleaved word by word.  One pair alone is a chain of 2*DW DEPENDENT v_bcnt (each adds onto the
// previous total); the compiler ran the four rows of a trip one after the other, so a wave issued one dependent bit-count after the other (~5-6 cycles
// each instead of 4.3).  `asm volatile` keeps the a0 / a1 alternation in program order.
template <int DW, bool MASKED>
__device__ __forceinline__ void pair_total_x2(const uint32_t* q, const uint32_t* qm, const uint4* t0, const uint4* m0, const uint4* t1, const uint4* m1,
                                              uint32_t& a0, uint32_t& a1) {
#pragma unroll
	for (int w4 = 0; w4 < DW / 4; ++w4) {
		const uint4 tv0 = t0[w4], tv1 = t1[w4];
		const uint32_t tw0[4] = {tv0.x, tv0.y, tv0.z, tv0.w}, tw1[4] = {tv1.x, tv1.y, tv1.z, tv1.w};
		uint32_t mw0[4] = {0, 0, 0, 0}, mw1[4] = {0, 0, 0, 0};
		if (MASKED) {
			const uint4 mv0 = m0[w4], mv1 = m1[w4];
			mw0[0] = mv0.x; mw0[1] = mv0.y; mw0[2] = mv0.z; mw0[3] = mv0.w;
			mw1[0] = mv1.x; mw1[1] = mv1.y; mw1[2] = mv1.z; mw1[3] = mv1.w;
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int w = 4 * w4 + k;
			const uint32_t x0 = q[w] ^ tw0[k], x1 = q[w] ^ tw1[k];
			if (MASKED) {
				const uint32_t xa0 = x0 & qm[w], xa1 = x1 & qm[w], xb0 = x0 & mw0[k], xb1 = x1 & mw1[k];
				if (w == 0) { asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(a0) : "v"(xa0)); asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(a1) : "v"(xa1)); }
				else { asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a0) : "v"(xa0)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a1) : "v"(xa1)); }
				asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a0) : "v"(xb0));
				asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a1) : "v"(xb1));
			} else {
				if (w == 0) { asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(a0) : "v"(x0)); asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(a1) : "v"(x1)); }
				else { asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a0) : "v"(x0)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a1) : "v"(x1)); }
			}
		}
	}
}

template <int DW, bool MASKED>
__device__ __forceinline__ int hamming(const uint32_t* q, const uint32_t* qm, const uint32_t* t, const uint32_t* tm) {
	const uint32_t acc = pair_total<DW, MASKED>(q, qm, reinterpret_cast<const uint4*>(t), reinterpret_cast<const uint4*>(tm));
	return MASKED ? (int)(acc >> 1) : (int)acc;   // static_cast<int>(dist / 2): ONE division of the total
}

template <int K, int DW, bool MASKED, bool COUNT, bool GROUP>
__global__ __launch_bounds__(256) void k_match_partial(MatchArgs a) {
	__shared__ __attribute__((aligned(16))) uint32_t td[MT * DW];
	__shared__ __attribute__((aligned(16))) uint32_t tm[MASKED ? MT * DW : 4];
	__shared__ __attribute__((aligned(16))) int tflag[MT + 4];        // camera group of the staged (eligible) rows
	__shared__ __attribute__((aligned(16))) uint32_t tidx[MT + 4];    // their original train index; 0xFFFFFFFF = padding
	__shared__ int wcnt[4];
	constexpr int CB = 16;      // candidate column depth per lane
	// slot e of lane tid at cand[e * 256 + tid]: whatever slots the lanes of a wave are at, lane l always hits bank l % 32 (a [tid][CB + 1] layout with
	// a +1 advance cost 38 M bank-conflict cycles per launch)
	__shared__ uint32_t cand[(CB + 1) * 256];
	const int tid = threadIdx.x;
	const int set = blockIdx.z, split = blockIdx.y;
	const int qi = blockIdx.x * 256 + tid;
	const RowMap QR{(size_t)(set % a.qmod) * a.qpitch, a.qblk, a.qbpitch}, TR{(size_t)((set / a.tdiv + a.toff) % a.tmod) * a.tpitch, a.tblk, a.tbpitch};
	bool qok = qi < a.nq;
	if (qok && a.qvalid) qok = a.qvalid[QR(qi)] != 0;
	uint32_t q[DW], qm[DW];
	int qg = 0;
	if (qok) {
		const uint32_t* qp = reinterpret_cast<const uint32_t*>(a.qd + QR(qi) * a.qstride);
#pragma unroll
		for (int w = 0; w < DW; ++w) q[w] = qp[w];
		if (MASKED) {
			const uint32_t* mp = reinterpret_cast<const uint32_t*>(a.qm + QR(qi) * a.qstride);
#pragma unroll
			for (int w = 0; w < DW; ++w) qm[w] = mp[w];
		}
		if (a.qgroup) qg = a.qgroup[QR(qi)];
	} else {
#pragma unroll
		for (int w = 0; w < DW; ++w) { q[w] = 0; qm[w] = 0; }
	}
	uint32_t best[K];
#pragma unroll
	for (int p = 0; p < K; ++p) best[p] = 0xFFFFFFFFu;
	int countLe = 0;
	constexpr bool useGroup = GROUP;   // = both sides carry camera groups (chosen by the launcher)
	// Candidate keys are first appended to a private LDS column (one ds_write per hit) and merged into the sorted
	// register list only when some lane's column is full or at the end: a sorted insert costs 2K VALU ops for the WHOLE
	// wave whenever ANY lane hits, which at K = 32 was more than the distance arithmetic itself.
	// `next` = index of the lane's next free slot in its column (tid, tid + 256, ...), kept as an index so that an append is
	// compare, select (slot or dump row), store, select + add (advance)
	// A candidate is kept as the RAW word total << 20 | index (masked: the un-halved popcount total); the exact key (total >> 1) << 20 | index is formed
	// when the column is merged.  Appending is: one shift-or, one compare against the raw limit, an unconditional LDS store to the lane's next slot and a
	// conditional advance of the slot — a word that does not qualify is simply overwritten by the next store.
	const uint32_t col0 = tid;
	uint32_t next = col0;
	auto exact_key = [](uint32_t w) { return MASKED ? (((w >> 21) << 20) | (w & 0xFFFFFu)) : w; };
	// Merge of the lane's candidate column into its sorted list.  K >= 16: the (<= CB = 16) candidates are loaded into registers
	// (empty slots = 0xFFFFFFFF), sorted by a 16-input bitonic network, folded against the upper half of the list
	// (m[i] = min(best[i], c[K-1-i]) holds the K smallest of both and is bitonic) and re-sorted by one bitonic MERGE — a fixed
	// ~340 VALU ops per flush, where inserting one candidate at a time through the sorted list cost 2K ops per candidate of the
	// fullest lane (~900 per flush at K = 32).  Smaller K keep the insertion loop.
	auto flush = [&]() {
		const int cnt = (int)((next - col0) >> 8);
		if (K >= CB) {
			uint32_t c[CB];
#pragma unroll
			for (int e = 0; e < CB; ++e) {   // unconditional reads (the whole column is the lane's own), stale slots masked afterwards
				const uint32_t raw = cand[e * 256 + tid];
				c[e] = e < cnt ? exact_key(raw) : 0xFFFFFFFFu;
			}
#pragma unroll
			for (int k = 2; k <= CB; k <<= 1)
#pragma unroll
				for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
					for (int i = 0; i < CB; ++i) {
						const int l = i ^ j;
						if (l > i) {
							const uint32_t lo = min(c[i], c[l]), hi = max(c[i], c[l]);
							const bool up = (i & k) == 0;
							c[i] = up ? lo : hi; c[l] = up ? hi : lo;
						}
					}
#pragma unroll
			for (int i = 0; i < CB; ++i) best[K - 1 - i] = min(best[K - 1 - i], c[i]);   // fold: ascending c against the descending tail
#pragma unroll
			for (int j = K >> 1; j > 0; j >>= 1)
#pragma unroll
				for (int i = 0; i < K; ++i) {
					const int l = i ^ j;
					if (l > i) { const uint32_t lo = min(best[i], best[l]), hi = max(best[i], best[l]); best[i] = lo; best[l] = hi; }
				}
		} else {
			int m = cnt;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
			for (int e = 0; e < m; ++e) {
				uint32_t key = e < cnt ? exact_key(cand[e * 256 + tid]) : 0xFFFFFFFFu;
				if (__any(key < best[K - 1])) {
#pragma unroll
					for (int p = 0; p < K; ++p) { const uint32_t lo = min(best[p], key), hi = max(best[p], key); best[p] = lo; key = hi; }
				}
			}
		}
		next = col0;
	};

	const int per = ((a.nt + a.splits - 1) / a.splits + 63) / 64 * 64;
	const int t0 = split * per, t1 = min(a.nt, t0 + per);
	const int lane = tid & 63, wv = tid >> 6;
	for (int base = t0; base < t1; base += MT) {
		// Stage only the ELIGIBLE rows of this step, compacted in order (ballot prefix), with their original index next to them;
		// the tail up to a multiple of 4 gets index 0xFFFFFFFF, which ORs every key to "empty".  The inner loop then needs no
		// per-row eligibility test at all (that test used to serialise every row behind a wave-uniform LDS read + branch).
		const int j = base + tid;
		int flag = -1;
		if (j < t1) {
			const bool ok = a.tvalid ? a.tvalid[TR(j)] != 0 : true;
			flag = ok ? (a.tgroup ? a.tgroup[TR(j)] : 0) : -1;
		}
		const unsigned long long bal = __ballot(flag >= 0);
		if (lane == 0) wcnt[wv] = __popcll(bal);
		__syncthreads();
		int pos = __popcll(bal & ((1ull << lane) - 1ull));
		for (int w = 0; w < wv; ++w) pos += wcnt[w];
		const int rows = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
		if (flag >= 0) {
			const uint32_t* tp = reinterpret_cast<const uint32_t*>(a.td + TR(j) * a.tstride);
#pragma unroll
			for (int w = 0; w < DW; ++w) td[pos * DW + w] = tp[w];
			if (MASKED) {
				const uint32_t* mp = reinterpret_cast<const uint32_t*>(a.tm + TR(j) * a.tstride);
#pragma unroll
				for (int w = 0; w < DW; ++w) tm[pos * DW + w] = mp[w];
			}
			tidx[pos] = (uint32_t)j;
			tflag[pos] = flag;
		}
		if (tid < 4) { tidx[rows + tid] = 0xFFFFFFFFu; tflag[rows + tid] = -1; }   // padding rows of the last trip
		__syncthreads();
		if (qok) {
			// 4 train rows per trip, two at a time with interleaved accumulate chains.  The distance threshold, the padding rows (index 0xFFFFFFFF ORs the
			// word to all ones) and "closer than the K-th best" are ONE unsigned compare of the raw word against rawLim.  Masked: the raw total t stands for
			// distance t >> 1, so "distance <= D" is t <= 2D + 1, i.e. word < (2D + 2) << 20; entries that tie the K-th best distance with a larger index
			// slip through and are dropped by the exact merge.
			const uint32_t dCap = a.maxDist >= 4095 ? 4095u : (uint32_t)a.maxDist;
			for (int r = 0; r < rows; r += 4) {
				const uint4 ti = *reinterpret_cast<const uint4*>(&tidx[r]);
				const uint32_t tiu[4] = {ti.x, ti.y, ti.z, ti.w};
				int tgu[4] = {0, 0, 0, 0};
				if (useGroup) { const int4 tg = *reinterpret_cast<const int4*>(&tflag[r]); tgu[0] = tg.x; tgu[1] = tg.y; tgu[2] = tg.z; tgu[3] = tg.w; }
				const uint4* trow = reinterpret_cast<const uint4*>(&td[r * DW]);
				const uint4* mrow = reinterpret_cast<const uint4*>(&tm[MASKED ? r * DW : 0]);
				uint32_t rawLim;
				if (MASKED) {
					const uint32_t dl = min(best[K - 1] >> 20, dCap);
					rawLim = dl >= 2047u ? 0xFFFFFFFFu : ((2u * dl + 2u) << 20);
				} else rawLim = min(best[K - 1], dCap >= 4095u ? 0xFFFFFFFFu : ((dCap + 1u) << 20));
				uint32_t acc[4];
				constexpr int RS = DW / 4;   // uint4 per row
				pair_total_x2<DW, MASKED>(q, qm, trow, mrow, trow + RS, mrow + (MASKED ? RS : 0), acc[0], acc[1]);
				pair_total_x2<DW, MASKED>(q, qm, trow + 2 * RS, mrow + (MASKED ? 2 * RS : 0), trow + 3 * RS, mrow + (MASKED ? 3 * RS : 0), acc[2], acc[3]);
				uint32_t w[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					w[u] = (acc[u] << 20) | tiu[u];
					if (useGroup) w[u] = tgu[u] == qg ? w[u] : 0xFFFFFFFFu;
					if (COUNT) countLe += (tiu[u] != 0xFFFFFFFFu && w[u] != 0xFFFFFFFFu && (int)(MASKED ? acc[u] >> 1 : acc[u]) <= a.countThresh) ? 1 : 0;
				}
				// (a wave vote on the smallest of the four words before the appends was measured: slower — some lane qualifies on practically every trip)
#pragma unroll
				for (int u = 0; u < 4; ++u) { cand[next] = w[u]; next += w[u] < rawLim ? 256u : 0u; }
				if (__any(next > col0 + (CB - 4) * 256)) flush();
			}
		}
		__syncthreads();
	}
	flush();
	if (qi < a.nq) {
		// splits == 1: these are the final lists; otherwise a partial list per split
		uint32_t* dst = a.splits == 1 ? a.keys + (size_t)set * K * a.nq : a.partial + ((size_t)set * a.splits + split) * K * a.nq;
#pragma unroll
		for (int p = 0; p < K; ++p) dst[(size_t)p * a.nq + qi] = best[p];
		if (a.splits == 1) a.outCount[(size_t)set * a.nq + qi] = countLe;
		else a.partialCount[((size_t)set * a.splits + split) * a.nq + qi] = countLe;
	}
}

template <int K>
__global__ __launch_bounds__(256) void k_match_merge(MatchArgs a) {
	const int qi = blockIdx.x * 256 + threadIdx.x;
	const int set = blockIdx.z;
	if (qi >= a.nq) return;
	uint32_t best[K];
#pragma unroll
	for (int p = 0; p < K; ++p) best[p] = 0xFFFFFFFFu;
	int countLe = 0;
	// A split's K entries are requested TOGETHER, and the next split's while this one is merged (round 5): entry by entry behind the `break` every thread paid a
	// dependent global round trip per entry — 77 us for the 12 splits of a single keyframe pair, the second largest piece of one multi-frame's matching latency.
	static_assert((K & (K - 1)) == 0, "the merge network needs a power of two");
	uint32_t cur[K], nxt[K];
	auto request = [&](int s, uint32_t (&dstv)[K]) {
		const uint32_t* src = a.partial + ((size_t)set * a.splits + s) * K * a.nq + qi;
#pragma unroll
		for (int e = 0; e < K; ++e) dstv[e] = src[(size_t)e * a.nq];
	};
	request(0, cur);
	for (int s = 0; s < a.splits; ++s) {
		countLe += a.partialCount[((size_t)set * a.splits + s) * a.nq + qi];
		if (s + 1 < a.splits) request(s + 1, nxt);
		// both lists ascend: the K smallest of their union are min(best[i], cur[K - 1 - i]) — a bitonic sequence —, sorted by one bitonic merge (log2 K stages of K / 2
		// compare-exchanges; K is a power of two).  Branch-free: 32 + 160 operations per split for K = 32 where insertion took up to 2048.
#pragma unroll
		for (int e = 0; e < K; ++e) best[K - 1 - e] = min(best[K - 1 - e], cur[e]);
#pragma unroll
		for (int j = K >> 1; j > 0; j >>= 1)
#pragma unroll
			for (int i = 0; i < K; ++i) {
				const int l = i ^ j;
				if (l > i) { const uint32_t lo = min(best[i], best[l]), hi = max(best[i], best[l]); best[i] = lo; best[l] = hi; }
			}
#pragma unroll
		for (int e = 0; e < K; ++e) cur[e] = nxt[e];
	}
	uint32_t* dst = a.keys + (size_t)set * K * a.nq;
#pragma unroll
	for (int p = 0; p < K; ++p) dst[(size_t)p * a.nq + qi] = best[p];
	a.outCount[(size_t)set * a.nq + qi] = countLe;
}

// packed [set][K][nq] keys -> the public [set][nq][K] (dist, idx) arrays of mcs_match_topk
__global__ __launch_bounds__(256) void k_match_unpack(MatchArgs a) {
	const int qi = blockIdx.x * 256 + threadIdx.x;
	const int set = blockIdx.z;
	if (qi >= a.nq) return;
	const uint32_t* src = a.keys + (size_t)set * a.K * a.nq;
	const size_t o = ((size_t)set * a.nq + qi) * a.K;
	for (int p = 0; p < a.K; ++p) {
		const uint32_t k = src[(size_t)p * a.nq + qi];
		const bool none = k == 0xFFFFFFFFu;
		a.outDist[o + p] = none ? 0x7FFFFFFF : (int)(k >> 20);
		a.outIdx[o + p] = none ? -1 : (int)(k & 0xFFFFFu);
	}
}

// after the partial-list kernel: merge of the per-split lists, the public (dist, idx) form
template <int K>
static void launch_post(const MatchArgs& a, hipStream_t s) {
	if (a.splits > 1) hipLaunchKernelGGL((k_match_merge<K>), dim3((a.nq + 255) / 256, 1, a.nsets), dim3(256), 0, s, a);
	if (a.outDist && a.outIdx) hipLaunchKernelGGL(k_match_unpack, dim3((a.nq + 255) / 256, 1, a.nsets), dim3(256), 0, s, a);
}

template <int K, int DW>
static void launch_kd(const MatchArgs& a, hipStream_t s) {
	dim3 grid((a.nq + 255) / 256, a.splits, a.nsets);
	const bool masked = a.qm && a.tm, count = a.countThresh >= 0;   // the searches do not need count_le: skip its VALU ops per pair
	const bool group = a.qgroup != nullptr && a.tgroup != nullptr;
#define MCS_LAUNCH_PARTIAL(M, C, G) hipLaunchKernelGGL((k_match_partial<K, DW, M, C, G>), grid, dim3(256), 0, s, a)
	if (masked) { if (count) { if (group) MCS_LAUNCH_PARTIAL(true, true, true); else MCS_LAUNCH_PARTIAL(true, true, false); }
	              else { if (group) MCS_LAUNCH_PARTIAL(true, false, true); else MCS_LAUNCH_PARTIAL(true, false, false); } }
	else { if (count) { if (group) MCS_LAUNCH_PARTIAL(false, true, true); else MCS_LAUNCH_PARTIAL(false, true, false); }
	       else { if (group) MCS_LAUNCH_PARTIAL(false, false, true); else MCS_LAUNCH_PARTIAL(false, false, false); } }
#undef MCS_LAUNCH_PARTIAL
	launch_post<K>(a, s);
}

template <int K>
static void launch_k(const MatchArgs& a, hipStream_t s) {
	if (a.dim == 16) launch_kd<K, 4>(a, s);
	else if (a.dim == 32) launch_kd<K, 8>(a, s);
	else launch_kd<K, 16>(a, s);
}

// mcs_match_mfma.hip: the same partial lists from the matrix cores (16 / 32-byte descriptors, no count_le, no camera groups)
bool match_mfma_serves(const MatchArgs& a);
void launch_match_mfma(const MatchArgs& a, hipStream_t s);

void launch_match(const MatchArgs& a, hipStream_t s) {
	static const bool valuOnly = getenv("MCS_MATCH_VALU") != nullptr;   // run-time switch: the v_bcnt kernel for every shape (tests compare the two)
	if (!valuOnly && match_mfma_serves(a)) {
		launch_match_mfma(a, s);
		switch (a.K) {
			case 1: launch_post<1>(a, s); break;
			case 2: launch_post<2>(a, s); break;
			case 4: launch_post<4>(a, s); break;
			case 8: launch_post<8>(a, s); break;
			case 16: launch_post<16>(a, s); break;
			default: launch_post<32>(a, s); break;
		}
		return;
	}
	switch (a.K) {
		case 1: launch_k<1>(a, s); break;
		case 2: launch_k<2>(a, s); break;
		case 4: launch_k<4>(a, s); break;
		case 8: launch_k<8>(a, s); break;
		case 16: launch_k<16>(a, s); break;
		default: launch_k<32>(a, s); break;
	}
}

__global__ void k_single_distance(const uint32_t* x, const uint32_t* y, const uint32_t* mx, const uint32_t* my, int dw, int* out) {
	if (threadIdx.x == 0) {
		int acc = 0;
		for (int w = 0; w < dw; ++w) {
			const uint32_t v = x[w] ^ y[w];
			if (mx) { acc += __popc(v & mx[w]); acc += __popc(v & my[w]); }
			else acc += __popc(v);
		}
		*out = mx ? acc >> 1 : acc;
	}
}

void launch_single_distance(const uint8_t* a, const uint8_t* b, const uint8_t* ma, const uint8_t* mb, int dim, int* out, hipStream_t s) {
	hipLaunchKernelGGL(k_single_distance, dim3(1), dim3(64), 0, s, (const uint32_t*)a, (const uint32_t*)b, (const uint32_t*)ma,
	                   (const uint32_t*)mb, dim / 4, out);
}

}  // namespace mcs

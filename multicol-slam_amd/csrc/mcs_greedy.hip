// mcs_greedy.hip — M4 / M4' / M5: the greedy, order-dependent part of cORBmatcher's brute-force searches, on the device.
// Reference: SearchByBoW(KF,KF) src/cORBmatcher.cpp:885-966 (accept best < TH_LOW_ and best < ratio*second, mark the train
// matched), SearchByBoW(KF,F) :179-323 without the BoW-node restriction (accept best <= TH_LOW_, output indexed by the frame
// feature), SearchForTriangulationRaw :968-1155 (candidates dist <= TH_LOW_ sorted by (dist, idx), DistTh = 2*best, first one
// passing CheckDistEpipolarLine src/misc.cpp:53-69 wins).  Queries are processed strictly in index order because every
// accepted match removes a train row from all later queries (vbMatched2 / vpMapPointMatches).
//
// One wave64 per (query set, train set) pair: the "matched" bitmap lives in LDS, lane e holds entry e of the query's
// top-K list (sorted by (distance, index) by mcs_match.hip), a __ballot over "entry still free" yields best and second.
// When the K entries cannot decide (too many of them already taken), the wave rescans the whole train set for that one
// query — exact for any K, K only trades list size against rescans.  Integer + a few FP64 mul/div: bit-exact.
#include "mcs_common.h"

#include <cstdlib>
#include <type_traits>

namespace mcs {

constexpr int kBitmapWords = 4096;   // nt <= 131072 train rows per set

template <int DW, bool MASKED>
__device__ __forceinline__ int hamming_g(const uint32_t* q, const uint32_t* qm, const uint32_t* t, const uint32_t* tm) {
	int acc = 0;
#pragma unroll
	for (int w = 0; w < DW; ++w) {
		const uint32_t x = q[w] ^ t[w];
		if (MASKED) { acc += __popc(x & qm[w]); acc += __popc(x & tm[w]); }
		else acc += __popc(x);
	}
	return MASKED ? acc >> 1 : acc;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { const uint32_t y = __shfl_xor(v, o); v = y < v ? y : v; }
	return v;
}

__device__ __forceinline__ bool check_epipolar(const double* ray1, const double* ray2, const double* E, double thresh) {
	double t[3];
	for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += ray2[k] * E[3 * k + j]; t[j] = s; }
	double nom = 0;
	for (int k = 0; k < 3; ++k) nom += t[k] * ray1[k];
	double Ex1[3], Etx2[3];
	for (int i = 0; i < 3; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += E[3 * i + k] * ray1[k]; Ex1[i] = s; }
	for (int i = 0; i < 3; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += E[3 * k + i] * ray2[k]; Etx2[i] = s; }
	const double den = Ex1[0] * Ex1[0] + Ex1[1] * Ex1[1] + Ex1[2] * Ex1[2] + Etx2[0] * Etx2[0] + Etx2[1] * Etx2[1] + Etx2[2] * Etx2[2];
	if (den == 0.0) return false;
	const double dsqr = (nom * nom) / den;
	return dsqr < thresh;
}

template <int DW, bool MASKED>
__global__ __launch_bounds__(64) void k_greedy(GreedyArgs g) {
	__shared__ uint32_t matched[kBitmapWords];
	const int set = blockIdx.x, lane = threadIdx.x;
	const RowMap QR{(size_t)(set % g.qmod) * g.qpitch, g.qblk, g.qbpitch}, TR{(size_t)((set / g.tdiv + g.toff) % g.tmod) * g.tpitch, g.tblk, g.tbpitch};
	const int K = g.K;
	for (int i = lane; i < (g.nt + 31) / 32; i += 64) matched[i] = 0;
	int* outM = g.outMatch + (size_t)set * (g.mode == 1 ? g.nt : g.nq);
	if (g.mode == 1) for (int j = lane; j < g.nt; j += 64) outM[j] = -1;
	__syncthreads();
	const bool useGroup = g.qgroup != nullptr && g.tgroup != nullptr;
	int nmatches = 0, nfallback = 0;

	for (int i = 0; i < g.nq; ++i) {
		const bool qok = g.qvalid ? g.qvalid[QR(i)] != 0 : true;   // uniform
		if (!qok) { if (g.mode != 1 && lane == 0) outM[i] = -1; continue; }
		int d = 0x7FFFFFFF, idx = -1;
		if (lane < K) {
			const uint32_t k = g.keys[((size_t)set * K + lane) * g.nq + i];
			if (k != 0xFFFFFFFFu) { d = (int)(k >> 20); idx = (int)(k & 0xFFFFFu); }
		}
		const bool freeE = idx >= 0 && !((matched[idx >> 5] >> (idx & 31)) & 1u);
		const unsigned long long bal = __ballot(freeE);
		const int lastIdx = __shfl(idx, K - 1), dK = __shfl(d, K - 1);
		const bool full = lastIdx >= 0;
		const int qg = useGroup ? g.qgroup[QR(i)] : 0;

		// lazily loaded query row for rescans
		uint32_t q[DW], qm[DW];
		bool qLoaded = false;
		auto load_q = [&]() {
			if (qLoaded) return;
			const uint32_t* qp = reinterpret_cast<const uint32_t*>(g.qd + QR(i) * g.qstride);
#pragma unroll
			for (int w = 0; w < DW; ++w) q[w] = qp[w];
			if (MASKED) {
				const uint32_t* mp = reinterpret_cast<const uint32_t*>(g.qm + QR(i) * g.qstride);
#pragma unroll
				for (int w = 0; w < DW; ++w) qm[w] = mp[w];
			}
			qLoaded = true;
		};
		// smallest key > after with distance <= bound among free, eligible train rows (0xFFFFFFFF if none); also second smallest
		auto rescan = [&](uint32_t after, int bound, uint32_t& k1, uint32_t& k2) {
			load_q();
			uint32_t a = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu;
			for (int j = lane; j < g.nt; j += 64) {
				if ((matched[j >> 5] >> (j & 31)) & 1u) continue;
				if (g.tvalid && g.tvalid[TR(j)] == 0) continue;
				if (useGroup && g.tgroup[TR(j)] != qg) continue;
				const uint32_t* tp = reinterpret_cast<const uint32_t*>(g.td + TR(j) * g.tstride);
				const uint32_t* mp = MASKED ? reinterpret_cast<const uint32_t*>(g.tm + TR(j) * g.tstride) : tp;
				const int dist = hamming_g<DW, MASKED>(q, qm, tp, mp);
				if (dist > bound) continue;
				const uint32_t key = ((uint32_t)dist << 20) | (uint32_t)j;
				if (key <= after && after != 0xFFFFFFFFu) continue;
				if (key < a) { b2 = a; a = key; } else if (key < b2) b2 = key;
			}
			const uint32_t m1 = wave_min_u32(a);
			const uint32_t m2 = wave_min_u32(a == m1 ? b2 : a);   // keys are unique (index bits), so exactly one lane owns m1
			k1 = m1; k2 = m2;
		};

		if (g.mode != 2) {
			const int n = __popcll(bal);
			int best = 0x7FFFFFFF, bestIdx = -1, second = 0x7FFFFFFF;
			bool needScan = false, reject = false;
			if (n >= 1) { const int e0 = __ffsll((long long)bal) - 1; best = __shfl(d, e0); bestIdx = __shfl(idx, e0); }
			if (n >= 2) { const int e1 = __ffsll((long long)(bal & (bal - 1))) - 1; second = __shfl(d, e1); }
			if (n < 2 && full) {   // the list may continue beyond K entries, every hidden entry has distance >= dK
				if (n == 1) {
					const bool pass = g.thInclusive ? best <= g.thLow : best < g.thLow;
					if (!pass) reject = true;
					else if (static_cast<double>(best) < g.ratio * static_cast<double>(dK)) second = dK;   // accepted for any second >= dK
					else needScan = true;
				} else {
					const bool pass = g.thInclusive ? dK <= g.thLow : dK < g.thLow;
					if (!pass) reject = true; else needScan = true;
				}
			}
			if (needScan) {
				uint32_t k1, k2;
				rescan(0xFFFFFFFFu, 0x7FF, k1, k2);
				++nfallback;
				best = k1 == 0xFFFFFFFFu ? 0x7FFFFFFF : (int)(k1 >> 20);
				bestIdx = k1 == 0xFFFFFFFFu ? -1 : (int)(k1 & 0xFFFFFu);
				second = k2 == 0xFFFFFFFFu ? 0x7FFFFFFF : (int)(k2 >> 20);
			}
			bool accept = false;
			if (!reject && bestIdx >= 0) {
				const bool pass = g.thInclusive ? best <= g.thLow : best < g.thLow;
				accept = pass && (static_cast<double>(best) < g.ratio * static_cast<double>(second));
			}
			if (accept) {
				if (lane == 0) {
					matched[bestIdx >> 5] |= 1u << (bestIdx & 31);
					if (g.mode == 0) outM[i] = bestIdx; else outM[bestIdx] = i;
				}
				++nmatches;
			} else if (g.mode == 0 && lane == 0) outM[i] = -1;
			__syncthreads();   // single wave: orders the LDS bitmap update before the next query's reads
		} else {
			// ---- SearchForTriangulationRaw
			int bestDist = -1, distTh = 0x7FFFFFFF, found = -1;
			uint32_t lastKey = 0xFFFFFFFFu;
			bool exhausted = true;   // ran off a full list without a stop condition
			const double* ray1 = g.rays1 + QR(i) * 3;
			const double* Em = g.E + (size_t)set * g.Epitch + (size_t)9 * ((size_t)qg * g.nrCams + qg);   // same-camera rule: camIdx2 == camIdx1
			for (int e = 0; e < K; ++e) {
				const int de = __shfl(d, e), ie = __shfl(idx, e);
				if (ie < 0 || de > g.thLow) { exhausted = false; break; }
				lastKey = ((uint32_t)de << 20) | (uint32_t)ie;
				if ((matched[ie >> 5] >> (ie & 31)) & 1u) continue;
				if (bestDist < 0) { bestDist = de; distTh = 2 * de; }
				if (de > distTh) { exhausted = false; break; }
				if (check_epipolar(ray1, g.rays2 + TR(ie) * 3, Em, 1e-2)) { found = ie; exhausted = false; break; }
			}
			if (exhausted && full) {
				for (int guard = 0; guard < g.nt; ++guard) {
					uint32_t k1, k2;
					const int bound = bestDist < 0 ? g.thLow : (distTh < g.thLow ? distTh : g.thLow);
					rescan(lastKey, bound, k1, k2);
					++nfallback;
					if (k1 == 0xFFFFFFFFu) break;
					const int de = (int)(k1 >> 20), ie = (int)(k1 & 0xFFFFFu);
					if (bestDist < 0) { bestDist = de; distTh = 2 * de; }
					if (de > distTh) break;
					if (check_epipolar(ray1, g.rays2 + TR(ie) * 3, Em, 1e-2)) { found = ie; break; }
					lastKey = k1;
				}
			}
			if (found >= 0) { if (lane == 0) matched[found >> 5] |= 1u << (found & 31); ++nmatches; }
			if (lane == 0) outM[i] = found;
			__syncthreads();
		}
	}
	if (lane == 0) {
		g.outCount[set] = nmatches;
		if (g.outFallbacks) g.outFallbacks[set] = nfallback;
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// Speculative wave-parallel form of the same greedy for SearchByBoW (modes 0 and 1).  64 consecutive queries are handled
// at once, one per lane, each lane holding its own top-K list in registers:
//   round:  every unresolved lane derives its decision (first two FREE entries of its list -> best / second -> threshold
//           and ratio tests) against the current "matched" bitmap;  accepting lanes publish their target with an LDS
//           atomicMin of the lane id;  a lane's decision is FINAL if no LOWER lane of this round claimed its best or its
//           second entry (those two rows are the only free rows the decision looked at);  all lanes below the first
//           non-final lane commit in one step, the others retry against the updated bitmap.
// The lowest unresolved lane is always final, so every round commits at least one query; the result is identical to the
// reference's strictly sequential loop (proof sketch: a committed lane saw exactly the bitmap the sequential loop would
// have had, because every lower lane either committed earlier or commits in the same step without touching its rows).
// A lane whose list cannot decide (its K entries are used up) is rescanned cooperatively by the wave when it becomes the
// lowest unresolved lane, and blocks the lanes above it until then.
#ifndef MCS_GREEDY_GROUP
#define MCS_GREEDY_GROUP 1   // the walk over a list's entries four at a time (0: entry by entry, for A/B)
#endif
constexpr int kClaimRows = 16384;   // train rows per set supported by the speculative kernel (LDS claim table)

// SearchForTriangulationRaw (TRI) in the same scheme: a query's outcome needs two rows of its sorted candidate list — a = the first FREE candidate
// (BestDist, DistTh = 2*BestDist) and b = the first free candidate that passes CheckDistEpipolarLine; b wins iff dist(b) <= DistTh.  The epipolar test
// of an entry does not depend on the matched state, so every lane evaluates it once for its K entries (a K-bit mask); a round then only re-reads the
// bitmap.  A decision is final if no lower lane of the round claimed a or b (rows between them can only turn from "free, fails the test" to "taken").
// The exact rescan of the lowest lane finds a and b in ONE pass over the train rows (the epipolar test runs inside the pass).
// The workgroup has kSpecWaves waves: wave 0 runs the rounds above, the other waves only help with the exact rescans — one wave alone pays the full
// global-memory latency of every trip over the train rows (~30 us per rescan), eight waves split the rows and overlap it.  Protocol per RESCAN: wave 0
// posts the query in LDS, block barrier A (where the helper waves wait), every wave scans its slice and posts its two smallest keys, block barrier B, wave 0 merges.
// Everything else in a round touches LDS from wave 0 only and is ordered by workgroup fences instead of barriers.
// wave 0 decides, the others scan.  Few set pairs (a frame ring: 64): 8 waves, the rescans are what takes time; a database sweep (thousands): 4 waves, four
// workgroups per CU — the kernel is a chain of dependent LDS / memory round trips per workgroup, and what speeds it up is workgroups in flight.
constexpr int kSpecWaves = 8, kSpecWavesMany = 4, kManySets = 512;
template <int K, int DW, bool MASKED, bool TRI>
// registers: 96 (five 4-wave workgroups per CU) where that costs no spills — the SearchByBoW forms on 16 / 32-byte descriptors —, 128 otherwise
__attribute__((amdgpu_waves_per_eu((DW <= 8 && !TRI) ? 5 : 4, (DW <= 8 && !TRI) ? 5 : 4)))
__global__ __launch_bounds__(64 * kSpecWaves) void k_greedy_spec(GreedyArgs g) {
	extern __shared__ uint32_t greedy_lds[];               // claim[nt], matched[ceil(nt / 32)]: sized by the launch, so that short sets leave room for more workgroups per CU
	uint32_t* claim = greedy_lds;
	uint32_t* matched = greedy_lds + g.nt;
	__shared__ int reqQ;                                   // query to rescan this round, -1 none, -2 the set is finished
	__shared__ uint32_t partA[kSpecWaves - 1], partB[kSpecWaves - 1];
	const int nscan = (int)(blockDim.x >> 6) - 1;
	const int set = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const RowMap QR{(size_t)(set % g.qmod) * g.qpitch, g.qblk, g.qbpitch}, TR{(size_t)((set / g.tdiv + g.toff) % g.tmod) * g.tpitch, g.tblk, g.tbpitch};
	constexpr uint32_t EMPTY = 0xFFFFFFFFu;
	for (int i = threadIdx.x; i < (g.nt + 31) / 32; i += blockDim.x) matched[i] = 0;
	for (int i = threadIdx.x; i < g.nt; i += blockDim.x) claim[i] = 0xFFFFFFFFu;
	int* outM = g.outMatch + (size_t)set * (g.mode == 1 ? g.nt : g.nq);
	if (g.mode == 1) for (int j = threadIdx.x; j < g.nt; j += blockDim.x) outM[j] = -1;
	__syncthreads();
	const bool grouped = g.qgroup != nullptr && g.tgroup != nullptr;
	// this wave's share of the exact rescan of query qi: rows (wave-1)*256 + lane + 64*u + 256*nscan*trip.  non-TRI: the slice's two smallest keys of
	// free eligible rows; TRI: its smallest candidate key and its smallest candidate key that passes the epipolar test.
	auto scan_slice = [&](int qi) {
		uint32_t q[DW], qm[DW];
		const uint32_t* qp = reinterpret_cast<const uint32_t*>(g.qd + QR(qi) * g.qstride);
#pragma unroll
		for (int w = 0; w < DW; ++w) q[w] = qp[w];
		if (MASKED) {
			const uint32_t* mp = reinterpret_cast<const uint32_t*>(g.qm + QR(qi) * g.qstride);
#pragma unroll
			for (int w = 0; w < DW; ++w) qm[w] = mp[w];
		}
		const int qgLow = grouped ? g.qgroup[QR(qi)] : 0;
		uint32_t a = EMPTY, b2 = EMPTY;
		double r1[3] = {0.0, 0.0, 0.0};
		const double* EmLow = g.E;
		if (TRI) {
#pragma unroll
			for (int c = 0; c < 3; ++c) r1[c] = g.rays1[QR(qi) * 3 + c];
			EmLow = g.E + (size_t)set * g.Epitch + (size_t)9 * ((size_t)qgLow * g.nrCams + qgLow);
		}
		// branch-free body, 4 rows per lane and trip: all global loads of a trip are issued before the first use
		for (int j0 = (wave - 1) * 256 + lane; j0 < g.nt; j0 += 256 * nscan) {
			uint32_t kk[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const int j = j0 + 64 * u;
				const int jc = j < g.nt ? j : g.nt - 1;
				const uint32_t* tp = reinterpret_cast<const uint32_t*>(g.td + TR(jc) * g.tstride);
				const uint32_t* mp = MASKED ? reinterpret_cast<const uint32_t*>(g.tm + TR(jc) * g.tstride) : tp;
				const bool ok = j < g.nt && !((matched[jc >> 5] >> (jc & 31)) & 1u) && (g.tvalid ? g.tvalid[TR(jc)] != 0 : true) &&
				                (!grouped || g.tgroup[TR(jc)] == qgLow);   // same camera / FeatureVector node only
				const int dist = hamming_g<DW, MASKED>(q, qm, tp, mp);
				const uint32_t k = ((uint32_t)dist << 20) | (uint32_t)jc;
				kk[u] = ok && (!TRI || dist <= g.thLow) ? k : EMPTY;
			}
			if (TRI) {   // the epipolar test is evaluated only where it can lower b2 (testing every row up front to get the ray loads out early was slower)
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const uint32_t k = kk[u];
					if (k < a) a = k;
					if (k < b2 && check_epipolar(r1, g.rays2 + TR((int)(k & 0xFFFFFu)) * 3, EmLow, 1e-2)) b2 = k;
				}
			} else {
#pragma unroll
				for (int u = 0; u < 4; ++u) { const uint32_t k = kk[u]; if (k < a) { b2 = a; a = k; } else if (k < b2) b2 = k; }
			}
		}
		const uint32_t m1 = wave_min_u32(a);
		const uint32_t m2 = TRI ? wave_min_u32(b2) : wave_min_u32(a == m1 ? b2 : a);
		if (lane == 0) { partA[wave - 1] = m1; partB[wave - 1] = m2; }
	};
	if (wave > 0) {   // helper waves
		for (;;) {
			__syncthreads();   // A
			const int rq = reqQ;
			if (rq == -2) return;
			if (rq >= 0) scan_slice(rq);
			__syncthreads();   // B
		}
	}
	int nmatches = 0, nfallback = 0;

	for (int i0 = 0; i0 < g.nq; i0 += 64) {
		const int i = i0 + lane;
		const bool inRange = i < g.nq;
		bool qok = inRange;
		if (qok && g.qvalid) qok = g.qvalid[QR(i)] != 0;
		uint32_t key[K];
#pragma unroll
		for (int e = 0; e < K; ++e) key[e] = EMPTY;
		if (qok) {
			const uint32_t* src = g.keys + (size_t)set * K * g.nq + i;   // [K][nq]: coalesced across lanes
#pragma unroll
			for (int e = 0; e < K; ++e) key[e] = src[(size_t)e * g.nq];
		}
		bool resolved = !qok;
		// The lane's result is written after the rounds (round 5: a workgroup-scope fence waits for every vector-memory operation of the wave on this chip, so a store
		// inside a round sits in front of the next fence).  Traced with the cycle counter on ONE keyframe pair (48 chunks one after the other, 245 us, the largest
		// piece of one multi-frame's matching latency): 107 rounds, 59 % of the cycles in the walk over the list entries below (≈ 3000 cycles per round: a chain of
		// dependent LDS bitmap reads as long as the unluckiest lane's prefix of taken rows), 15 % claims, 13 % list loads, 3 % commits.  Measured against it without
		// gain: the next chunk's lists requested a chunk ahead (245 us), these deferred stores (244), all K bitmap bits read at once and the two free entries picked
		// with bit arithmetic (317 with per-entry branches, 300 branch-free: 64 lanes x K entries of VALU work cost more than the chain they replace).
		int myOut = -1;            // modes 0 / 2: the row this query takes; mode 1: the row whose slot receives this query's index

		// TRI: a list that ends with an empty slot or an entry beyond TH_LOW holds every candidate; otherwise rows of distance >= dK may be hidden
		const bool full = TRI ? (key[K - 1] != EMPTY && (int)(key[K - 1] >> 20) <= g.thLow) : key[K - 1] != EMPTY;
		const int dK = full ? (int)(key[K - 1] >> 20) : 0x7FFFFFFF;
		uint32_t epi = 0;           // TRI: bit e = entry e is a candidate (distance <= TH_LOW) that passes the epipolar test
		double ray1[3] = {0.0, 0.0, 0.0};
		const double* Em = g.E;
		if (TRI && qok && g.nt > 0) {   // (an empty train set has no ray rows to read)
			const int qg = g.qgroup[QR(i)];
			Em = g.E + (size_t)set * g.Epitch + (size_t)9 * ((size_t)qg * g.nrCams + qg);   // same-camera rule: camIdx2 == camIdx1 (the lists are built per camera)
#pragma unroll
			for (int c = 0; c < 3; ++c) ray1[c] = g.rays1[QR(i) * 3 + c];
			// the rays of up to 8 list entries are fetched before the first test (a dependent global round trip per entry otherwise)
			constexpr int CH = K < 8 ? K : 8;
#pragma unroll
			for (int e0 = 0; e0 < K; e0 += CH) {
				double r2[CH][3];
#pragma unroll
				for (int u = 0; u < CH; ++u) {
					const uint32_t k = key[e0 + u];
					const double* rp = g.rays2 + TR(k != EMPTY ? (int)(k & 0xFFFFFu) : 0) * 3;
#pragma unroll
					for (int c = 0; c < 3; ++c) r2[u][c] = rp[c];
				}
#pragma unroll
				for (int u = 0; u < CH; ++u) {
					const uint32_t k = key[e0 + u];
					if (k != EMPTY && (int)(k >> 20) <= g.thLow && check_epipolar(ray1, r2[u], Em, 1e-2)) epi |= 1u << (e0 + u);
				}
			}
		}

		for (int round = 0; round < 130; ++round) {
			const unsigned long long pend = __ballot(!resolved);
			if (pend == 0ull) break;
			const int low = __ffsll((long long)pend) - 1;
			// ---- tentative decision of every unresolved lane
			int state = 0;            // 0 reject, 1 accept, 2 needs rescan
			int best = 0x7FFFFFFF, bestIdx = -1, second = 0x7FFFFFFF, secondIdx = -1;
			// bestIdx = the row this lane takes if it accepts; bestIdx / secondIdx = the rows whose capture by a lower lane voids the decision
			if (!resolved && TRI) {
				int aIdx = -1, aDist = 0, bIdx = -1, bDist = 0;
#pragma unroll
				for (int e = 0; e < K; ++e) {
					uint32_t k = key[e];
					asm volatile("" : "+v"(k));   // as below
					if (k != EMPTY && (int)(k >> 20) <= g.thLow && bIdx < 0) {
						const int idx = (int)(k & 0xFFFFFu);
						if (!((matched[idx >> 5] >> (idx & 31)) & 1u)) {
							if (aIdx < 0) { aIdx = idx; aDist = (int)(k >> 20); }
							if ((epi >> e) & 1u) { bIdx = idx; bDist = (int)(k >> 20); }
						}
					}
				}
				secondIdx = aIdx;
				if (bIdx >= 0) { state = bDist <= 2 * aDist ? 1 : 0; bestIdx = bIdx; }
				else if (!full) state = 0;                             // every candidate is in the list and none qualifies
				else if (aIdx >= 0 && dK > 2 * aDist) state = 0;       // hidden rows lie beyond DistTh (as long as a stays free)
				else state = 2;
			} else if (!resolved) {
				int n = 0;
				bool stop = false;   // wave-uniform: every active lane has its two free entries (the lists are sorted: usually within the first few)
#pragma unroll
				for (int e0 = 0; e0 < K; e0 += 4) {
					if (!stop) {
#if MCS_GREEDY_GROUP
						// the bitmap words of the group's four entries are requested TOGETHER (their addresses come from registers), then the first free ones are
						// picked branch-free: one LDS round trip per four entries where the entry-by-entry walk paid one per entry
						uint32_t kk[4], ww[4];
#pragma unroll
						for (int u = 0; u < 4; ++u) {
							uint32_t k = key[e0 + u < K ? e0 + u : K - 1];
							asm volatile("" : "+v"(k));   // (index, bitmap word and bit of every entry hoisted out of the round loop cost 100 registers)
							kk[u] = e0 + u < K ? k : EMPTY;
							ww[u] = matched[kk[u] != EMPTY ? (kk[u] & 0xFFFFFu) >> 5 : 0u];
						}
#pragma unroll
						for (int u = 0; u < 4; ++u) {
							const uint32_t k = kk[u];
							const int idx = (int)(k & 0xFFFFFu);
							const bool fr = k != EMPTY && !((ww[u] >> (idx & 31)) & 1u);
							const bool t0 = fr && n == 0, t1 = fr && n == 1;
							best = t0 ? (int)(k >> 20) : best; bestIdx = t0 ? idx : bestIdx;
							second = t1 ? (int)(k >> 20) : second; secondIdx = t1 ? idx : secondIdx;
							n += (fr && n < 2) ? 1 : 0;
						}
#else
#pragma unroll
						for (int e = e0; e < e0 + 4 && e < K; ++e) {
							uint32_t k = key[e];
							asm volatile("" : "+v"(k));   // (index, bitmap word and bit of every entry hoisted out of the round loop cost 100 registers)
							if (k != EMPTY && n < 2) {
								const int idx = (int)(k & 0xFFFFFu);
								if (!((matched[idx >> 5] >> (idx & 31)) & 1u)) {
									if (n == 0) { best = (int)(k >> 20); bestIdx = idx; } else { second = (int)(k >> 20); secondIdx = idx; }
									++n;
								}
							}
						}
#endif
						if (e0 + 4 < K) stop = __all(n >= 2 || key[e0 + 3 < K ? e0 + 3 : K - 1] == EMPTY);
					}
				}
				if (n >= 2 || !full) {
					const bool pass = bestIdx >= 0 && (g.thInclusive ? best <= g.thLow : best < g.thLow);
					state = (pass && static_cast<double>(best) < g.ratio * static_cast<double>(second)) ? 1 : 0;
				} else if (n == 1) {   // hidden rows beyond the list all have distance >= dK
					const bool pass = g.thInclusive ? best <= g.thLow : best < g.thLow;
					if (!pass) state = 0;
					else if (static_cast<double>(best) < g.ratio * static_cast<double>(dK)) state = 1;
					else state = 2;
				} else {
					const bool pass = g.thInclusive ? dK <= g.thLow : dK < g.thLow;
					state = pass ? 2 : 0;
				}
			}
			// ---- the lowest unresolved lane may need an exact rescan of the whole train set (all waves of the workgroup)
			// (the helper waves wait at barrier A; wave 0 joins them there only when a rescan is due — needScan is wave-uniform — so the usual round costs no
			// workgroup barrier at all: 2.67 -> 2.56 ms on the configs[2] sweep)
			const bool needScan = __shfl(state, low) == 2;
			if (needScan) {
				if (lane == 0) reqQ = i0 + low;
				__syncthreads();   // A
				__syncthreads();   // B  (wave 0 does not scan: with its lists live across the scan the kernel needed 254 registers — one workgroup per CU)
				uint32_t m1 = EMPTY, m2 = EMPTY;
				for (int w = 0; w < nscan; ++w) {
					const uint32_t pa = partA[w], pb = partB[w];
					if (TRI) { m1 = pa < m1 ? pa : m1; m2 = pb < m2 ? pb : m2; }
					else {   // the two smallest of all slices' two smallest
						if (pa < m1) { m2 = m1; m1 = pa; } else if (pa < m2) m2 = pa;
						if (pb < m2) m2 = pb;
					}
				}
				++nfallback;
				if (lane == low) {
					if (TRI) {
						secondIdx = -1;
						bestIdx = m2 == EMPTY ? -1 : (int)(m2 & 0xFFFFFu);
						state = (m2 != EMPTY && (int)(m2 >> 20) <= 2 * (int)(m1 >> 20)) ? 1 : 0;
					} else {
						best = m1 == EMPTY ? 0x7FFFFFFF : (int)(m1 >> 20);
						bestIdx = m1 == EMPTY ? -1 : (int)(m1 & 0xFFFFFu);
						second = m2 == EMPTY ? 0x7FFFFFFF : (int)(m2 >> 20);
						secondIdx = -1;   // exact: depends on nothing a lower lane can still change (there is no lower pending lane)
						const bool pass = bestIdx >= 0 && (g.thInclusive ? best <= g.thLow : best < g.thLow);
						state = (pass && static_cast<double>(best) < g.ratio * static_cast<double>(second)) ? 1 : 0;
					}
				}
			}
			// ---- claims and finality
			if (!resolved && state == 1) atomicMin(&claim[bestIdx], (uint32_t)lane);
			__threadfence_block();   // wave 0 only from here to the end of the round: LDS operations of one wave are carried out in order
			bool blocked = false;
			if (!resolved) {
				if (state == 2) blocked = true;
				else {
					if (bestIdx >= 0 && claim[bestIdx] < (uint32_t)lane) blocked = true;
					if (secondIdx >= 0 && claim[secondIdx] < (uint32_t)lane) blocked = true;
				}
			}
			const unsigned long long blk = __ballot(blocked);
			const int firstBlocked = blk ? __ffsll((long long)blk) - 1 : 64;
			__threadfence_block();
			if (!resolved && state == 1) claim[bestIdx] = 0xFFFFFFFFu;   // reset own claim (all claims of this round)
			const bool commit = !resolved && lane < firstBlocked;
			if (commit) {
				if (state == 1) {
					atomicOr(&matched[bestIdx >> 5], 1u << (bestIdx & 31));
					myOut = bestIdx;
				}
				resolved = true;
			}
			nmatches += __popcll(__ballot(commit && state == 1));
			__threadfence_block();
		}
		if (g.mode != 1) { if (inRange) outM[i] = myOut; }   // (a query without a good map point: -1)
		else if (myOut >= 0) outM[myOut] = i;
	}
	if (lane == 0) {
		g.outCount[set] = nmatches;
		if (g.outFallbacks) g.outFallbacks[set] = nfallback;
		reqQ = -2;
	}
	__syncthreads();   // A: releases the helper waves
}

// ---------------------------------------------------------------------------------------------------------------------
// The same greedy as a FIXPOINT, for FEW set pairs (one multi-frame against one keyframe: the latency of a live tracker's call).  The sequential loop is a
// lower-triangular system: query i's outcome A_i (the row it takes, or none) is a function f_i of { A_j : j < i } — a row is "taken" for i iff some earlier
// query takes it.  Iterating  A_i <- f_i({A_j : j < i})  for ALL queries at once converges to the unique solution of that system (after t sweeps the first t
// queries are final), which is the sequential result; dependencies are short chains (a query depends on the few earlier queries that compete for its two
// nearest free rows), so a handful of sweeps suffices where k_greedy_spec walks 48 chunks of 64 queries one after the other (0.25 ms for ONE pair).
//   sweep:  owner[r] = the lowest query that currently takes row r (LDS, atomicMin);  then every query re-derives its decision from its sorted list with
//           "taken" = owner[r] < i — the same best / second / threshold / ratio rules as above;  queries whose list cannot decide (its K entries used up) keep
//           their previous outcome during the sweeps and are rescanned EXACTLY (all waves, the whole train set, free = owner >= i) once the sweeps are stable;
//           a rescan that changes an outcome restarts the sweeps.  The loop ends only after a pass in which nothing changed and every such query was rescanned
//           against the final owners: every equation of the system holds.
// One workgroup of 1024 threads per set pair; a thread owns queries tid, tid + 1024, ...; the first kJacCache entries of every list sit in LDS.
#ifndef MCS_JAC_EXT
#define MCS_JAC_EXT 4
#endif
constexpr int kJacBatch = 4;   // undecidable queries rescanned per pass over the train rows
constexpr int kJacOwn = 3, kJacExt = MCS_JAC_EXT;   // queries per thread whose list entries 8 .. 8 + kJacExt - 1 sit in registers (sets of up to 3072 queries: all of them)
constexpr int kJacThreads = 1024, kJacCache = 8;   // list entries per query kept in LDS (a walk beyond them reads the list in memory: a dependent round trip per entry)
__host__ __device__ constexpr size_t jacobi_lds_words(int nq, int nt, int K) { return (size_t)2 * (nt + 1) + nq + (size_t)(K < kJacCache ? K : kJacCache) * nq + nq; }
template <int K, int DW, bool MASKED>
__global__ __launch_bounds__(kJacThreads) void k_greedy_jacobi(GreedyArgs g) {
	// LDS: owner[2][nt] — the lowest query that takes a row, as  (0xFFFF - sweep tag) << 16 | query  under atomicMin: entries of older sweeps compare larger and are
	// simply ignored, so nothing is cleared, and the claims of sweep t + 1 go into the other buffer while sweep t is still being read: ONE barrier per sweep —,
	// A[nq] the outcomes, head[C][nq] the first C list entries, last[nq] the K-th entry
	extern __shared__ uint32_t greedy_lds[];
	constexpr int C = K < kJacCache ? K : kJacCache;
	uint32_t* owner = greedy_lds;
	const int nt1 = g.nt + 1;   // a claim buffer = nt rows + one sentinel slot that always reads 0 ("taken for everybody"): where list entries past a list's end point
	int* A = reinterpret_cast<int*>(greedy_lds + 2 * (size_t)nt1);
	uint32_t* head = greedy_lds + 2 * (size_t)nt1 + g.nq;
	uint32_t* lastK = head + (size_t)C * g.nq;
	__shared__ int changed[3], nRescan[3], rescanQ[3][64];   // per sweep, slot = sweep % 3: a slot is cleared two sweeps before it is used (one barrier per sweep)
	__shared__ uint32_t partA[kJacThreads / 64], partB[kJacThreads / 64];
	const int set = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const RowMap QR{(size_t)(set % g.qmod) * g.qpitch, g.qblk, g.qbpitch}, TR{(size_t)((set / g.tdiv + g.toff) % g.tmod) * g.tpitch, g.tblk, g.tbpitch};
	constexpr uint32_t EMPTY = 0xFFFFFFFFu;
	const bool grouped = g.qgroup != nullptr && g.tgroup != nullptr;
	const uint32_t* keys = g.keys + (size_t)set * K * g.nq;   // [K][nq]
	int* outM = g.outMatch + (size_t)set * (g.mode == 1 ? g.nt : g.nq);
	for (int j = tid; j < 2 * nt1; j += kJacThreads) owner[j] = (j == g.nt || j == 2 * nt1 - 1) ? 0u : EMPTY;
	if (g.mode == 1) for (int j = tid; j < g.nt; j += kJacThreads) outM[j] = -1;
	for (int i = tid; i < g.nq; i += kJacThreads) {
		const bool ok = !g.qvalid || g.qvalid[QR(i)] != 0;   // a query without a map point never takes a row: an empty list
		A[i] = -1;
#pragma unroll
		for (int e = 0; e < C; ++e) head[(size_t)e * g.nq + i] = ok ? keys[(size_t)e * g.nq + i] : EMPTY;
		lastK[i] = ok ? keys[(size_t)(K - 1) * g.nq + i] : EMPTY;
	}
	if (tid < 3) { changed[tid] = 0; nRescan[tid] = 0; }
	uint32_t extq[kJacOwn][kJacExt];   // entries C .. C + kJacExt - 1 of the thread's first kJacOwn queries
#pragma unroll
	for (int u = 0; u < kJacOwn; ++u) {
		const int i = tid + u * kJacThreads;
		const bool ok = i < g.nq && (!g.qvalid || g.qvalid[QR(i)] != 0);
#pragma unroll
		for (int e = 0; e < kJacExt; ++e) extq[u][e] = (ok && C + e < K) ? keys[(size_t)(C + e) * g.nq + i] : EMPTY;
	}
	__syncthreads();
	// decision of query i from its list under "taken = a lower query claims the row in buffer `own` with tag `tag`": 0 none, 1 takes *row, 2 the list cannot decide
	// "taken for query i" in ONE unsigned compare (round 6): tags fall from sweep to sweep and a buffer is always read with the newest tag it holds, so a claim word of an
	// older sweep (larger tag), and EMPTY, compare ABOVE (tag << 16 | i), and so does a current claim by a query >= i; only current claims by lower queries fall below
	auto taken = [&](const uint32_t* own, uint32_t tag, int row, int i) { return own[row] < ((tag << 16) | (uint32_t)i); };
	// the rules on (free rows found, best, second, the list's K-th entry): 0 none, 1 take the best row, 2 the list cannot decide
	auto verdict = [&](int n, int best, int bestIdx, int second, uint32_t last) {
		const bool full = last != EMPTY;
		const int dK = full ? (int)(last >> 20) : 0x7FFFFFFF;
		if (n >= 2 || !full) {
			const bool pass = bestIdx >= 0 && (g.thInclusive ? best <= g.thLow : best < g.thLow);
			return (pass && static_cast<double>(best) < g.ratio * static_cast<double>(second)) ? 1 : 0;
		}
		if (n == 1) {   // hidden rows beyond the list all have distance >= dK
			const bool pass = g.thInclusive ? best <= g.thLow : best < g.thLow;
			if (!pass) return 0;
			return static_cast<double>(best) < g.ratio * static_cast<double>(dK) ? 1 : 2;
		}
		return (g.thInclusive ? dK <= g.thLow : dK < g.thLow) ? 2 : 0;
	};
	// The sweeps are bound by the INSTRUCTION COUNT of this function (3000 queries on the four SIMDs of one CU: 5 - 10 us per sweep at ~200 instructions per query, the LDS
	// round trips hidden by the other waves), so it is written for few instructions: the C cached entries and their claim words in two batches of LDS reads, then ONE
	// pass from the last entry to the first that keeps the (first, second) free entries seen so far — a free entry e becomes the first and pushes the old first to second:
	// two selects per entry under the entry's compare mask, no counters, no halves (every wave met a lane that needed the second half anyway).
	auto decide = [&](const uint32_t* own, uint32_t tag, int i, int* row, const uint32_t* ext = nullptr) {   // ext: entries C .. C + kJacExt - 1 of the list, held by the caller
		const uint32_t lim = (tag << 16) | (uint32_t)i;
		uint32_t kk[C], vv[C];
#pragma unroll
		for (int e = 0; e < C; ++e) kk[e] = head[(size_t)e * g.nq + i];
#pragma unroll
		for (int e = 0; e < C; ++e) vv[e] = own[min(kk[e] & 0xFFFFFu, (uint32_t)g.nt)];   // EMPTY -> the buffer's sentinel slot [nt] = 0: "taken" for everybody
		uint32_t k1 = EMPTY, k2 = EMPTY;   // the first and the second free entry (keys: distance << 20 | row; the lists are sorted, EMPTY pads their ends)
#pragma unroll
		for (int e = C - 1; e >= 0; --e) {
			const bool fr = vv[e] >= lim;
			k2 = fr ? k1 : k2;
			k1 = fr ? kk[e] : k1;
		}
		if (k2 == EMPTY && kk[C - 1] != EMPTY) {   // fewer than two free rows in the cached head and the list goes on (a query whose nearest rows are mostly taken by lower queries: the late ones)
			// several entries per round trip (entry by entry this was a dependent memory round trip each, and the slowest lane of the workgroup sets the pace of every
			// sweep: 13 us per sweep): first the caller's register copy of entries C .. C + kJacExt - 1, then the list in memory, eight at a time
			bool ended = false;
			auto batch = [&](const uint32_t* kb, auto wc) {
				constexpr int W = decltype(wc)::value;
				uint32_t vb[W];
#pragma unroll
				for (int u = 0; u < W; ++u) vb[u] = own[kb[u] != EMPTY ? (kb[u] & 0xFFFFFu) : 0u];
#pragma unroll
				for (int u = 0; u < W; ++u) {   // in list order: the first free entry fills k1, the next k2
					ended = ended || kb[u] == EMPTY;
					const bool fr = !ended && vb[u] >= lim;
					const bool t0 = fr && k1 == EMPTY, t1 = fr && !t0 && k2 == EMPTY;
					k1 = t0 ? kb[u] : k1;
					k2 = t1 ? kb[u] : k2;
				}
			};
			int e0 = C;
			if (ext) { batch(ext, std::integral_constant<int, kJacExt>()); e0 = C + kJacExt; }
			for (; e0 < K && k2 == EMPTY && !ended; e0 += 8) {
				uint32_t k8[8];
#pragma unroll
				for (int u = 0; u < 8; ++u) k8[u] = e0 + u < K ? keys[(size_t)(e0 + u) * g.nq + i] : EMPTY;
				batch(k8, std::integral_constant<int, 8>());
			}
		}
		const int n = (k1 != EMPTY ? 1 : 0) + (k2 != EMPTY ? 1 : 0);
		*row = k1 != EMPTY ? (int)(k1 & 0xFFFFFu) : -1;
		return verdict(n, k1 != EMPTY ? (int)(k1 >> 20) : 0x7FFFFFFF, *row, k2 != EMPTY ? (int)(k2 >> 20) : 0x7FFFFFFF, lastK[i]);
	};
	// exact rescan of query qi by the whole workgroup: the two smallest keys among the eligible rows no lower query takes -> the query's outcome
	auto rescan = [&](const uint32_t* own, uint32_t tag, int qi) -> int {
		uint32_t q[DW], qm[DW];
		const uint32_t* qp = reinterpret_cast<const uint32_t*>(g.qd + QR(qi) * g.qstride);
#pragma unroll
		for (int w = 0; w < DW; ++w) q[w] = qp[w];
		if (MASKED) {
			const uint32_t* mp = reinterpret_cast<const uint32_t*>(g.qm + QR(qi) * g.qstride);
#pragma unroll
			for (int w = 0; w < DW; ++w) qm[w] = mp[w];
		}
		const int qg = grouped ? g.qgroup[QR(qi)] : 0;
		uint32_t a = EMPTY, b2 = EMPTY;
		// rows as 16-byte words where the layout allows it (a 32-byte row is two loads instead of eight)
		const bool wide = DW % 4 == 0 && ((((uintptr_t)g.td | (uintptr_t)(MASKED ? g.tm : g.td)) | (uintptr_t)g.tstride) & 15u) == 0;
		for (int j = tid; j < g.nt; j += kJacThreads) {
			const uint8_t* tb = g.td + TR(j) * g.tstride;
			const uint8_t* mb = MASKED ? g.tm + TR(j) * g.tstride : tb;
			uint32_t tw[DW], mw[DW];
			if (wide) {
#pragma unroll
				for (int w = 0; w < DW / 4; ++w) {
					const uint4 v = reinterpret_cast<const uint4*>(tb)[w];
					tw[4 * w] = v.x; tw[4 * w + 1] = v.y; tw[4 * w + 2] = v.z; tw[4 * w + 3] = v.w;
					if (MASKED) { const uint4 u = reinterpret_cast<const uint4*>(mb)[w]; mw[4 * w] = u.x; mw[4 * w + 1] = u.y; mw[4 * w + 2] = u.z; mw[4 * w + 3] = u.w; }
				}
			} else {
#pragma unroll
				for (int w = 0; w < DW; ++w) { tw[w] = reinterpret_cast<const uint32_t*>(tb)[w]; if (MASKED) mw[w] = reinterpret_cast<const uint32_t*>(mb)[w]; }
			}
			const bool ok = !taken(own, tag, j, qi) && (g.tvalid ? g.tvalid[TR(j)] != 0 : true) && (!grouped || g.tgroup[TR(j)] == qg);
			const uint32_t k = ok ? (((uint32_t)hamming_g<DW, MASKED>(q, qm, tw, MASKED ? mw : tw) << 20) | (uint32_t)j) : EMPTY;
			if (k < a) { b2 = a; a = k; } else if (k < b2) b2 = k;
		}
		const uint32_t m1 = wave_min_u32(a);
		const uint32_t m2 = wave_min_u32(a == m1 ? b2 : a);
		if (lane == 0) { partA[wave] = m1; partB[wave] = m2; }
		__syncthreads();
		uint32_t r1 = EMPTY, r2 = EMPTY;
		for (int w = 0; w < kJacThreads / 64; ++w) {
			const uint32_t pa = partA[w], pb = partB[w];
			if (pa < r1) { r2 = r1; r1 = pa; } else if (pa < r2) r2 = pa;
			if (pb < r2) r2 = pb;
		}
		__syncthreads();
		const int best = r1 == EMPTY ? 0x7FFFFFFF : (int)(r1 >> 20), second = r2 == EMPTY ? 0x7FFFFFFF : (int)(r2 >> 20);
		const bool pass = r1 != EMPTY && (g.thInclusive ? best <= g.thLow : best < g.thLow);
		return (pass && static_cast<double>(best) < g.ratio * static_cast<double>(second)) ? (int)(r1 & 0xFFFFFu) : -1;
	};
	// the same for up to kJacBatch undecidable queries in ONE pass over the train rows (a row is loaded once and compared with every query of the batch; one pair of
	// barriers for all of them): calls with a handful of such queries paid 9 us for each
	__shared__ uint32_t bq[kJacBatch][2 * DW];
	__shared__ int bqi[kJacBatch], bgrp[kJacBatch], bres[kJacBatch];
	__shared__ uint32_t bpart[kJacBatch][kJacThreads / 64][2];
	auto rescan_batch = [&](const uint32_t* own, uint32_t tag, const int* ql, int nb) {   // results in bres[0 .. nb)
		for (int x = tid; x < nb * 2 * DW; x += kJacThreads) {
			const int bb = x / (2 * DW), w = x - bb * 2 * DW, qi = ql[bb];
			bq[bb][w] = w < DW ? reinterpret_cast<const uint32_t*>(g.qd + QR(qi) * g.qstride)[w]
			                   : (MASKED ? reinterpret_cast<const uint32_t*>(g.qm + QR(qi) * g.qstride)[w - DW] : 0u);
		}
		if (tid < nb) { bqi[tid] = ql[tid]; bgrp[tid] = grouped ? g.qgroup[QR(ql[tid])] : 0; }
		__syncthreads();
		uint32_t a[kJacBatch], b2[kJacBatch];
#pragma unroll
		for (int bb = 0; bb < kJacBatch; ++bb) { a[bb] = EMPTY; b2[bb] = EMPTY; }
		const bool wide = DW % 4 == 0 && ((((uintptr_t)g.td | (uintptr_t)(MASKED ? g.tm : g.td)) | (uintptr_t)g.tstride) & 15u) == 0;
		for (int j = tid; j < g.nt; j += kJacThreads) {
			const uint8_t* tb = g.td + TR(j) * g.tstride;
			const uint8_t* mb = MASKED ? g.tm + TR(j) * g.tstride : tb;
			uint32_t tw[DW], mw[DW];
			if (wide) {
#pragma unroll
				for (int w = 0; w < DW / 4; ++w) {
					const uint4 v = reinterpret_cast<const uint4*>(tb)[w];
					tw[4 * w] = v.x; tw[4 * w + 1] = v.y; tw[4 * w + 2] = v.z; tw[4 * w + 3] = v.w;
					if (MASKED) { const uint4 u = reinterpret_cast<const uint4*>(mb)[w]; mw[4 * w] = u.x; mw[4 * w + 1] = u.y; mw[4 * w + 2] = u.z; mw[4 * w + 3] = u.w; }
				}
			} else {
#pragma unroll
				for (int w = 0; w < DW; ++w) { tw[w] = reinterpret_cast<const uint32_t*>(tb)[w]; if (MASKED) mw[w] = reinterpret_cast<const uint32_t*>(mb)[w]; }
			}
			const uint32_t v = own[j];
			const bool valid = g.tvalid ? g.tvalid[TR(j)] != 0 : true;
			const int tg = grouped ? g.tgroup[TR(j)] : 0;
#pragma unroll
			for (int bb = 0; bb < kJacBatch; ++bb) {
				if (bb < nb) {
					const bool ok = valid && !(v < ((tag << 16) | (uint32_t)bqi[bb])) && (!grouped || tg == bgrp[bb]);   // (`taken`, on the word already loaded)
					const uint32_t k = ok ? (((uint32_t)hamming_g<DW, MASKED>(bq[bb], bq[bb] + DW, tw, MASKED ? mw : tw) << 20) | (uint32_t)j) : EMPTY;
					if (k < a[bb]) { b2[bb] = a[bb]; a[bb] = k; } else if (k < b2[bb]) b2[bb] = k;
				}
			}
		}
#pragma unroll
		for (int bb = 0; bb < kJacBatch; ++bb) {
			if (bb < nb) {
				const uint32_t m1 = wave_min_u32(a[bb]);
				const uint32_t m2 = wave_min_u32(a[bb] == m1 ? b2[bb] : a[bb]);
				if (lane == 0) { bpart[bb][wave][0] = m1; bpart[bb][wave][1] = m2; }
			}
		}
		__syncthreads();
		if (tid < nb) {
			uint32_t r1 = EMPTY, r2 = EMPTY;
			for (int w = 0; w < kJacThreads / 64; ++w) {
				const uint32_t pa = bpart[tid][w][0], pb = bpart[tid][w][1];
				if (pa < r1) { r2 = r1; r1 = pa; } else if (pa < r2) r2 = pa;
				if (pb < r2) r2 = pb;
			}
			const int best = r1 == EMPTY ? 0x7FFFFFFF : (int)(r1 >> 20), second = r2 == EMPTY ? 0x7FFFFFFF : (int)(r2 >> 20);
			const bool pass = r1 != EMPTY && (g.thInclusive ? best <= g.thLow : best < g.thLow);
			bres[tid] = (pass && static_cast<double>(best) < g.ratio * static_cast<double>(second)) ? (int)(r1 & 0xFFFFFu) : -1;
		}
		__syncthreads();
	};
	// sweep t reads the claims of buffer t & 1 made with tag t (none for t = 0: every row free) and writes the claims of the new outcomes into the other buffer
	// Budget (round 6): the fixpoint needs as many sweeps as the longest dependency chain — 9 to 14 on real frames — but nothing bounds a chain below nq, every
	// stable state with undecidable queries costs an exact rescan each, and a rescan that changes an outcome restarts the sweeps: on clustered descriptors that can
	// reach (undecidable queries) x (chain length) iterations.  Past kMaxSweeps sweeps or nq rescans the loop is given up and the set is resolved IN ORDER by exact
	// rescans (the sequential greedy itself, about nq passes over the train rows by the whole workgroup), which cannot fail; `converged` says which way it ended.
#ifdef MCS_JAC_DEBUG   // A/B builds only: device timestamps (100 MHz) of the sweeps, printed by thread 0 (tools/jd_probe.py)
	long long stamp[40]; int nstamp = 0;
	const long long tStart = (long long)wall_clock64();
#define JSTAMP() do { if (nstamp < 40) stamp[nstamp++] = (long long)wall_clock64() - tStart; } while (0)
#else
#define JSTAMP() do {} while (0)
#endif
	JSTAMP();
	const int maxSweeps = g.jacMaxSweeps > 0 ? g.jacMaxSweeps : 256;
	bool converged = false;
	int rescansDone = 0;
	int nfallback = 0;
	uint32_t t = 0;
	for (int guard = 0; guard < maxSweeps && rescansDone <= g.nq; ++guard) {
		const uint32_t* own = owner + (size_t)(t & 1u) * nt1;
		uint32_t* nxt = owner + (size_t)((t + 1u) & 1u) * nt1;
		const uint32_t tag = 0xFFFFu - (t & 0xFFFFu), ntag = 0xFFFFu - ((t + 1u) & 0xFFFFu);   // (tags repeat after 65536 sweeps; a sweep count is bounded by nq + passes, far below)
		const int slot = (int)(t % 3u);
		bool ch = false;
		// a thread's queries; the first kJacOwn of them with entries C .. C + 7 of their lists in registers (loaded once): late queries find most of their nearest rows
		// taken by lower queries and walk a dozen entries — from memory that was a round trip per eight entries in every sweep, and the slowest wave sets the sweep's pace
		auto one = [&](int i, const uint32_t* ext) {
			int row, na = A[i];
			const int old = na;
			const int st = decide(own, tag, i, &row, ext);
			if (st == 2) { const int at = atomicAdd(&nRescan[slot], 1); if (at < 64) rescanQ[slot][at] = i; }   // keeps its outcome until the exact rescan below
			else { na = st == 1 ? row : -1; if (na != old) { A[i] = na; ch = true; } }
			if (na >= 0) atomicMin(&nxt[na], (ntag << 16) | (uint32_t)i);
		};
#pragma unroll
		for (int u = 0; u < kJacOwn; ++u) { const int i = tid + u * kJacThreads; if (i < g.nq) one(i, K > C ? extq[u] : nullptr); }
		for (int i = tid + kJacOwn * kJacThreads; i < g.nq; i += kJacThreads) one(i, nullptr);
		if (ch) changed[slot] = 1;
		__syncthreads();
		JSTAMP();
		const bool again = changed[slot] != 0;
		const int nr = nRescan[slot];
		if (tid == 0) { const int s2 = (int)((t + 2u) % 3u); changed[s2] = 0; nRescan[s2] = 0; }   // the flags of the sweep after the next (last read before this barrier, next written after the next one)
		++t;
		if (again) continue;
		// ---- stable: the queries whose lists could not decide, exactly, against the claims of the stable state (= the next buffer: every outcome re-claimed there)
		if (nr == 0) { converged = true; break; }
		rescansDone += nr;
		bool any = false;
		auto settle = [&](int qi) {   // (uniform: every thread sees the same old and new outcome)
			const int na = rescan(nxt, ntag, qi);
			const int old = A[qi];
			__syncthreads();
			if (na != old) { any = true; if (tid == 0) A[qi] = na; }
		};
		nfallback = 0;
		if (nr > 64) {   // (more than the queue holds: found again one by one)
			for (int i = 0; i < g.nq; ++i) {
				if (tid == 0) { int row = -1; rescanQ[slot][0] = decide(nxt, ntag, i, &row); }
				__syncthreads();
				const int st = rescanQ[slot][0];
				__syncthreads();
				if (st != 2) continue;
				++nfallback;
				settle(i);
			}
		} else {
			for (int k0 = 0; k0 < nr; k0 += kJacBatch) {
				const int nb = min(kJacBatch, nr - k0);
				rescan_batch(nxt, ntag, &rescanQ[slot][k0], nb);
				for (int k = 0; k < nb; ++k) {   // (uniform: every thread sees the same old and new outcome)
					const int qi = rescanQ[slot][k0 + k], na = bres[k], old = A[qi];
					__syncthreads();
					if (na != old) { any = true; if (tid == 0) A[qi] = na; }
				}
				__syncthreads();   // (bres and the queue entries are read before the next batch overwrites them)
			}
			nfallback = nr;
		}
		__syncthreads();
		if (!any) { converged = true; break; }
		// a rescan changed an outcome: the claims of buffer `nxt` are stale for that query — rebuild them under a fresh tag by one more pass over the outcomes
		++t;   // (skip a tag: buffer (t & 1) is `own` again, written with the tag of sweep t)
		{
			uint32_t* cur = owner + (size_t)(t & 1u) * nt1;
			const uint32_t ctag = 0xFFFFu - (t & 0xFFFFu);
			for (int i = tid; i < g.nq; i += kJacThreads) { const int r = A[i]; if (r >= 0) atomicMin(&cur[r], (ctag << 16) | (uint32_t)i); }
			if (tid < 3) { changed[tid] = 0; nRescan[tid] = 0; }
			__syncthreads();
		}
	}
	if (!converged) {
		// ---- the budget ran out: the sequential greedy itself.  Query after query in ascending order, each by an exact rescan of the whole train set against the
		// claims of the queries before it (a fresh buffer and tag: nothing of the sweeps is read), its row claimed before the next one looks.
		__syncthreads();
		t += 2;
		uint32_t* cur = owner + (size_t)(t & 1u) * nt1;
		const uint32_t ctag = 0xFFFFu - (t & 0xFFFFu);
		for (int i = 0; i < g.nq; ++i) {
			const bool ok = !g.qvalid || g.qvalid[QR(i)] != 0;
			const int na = ok ? rescan(cur, ctag, i) : -1;   // (uniform over the workgroup; barriers inside)
			if (tid == 0) { A[i] = na; if (na >= 0) cur[na] = (ctag << 16) | (uint32_t)i; }
			__syncthreads();
		}
		nfallback = g.nq;
	}
	JSTAMP();
	// ---- results
	int nm = 0;
	for (int i = tid; i < g.nq; i += kJacThreads) {
		const int r = A[i];
		if (g.mode != 1) outM[i] = r;
		else if (r >= 0) outM[r] = i;
		nm += r >= 0 ? 1 : 0;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) nm += __shfl_xor(nm, o);
	if (lane == 0) partA[wave] = (uint32_t)nm;
	__syncthreads();
	if (tid == 0) {
		int tot = 0;
		for (int w = 0; w < kJacThreads / 64; ++w) tot += (int)partA[w];
		g.outCount[set] = tot;
#ifdef MCS_JAC_DEBUG
		JSTAMP();
		printf("jac: sweeps %d fb %d stamps(10ns):", (int)t, nfallback);
		for (int k = 0; k < nstamp; ++k) printf(" %lld", stamp[k]);
		printf("\n");
#endif
		if (g.outFallbacks) g.outFallbacks[set] = nfallback;
	}
}

template <int K, int DW, bool MASKED, bool TRI>
static void launch_spec(const GreedyArgs& g, hipStream_t s) {
	const size_t lds = (size_t)(g.nt + (g.nt + 31) / 32) * 4;   // claim[nt] + matched bitmap
	if (lds > 60 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_greedy_spec<K, DW, MASKED, TRI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipLaunchKernelGGL((k_greedy_spec<K, DW, MASKED, TRI>), dim3(g.nsets), dim3(64 * (g.nsets >= kManySets ? kSpecWavesMany : kSpecWaves)), lds, s, g);
}

// how much dynamic LDS a workgroup of this device may ask for beside the kernel's ~4 KB of static arrays (queried once per device; 160 KB per CU on gfx950)
static size_t jacobi_lds_limit() {
	static thread_local int dev = -1;
	static thread_local size_t limit = 0;
	int d = 0;
	if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return 0; }
	if (d != dev) {
		int optin = 0;
		if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, d) != hipSuccess) { (void)hipGetLastError(); optin = 64 * 1024; }
		dev = d;
		limit = optin > 8 * 1024 ? (size_t)optin - 8 * 1024 : 0;
	}
	return limit;
}

template <int K, int DW, bool MASKED>
static bool launch_jacobi(const GreedyArgs& g, hipStream_t s) {   // false: this device cannot give the workgroup its LDS — the caller takes k_greedy_spec
	const size_t lds = jacobi_lds_words(g.nq, g.nt, K) * 4;
	if (lds > jacobi_lds_limit()) return false;
	// (the attribute belongs to the function ON THE CURRENT DEVICE, and one process may drive several devices from several threads: set before every launch that needs it)
	if (lds > 60 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_greedy_jacobi<K, DW, MASKED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
		(void)hipGetLastError();
		return false;
	}
	hipLaunchKernelGGL((k_greedy_jacobi<K, DW, MASKED>), dim3(g.nsets), dim3(kJacThreads), lds, s, g);
	return true;
}

template <int K, int DW>
static void launch_spec_kd(const GreedyArgs& g, hipStream_t s) {
	const bool masked = g.qm && g.tm;
	// few set pairs (a tracker's one multi-frame against one keyframe): the fixpoint form, a whole workgroup per pair; MCS_GREEDY_JACOBI = the largest number of
	// pairs that takes it (0: never, for A/B and tests)
	static const int jacSets = getenv("MCS_GREEDY_JACOBI") ? atoi(getenv("MCS_GREEDY_JACOBI")) : 8;
	if (g.mode != 2 && g.nsets <= jacSets && g.nq < 65536 && jacobi_lds_words(g.nq, g.nt, K) * 4 <= 150 * 1024) {
		if (masked ? launch_jacobi<K, DW, true>(g, s) : launch_jacobi<K, DW, false>(g, s)) return;
	}
	if (g.mode == 2) { if (masked) launch_spec<K, DW, true, true>(g, s); else launch_spec<K, DW, false, true>(g, s); }
	else if (masked) launch_spec<K, DW, true, false>(g, s);
	else launch_spec<K, DW, false, false>(g, s);
}

template <int DW>
static void launch_dw(const GreedyArgs& g, hipStream_t s) {
	if (g.nt <= kClaimRows && (g.mode != 2 || (g.qgroup && g.tgroup))) {
		switch (g.K) {
			case 1: launch_spec_kd<1, DW>(g, s); return;
			case 2: launch_spec_kd<2, DW>(g, s); return;
			case 4: launch_spec_kd<4, DW>(g, s); return;
			case 8: launch_spec_kd<8, DW>(g, s); return;
			case 16: launch_spec_kd<16, DW>(g, s); return;
			default: launch_spec_kd<32, DW>(g, s); return;
		}
	}
	if (g.qm && g.tm) hipLaunchKernelGGL((k_greedy<DW, true>), dim3(g.nsets), dim3(64), 0, s, g);
	else hipLaunchKernelGGL((k_greedy<DW, false>), dim3(g.nsets), dim3(64), 0, s, g);
}

void launch_greedy(const GreedyArgs& g, hipStream_t s) {
	if (g.dim == 16) launch_dw<4>(g, s);
	else if (g.dim == 32) launch_dw<8>(g, s);
	else launch_dw<16>(g, s);
}

__global__ void k_rows_valid(const int* nkp, int nimg, int cap, uint8_t* valid) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nimg * cap) valid[i] = (i % cap) < nkp[i / cap] ? 1 : 0;
}

// Exchange blocks of the camera-sharded rig (rig.py): image block = cap descriptor rows + ONE header row whose first 4 bytes hold the image's keypoint count.
__global__ void k_rig_pack_headers(const int* nkp, int nimg, int cap, uint8_t* blocks, int rowStride) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nimg) *reinterpret_cast<int*>(blocks + ((size_t)i * (cap + 1) + cap) * rowStride) = nkp[i];
}
__global__ void k_rig_rows_valid(const uint8_t* blocks, int nimg, int cap, int rowStride, uint8_t* valid, int* nkpOut) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nimg * (cap + 1)) return;
	const int img = i / (cap + 1), k = i - img * (cap + 1);
	const int n = *reinterpret_cast<const int*>(blocks + ((size_t)img * (cap + 1) + cap) * rowStride);
	valid[i] = k < n && k < cap ? 1 : 0;   // the header row itself is never a feature
	if (nkpOut && k == 0) nkpOut[img] = n;
}
void launch_rig_pack_headers(const int* nkp, int nimg, int cap, uint8_t* blocks, int rowStride, hipStream_t s) {
	hipLaunchKernelGGL(k_rig_pack_headers, dim3((nimg + 255) / 256), dim3(256), 0, s, nkp, nimg, cap, blocks, rowStride);
}
void launch_rig_rows_valid(const uint8_t* blocks, int nimg, int cap, int rowStride, uint8_t* valid, int* nkpOut, hipStream_t s) {
	hipLaunchKernelGGL(k_rig_rows_valid, dim3((nimg * (cap + 1) + 255) / 256), dim3(256), 0, s, blocks, nimg, cap, rowStride, valid, nkpOut);
}

void launch_rows_valid(const int* nkp, int nimg, int cap, uint8_t* valid, hipStream_t s) {
	hipLaunchKernelGGL(k_rows_valid, dim3((nimg * cap + 255) / 256), dim3(256), 0, s, nkp, nimg, cap, valid);
}

}  // namespace mcs

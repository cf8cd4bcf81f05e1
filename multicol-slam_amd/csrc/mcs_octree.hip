// mcs_octree.hip — E3/E4: DistributeOctTree on the GPU, one 256-thread workgroup per (image, pyramid level).
// Reference: src/mdBRIEFextractorOct.cpp:569-629 (DivideNode), 631-861 (DistributeOctTree) — a sequential std::list
// walk that push_front()s children.  Here the node list is REBUILT once per pass from prefix sums and keys never move:
// a key only stores the list position of its node (they keep their original order inside a node, exactly like the
// reference's vKeys).  tests/octree_array_model.py is the executable specification of this formulation and
// tests/test_octree_model.py proves it equal to the literal list-based oracle.  Pure integer arithmetic (FAST
// coordinates are integers), so the result is bit-exact by construction.
//
//   pass, phase A (:698-763):  every node with >1 key is split, in list order.
//   pass, phase B (:774-835):  nodes created by the previous pass with >1 key, largest first (ties: later created
//                              first — the documented stand-in for the reference's heap-address tie-break), stop as
//                              soon as the list holds >= N nodes.
//   new list = children of the LAST processed node first (each node: n4,n3,n2,n1), then the untouched nodes in order.
//   finally one key per node: max response, first wins ties (:840-858); output order = list order.
//
// Prologue: the per-cell candidate slots written by the FAST kernel are compacted into the level's dense list in the
// reference's order (cell row-major, then row-major inside the cell) with a prefix sum over the cell counts.
#include "mcs_common.h"
#include "mcs_orient.h"

namespace mcs {

// MAXN = node capacity (host guarantees nfeat+3 <= MAXN and 4*nIni <= MAXN); KCACHE = keys kept in LDS (larger levels use HBM)

template <int MAXN>
struct NodeBuf {
	short x0[MAXN], x1[MAXN], y0[MAXN], y1[MAXN];
	int cnt[MAXN];
	short cre[MAXN];   // creation index inside the pass that made the node if cnt > 1, else -1
};

// inclusive prefix sum over the 64 lanes of a wave on the DPP path (no LDS round trips): four row_shr steps inside the rows of 16, then row_bcast:15 / :31
__device__ __forceinline__ int oct_wave_incl_scan(int x) {
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
	return x;
}
// a wave's own accesses to LDS stay in program order; this keeps the compiler from moving them across a step of the node-level passes
#define WFENCE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// exclusive scan of a[0..n) (n <= 64 * IPTW) by ONE wave, IPTW consecutive entries per lane; returns the total.  No workgroup barrier.
template <int IPTW>
__device__ __forceinline__ int wave_exscan(int* a, int n) {
	const int lane = threadIdx.x & 63, base = lane * IPTW;
	int v[IPTW];
	int s = 0;
#pragma unroll
	for (int j = 0; j < IPTW; ++j) { v[j] = base + j < n ? a[base + j] : 0; s += v[j]; }
	const int x = oct_wave_incl_scan(s);
	const int total = __builtin_amdgcn_readlane(x, 63);
	int ex = x - s;
#pragma unroll
	for (int j = 0; j < IPTW; ++j) { if (base + j < n) a[base + j] = ex; ex += v[j]; }
	WFENCE();
	return total;
}

// exclusive scan of a[0..n) (n <= 256 * IPT) by the whole 256-thread block, IPT consecutive entries per thread; returns the total.  Caller must have synced.
template <int IPT>
__device__ int block_exscan(int* a, int n, int* wsum) {
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int base = tid * IPT;
	int v[IPT];
	int s = 0;
#pragma unroll
	for (int j = 0; j < IPT; ++j) { v[j] = base + j < n ? a[base + j] : 0; s += v[j]; }
	int x = s;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		int y = __shfl_up(x, o);
		if (lane >= o) x += y;
	}
	if (lane == 63) wsum[wave] = x;
	__syncthreads();
	int woff = 0;
	for (int w = 0; w < wave; ++w) woff += wsum[w];
	const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
	int ex = woff + x - s;
#pragma unroll
	for (int j = 0; j < IPT; ++j) { if (base + j < n) a[base + j] = ex; ex += v[j]; }
	__syncthreads();
	return total;
}

#ifdef MCS_OCT_TRACE   // debugging aid: workgroup 0 prints where its time goes (100 MHz ticks), three launches
__device__ int g_octTraceLaunches = 0;
#define OCT_T(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (k) < 24) octT[(k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define OCT_T(k) do { } while (0)
#endif
template <int MAXN, int KCACHE>
__attribute__((amdgpu_waves_per_eu(5, 5)))   // 102 registers: five workgroups per CU stay possible (the orientation tail in its round-4 form would take 113)
__global__ __launch_bounds__(256) void k_octree(ExtractBuffers b, int nimg, int level0) {
	__shared__ NodeBuf<MAXN> nb[2];
	__shared__ uint32_t kd[KCACHE];          // LDS copy of the level's candidate records ...
	__shared__ unsigned short kn[KCACHE];    // ... and of the node position of every key, when the level fits
	__shared__ int cc[MAXN * 4];
	__shared__ short mapq[MAXN * 4];
	__shared__ int scanA[MAXN];
	__shared__ int scanB[MAXN];
	__shared__ short crank[MAXN];
	__shared__ short byRank[MAXN];
	__shared__ int wsum[4];
	__shared__ int shR;

#ifdef MCS_OCT_TRACE
	__shared__ unsigned long long octT[24];
	int octPass = 0;
#endif
	OCT_T(0);
	constexpr int IPT = MAXN / 256 < 4 ? 4 : MAXN / 256;   // scan entries per thread: every scan below is over at most MAXN entries
	const PyrDesc& d = *b.desc;
	// level-major over the launch's level range: the big levels (most candidates, most passes) start first and the small ones fill the tail
	const int level = level0 + blockIdx.x / nimg;
	const int img = blockIdx.x - (level - level0) * nimg;
	const LevelInfo& Lv = d.lv[level];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int N = Lv.nfeat;
	// the two exact-pass list counters of the descriptor stage start every batch at zero: cleared here (the previous batch's list kernels are done — stream order —,
	// k_orient_b, their first writer, comes after this kernel) instead of by a memset launch of their own on the critical path
	if (blockIdx.x == 0 && tid == 0 && b.fbCount) { b.fbCount[0] = 0; b.fbCount[1] = 0; b.fbCount[2] = 0; }   // fallback list, pre-list, tie list (mcs_describe.hip)

	uint32_t* denseG = b.dense + (size_t)img * d.densePerImage + Lv.denseBase;
	unsigned short* knodeG = b.knode + (size_t)img * d.densePerImage + Lv.denseBase;
	uint32_t* sel = b.sel + (size_t)img * d.selPerImage + Lv.selBase;
	int* selCount = b.selCount + (size_t)img * d.nlevels + level;
	int* denseCount = b.denseCount + (size_t)img * d.nlevels + level;

	// ---------------------------------------------------------------- compaction of the cell slots (ordered)
	const int ncell = Lv.nCols * Lv.nRows;
	const int* cellCount = b.cellCount + (size_t)img * d.cellsPerImage + Lv.cellBase;
	const uint32_t* slots = b.slots + (size_t)img * d.slotsPerImage + Lv.slotBase;
	int n = 0;
	for (int c0 = 0; c0 < ncell; c0 += MAXN) {
		const int m = min(MAXN, ncell - c0);
		__syncthreads();
		for (int i = tid; i < m; i += 256) scanA[i] = cellCount[c0 + i];
		__syncthreads();
		const int tot = block_exscan<IPT>(scanA, m, wsum);
#ifdef MCS_OCT_WAVE_CELLS   // A/B: round 3's form, a wave per cell — a hundred dependent trips per wave (count -> records -> store) with a handful of lanes each
		for (int c = wave; c < m; c += 4) {
			const int cnt = cellCount[c0 + c];
			const int off = n + scanA[c];
			const uint32_t* sp = slots + (size_t)(c0 + c) * Lv.capc;
			for (int j = lane; j < cnt; j += 64) denseG[off + j] = sp[j];
		}
#else
		// a THREAD per cell (a cell holds a handful of records): every thread's loads are independent of everybody else's, four records in flight per thread
		for (int c = tid; c < m; c += 256) {
			const int cnt = cellCount[c0 + c];
			const int off = n + scanA[c];
			const uint32_t* sp = slots + (size_t)(c0 + c) * Lv.capc;
			for (int j = 0; j < cnt; j += 4) {
				uint32_t v[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) v[u] = j + u < cnt ? sp[j + u] : 0u;
#pragma unroll
				for (int u = 0; u < 4; ++u)
					if (j + u < cnt) {
						denseG[off + j + u] = v[u];
						if (off + j + u < KCACHE) kd[off + j + u] = v[u];   // the LDS copy at once (used if the whole level fits): no trip to memory and back
					}
			}
		}
#endif
		n += tot;
	}
	__syncthreads();
	if (tid == 0) *denseCount = n;
	if (n == 0 || Lv.nIni < 1) { if (tid == 0) *selCount = 0; return; }   // (no root: a level more than twice as tall as wide, see mcs_capi.hip)
	OCT_T(1);
	// keys of small levels live in LDS for the passes below (each pass walks all keys twice; from HBM that is the kernel's latency)
	const bool inLds = n <= KCACHE;
#ifdef MCS_OCT_WAVE_CELLS
	if (inLds) for (int k = tid; k < n; k += 256) kd[k] = denseG[k];
#endif
	const uint32_t* dense = inLds ? kd : denseG;
	unsigned short* knode = inLds ? kn : knodeG;
	__syncthreads();

	// ---------------------------------------------------------------- roots (:641-683)
	const int nIni = Lv.nIni;
	const double hX = Lv.hX;
	const int H = Lv.h - 2 * kMinBorder;   // maxY - minY
	for (int i = tid; i < kMaxRoots; i += 256) scanA[i] = 0;
	__syncthreads();
	for (int k = tid; k < n; k += 256) {
		const int x = dense[k] & 0xFFF;
		int r = (int)((double)x / hX);    // vpIniNodes[kp.pt.x / hX]
		r = r < nIni - 1 ? r : nIni - 1;
		knode[k] = (unsigned short)r;
		atomicAdd(&scanA[r], 1);
	}
	__syncthreads();
	if (tid == 0) {
		int L0 = 0;
		for (int i = 0; i < nIni; ++i) {
			if (scanA[i] > 0) {
				nb[0].x0[L0] = (short)Lv.rootX[i]; nb[0].x1[L0] = (short)Lv.rootX[i + 1];
				nb[0].y0[L0] = 0; nb[0].y1[L0] = (short)H;
				nb[0].cnt[L0] = scanA[i]; nb[0].cre[L0] = -1;
				scanB[i] = L0++;
			}
		}
		shR = L0;
	}
	__syncthreads();
	int L = shR;
	for (int k = tid; k < n; k += 256) knode[k] = (unsigned short)scanB[knode[k]];
	__syncthreads();

	OCT_T(2);
	// ---------------------------------------------------------------- passes
	int cur = 0;
	bool phaseB = false;
#ifdef MCS_OCT_BLOCK_PASSES   // A/B: round 3's passes — every node-level step by all 256 threads with a workgroup barrier (or two) behind it: ~25 per pass
	for (int pass = 0; pass < 64; ++pass) {
		NodeBuf<MAXN>& A = nb[cur];
		NodeBuf<MAXN>& Bn = nb[cur ^ 1];
		const int prevL = L;
		// (1) candidate ranks
		int M;
		if (!phaseB) {
			for (int i = tid; i < L; i += 256) scanA[i] = A.cnt[i] > 1 ? 1 : 0;
			__syncthreads();
			M = block_exscan<IPT>(scanA, L, wsum);
			for (int i = tid; i < L; i += 256) crank[i] = A.cnt[i] > 1 ? (short)scanA[i] : (short)-1;
		} else {
			for (int i = tid; i < L; i += 256) scanA[i] = A.cre[i] >= 0 ? 1 : 0;
			__syncthreads();
			M = block_exscan<IPT>(scanA, L, wsum);
			for (int i = tid; i < L; i += 256) {   // rank = number of candidates with a larger (cnt, cre) key
				short r = -1;
				if (A.cre[i] >= 0) {
					const unsigned long long key = ((unsigned long long)A.cnt[i] << 16) | (unsigned)(unsigned short)A.cre[i];
					int g = 0;
					for (int j = 0; j < L; ++j) {
						const int cj = A.cre[j];
						const unsigned long long kj = ((unsigned long long)A.cnt[j] << 16) | (unsigned)(unsigned short)cj;
						g += (cj >= 0 && kj > key) ? 1 : 0;
					}
					r = (short)g;
				}
				crank[i] = r;
			}
		}
		for (int i = tid; i < L * 4; i += 256) cc[i] = 0;
		__syncthreads();
		if (M == 0) break;   // nothing can be split: lNodes.size() == prevSize (:767,832)
		// (2) child key counts of every candidate
		for (int k = tid; k < n; k += 256) {
			const int i = knode[k];
			if (crank[i] >= 0) {
				const uint32_t rec = dense[k];
				const int x = rec & 0xFFF, y = (rec >> 12) & 0xFFF;
				const int mx = A.x0[i] + ((A.x1[i] - A.x0[i] + 1) >> 1);
				const int my = A.y0[i] + ((A.y1[i] - A.y0[i] + 1) >> 1);
				const int q = (x < mx ? 0 : 1) + (y < my ? 0 : 2);
				atomicAdd(&cc[i * 4 + q], 1);
			}
		}
		for (int i = tid; i < L; i += 256)
			if (crank[i] >= 0) byRank[crank[i]] = (short)i;
		__syncthreads();
		// (3) how many candidates are processed: all (phase A) or up to the node that lifts the list to >= N (phase B, :828)
		int P = M;
		if (phaseB) {
			for (int r = tid; r < M; r += 256) {
				const int i = byRank[r];
				scanB[r] = (cc[i * 4] > 0) + (cc[i * 4 + 1] > 0) + (cc[i * 4 + 2] > 0) + (cc[i * 4 + 3] > 0) - 1;
			}
			if (tid == 0) shR = M - 1;
			__syncthreads();
			block_exscan<IPT>(scanB, M, wsum);
			for (int r = tid; r < M; r += 256) {
				const int i = byRank[r];
				const int nch = (cc[i * 4] > 0) + (cc[i * 4 + 1] > 0) + (cc[i * 4 + 2] > 0) + (cc[i * 4 + 3] > 0);
				const int sizeAfter = L + scanB[r] + nch - 1;
				if (sizeAfter >= N) atomicMin(&shR, r);
			}
			__syncthreads();
			P = shR + 1;
			__syncthreads();
		}
		// (4) prefix sums in processing order: children (list placement) and big children (creation index)
		for (int r = tid; r < P; r += 256) {
			const int i = byRank[r];
			scanA[r] = (cc[i * 4] > 0) + (cc[i * 4 + 1] > 0) + (cc[i * 4 + 2] > 0) + (cc[i * 4 + 3] > 0);
			scanB[r] = (cc[i * 4] > 1) + (cc[i * 4 + 1] > 1) + (cc[i * 4 + 2] > 1) + (cc[i * 4 + 3] > 1);
		}
		__syncthreads();
		const int T = block_exscan<IPT>(scanA, P, wsum);
		const int nToExpand = block_exscan<IPT>(scanB, P, wsum);
		const int newL = T + L - P;
		if (newL > MAXN) { if (tid == 0) atomicExch(b.status, MCS_ERR_CAPACITY); L = 0; break; }
		// children of processed nodes
		for (int r = tid; r < P; r += 256) {
			const int i = byRank[r];
			const int c0 = cc[i * 4], c1 = cc[i * 4 + 1], c2 = cc[i * 4 + 2], c3 = cc[i * 4 + 3];
			const int nch = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
			int pos = T - scanA[r] - nch;        // children of later-processed nodes sit in front
			int cre = scanB[r];
			const short x0 = A.x0[i], x1 = A.x1[i], y0 = A.y0[i], y1 = A.y1[i];
			const short mx = (short)(x0 + ((x1 - x0 + 1) >> 1)), my = (short)(y0 + ((y1 - y0 + 1) >> 1));
			short p0 = -1, p1 = -1, p2 = -1, p3 = -1;
			if (c3 > 0) p3 = (short)pos++;       // list order inside the node: n4, n3, n2, n1
			if (c2 > 0) p2 = (short)pos++;
			if (c1 > 0) p1 = (short)pos++;
			if (c0 > 0) p0 = (short)pos++;
			if (c0 > 0) { Bn.x0[p0] = x0; Bn.x1[p0] = mx; Bn.y0[p0] = y0; Bn.y1[p0] = my; Bn.cnt[p0] = c0; Bn.cre[p0] = c0 > 1 ? (short)cre++ : (short)-1; }
			if (c1 > 0) { Bn.x0[p1] = mx; Bn.x1[p1] = x1; Bn.y0[p1] = y0; Bn.y1[p1] = my; Bn.cnt[p1] = c1; Bn.cre[p1] = c1 > 1 ? (short)cre++ : (short)-1; }
			if (c2 > 0) { Bn.x0[p2] = x0; Bn.x1[p2] = mx; Bn.y0[p2] = my; Bn.y1[p2] = y1; Bn.cnt[p2] = c2; Bn.cre[p2] = c2 > 1 ? (short)cre++ : (short)-1; }
			if (c3 > 0) { Bn.x0[p3] = mx; Bn.x1[p3] = x1; Bn.y0[p3] = my; Bn.y1[p3] = y1; Bn.cnt[p3] = c3; Bn.cre[p3] = c3 > 1 ? (short)cre++ : (short)-1; }
			mapq[i * 4] = p0; mapq[i * 4 + 1] = p1; mapq[i * 4 + 2] = p2; mapq[i * 4 + 3] = p3;
		}
		__syncthreads();
		// untouched nodes keep their relative order behind the new children
		for (int i = tid; i < L; i += 256) scanA[i] = (crank[i] >= 0 && crank[i] < P) ? 0 : 1;
		__syncthreads();
		block_exscan<IPT>(scanA, L, wsum);
		for (int i = tid; i < L; i += 256) {
			if (!(crank[i] >= 0 && crank[i] < P)) {
				const int p = T + scanA[i];
				Bn.x0[p] = A.x0[i]; Bn.x1[p] = A.x1[i]; Bn.y0[p] = A.y0[i]; Bn.y1[p] = A.y1[i];
				Bn.cnt[p] = A.cnt[i]; Bn.cre[p] = -1;
				mapq[i * 4] = (short)p;
				crank[i] = -1;
			}
		}
		__syncthreads();
		// (6) keys follow their node
		for (int k = tid; k < n; k += 256) {
			const int i = knode[k];
			int q = 0;
			if (crank[i] >= 0) {
				const uint32_t rec = dense[k];
				const int x = rec & 0xFFF, y = (rec >> 12) & 0xFFF;
				const int mx = A.x0[i] + ((A.x1[i] - A.x0[i] + 1) >> 1);
				const int my = A.y0[i] + ((A.y1[i] - A.y0[i] + 1) >> 1);
				q = (x < mx ? 0 : 1) + (y < my ? 0 : 2);
			}
			knode[k] = (unsigned short)mapq[i * 4 + q];
		}
		__syncthreads();
		cur ^= 1;
		L = newL;
		// (7) termination (:767-771, :832)
		if (L >= N || L == prevL) break;
		if (!phaseB && L + 3 * nToExpand > N) phaseB = true;
	}
#else
	// Round 4: a pass crosses FOUR workgroup barriers instead of ~25.  The levels of a 754 x 480 image hold 300-900 candidates and at most 256 nodes: the two loops
	// over the KEYS use all four waves, but every step over the NODES (ranks, processing order, where the list reaches N, the two prefix sums, the children, the
	// untouched nodes) is a few entries per lane of ONE wave — wave 0 runs them back to back with DPP prefix sums, its LDS accesses stay in program order, and
	// nothing waits for a barrier in between (the kernel was a chain of barrier and LDS round-trip latencies: 100 us for ~40 us of instructions).
	__shared__ int shM, shNewL, shNExp;
	constexpr int IPTW = MAXN / 64;   // node entries per lane of wave 0
	for (int pass = 0; pass < 64; ++pass) {
		NodeBuf<MAXN>& A = nb[cur];
		NodeBuf<MAXN>& Bn = nb[cur ^ 1];
		const int prevL = L;
		// (1) candidates and their ranks
		if (wave == 0) {
			for (int i = lane; i < L; i += 64) scanA[i] = (phaseB ? A.cre[i] >= 0 : A.cnt[i] > 1) ? 1 : 0;
			WFENCE();
			const int M_ = wave_exscan<IPTW>(scanA, L);
			if (!phaseB) for (int i = lane; i < L; i += 64) crank[i] = A.cnt[i] > 1 ? (short)scanA[i] : (short)-1;
			if (lane == 0) shM = M_;
		}
		if (phaseB) {
			for (int i = tid; i < L; i += 256) {   // rank = number of candidates with a larger (cnt, cre) key: all threads (reads the previous pass's nodes only)
				short r = -1;
				if (A.cre[i] >= 0) {
					const unsigned long long key = ((unsigned long long)A.cnt[i] << 16) | (unsigned)(unsigned short)A.cre[i];
					int g = 0;
					for (int j = 0; j < L; ++j) {
						const int cj = A.cre[j];
						const unsigned long long kj = ((unsigned long long)A.cnt[j] << 16) | (unsigned)(unsigned short)cj;
						g += (cj >= 0 && kj > key) ? 1 : 0;
					}
					r = (short)g;
				}
				crank[i] = r;
			}
		}
		for (int i = tid; i < L * 4; i += 256) cc[i] = 0;
		__syncthreads();   // barrier 1
		const int M = shM;
		if (M == 0) break;   // nothing can be split: lNodes.size() == prevSize (:767,832)
		// (2) child key counts of every candidate
		for (int k = tid; k < n; k += 256) {
			const int i = knode[k];
			if (crank[i] >= 0) {
				const uint32_t rec = dense[k];
				const int x = rec & 0xFFF, y = (rec >> 12) & 0xFFF;
				const int mx = A.x0[i] + ((A.x1[i] - A.x0[i] + 1) >> 1);
				const int my = A.y0[i] + ((A.y1[i] - A.y0[i] + 1) >> 1);
				const int q = (x < mx ? 0 : 1) + (y < my ? 0 : 2);
				atomicAdd(&cc[i * 4 + q], 1);
			}
		}
		__syncthreads();   // barrier 2
		if (wave == 0) {
			auto nchild = [&](int i) { return (cc[i * 4] > 0) + (cc[i * 4 + 1] > 0) + (cc[i * 4 + 2] > 0) + (cc[i * 4 + 3] > 0); };
			for (int i = lane; i < L; i += 64)
				if (crank[i] >= 0) byRank[crank[i]] = (short)i;
			WFENCE();
			// (3) how many candidates are processed: all (phase A) or up to the node that lifts the list to >= N (phase B, :828)
			int P = M;
			if (phaseB) {
				for (int r = lane; r < M; r += 64) scanB[r] = nchild(byRank[r]) - 1;
				WFENCE();
				wave_exscan<IPTW>(scanB, M);
				int first = M - 1;
				for (int r = lane; r < M; r += 64) {
					const int sizeAfter = L + scanB[r] + nchild(byRank[r]) - 1;
					if (sizeAfter >= N) first = min(first, r);
				}
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
				P = first + 1;
				WFENCE();
			}
			// (4) prefix sums in processing order: children (list placement) and big children (creation index)
			for (int r = lane; r < P; r += 64) {
				const int i = byRank[r];
				scanA[r] = nchild(i);
				scanB[r] = (cc[i * 4] > 1) + (cc[i * 4 + 1] > 1) + (cc[i * 4 + 2] > 1) + (cc[i * 4 + 3] > 1);
			}
			WFENCE();
			const int T = wave_exscan<IPTW>(scanA, P);
			const int nExp = wave_exscan<IPTW>(scanB, P);
			const int newL_ = T + L - P;
			if (newL_ > MAXN) { if (lane == 0) shNewL = -1; }
			else {
				// children of processed nodes
				for (int r = lane; r < P; r += 64) {
					const int i = byRank[r];
					const int c0 = cc[i * 4], c1 = cc[i * 4 + 1], c2 = cc[i * 4 + 2], c3 = cc[i * 4 + 3];
					const int nch = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
					int pos = T - scanA[r] - nch;        // children of later-processed nodes sit in front
					int cre = scanB[r];
					const short x0 = A.x0[i], x1 = A.x1[i], y0 = A.y0[i], y1 = A.y1[i];
					const short mx = (short)(x0 + ((x1 - x0 + 1) >> 1)), my = (short)(y0 + ((y1 - y0 + 1) >> 1));
					short p0 = -1, p1 = -1, p2 = -1, p3 = -1;
					if (c3 > 0) p3 = (short)pos++;       // list order inside the node: n4, n3, n2, n1
					if (c2 > 0) p2 = (short)pos++;
					if (c1 > 0) p1 = (short)pos++;
					if (c0 > 0) p0 = (short)pos++;
					if (c0 > 0) { Bn.x0[p0] = x0; Bn.x1[p0] = mx; Bn.y0[p0] = y0; Bn.y1[p0] = my; Bn.cnt[p0] = c0; Bn.cre[p0] = c0 > 1 ? (short)cre++ : (short)-1; }
					if (c1 > 0) { Bn.x0[p1] = mx; Bn.x1[p1] = x1; Bn.y0[p1] = y0; Bn.y1[p1] = my; Bn.cnt[p1] = c1; Bn.cre[p1] = c1 > 1 ? (short)cre++ : (short)-1; }
					if (c2 > 0) { Bn.x0[p2] = x0; Bn.x1[p2] = mx; Bn.y0[p2] = my; Bn.y1[p2] = y1; Bn.cnt[p2] = c2; Bn.cre[p2] = c2 > 1 ? (short)cre++ : (short)-1; }
					if (c3 > 0) { Bn.x0[p3] = mx; Bn.x1[p3] = x1; Bn.y0[p3] = my; Bn.y1[p3] = y1; Bn.cnt[p3] = c3; Bn.cre[p3] = c3 > 1 ? (short)cre++ : (short)-1; }
					mapq[i * 4] = p0; mapq[i * 4 + 1] = p1; mapq[i * 4 + 2] = p2; mapq[i * 4 + 3] = p3;
				}
				WFENCE();
				// untouched nodes keep their relative order behind the new children
				for (int i = lane; i < L; i += 64) scanA[i] = (crank[i] >= 0 && crank[i] < P) ? 0 : 1;
				WFENCE();
				wave_exscan<IPTW>(scanA, L);
				for (int i = lane; i < L; i += 64) {
					if (!(crank[i] >= 0 && crank[i] < P)) {
						const int p = T + scanA[i];
						Bn.x0[p] = A.x0[i]; Bn.x1[p] = A.x1[i]; Bn.y0[p] = A.y0[i]; Bn.y1[p] = A.y1[i];
						Bn.cnt[p] = A.cnt[i]; Bn.cre[p] = -1;
						mapq[i * 4] = (short)p;
						crank[i] = -1;
					}
				}
				if (lane == 0) { shNewL = newL_; shNExp = nExp; }
			}
		}
		__syncthreads();   // barrier 3
		const int newL = shNewL, nToExpand = shNExp;
		if (newL < 0) { if (tid == 0) atomicExch(b.status, MCS_ERR_CAPACITY); L = 0; break; }
		// (6) keys follow their node
		for (int k = tid; k < n; k += 256) {
			const int i = knode[k];
			int q = 0;
			if (crank[i] >= 0) {
				const uint32_t rec = dense[k];
				const int x = rec & 0xFFF, y = (rec >> 12) & 0xFFF;
				const int mx = A.x0[i] + ((A.x1[i] - A.x0[i] + 1) >> 1);
				const int my = A.y0[i] + ((A.y1[i] - A.y0[i] + 1) >> 1);
				q = (x < mx ? 0 : 1) + (y < my ? 0 : 2);
			}
			knode[k] = (unsigned short)mapq[i * 4 + q];
		}
		__syncthreads();   // barrier 4
#ifdef MCS_OCT_TRACE
		OCT_T(3 + octPass); ++octPass;
#endif
		cur ^= 1;
		L = newL;
		// (7) termination (:767-771, :832)
		if (L >= N || L == prevL) break;
		if (!phaseB && L + 3 * nToExpand > N) phaseB = true;
	}
#endif
	__syncthreads();

	OCT_T(14);
	// ---------------------------------------------------------------- best key per node (:840-858)
	unsigned* best = reinterpret_cast<unsigned*>(scanA);
	for (int i = tid; i < L; i += 256) best[i] = 0;
	__syncthreads();
	for (int k = tid; k < n; k += 256) {
		const unsigned v = (dense[k] & 0xFF000000u) | (0xFFFFFFu - (unsigned)k);   // max response, lowest index wins ties
		atomicMax(&best[knode[k]], v);
	}
	__syncthreads();
	if (L > Lv.selCap) { if (tid == 0) { atomicExch(b.status, MCS_ERR_CAPACITY); *selCount = 0; } return; }
	uint32_t* selL = reinterpret_cast<uint32_t*>(cc);   // the selected records stay in LDS for the orientation tail (cc is dead): one memory round trip less per trip of the tail
	for (int i = tid; i < L; i += 256) { const uint32_t r = dense[0xFFFFFFu - (best[i] & 0xFFFFFFu)]; sel[i] = r; selL[i] = r; }
	if (tid == 0) *selCount = L;
	// ---------------------------------------------------------------- E5: orientation of the selected keys (IC_Angle, :221-248)
	// Here rather than in a kernel of its own: a separate launch ran 0.115 ms with nothing beside it, three dependent memory round trips per key; in this
	// tail they hide behind the other (image, level) workgroups' passes.
	OCT_T(15);
	__shared__ __attribute__((aligned(16))) uint32_t otab[kOrientTabWords];
	orient_table(d.umax, otab);
	__syncthreads();   // the selection and the table are complete
	int rstride;
	const uint8_t* raw = level_ptr(b, d, img, level, &rstride);
#ifndef MCS_OCT_AB   // A/B (timing only): the kernel without its orientation tail (101 against 157 us before the tail's byte dot products)
	orient_selected(raw, rstride, otab, selL, L, b.selAngle + (size_t)img * d.selPerImage + Lv.selBase);
#endif
#ifdef MCS_OCT_TRACE
	__syncthreads();
	OCT_T(16);
	if (blockIdx.x == 0 && tid == 0 && atomicAdd(&g_octTraceLaunches, 1) < 3) {
		printf("octT n=%d L=%d passes=%d | compact %llu roots %llu |", n, L, octPass, octT[1] - octT[0], octT[2] - octT[1]);
		for (int q = 0; q < octPass && q < 10; ++q) printf(" p%d %llu", q, octT[3 + q] - (q ? octT[2 + q] : octT[2]));
		printf(" | best %llu orient %llu total %llu (x10 ns)\n", octT[15] - octT[14], octT[16] - octT[15], octT[16] - octT[0]);
	}
#endif
}

void launch_octree(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s, int level0, int level1) {   // levels [level0, level1)
	if (level1 > hd.nlevels) level1 = hd.nlevels;
	if (level0 >= level1) return;
	const int nl = level1 - level0;
	int need = 0;
	for (int l = 0; l < hd.nlevels; ++l) need = need > hd.lv[l].nfeat + 3 ? need : hd.lv[l].nfeat + 3, need = need > 4 * hd.lv[l].nIni ? need : 4 * hd.lv[l].nIni;
	if (need <= 256) hipLaunchKernelGGL((k_octree<256, 2048>), dim3(nimg * nl), dim3(256), 0, s, b, nimg, level0);   // N = 1000 over 8 levels needs 220 nodes: 29 KB LDS, five workgroups per CU (0.195 -> 0.162 ms; key cache 3072: four per CU, 0.175; 1536: 0.176)
	else if (need <= 512) hipLaunchKernelGGL((k_octree<512, 3072>), dim3(nimg * nl), dim3(256), 0, s, b, nimg, level0);   // 50 KB LDS: 3 workgroups per CU
	else if (need <= 1024) hipLaunchKernelGGL((k_octree<1024, 2048>), dim3(nimg * nl), dim3(256), 0, s, b, nimg, level0);
	else hipLaunchKernelGGL((k_octree<2048, 2048>), dim3(nimg * nl), dim3(256), 0, s, b, nimg, level0);   // nFeatures up to ~9400 (the reference has no limit, :167-179): 140 KB of LDS, one workgroup per CU
}

}  // namespace mcs

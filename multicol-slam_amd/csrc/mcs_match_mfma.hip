// mcs_match_mfma.hip — the brute-force top-K matcher of mcs_match.hip with the pair distances on the matrix cores.
// Reference: DescriptorDistance64 / DescriptorDistance64Masked src/cORBmatcher.cpp:2438-2474 (see mcs_match.hip for the call sites).
//
// A (masked) Hamming total is a bilinear form in the bits.  With x = q ^ t:
//     popc(x & mq) = popc(mq & q) + sum_i t_i * mq_i (1 - 2 q_i)          popc(x & mt) = popc(mt & t) + sum_i q_i * mt_i (1 - 2 t_i)
// so   total(q, t) = cq + ct + < [uq ; q] , [t ; vt] >,   uq = mq (1 - 2q), vt = mt (1 - 2t) in {-1, 0, +1},  cq = popc(mq & q), ct = popc(mt & t)
// (unmasked: popc(q ^ t) = popc(q) + < 1 - 2q, t >).  Entries 0 / +-1 are exact in FP4 (E2M1: +1 = 0x2, -1 = 0xA), products and sums of at most 1024 of
// them are exact in the f32 accumulators: v_mfma_f32_32x32x64_f8f6f4 gives 32 x 32 exact totals per instruction and 64 bits of K.  A 32-byte masked pair
// is K = 512: 8 instructions (256 cycles) per 1024 pairs, where the v_xor / v_and / v_bcnt chain of mcs_match.hip issues for ~1700.
// (tools/mfma_fp4_probe.hip pins the operand and result layout used here on the device.)
//
// Workgroup = 4 waves, 64 queries per wave as two 32-column operand sets (B operand lane = column + 32 * k-half; 4 VGPRs per K step and set, in
// registers for the whole kernel).  The train sets are prepared ONCE per call by k_expand_train (eligible rows only, compacted in order, expanded bit ->
// nibble through a 256-entry LDS table into the A-operand layout [stage of 64 rows][tile of 32][K step][lane]); the list kernel copies a stage at a time into
// LDS with global_load_lds, where every wave's operand read is one conflict-free ds_read_b128, used for both query sets (round 2 staged, compacted and expanded
// the rows in every workgroup: two thirds of its time).  An MFMA result puts 16 of a column's 32 rows in lane c and the other 16 in lane c + 32; one
// v_permlane32_swap per register between the two sets' results leaves lane c with all 32 rows of query c of the first set and lane c + 32 with all 32
// rows of query c of the second: ONE query per lane, so its K-best list sees every train row (two half lists per query appended 1.7x as many
// candidates, and merged as often).
// The candidate word is a FLOAT and the matrix core forms it: the accumulators start from a per-row word (ct + 256) + index / 2^14 (exact in f32: 10 +
// 14 bits), the +-1 products add the dot product, so an MFMA result IS the key "total - cq + 256, then index" — positive floats order like their bit
// patterns, so the append is one unsigned compare against the limit's bits, a store and an add under the compare's lane mask; no conversion, no shift-
// or per pair.  (Any order of the additions inside the instruction is exact: every partial sum is a multiple of 2^-14 in [0, 769).)  The exact integer
// key (total << 20 | index, as in mcs_match.hip) is formed when a column is merged.  Train sets of more than 2^14 rows go to mcs_match.hip.
#include "mcs_common.h"

#ifndef MCS_MM_AB
#define MCS_MM_AB 0
#endif
#ifndef MCS_MM_WAVES
#define MCS_MM_WAVES 3
#endif
#ifndef MCS_MM_WGW
#define MCS_MM_WGW 4
#endif
#ifndef MCS_MM_TPS
#define MCS_MM_TPS 2
#endif
#ifndef MCS_MM_NBUF
#define MCS_MM_NBUF 2
#endif
namespace mcs {

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

constexpr int WGW = MCS_MM_WGW;   // waves per workgroup: they share the staged train operands and meet at one barrier per stage
constexpr int TPS = MCS_MM_TPS;   // 32-row tiles per stage
constexpr int XQ = 64 * WGW;      // queries per workgroup
constexpr int kColStride = XQ * 4;   // bytes between the slots of a lane's candidate column
// Words in the candidate columns are BIASED: dot + ct + 256 (+ index / 2^14), i.e. the total minus the query's own cq (<= 256) plus 256 — never negative —
// so that cq costs nothing per pair: it is subtracted from the limit once per group and added back when a column is merged.  The f32 dot product of any
// staged row (real rows; the rest of a set's last stage holds zero operands) lies in [-512, 512], so a padding row's word (kPadWord) ends in [2560, 3584]: never below a
// biased limit (<= kLimCap + 256 = 2560); real totals (<= 512) stay below kLimCap.
constexpr float kPadWord = 3072.f, kIdxUnit = 1.f / 16384.f;
constexpr uint32_t kLimCap = 0x900u, kBias = 256u;
constexpr int kMaxTrainRows = 16384;

// bit k of the byte -> nibble k = 1
__device__ __forceinline__ uint32_t spread8(uint32_t b) {
	uint32_t v = b;
	v = (v | (v << 12)) & 0x000F000Fu;
	v = (v | (v << 6)) & 0x03030303u;
	v = (v | (v << 3)) & 0x11111111u;
	return v;
}

// 32 bits -> 32 FP4 values (4 dwords).  plain: bit -> +1.0;  signed: m & ~x -> +1.0, m & x -> -1.0, ~m -> 0
__device__ __forceinline__ uint4 expand01(const uint32_t* lut, uint32_t x) {
	uint4 o;
	o.x = lut[x & 0xff] << 1; o.y = lut[(x >> 8) & 0xff] << 1; o.z = lut[(x >> 16) & 0xff] << 1; o.w = lut[x >> 24] << 1;
	return o;
}
__device__ __forceinline__ uint4 expandpm(const uint32_t* lut, uint32_t m, uint32_t x) {
	const uint32_t n = m & x;
	uint4 o;
	o.x = (lut[m & 0xff] << 1) | (lut[n & 0xff] << 3);
	o.y = (lut[(m >> 8) & 0xff] << 1) | (lut[(n >> 8) & 0xff] << 3);
	o.z = (lut[(m >> 16) & 0xff] << 1) | (lut[(n >> 16) & 0xff] << 3);
	o.w = (lut[m >> 24] << 1) | (lut[n >> 24] << 3);
	return o;
}

// *(uint32_t*)next = w (next: LDS byte address) and next += kColStride in the lanes with w < lim (unsigned compare of float bits): the compare writes the lane
// mask, the store and the add run under it — one compare + one add on the VALU per pair (compare / select / shift-add / add as plain C++).  All 64
// lanes are active at every call site (wave-uniform control flow, full workgroups).
__device__ __forceinline__ void append(uint32_t& next, uint32_t w, uint32_t lim) {
	asm volatile(
		"v_cmpx_lt_u32_e32 vcc, %1, %2\n\t"
		"ds_write_b32 %0, %1\n\t"
		"v_add_u32_e32 %0, %3, %0\n\t"
		"s_mov_b64 exec, -1"
		: "+v"(next) : "v"(w), "v"(lim), "i"(kColStride) : "vcc", "memory");
}

// ---- train side, once per call: eligible rows compacted in order and expanded bit -> FP4 nibble into the A-operand layout ----------------------------
// exA[set][stage of 64 rows][tile of 32][K step][lane] (16 bytes: lane = row + 32 * k-half), exW[set][stage * 64 + row] = the row's word
// (ct + 256) + original index / 2^14, exRows[set] = eligible rows.  Every workgroup of the matcher that meets the set (a database sweep: hundreds) then
// copies finished operands instead of staging, compacting and expanding the rows itself — that was two thirds of the matcher's time.
template <int DW, bool MASKED>
__global__ __launch_bounds__(256) void k_expand_train(MatchArgs a) {
	constexpr int HS = DW / 2, NS = (MASKED ? 2 : 1) * HS;
	__shared__ uint32_t lut[256];
	__shared__ int wcnt[4], wbefore[4];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ts = blockIdx.x, base = blockIdx.y * 256;
	const RowMap TR{(size_t)ts * a.tpitch, a.tblk, a.tbpitch};
	lut[tid] = spread8((uint32_t)tid);
	uint4* dstA = a.exA + (size_t)ts * a.exStages * (2 * NS * 64);
	float* dstW = a.exW + (size_t)ts * a.exStages * 64;
	// a workgroup per 256 rows; where its rows go = the eligible rows before them, counted from the flags (at most nt bytes)
	int mine = 0;
	if (a.tvalid) for (int j = tid; j < base; j += 256) mine += __popcll(__ballot(a.tvalid[TR(j)] != 0));
	const int j = base + tid;
	bool ok = false;
	uint32_t tw[DW], mw[DW];
	int ct = 0;
	if (j < a.nt) {
		const size_t row = TR(j);
		const uint32_t* tp = reinterpret_cast<const uint32_t*>(a.td + row * a.tstride);
#pragma unroll
		for (int w = 0; w < DW; ++w) tw[w] = tp[w];
		if (MASKED) {
			const uint32_t* mp = reinterpret_cast<const uint32_t*>(a.tm + row * a.tstride);
#pragma unroll
			for (int w = 0; w < DW; ++w) { mw[w] = mp[w]; ct += __popc(mw[w] & tw[w]); }
		}
		ok = a.tvalid ? a.tvalid[row] != 0 : true;
	}
	const unsigned long long bal = __ballot(ok);
	if (lane == 0) { wcnt[wv] = __popcll(bal); wbefore[wv] = mine; }
	__syncthreads();
	const int before = a.tvalid ? wbefore[0] + wbefore[1] + wbefore[2] + wbefore[3] : base;
	int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
	for (int w = 0; w < wv; ++w) pos += wcnt[w];
	if (ok) {
		uint4* d = dstA + (size_t)(pos >> 5) * (NS * 64) + (pos & 31);
#pragma unroll
		for (int w = 0; w < DW; ++w) {   // step (segment * HS + w / 2), k-half w & 1
			d[((w >> 1) * 64) + 32 * (w & 1)] = expand01(lut, tw[w]);
			if (MASKED) d[((HS + (w >> 1)) * 64) + 32 * (w & 1)] = expandpm(lut, mw[w], tw[w]);
		}
		dstW[pos] = (float)(ct + (int)kBias) + (float)j * kIdxUnit;
	}
	if (base + 256 >= a.nt) {
		// the set's last workgroup: the rest of the last stage gets zero operands and a word no limit can reach
		const int rows = before + wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3], p2 = rows + tid;
		if (tid < 64 && p2 < ((rows + 63) & ~63)) {
			uint4* d = dstA + (size_t)(p2 >> 5) * (NS * 64) + (p2 & 31);
#pragma unroll
			for (int st = 0; st < NS; ++st) { d[st * 64] = uint4{0, 0, 0, 0}; d[st * 64 + 32] = uint4{0, 0, 0, 0}; }
			dstW[p2] = kPadWord;
		}
		if (tid == 0) a.exRows[ts] = rows;
	}
}

// ---- the lists ---------------------------------------------------------------------------------------------------------------------------------------
// Three waves per SIMD (168 registers; the LDS allows three workgroups per CU): left alone the compiler takes more registers for K = 32 and two waves.
template <int K, int DW, bool MASKED>
__attribute__((amdgpu_waves_per_eu(MCS_MM_WAVES, MCS_MM_WAVES)))
__global__ __launch_bounds__(XQ) void k_match_mfma(MatchArgs a) {
	constexpr int HS = DW / 2;                       // K steps per segment (64 bits each)
	constexpr int NS = (MASKED ? 2 : 1) * HS;        // K steps per pair
#ifndef MCS_MM_CB
#define MCS_MM_CB 16
#endif
	constexpr int CB = MCS_MM_CB;                    // candidate column depth per lane, a power of two (A/B, round 4: 32 with one-tile stages — the same LDS — 10.6 against 7.07 ms on configs[2]: 30 spilled registers, sort network of 32)
	constexpr int SLABS = TPS * NS;                  // 1-KB operand slabs (tile, K step) per stage
	// A operands of two stages: stage g + 1 arrives (global_load_lds: global -> LDS without passing registers) while stage g is multiplied
	__shared__ __attribute__((aligned(16))) uint4 ex[MCS_MM_NBUF][TPS][NS][64];
	__shared__ __attribute__((aligned(16))) float wrow[MCS_MM_NBUF][64];
	__shared__ uint32_t lut[256];
	__shared__ uint32_t cand[(CB + 1) * XQ];

	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, kh = lane >> 5;
	// Workgroups are dealt to the 8 XCDs round robin; each XCD has its own L2.  Give XCD x the x-th contiguous eighth of the (set, split, query tile) order, so
	// that the workgroups that read one train set run on one XCD (a set's expanded operands are fetched into one L2, not eight).
	int bx = blockIdx.x, split = blockIdx.y, set = blockIdx.z;
	{
		const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
		if ((total & 7u) == 0) {
			const unsigned lin = (blockIdx.z * ny + blockIdx.y) * nx + blockIdx.x, m = (lin & 7u) * (total >> 3) + (lin >> 3);
			bx = (int)(m % nx); split = (int)((m / nx) % ny); set = (int)(m / (nx * ny));
		}
	}
	const int qbase = bx * XQ + wv * 64;
	const int qi = qbase + lane;   // the query this lane OWNS after the half swap (set 0: lanes 0..31, set 1: lanes 32..63)
	const RowMap QR{(size_t)(set % a.qmod) * a.qpitch, a.qblk, a.qbpitch};
	const int ts = a.tsets > 1 ? (set / a.tdiv + a.toff) % a.tmod : 0;
	const int rows = a.exRows[ts];
	const int per = (a.exStages * (2 / TPS) + a.splits - 1) / a.splits;   // a.exStages counts 64-row units
	const int g0 = split * per, g1 = min((rows + 32 * TPS - 1) / (32 * TPS), g0 + per);
	const uint4* srcA = a.exA + (size_t)ts * a.exStages * (2 * NS * 64);
	const float* srcW = a.exW + (size_t)ts * a.exStages * 64;
	// (lane indices and addresses are re-derived from an opaque copy of the thread index per stage: held across the loop they were spilled, and a spill
	// reload waits for every load in flight — the next stage's operands included)
	auto request = [&](int g, int buf, int ln, int wave) {   // this wave's share of stage g's slabs (and the words, wave 0)
#pragma unroll
		for (int i = 0; i < (SLABS + WGW - 1) / WGW; ++i) {
			const int slab = wave + i * WGW;
			if (SLABS % WGW == 0 || slab < SLABS)
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA + ((size_t)g * SLABS + slab) * 64 + ln),
			                                 (__attribute__((address_space(3))) void*)(&ex[buf][0][0][0] + slab * 64), 16, 0, 0);
		}
		if (wave == 0)
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW + (size_t)g * (32 * TPS) + ln),
			                                 (__attribute__((address_space(3))) void*)(&wrow[buf][0]), 4, 0, 0);
	};
	if (MCS_MM_NBUF == 2 && g0 < g1) request(g0, 0, lane, wv);
	for (int i = tid; i < 256; i += XQ) lut[i] = spread8((uint32_t)i);
	__syncthreads();

	// the operands of query (set u, column col): step s = segment * HS + j covers dwords 2j (k-half 0) and 2j + 1 (k-half 1) of the segment's bit vector
	v8i_t bq[2][NS];
	uint32_t cq = 0;     // popc(mq & q) of the lane's OWN query (k-half = set)
	bool qok = false;
#pragma unroll
	for (int u = 0; u < 2; ++u) {
		const int qu = qbase + 32 * u + col;
		bool ok = qu < a.nq;
		if (ok && a.qvalid) ok = a.qvalid[QR(qu)] != 0;
		if (u == kh) qok = ok;
		uint32_t q[DW], qm[DW];
#pragma unroll
		for (int w = 0; w < DW; ++w) { q[w] = 0; qm[w] = 0; }
		if (ok) {
			const uint32_t* qp = reinterpret_cast<const uint32_t*>(a.qd + QR(qu) * a.qstride);
#pragma unroll
			for (int w = 0; w < DW; ++w) q[w] = qp[w];
			if (MASKED) {
				const uint32_t* mp = reinterpret_cast<const uint32_t*>(a.qm + QR(qu) * a.qstride);
#pragma unroll
				for (int w = 0; w < DW; ++w) qm[w] = mp[w];
			}
		}
		int c = 0;
#pragma unroll
		for (int w = 0; w < DW; ++w) c += __popc(MASKED ? (q[w] & qm[w]) : q[w]);
		if (u == kh) cq = (uint32_t)c;
#pragma unroll
		for (int j = 0; j < HS; ++j) {
			const uint32_t x = kh ? q[2 * j + 1] : q[2 * j], m = MASKED ? (kh ? qm[2 * j + 1] : qm[2 * j]) : 0xFFFFFFFFu;
			const uint4 e = expandpm(lut, m, x);
			bq[u][j] = v8i_t{(int)e.x, (int)e.y, (int)e.z, (int)e.w, 0, 0, 0, 0};
			if (MASKED) {
				const uint4 p = expand01(lut, x);
				bq[u][HS + j] = v8i_t{(int)p.x, (int)p.y, (int)p.z, (int)p.w, 0, 0, 0, 0};
			}
		}
	}

	uint32_t best[K];
#pragma unroll
	for (int p = 0; p < K; ++p) best[p] = 0xFFFFFFFFu;
	const uint32_t col0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(cand + tid);   // LDS byte address of the lane's column; slot e is kColStride bytes further
	uint32_t next = col0;
	auto exact_key = [&](uint32_t bits) {   // float word -> un-biased integer key as in mcs_match.hip
		const uint32_t k = (uint32_t)(__uint_as_float(bits) * 16384.f);   // exact: at most 14 fractional bits
		const uint32_t t = (k >> 14) - kBias + cq, idx = k & 16383u;
		return ((MASKED ? (t >> 1) : t) << 20) | idx;
	};
	auto bitonic_merge_best = [&]() {
#pragma unroll
		for (int j = K >> 1; j > 0; j >>= 1)
#pragma unroll
			for (int i = 0; i < K; ++i) {
				const int l = i ^ j;
				if (l > i) { const uint32_t lo = min(best[i], best[l]), hi = max(best[i], best[l]); best[i] = lo; best[l] = hi; }
			}
	};
	auto flush = [&]() {   // as in mcs_match.hip
#if MCS_MM_AB == 2
		next = col0; return;
#endif
		const int cnt = (int)((next - col0) / kColStride);
		if (K >= CB) {
			uint32_t c[CB];
#pragma unroll
			for (int e = 0; e < CB; ++e) c[e] = cand[e * XQ + tid];
#pragma unroll
			for (int e = 0; e < CB; ++e) {
				asm volatile("" : "+v"(c[e]));   // the reads stay unconditional and in flight together (a stale slot's bits convert to some key that the select drops)
				c[e] = e < cnt ? exact_key(c[e]) : 0xFFFFFFFFu;
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
			for (int i = 0; i < CB; ++i) best[K - 1 - i] = min(best[K - 1 - i], c[i]);
			bitonic_merge_best();
		} else {
			int m = cnt;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
			for (int e = 0; e < m; ++e) {
				uint32_t key = e < cnt ? exact_key(cand[e * XQ + tid]) : 0xFFFFFFFFu;
				if (__any(key < best[K - 1])) {
#pragma unroll
					for (int p = 0; p < K; ++p) { const uint32_t lo = min(best[p], key), hi = max(best[p], key); best[p] = lo; key = hi; }
				}
			}
		}
		next = col0;
	};

	const uint32_t dCap = a.maxDist >= 4095 ? 4095u : (uint32_t)a.maxDist;
	// The distance threshold, the padding rows and "closer than the K-th best" are ONE unsigned compare of the biased word's bits against rawLim (masked:
	// the raw total t stands for distance t >> 1, so "distance <= D" is t <= 2D + 1).  The K-th best only changes in a merge: the limit is recomputed there.
	auto limit = [&]() -> uint32_t {
		uint32_t lim, frac = 0;   // "field < lim, or field == lim and index < frac"
		const uint32_t kth = best[K - 1];
		if (MASKED) {
			const uint32_t dl = min(kth >> 20, dCap);
			lim = dl >= 1151u ? kLimCap : 2u * dl + 2u;
		} else {
			lim = min(dCap >= 4095u ? kLimCap : dCap + 1u, kLimCap);
			if ((kth >> 20) < lim) { lim = kth >> 20; frac = kth & 0xFFFFFu; }
		}
		// the biased words carry total - cq + 256: lim + 256 - cq >= 2 (masked) / >= 0
		return qok ? __float_as_uint((float)(lim + kBias - cq) + (float)frac * kIdxUnit) : 0u;
	};
	uint32_t rawLim = limit();
	for (int g = g0; g < g1; ++g) {
#if MCS_MM_NBUF == 2
		const int buf = (g - g0) & 1;
		// stage g has landed once every wave's own LDS-DMA loads are complete (hipcc does not count them before a barrier: the wait is explicit) and the
		// waves have met; every wave is then through with stage g - 1 as well
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		int t2 = threadIdx.x;
		asm volatile("" : "+v"(t2));
		const int ln = t2 & 63, kh2 = (t2 >> 5) & 1;
		if (g + 1 < g1) request(g + 1, buf ^ 1, ln, t2 >> 6);
#else
		const int buf = 0;
		int t2 = threadIdx.x;
		asm volatile("" : "+v"(t2));
		const int ln = t2 & 63, kh2 = (t2 >> 5) & 1;
		__syncthreads();
		request(g, 0, ln, t2 >> 6);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
#endif
#pragma unroll
		for (int tile = 0; tile < TPS; ++tile) {
			const int row0 = (g * TPS + tile) << 5;
			if (row0 < rows) {
				// accumulator r of a lane is train row (r & 3) + 8 (r >> 2) + 4 * k-half of the tile: both sets start from the rows' words
				v16f_t acc0, acc1;
#pragma unroll
				for (int jp = 0; jp < 4; ++jp) {
					const float4 w4 = *reinterpret_cast<const float4*>(&wrow[buf][(tile << 5) + 8 * jp + 4 * kh2]);
					acc0[4 * jp] = w4.x; acc0[4 * jp + 1] = w4.y; acc0[4 * jp + 2] = w4.z; acc0[4 * jp + 3] = w4.w;
				}
				acc1 = acc0;
#pragma unroll
				for (int s = 0; s < NS; ++s) {
					const uint4 av = ex[buf][tile][s][ln];
					const v8i_t va{(int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0};
#if MCS_MM_AB == 3
					acc0[s] += __int_as_float(va[0]); acc1[s] += __int_as_float(va[1] ^ bq[1][s][0]);
#else
					acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, bq[0][s], acc0, 4, 4, 0, 0, 0, 0);
					acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, bq[1][s], acc1, 4, 4, 0, 0, 0, 0);
#endif
				}
				// result register r of a lane = the key of train row (r & 3) + 8 (r >> 2) + 4 * k-half of the tile, for column (lane & 31) of its set.
				// The swap exchanges the upper half of set 0's register with the lower half of set 1's: afterwards lo[r] is row (r & 3) + 8 (r >> 2)
				// and hi[r] row (r & 3) + 8 (r >> 2) + 4 of the lane's OWN query, in every lane
#pragma unroll
				for (int jp = 0; jp < 4; ++jp) {
					uint32_t lo[4], hi[4];
#pragma unroll
					for (int u = 0; u < 4; ++u) {
						const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc0[4 * jp + u]), __float_as_uint(acc1[4 * jp + u]), false, false);
						lo[u] = sw[0]; hi[u] = sw[1];
					}
#pragma unroll
					for (int h = 0; h < 2; ++h) {   // four rows at a time: a column has room for 16, so a merge is due once a lane holds more than 12
#pragma unroll
						for (int u = 0; u < 4; ++u)
#if MCS_MM_AB == 4
							next += (h ? hi[u] : lo[u]) == 0x12345u ? 1024u : 0u;
#else
							append(next, h ? hi[u] : lo[u], rawLim);
#endif
						if (__any(next > col0 + (CB - 4) * kColStride)) { flush(); rawLim = limit(); }
					}
				}
			}
		}
	}
	flush();
	if (qi < a.nq) {
		// splits == 1: these are the final lists; otherwise a partial list per split
		uint32_t* dst = a.splits == 1 ? a.keys + (size_t)set * K * a.nq : a.partial + ((size_t)set * a.splits + split) * K * a.nq;
#pragma unroll
		for (int p = 0; p < K; ++p) dst[(size_t)p * a.nq + qi] = best[p];
		if (a.splits == 1) a.outCount[(size_t)set * a.nq + qi] = 0;
		else a.partialCount[((size_t)set * a.splits + split) * a.nq + qi] = 0;
	}
}

template <int K, int DW>
static void launch_mfma_kd(const MatchArgs& a, hipStream_t s) {
	dim3 grid((a.nq + XQ - 1) / XQ, a.splits, a.nsets);
	if (a.qm && a.tm) hipLaunchKernelGGL((k_match_mfma<K, DW, true>), grid, dim3(XQ), 0, s, a);
	else hipLaunchKernelGGL((k_match_mfma<K, DW, false>), grid, dim3(XQ), 0, s, a);
}

template <int K>
static void launch_mfma_k(const MatchArgs& a, hipStream_t s) {
	if (a.dim == 16) launch_mfma_kd<K, 4>(a, s);
	else launch_mfma_kd<K, 8>(a, s);
}

// the partial-list kernel of launch_match() for the shapes the matrix-core form serves: 16 / 32-byte descriptors, no count_le output, no camera groups,
// at most 2^14 train rows per set (the index rides in the low 14 bits of a float)
bool match_mfma_shape(const MatchArgs& a) {
	return (a.dim == 16 || a.dim == 32) && a.countThresh < 0 && !(a.qgroup && a.tgroup) && a.nt <= kMaxTrainRows;
}
// scratch of the expanded train sets (the caller allocates; a.exA == nullptr sends the call to mcs_match.hip)
void match_mfma_scratch(const MatchArgs& a, int tsets, size_t* bytesA, size_t* bytesW, int* stages) {
	const int st = (a.nt + 63) / 64, ns = (a.qm && a.tm ? 2 : 1) * (a.dim / 8);
	*stages = st;
	*bytesA = (size_t)tsets * st * 2 * ns * 64 * sizeof(uint4);
	*bytesW = ((size_t)tsets * st * 64 + 64) * sizeof(float);   // (a stage's words are fetched 64 at a time)
}
bool match_mfma_serves(const MatchArgs& a) { return match_mfma_shape(a) && a.exA && a.exW && a.exRows && a.tsets >= 1; }

// the train sets' pass; launch_match_mfma runs it itself unless the caller has (a.exDone: on another stream, ordered before s)
void launch_match_expand(const MatchArgs& a, hipStream_t s) {
	const bool masked = a.qm && a.tm;
	if (a.dim == 16) { if (masked) hipLaunchKernelGGL((k_expand_train<4, true>), dim3(a.tsets, (a.nt + 255) / 256), dim3(256), 0, s, a); else hipLaunchKernelGGL((k_expand_train<4, false>), dim3(a.tsets, (a.nt + 255) / 256), dim3(256), 0, s, a); }
	else { if (masked) hipLaunchKernelGGL((k_expand_train<8, true>), dim3(a.tsets, (a.nt + 255) / 256), dim3(256), 0, s, a); else hipLaunchKernelGGL((k_expand_train<8, false>), dim3(a.tsets, (a.nt + 255) / 256), dim3(256), 0, s, a); }
}

void launch_match_mfma(const MatchArgs& a, hipStream_t s) {
	if (!a.exDone) launch_match_expand(a, s);
	switch (a.K) {
		case 1: launch_mfma_k<1>(a, s); break;
		case 2: launch_mfma_k<2>(a, s); break;
		case 4: launch_mfma_k<4>(a, s); break;
		case 8: launch_mfma_k<8>(a, s); break;
		case 16: launch_mfma_k<16>(a, s); break;
		default: launch_mfma_k<32>(a, s); break;
	}
}

}  // namespace mcs

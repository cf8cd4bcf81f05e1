// mcs_fast.hip — E2: grid-cell FAST (9/16, and the 7/12 and 5/8 rings) + non-max suppression + mirror-mask filter.
// Reference: src/mdBRIEFextractorOct.cpp:863-949 (one cv::FastFeatureDetector(th, nonmax=true, TYPE_9_16)->detect()
// per ~30x30 cell view with its mask view); FAST arithmetic per SURVEY Appendix A.3.
//
// One 256-thread workgroup per (image, cell): the cell's processed region plus its 3-px ring is staged in LDS once,
// every pixel's corner score is computed from LDS, NMS runs on an LDS score tile whose 1-px frame is zero — which is
// exactly the reference's behaviour: every cell is its own FAST() call, so a neighbour outside the cell's processed
// region counts as score 0 and corners are never suppressed across a cell seam.
// Score (closed form of cv::cornerScore<16>, proven equal by tests/test_oracle_kat.py):
//     d[k] = v - I[k];  A = max over the 16 arcs of 9 contiguous k of min d;  B = max over arcs of min(-d)
//     corner <=> max(A,B) > t;   score = max(A,B) - 1  (stored as u8, 0 = no corner)
// Survivors are emitted in the reference's order (row-major inside the cell) with wave64 ballots + prefix counts into
// the cell's private slot range, so the later compaction is a pure prefix sum over cells (cell row-major order).
// Candidate record: x | y<<12 | score<<24 with x,y relative to minBorder (22), like vToDistributeKeys.
#include "mcs_common.h"

#include <algorithm>

namespace mcs {

// LDS geometry for cells of at most CW x CW processed pixels (the level's wCell / hCell): tile = cell + 3-px ring, score tile = cell + 1-px zero frame.
// Two instances: CW = 40 (7.8 KB per workgroup: every level with four or more cell columns AND rows — a 30-px grid on w px gives cells of
// ceil(w / floor(w / 30)) <= 40 from 120 px on) and CW = 60 (any cell).
#ifndef MCS_FAST_BS
#define MCS_FAST_BS 128
#endif
template <int CW> struct FastGeom {
	static constexpr int kTileX = 4;   // tile column of the cell's first processed pixel: a 4-byte left margin (3 ring pixels + 1), so that groups of 4 pixels are aligned dwords
	static constexpr int kTilePitch = (CW + kTileX + 3 + 4 + 3) / 4 * 4, kTileRows = CW + 6;   // + 4: the packed compass test reads one dword past the right ring
	static constexpr int kScPitch = (CW + 2 + 3) / 4 * 4, kScRows = CW + 2;
	static constexpr int kBitWords = (CW * CW + 63) / 64 * 2, kGroups = (CW * CW + 63) / 64;
};

// ---- the two small rings (FastFeatureDetector TYPE_7_12 / TYPE_5_8, reference src/mdBRIEFextractorOct.cpp:869-872) -------------------------------------
// OpenCV 3.x's FAST_t<patternSize> keeps the 3-pixel border for every ring, and its quick rejection test always reads entries 0|8, 2|10, 4|12, 6|14, 1|9,
// 3|11, 5|13, 7|15 of a 25-entry offset table that has wrapped around for the small rings (pixel[k] = pixel[k - patternSize]).  So a pixel is a corner iff
//   ring 12:  the pairs (0,8) (2,10) (4,0) (6,2) (1,9) (3,11) (5,1) (7,3) each hold a darker pixel AND a run of 7 contiguous darker pixels exists (or the same
//             with brighter);
//   ring 8:   all 8 ring pixels are darker, or all are brighter (the wrapped table pairs every entry with itself; the 5-run then always exists).
// Score: cornerScore<12> / <8> = max over the arcs of 7 / 5 contiguous ring pixels of min d (or of min -d), minus 1.
template <int P, int kTilePitch> struct Ring;
template <int kTilePitch> struct Ring<12, kTilePitch> {
	static __device__ __forceinline__ void diffs(const uint8_t* c, int (&d)[12]) {
		const int v = c[0];
		d[0] = v - c[2 * kTilePitch]; d[1] = v - c[2 * kTilePitch + 1]; d[2] = v - c[kTilePitch + 2]; d[3] = v - c[2];
		d[4] = v - c[-kTilePitch + 2]; d[5] = v - c[-2 * kTilePitch + 1]; d[6] = v - c[-2 * kTilePitch]; d[7] = v - c[-2 * kTilePitch - 1];
		d[8] = v - c[-kTilePitch - 2]; d[9] = v - c[-2]; d[10] = v - c[kTilePitch - 2]; d[11] = v - c[2 * kTilePitch - 1];
	}
};
template <int kTilePitch> struct Ring<8, kTilePitch> {
	static __device__ __forceinline__ void diffs(const uint8_t* c, int (&d)[8]) {
		const int v = c[0];
		d[0] = v - c[kTilePitch]; d[1] = v - c[kTilePitch + 1]; d[2] = v - c[1]; d[3] = v - c[-kTilePitch + 1];
		d[4] = v - c[-kTilePitch]; d[5] = v - c[-kTilePitch - 1]; d[6] = v - c[-1]; d[7] = v - c[kTilePitch - 1];
	}
};
// the quick test of the small rings: bit 0 = the darker branch may hold a corner, bit 1 = the brighter branch
template <int P>
__device__ __forceinline__ int small_ring_pretest(const int (&d)[P], int t) {
	if (P == 8) {
		bool dk = true, br = true;
#pragma unroll
		for (int k = 0; k < 8; ++k) { dk &= d[k] > t; br &= d[k] < -t; }
		return (dk ? 1 : 0) | (br ? 2 : 0);
	}
	constexpr int pa[8] = {0, 2, 4, 6, 1, 3, 5, 7}, pb[8] = {8, 10, 0, 2, 9, 11, 1, 3};   // entries 8..15 of the wrapped table, modulo 12
	bool dk = true, br = true;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		dk &= (d[pa[i] % P] > t) | (d[pb[i] % P] > t);
		br &= (d[pa[i] % P] < -t) | (d[pb[i] % P] < -t);
	}
	return (dk ? 1 : 0) | (br ? 2 : 0);
}
template <int P, int kTilePitch>
__device__ __forceinline__ bool small_ring_quick(const uint8_t* c, int t) {
	int d[P];
	Ring<P, kTilePitch>::diffs(c, d);
	return small_ring_pretest<P>(d, t) != 0;
}
template <int P, int kTilePitch>
__device__ __forceinline__ int small_ring_score(const uint8_t* c, int t) {
	int d[P];
	Ring<P, kTilePitch>::diffs(c, d);
	const int pre = small_ring_pretest<P>(d, t);
	constexpr int R = P / 2 + 1;   // arc length: 7 of 12, 5 of 8
	int A = -256, Bn = 256;
#pragma unroll
	for (int k = 0; k < P; ++k) {
		int lo = d[k], hi = d[k];
#pragma unroll
		for (int j = 1; j < R; ++j) { lo = min(lo, d[(k + j) % P]); hi = max(hi, d[(k + j) % P]); }
		A = max(A, lo);
		Bn = min(Bn, hi);
	}
	const bool corner = ((pre & 1) && A > t) || ((pre & 2) && -Bn > t);
	return corner ? max(A, -Bn) - 1 : 0;
}

// Necessary condition for a 9-of-16 arc: it covers at least two ADJACENT compass points (k = 0, 4, 8, 12), so two
// adjacent compass pixels must both be darker (d > t) or both be brighter (d < -t) than the centre.  Stricter than
// cv::FAST's opposite-pair test and never rejects a corner.
template <int kTilePitch>
__device__ __forceinline__ bool fast_quick(const uint8_t* c, int t) {
	const int v = c[0];
	const int d0 = v - c[3 * kTilePitch], d8 = v - c[-3 * kTilePitch], d4 = v - c[3], d12 = v - c[-3];
	const bool h0 = d0 > t, h4 = d4 > t, h8 = d8 > t, h12 = d12 > t;
	const bool l0 = d0 < -t, l4 = d4 < -t, l8 = d8 < -t, l12 = d12 < -t;
	return (h0 & h4) | (h4 & h8) | (h8 & h12) | (h12 & h0) | (l0 & l4) | (l4 & l8) | (l8 & l12) | (l12 & l0);
}

// The same test for 4 horizontally adjacent pixels at once (an aligned group of the tile row): five aligned LDS dwords instead of 20 byte reads, the
// differences as packed 16-bit pairs (v_pk_sub_i16 / v_pk_min_i16 / v_pk_max_i16).  With d = centre - ring pixel: two adjacent compass points darker
// <=> min(d_a, d_b) > t, brighter <=> min(-d_a, -d_b) > t, so pass <=> max over the four adjacent pairs of max(min(d_a, d_b), -max(d_a, d_b)) > t.
// Returns bit j = pixel 4g + j passes.
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s as_v2s(uint32_t x) { union { uint32_t u; v2s v; } c; c.u = x; return c.v; }
template <int kTilePitch>
__device__ __forceinline__ int fast_quick4(const uint8_t* rowc /* tile row of the centres, at the group's first pixel (4-aligned) */, int t) {
	const uint32_t A = *reinterpret_cast<const uint32_t*>(rowc - 4), Cc = *reinterpret_cast<const uint32_t*>(rowc), B = *reinterpret_cast<const uint32_t*>(rowc + 4);
	const uint32_t U = *reinterpret_cast<const uint32_t*>(rowc - 3 * kTilePitch), D = *reinterpret_cast<const uint32_t*>(rowc + 3 * kTilePitch);
	const uint32_t Lf = __builtin_amdgcn_alignbyte(Cc, A, 1);   // the pixels 3 to the left of each centre:  A[1] A[2] A[3] C[0]
	const uint32_t Rt = __builtin_amdgcn_alignbyte(B, Cc, 3);   // 3 to the right:                           C[3] B[0] B[1] B[2]
	int bits = 0;
#pragma unroll
	for (int h = 0; h < 2; ++h) {   // h = 0: bytes 0 and 2 of every dword, h = 1: bytes 1 and 3
		const uint32_t m = 0x00FF00FFu;
		const v2s v = as_v2s((Cc >> (8 * h)) & m);
		const v2s d0 = v - as_v2s((D >> (8 * h)) & m), d8 = v - as_v2s((U >> (8 * h)) & m), d4 = v - as_v2s((Rt >> (8 * h)) & m), d12 = v - as_v2s((Lf >> (8 * h)) & m);
		const v2s lo = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_min(d0, d4), __builtin_elementwise_min(d4, d8)),
		                                         __builtin_elementwise_max(__builtin_elementwise_min(d8, d12), __builtin_elementwise_min(d12, d0)));
		const v2s hi = __builtin_elementwise_min(__builtin_elementwise_min(__builtin_elementwise_max(d0, d4), __builtin_elementwise_max(d4, d8)),
		                                         __builtin_elementwise_min(__builtin_elementwise_max(d8, d12), __builtin_elementwise_max(d12, d0)));
		const v2s best = __builtin_elementwise_max(lo, -hi);
		bits |= (best.x > t ? 1 : 0) << h;
		bits |= (best.y > t ? 1 : 0) << (2 + h);
	}
	return bits;
}

template <int kTilePitch>
__device__ __forceinline__ int fast_score(const uint8_t* c /* centre in LDS tile */, int t) {
	const int v = c[0];
	int d[16];
	d[0] = v - c[3 * kTilePitch];
	d[8] = v - c[-3 * kTilePitch];
	d[4] = v - c[3];
	d[12] = v - c[-3];
	d[1] = v - c[3 * kTilePitch + 1];
	d[2] = v - c[2 * kTilePitch + 2];
	d[3] = v - c[1 * kTilePitch + 3];
	d[5] = v - c[-1 * kTilePitch + 3];
	d[6] = v - c[-2 * kTilePitch + 2];
	d[7] = v - c[-3 * kTilePitch + 1];
	d[9] = v - c[-3 * kTilePitch - 1];
	d[10] = v - c[-2 * kTilePitch - 2];
	d[11] = v - c[-1 * kTilePitch - 3];
	d[13] = v - c[1 * kTilePitch - 3];
	d[14] = v - c[2 * kTilePitch - 2];
	d[15] = v - c[3 * kTilePitch - 1];
	int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
	for (int k = 0; k < 16; ++k) { lo2[k] = min(d[k], d[(k + 1) & 15]); hi2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
	for (int k = 0; k < 16; ++k) { lo4[k] = min(lo2[k], lo2[(k + 2) & 15]); hi4[k] = max(hi2[k], hi2[(k + 2) & 15]); }
	int A = -256, Bn = 256;
#pragma unroll
	for (int k = 0; k < 16; ++k) {
		int lo9 = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);   // min d[k..k+8]
		int hi9 = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);   // max d[k..k+8]
		A = max(A, lo9);
		Bn = min(Bn, hi9);
	}
	const int best = max(A, -Bn);
	return best > t ? best - 1 : 0;
}

// The same score for TWO survivors at once, one in each half of packed 16-bit lanes (differences fit 9 bits): the min / max network is issued once for the pair
// (v_pk_min_i16 / v_pk_max_i16 cost what the 32-bit forms cost), so a cell's ~150 survivors take one trip of the workgroup instead of two, the second a sixth full.
template <int kTilePitch>
__device__ __forceinline__ uint32_t fast_score2(const uint8_t* ca, const uint8_t* cb, int t) {
	auto pk = [&](int off) { return as_v2s((uint32_t)ca[off] | ((uint32_t)cb[off] << 16)); };
	const v2s v = pk(0);
	v2s d[16];
	d[0] = v - pk(3 * kTilePitch);
	d[8] = v - pk(-3 * kTilePitch);
	d[4] = v - pk(3);
	d[12] = v - pk(-3);
	d[1] = v - pk(3 * kTilePitch + 1);
	d[2] = v - pk(2 * kTilePitch + 2);
	d[3] = v - pk(1 * kTilePitch + 3);
	d[5] = v - pk(-1 * kTilePitch + 3);
	d[6] = v - pk(-2 * kTilePitch + 2);
	d[7] = v - pk(-3 * kTilePitch + 1);
	d[9] = v - pk(-3 * kTilePitch - 1);
	d[10] = v - pk(-2 * kTilePitch - 2);
	d[11] = v - pk(-1 * kTilePitch - 3);
	d[13] = v - pk(1 * kTilePitch - 3);
	d[14] = v - pk(2 * kTilePitch - 2);
	d[15] = v - pk(3 * kTilePitch - 1);
	v2s lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
	for (int k = 0; k < 16; ++k) { lo2[k] = __builtin_elementwise_min(d[k], d[(k + 1) & 15]); hi2[k] = __builtin_elementwise_max(d[k], d[(k + 1) & 15]); }
#pragma unroll
	for (int k = 0; k < 16; ++k) { lo4[k] = __builtin_elementwise_min(lo2[k], lo2[(k + 2) & 15]); hi4[k] = __builtin_elementwise_max(hi2[k], hi2[(k + 2) & 15]); }
	v2s A = {-256, -256}, Bn = {256, 256};
#pragma unroll
	for (int k = 0; k < 16; ++k) {
		const v2s lo9 = __builtin_elementwise_min(__builtin_elementwise_min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);
		const v2s hi9 = __builtin_elementwise_max(__builtin_elementwise_max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);
		A = __builtin_elementwise_max(A, lo9);
		Bn = __builtin_elementwise_min(Bn, hi9);
	}
	const v2s best = __builtin_elementwise_max(A, -Bn);
	const int ba = best.x, bb = best.y;
	return (uint32_t)(ba > t ? ba - 1 : 0) | ((uint32_t)(bb > t ? bb - 1 : 0) << 8);
}

template <int CW, int kFastBS, int P>   // P = ring size: 16 (TYPE_9_16), 12 (TYPE_7_12), 8 (TYPE_5_8)
__global__ __launch_bounds__(kFastBS) void k_fast_cells(ExtractBuffers b, int nimg, int nblocks, int perXcd, int cell0, int ncells) {
	typedef FastGeom<CW> Geo;
	constexpr int kTilePitch = Geo::kTilePitch, kTileRows = Geo::kTileRows, kScPitch = Geo::kScPitch, kScRows = Geo::kScRows;
	__shared__ __attribute__((aligned(16))) uint8_t tile[kTileRows * kTilePitch];
	__shared__ __attribute__((aligned(16))) uint8_t sc[kScRows * kScPitch];
	__shared__ uint32_t keepBits[Geo::kBitWords];   // NMS + mask verdict per pixel of the cell (row-major bit index)
	__shared__ int groupOff[64];         // exclusive prefix of the kept-pixel counts per 64-pixel group
	__shared__ int runBase;
	__shared__ int nSurv;
	__shared__ unsigned short surv[CW * CW];   // pixel indices that pass the compass test

	// XCD-aware mapping: hardware places block i on XCD i%8; give every XCD a contiguous run of cells so that the
	// overlapping cell rings / shared cache lines of neighbouring cells hit the same L2.
	const int logical = (blockIdx.x % kNumXCD) * perXcd + blockIdx.x / kNumXCD;
	if (logical >= nblocks) return;
	const PyrDesc& d = *b.desc;
	const int img = logical / ncells;                  // this launch covers cells [cell0, cell0 + ncells) of every image (a range of pyramid levels)
	const int ci = cell0 + (logical - img * ncells);
	const CellInfo cell = b.cells[ci];
	const LevelInfo& L = d.lv[cell.level];
	const int tid = threadIdx.x;
	const int cw = cell.cw, ch = cell.ch;
	int* countOut = b.cellCount + (size_t)img * d.cellsPerImage + ci;
	if (cw <= 0 || ch <= 0) { if (tid == 0) *countOut = 0; return; }

	int stride;
	const uint8_t* src = level_ptr(b, d, img, cell.level, &stride);
	src += (size_t)(cell.y0 - 3) * stride + (cell.x0 - Geo::kTileX);
	const int tw = cw + Geo::kTileX + 3, th = ch + 6;
	const int ndw = (tw + 3) >> 2;   // unaligned dword loads; the <= 3 bytes of over-read per row stay inside the image row
	// i / ndw by multiplication: ndw <= 17 and i < 17 * 66, so with M = ceil(2^16 / ndw) the error term i * (M*ndw - 2^16) < 2^16 and
	// (i * M) >> 16 is exact; 32-bit offsets keep the address arithmetic out of 64-bit multiplies.
	const unsigned rowM = (65536u + (unsigned)ndw - 1u) / (unsigned)ndw;
	// two dwords per thread and trip (a ~30x30 cell is 370 dwords: one trip): both loads are in flight before the first LDS store waits
	const int ndwTile = ndw * th;
	for (int i = tid; i < ndwTile; i += 2 * kFastBS) {
		const int i2 = i + kFastBS;
		const bool has2 = i2 < ndwTile;
		const unsigned ty = ((unsigned)i * rowM) >> 16, kx = (unsigned)i - ty * (unsigned)ndw;
		const unsigned ty2 = ((unsigned)i2 * rowM) >> 16, kx2 = (unsigned)i2 - ty2 * (unsigned)ndw;
		uint32_t v, v2 = 0;
		__builtin_memcpy(&v, src + (ty * (unsigned)stride + 4u * kx), 4);
		if (has2) __builtin_memcpy(&v2, src + (ty2 * (unsigned)stride + 4u * kx2), 4);
		*reinterpret_cast<uint32_t*>(&tile[ty * kTilePitch + 4 * kx]) = v;
		if (has2) *reinterpret_cast<uint32_t*>(&tile[ty2 * kTilePitch + 4 * kx2]) = v2;
	}
	const int sw = cw + 2, sh = ch + 2;
	for (int i = tid; i < sh * (kScPitch / 4); i += kFastBS) reinterpret_cast<uint32_t*>(sc)[i] = 0;   // score tile cleared as dwords
	for (int i = tid; i < Geo::kBitWords; i += kFastBS) keepBits[i] = 0;
	if (tid == 0) { runBase = 0; nSurv = 0; }
	__syncthreads();

	const int t = d.fastThreshold;
	const int npx = cw * ch;
	// p / cw by multiplication: cw <= 60 and p < 3600, so with M = ceil(2^18 / cw) the error term p * (M*cw - 2^18) < 3600 * 60 < 2^18
	// and (p * M) >> 18 is exact (a variable integer division is ~20 VALU instructions, and there were three per pixel).
	const unsigned divM = (262144u + (unsigned)cw - 1u) / (unsigned)cw;
	const int lane = tid & 63, wave = tid >> 6;
	// pass 1: cheap compass test on every pixel; survivors are compacted into an LDS list (order is irrelevant here) so that
	// pass 2 — the full 16-pixel arc score, ~10x the work — runs with all lanes busy instead of diverging inside each wave
	if constexpr (P == 16) {
		// pass 1, 16-pixel ring: four adjacent pixels per thread (fast_quick4), one list reservation per wave and trip for the four ballots
		const int gpr = (cw + 3) >> 2, ngrp = gpr * ch;   // groups per row: gpr <= 15, g < 15 * 60: M = ceil(2^16 / gpr) is exact
		const unsigned gM = (65536u + (unsigned)gpr - 1u) / (unsigned)gpr;
		for (int base = 0; base < ngrp; base += kFastBS) {
			const int g = base + tid;
			int bits = 0, py = 0, gx = 0;
			if (g < ngrp) {
				py = (int)(((unsigned)g * gM) >> 16); gx = g - py * gpr;
				bits = fast_quick4<kTilePitch>(&tile[(py + 3) * kTilePitch + 4 * gx + Geo::kTileX], t);
				const int left = cw - 4 * gx;   // pixels of the group inside the cell (the last group of a row may be partial)
				if (left < 4) bits &= (1 << left) - 1;
			}
			unsigned long long bal[4];
			int cnt[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) { bal[j] = __ballot((bits >> j) & 1); cnt[j] = __popcll(bal[j]); }
			const int total = cnt[0] + cnt[1] + cnt[2] + cnt[3];
			int wbase = 0;
			if (lane == 0 && total) wbase = atomicAdd(&nSurv, total);
			wbase = __shfl(wbase, 0);
			const int p0 = py * cw + 4 * gx;
			const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if ((bits >> j) & 1) surv[wbase + __popcll(bal[j] & below)] = (unsigned short)(p0 + j);
				wbase += cnt[j];
			}
		}
	} else
	for (int base = 0; base < npx; base += kFastBS) {
		const int p = base + tid;
		bool pass = false;
		if (p < npx) {
			const int py = (int)(((unsigned)p * divM) >> 18), px = p - py * cw;
			const uint8_t* c = &tile[(py + 3) * kTilePitch + px + Geo::kTileX];
			if constexpr (P == 16) pass = fast_quick<kTilePitch>(c, t);
			else pass = small_ring_quick<P, kTilePitch>(c, t);
		}
		const unsigned long long bal = __ballot(pass);
		int wbase = 0;
		if (lane == 0 && bal) wbase = atomicAdd(&nSurv, __popcll(bal));
		wbase = __shfl(wbase, 0);
		if (pass) surv[wbase + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)p;
	}
	__syncthreads();
	const int ns = nSurv;
	if constexpr (P == 16) {
		for (int i = 2 * tid; i < ns; i += 2 * kFastBS) {
			const int pa = surv[i], pb = surv[i + 1 < ns ? i + 1 : i];
			const int ya = (int)(((unsigned)pa * divM) >> 18), xa = pa - ya * cw, yb = (int)(((unsigned)pb * divM) >> 18), xb = pb - yb * cw;
			const uint32_t s2 = fast_score2<kTilePitch>(&tile[(ya + 3) * kTilePitch + xa + Geo::kTileX], &tile[(yb + 3) * kTilePitch + xb + Geo::kTileX], t);
			sc[(ya + 1) * kScPitch + xa + 1] = (uint8_t)s2;
			if (i + 1 < ns) sc[(yb + 1) * kScPitch + xb + 1] = (uint8_t)(s2 >> 8);
		}
	} else
	for (int i = tid; i < ns; i += kFastBS) {
		const int p = surv[i];
		const int py = (int)(((unsigned)p * divM) >> 18), px = p - py * cw;
		const uint8_t* c = &tile[(py + 3) * kTilePitch + px + Geo::kTileX];
		int score;
		if constexpr (P == 16) score = fast_score<kTilePitch>(c, t);
		else score = small_ring_score<P, kTilePitch>(c, t);
		sc[(py + 1) * kScPitch + px + 1] = (uint8_t)score;
	}
	__syncthreads();

	uint32_t* slots = b.slots + (size_t)img * d.slotsPerImage + cell.slot;
	const short* mapX = b.maskMap + L.mapX;
	const short* mapY = b.maskMap + L.mapY;
	const uint8_t* mask = b.mask0 ? b.mask0 + (size_t)img * b.mask0Pitch : nullptr;
	// pass 3a: non-max suppression + mirror mask, only for the pixels that have a score at all (the compass survivors); the verdicts go
	// into a bitmap indexed by the pixel's row-major number inside the cell
	for (int i = tid; i < ns; i += kFastBS) {
		const int p = surv[i];
		const int py = (int)(((unsigned)p * divM) >> 18), px = p - py * cw;
		const uint8_t* q = &sc[(py + 1) * kScPitch + px + 1];
		const int s0 = q[0];
		const int nb = max(max(max((int)q[-1], (int)q[1]), max((int)q[-kScPitch - 1], (int)q[-kScPitch])),
		                   max(max((int)q[-kScPitch + 1], (int)q[kScPitch - 1]), max((int)q[kScPitch], (int)q[kScPitch + 1])));
		bool keep = s0 > nb;   // strictly greater than all 8 neighbours (a score of 0 never is)
		if (keep && mask) {   // KeyPointsFilter::runByPixelsMask on the (nearest-neighbour) mask pyramid
			const int mx = mapX[cell.x0 + px], my = mapY[cell.y0 + py];
			keep = mask[(size_t)my * b.mask0Stride + mx] != 0;
		}
		if (keep) atomicOr(&keepBits[p >> 5], 1u << (p & 31));
	}
	__syncthreads();
	// the emission order (row-major inside the cell, the reference's) is an exclusive prefix sum over the kept-pixel counts of <= 57 groups of
	// 64 pixels (two bitmap words each; one wave does it) — two barriers for the whole cell instead of three per 256-pixel slab.
	const int ngroups = (npx + 63) >> 6;
	if (wave == 0) {
		const int v = lane < ngroups ? __popc(keepBits[2 * lane]) + __popc(keepBits[2 * lane + 1]) : 0;
		int incl = v;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
		groupOff[lane] = incl - v;
		if (lane == 63) runBase = incl;
	}
	__syncthreads();
	// every kept survivor writes its record at (kept pixels before it in row-major order): group prefix + kept bits below it in its own 64-pixel group
	for (int i = tid; i < ns; i += kFastBS) {
		const int p = surv[i];
		const uint32_t w = keepBits[p >> 5];
		if ((w >> (p & 31)) & 1u) {
			const int g = p >> 6;
			const uint32_t below = (p & 32) ? __popc(keepBits[2 * g]) + __popc(w & ((1u << (p & 31)) - 1u)) : __popc(w & ((1u << (p & 31)) - 1u));
			const int py = (int)(((unsigned)p * divM) >> 18), px = p - py * cw;
			const int s0 = sc[(py + 1) * kScPitch + px + 1];
			slots[groupOff[g] + (int)below] = (uint32_t)(cell.x0 + px - kMinBorder) | ((uint32_t)(cell.y0 + py - kMinBorder) << 12) | ((uint32_t)s0 << 24);
		}
	}
	if (tid == 0) *countOut = runBase;
	(void)sw;
}

// (A strip form — one workgroup per run of 8 cells of a cell row, per-wave survivor queues instead of the list + atomics, the score tile rescanned as dwords
// for the non-max suppression, four barriers per 8 cells — was built and measured in round 3: bit-exact, but 0.69 -> 1.10 ms per step under the profiler: 32 KB
// of LDS per workgroup halves the waves per CU, and a wave's 50 dependent trips of test -> ballot -> queue are a longer chain than the cell form's 10.)
// cells of pyramid levels [level0, level1): level 0 needs no pyramid, so the caller can run it beside the resize chain
void launch_fast(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s, int level0, int level1) {
	if (level1 > hd.nlevels) level1 = hd.nlevels;
	if (level0 >= level1) return;
	const int cell0 = hd.lv[level0].cellBase;
	const int cellEnd = level1 < hd.nlevels ? hd.lv[level1].cellBase : hd.cellsPerImage;
	const int ncells = cellEnd - cell0;
	const int nblocks = nimg * ncells;
	if (nblocks <= 0) return;
	const int perXcd = (nblocks + kNumXCD - 1) / kNumXCD;
	int cellMax = 0;
	for (int l = level0; l < level1; ++l) cellMax = std::max(cellMax, std::max(hd.lv[l].wCell, hd.lv[l].hCell));
	// two waves per cell for the small instance: alone 0.46 ms against 0.52 (one wave) and 0.55 (four waves); in the overlapped step one and two waves are within 1 %
	const dim3 grid(perXcd * kNumXCD);
#define MCS_FAST_LAUNCH(P)                                                                                                                      \
	do {                                                                                                                                        \
		if (cellMax <= 40) hipLaunchKernelGGL((k_fast_cells<40, MCS_FAST_BS, P>), grid, dim3(MCS_FAST_BS), 0, s, b, nimg, nblocks, perXcd, cell0, ncells);     \
		else hipLaunchKernelGGL((k_fast_cells<60, 256, P>), grid, dim3(256), 0, s, b, nimg, nblocks, perXcd, cell0, ncells);                   \
	} while (0)
	if (hd.fastRing == 16) MCS_FAST_LAUNCH(16);
	else if (hd.fastRing == 12) MCS_FAST_LAUNCH(12);
	else MCS_FAST_LAUNCH(8);
#undef MCS_FAST_LAUNCH
}

}  // namespace mcs

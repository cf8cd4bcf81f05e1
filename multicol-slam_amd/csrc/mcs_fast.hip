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
//
// AGAST (useAgast, reference :869-870, 912-914; cv::AgastFeatureDetector AGAST_5_8 / AGAST_7_12d / AGAST_7_12s / OAST_9_16) runs in the same kernel (template
// parameter AG): the detector's decision trees decide the plain segment test "N contiguous of the P ring pixels all brighter than v + t or all darker than v - t",
// so the closed form above on the type's ring IS the detector and its bisection score (tests/test_oracle_agast.py); what differs from FAST is the border (the
// ring's radius: 1 / 3 / 2 / 3, so the processed regions of neighbouring cells overlap and a corner can be reported by two cells, as in the reference) and the
// suppression: corners are merged into 4-connected regions by a forest of "dominated by" links, in raster order — one wave walks the corners that have a
// neighbour above or to the left, exactly the statements of cv::AGAST's loop (the indices of those neighbours come from a bitmap, in parallel).
#include "mcs_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace mcs {

// LDS geometry for cells of at most CW x CW processed pixels (the level's wCell / hCell): tile = cell + 3-px ring, score tile = cell + 1-px zero frame.
// Two instances: CW = 40 (7.8 KB per workgroup: every level with four or more cell columns AND rows — a 30-px grid on w px gives cells of
// ceil(w / floor(w / 30)) <= 40 from 120 px on) and CW = 60 (any cell).
#ifndef MCS_FAST_BS
#define MCS_FAST_BS 128
#endif
#ifndef MCS_FAST_SEG16
#define MCS_FAST_SEG16 1   // pass 1 of the 16-pixel ring on sixteen pixels per thread (round 5); 0: four pixels per thread (round 4), for A/B
#endif
template <int CW> struct FastGeom {
	static constexpr int kTileX = 4;   // tile column of the cell's first processed pixel: a 4-byte left margin (3 ring pixels + 1), so that groups of 4 pixels are aligned dwords
	static constexpr int kTilePitch = (CW + kTileX + 3 + 4 + 3) / 4 * 4, kTileRows = CW + 6;   // + 4: the packed compass test reads one dword past the right ring
	static constexpr int kScPitch = (CW + 2 + 3) / 4 * 4, kScRows = CW + 2;
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

// ---- AGAST: the plain segment test on the type's ring (cv::AgastFeatureDetector; what OpenCV 3.x computes is written out in DESIGN.md section 7) ----------------
// type 0 AGAST_5_8: the 8 neighbours, 5 contiguous; 1 AGAST_7_12d: the 12-pixel diamond of radius 3, 7 contiguous; 2 AGAST_7_12s: the 12-pixel square of radius 2
// (= FAST's 12 ring), 7 contiguous; 3 OAST_9_16: the 16-pixel circle, 9 contiguous (scored by fast_score2).  Border of the scan = the radius.
template <int AG> struct AgastGeom { static constexpr int P = AG == 0 ? 8 : (AG == 3 ? 16 : 12), N = P / 2 + 1, R = AG == 0 ? 1 : (AG == 2 ? 2 : 3); };
template <int AG, int kTilePitch>
__device__ __forceinline__ int agast_score(const uint8_t* c, int t) {
	constexpr int P = AgastGeom<AG>::P, N = AgastGeom<AG>::N;
	int d[P];
	if constexpr (AG == 1) {   // makeAgastOffsets AGAST_7_12d, in ring order
		const int v = c[0];
		d[0] = v - c[-3]; d[1] = v - c[kTilePitch - 2]; d[2] = v - c[2 * kTilePitch - 1]; d[3] = v - c[3 * kTilePitch];
		d[4] = v - c[2 * kTilePitch + 1]; d[5] = v - c[kTilePitch + 2]; d[6] = v - c[3]; d[7] = v - c[-kTilePitch + 2];
		d[8] = v - c[-2 * kTilePitch + 1]; d[9] = v - c[-3 * kTilePitch]; d[10] = v - c[-2 * kTilePitch - 1]; d[11] = v - c[-kTilePitch - 2];
	} else Ring<P, kTilePitch>::diffs(c, d);   // the same pixels in a circular order (the direction and the start do not matter for "contiguous")
	int A = -256, Bn = 256;
#pragma unroll
	for (int k = 0; k < P; ++k) {
		int lo = d[k], hi = d[k];
#pragma unroll
		for (int j = 1; j < N; ++j) { lo = min(lo, d[(k + j) % P]); hi = max(hi, d[(k + j) % P]); }
		A = max(A, lo);
		Bn = min(Bn, hi);
	}
	const int best = max(A, -Bn);
	return best > t ? best - 1 : 0;   // = the bisection of agast_cornerScore: the largest b for which the pixel is still a corner (t >= 1: a corner's score is >= 1)
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

// The same test for 4 horizontally adjacent pixels at once (an aligned group of the tile row): five aligned LDS dwords, the bytes spread into packed 16-bit
// pairs by v_perm_b32 (pixels 0 | 2 and 1 | 3).  Any of {0, 8} is adjacent to any of {4, 12}, so "two adjacent compass points darker" is
//   max(min(r0, r8), min(r4, r12)) < v - t      (the smaller of each opposite pair must both be dark <=> ... written on the ring values r, no differences)
// and brighter is  min(max(r0, r8), max(r4, r12)) > v + t:  six packed min / max, two packed subtractions, one min whose SIGN BITS are the verdicts.
// Returns the verdicts of pixels 0, 1, 2, 3 in bits 0, 1, 16, 17.
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s as_v2s(uint32_t x) { union { uint32_t u; v2s v; } c; c.u = x; return c.v; }
__device__ __forceinline__ uint32_t as_u32(v2s x) { union { uint32_t u; v2s v; } c; c.v = x; return c.u; }
// R = the ring's radius (3: the 16-pixel ring and AGAST's 12-pixel diamond; 2 / 1: AGAST's 12-pixel square and 8-pixel ring, whose arcs of 7 of 12 / 5 of 8 also cover two
// adjacent compass points): the selectors pick the pixels R to the right / left out of the three row dwords.
template <int kTilePitch, int R = 3>
__device__ __forceinline__ uint32_t fast_quick4(const uint8_t* rowc /* tile row of the centres, at the group's first pixel (4-aligned) */, v2s tt /* t in both halves */) {
	const uint32_t A = *reinterpret_cast<const uint32_t*>(rowc - 4), Cc = *reinterpret_cast<const uint32_t*>(rowc), B = *reinterpret_cast<const uint32_t*>(rowc + 4);
	const uint32_t U = *reinterpret_cast<const uint32_t*>(rowc - R * kTilePitch), D = *reinterpret_cast<const uint32_t*>(rowc + R * kTilePitch);
	// byte k of the concatenation Cc | B is pixel k of the group's row: pixel j + R for j = 0, 2 (h = 0) and j = 1, 3 (h = 1); byte k of A | Cc is pixel k - 4: pixel j - R
	constexpr uint32_t selR0 = 0x0c000c00u | (uint32_t)(R + 2) << 16 | (uint32_t)R, selR1 = 0x0c000c00u | (uint32_t)(R + 3) << 16 | (uint32_t)(R + 1);
	constexpr uint32_t selL0 = 0x0c000c00u | (uint32_t)(6 - R) << 16 | (uint32_t)(4 - R), selL1 = 0x0c000c00u | (uint32_t)(7 - R) << 16 | (uint32_t)(5 - R);
	static_assert(R != 3 || (selR0 == 0x0c050c03u && selR1 == 0x0c060c04u && selL0 == 0x0c030c01u && selL1 == 0x0c040c02u), "selectors of the 16-pixel ring");
	uint32_t sign[2];
#pragma unroll
	for (int h = 0; h < 2; ++h) {   // h = 0: pixels 0 and 2, h = 1: pixels 1 and 3.  v_perm_b32(s0, s1, sel): selector 0-3 = byte of s1, 4-7 = byte of s0, 0x0c = zero
		const uint32_t own = h ? 0x0c030c01u : 0x0c020c00u;
		const v2s v = as_v2s(__builtin_amdgcn_perm(0u, Cc, own));
		const v2s r0 = as_v2s(__builtin_amdgcn_perm(0u, D, own)), r8 = as_v2s(__builtin_amdgcn_perm(0u, U, own));
		const v2s r4 = as_v2s(__builtin_amdgcn_perm(B, Cc, h ? selR1 : selR0));    // R = 3: 3 to the right of pixel j: C[3] B[0] B[1] B[2]
		const v2s r12 = as_v2s(__builtin_amdgcn_perm(Cc, A, h ? selL1 : selL0));   //        3 to the left:             A[1] A[2] A[3] C[0]
		const v2s m1 = __builtin_elementwise_max(__builtin_elementwise_min(r0, r8), __builtin_elementwise_min(r4, r12));
		const v2s m2 = __builtin_elementwise_min(__builtin_elementwise_max(r0, r8), __builtin_elementwise_max(r4, r12));
		const v2s dark = m1 - (v - tt);      // < 0 <=> two adjacent compass points are darker than v - t
		const v2s bright = (v + tt) - m2;    // < 0 <=> ... brighter than v + t
		sign[h] = as_u32(__builtin_elementwise_min(dark, bright));
	}
	return ((sign[0] >> 15) & 0x00010001u) | ((sign[1] >> 14) & 0x00020002u);
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

// The same score for TWO survivors at once, one in each half of packed 16-bit lanes (differences fit 9 bits; v_pk_min_i16 / v_pk_max_i16 cost what the 32-bit
// forms cost).  The arcs are taken in pairs like cv::cornerScore does: with a_k = min d[k+1 .. k+8] for EVEN k, the arc from k is min(d[k], a_k) and
// the arc from k + 1 is min(a_k, d[k+9]), so  A = max over even k of min(a_k, max(d[k], d[k+9]))  — only the eight windows that start on an odd index are needed
// (three levels of eight packed minima), and the same with min / max exchanged for the brighter side: 110 packed operations per pair where all sixteen
// windows of both kinds took 176.
template <int kTilePitch>
__device__ __forceinline__ uint32_t fast_score2(const uint8_t* ca, const uint8_t* cb, int t) {
	auto pk = [&](int off) { v2s r; r.x = (short)ca[off]; r.y = (short)cb[off]; return r; };
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
	v2s lo2[8], hi2[8], lo4[8], hi4[8];   // windows starting at the odd index 2j + 1
#pragma unroll
	for (int j = 0; j < 8; ++j) { lo2[j] = __builtin_elementwise_min(d[2 * j + 1], d[(2 * j + 2) & 15]); hi2[j] = __builtin_elementwise_max(d[2 * j + 1], d[(2 * j + 2) & 15]); }
#pragma unroll
	for (int j = 0; j < 8; ++j) { lo4[j] = __builtin_elementwise_min(lo2[j], lo2[(j + 1) & 7]); hi4[j] = __builtin_elementwise_max(hi2[j], hi2[(j + 1) & 7]); }
	v2s A = {-256, -256}, Bn = {256, 256};
#pragma unroll
	for (int j = 0; j < 8; ++j) {   // k = 2j: a_k = min d[k+1 .. k+8] = the 8-window from odd index 2j + 1
		const v2s a = __builtin_elementwise_min(lo4[j], lo4[(j + 2) & 7]), bb = __builtin_elementwise_max(hi4[j], hi4[(j + 2) & 7]);
		const v2s e0 = d[2 * j], e9 = d[(2 * j + 9) & 15];
		A = __builtin_elementwise_max(A, __builtin_elementwise_min(a, __builtin_elementwise_max(e0, e9)));
		Bn = __builtin_elementwise_min(Bn, __builtin_elementwise_max(bb, __builtin_elementwise_min(e0, e9)));
	}
	const v2s best = __builtin_elementwise_max(A, -Bn);
	const int ba = best.x, bb = best.y;
	return (uint32_t)(ba > t ? ba - 1 : 0) | ((uint32_t)(bb > t ? bb - 1 : 0) << 8);
}

// inclusive prefix sum over the 64 lanes of a wave on the VALU data-parallel-primitive path (no LDS round trips): four row_shr steps inside the rows of 16, then
// row_bcast:15 / row_bcast:31 carry the row totals
__device__ __forceinline__ int wave_incl_scan(int x) {
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
	return x;
}

// A pixel of the cell is named by its code  row << 6 | column  (cells are at most 60 wide): row-major order, and the row / column come back with a shift and a mask.
template <int CW, int kFastBS, int P, int AG = -1>   // P = ring size: 16 (TYPE_9_16), 12 (TYPE_7_12), 8 (TYPE_5_8); AG >= 0: AGAST type AG (P = its ring size)
__global__ __launch_bounds__(kFastBS) void k_fast_cells(ExtractBuffers b, int nimg, int nblocks, int perXcd, int cell0, int ncells, unsigned ncellsM) {
	typedef FastGeom<CW> Geo;
	constexpr int kTilePitch = Geo::kTilePitch, kTileRows = Geo::kTileRows, kScPitch = Geo::kScPitch, kScRows = Geo::kScRows;
	// (AGAST: after the scores are in `sc` the tile is dead and holds the corners' codes in raster order, two bytes per pixel of the cell at most; the links of the
	// suppression walk take the place of `surv`)
	constexpr int kTileBytes = (AG >= 0 && 2 * CW * CW > kTileRows * kTilePitch) ? 2 * CW * CW : kTileRows * kTilePitch;
	__shared__ __attribute__((aligned(16))) uint8_t tile[kTileBytes];
	__shared__ __attribute__((aligned(16))) uint8_t sc[(kScRows * kScPitch + 15) / 16 * 16];
	__shared__ uint32_t keepBits[2 * CW];   // NMS + mask verdict per pixel of the cell: two words per row, bit = column
	__shared__ int rowOff[64];           // exclusive prefix of the kept-pixel counts per row
	__shared__ int runBase;
	__shared__ int nSurv;
	__shared__ unsigned short surv[CW * CW];   // codes of the pixels that pass the compass test
	constexpr int QR = AG >= 0 ? AgastGeom<AG < 0 ? 3 : AG>::R : 3;   // radius of the compass test
#ifndef MCS_FAST_ORDERED
#define MCS_FAST_ORDERED 1   // 0 (A/B): the round-5 form — survivors appended by atomics in any order, verdict bitmap + row prefix + a second loop for the emission
#endif
	// FAST on the 16-pixel ring: the survivor list is kept in ROW-MAJOR order (round 6), so that a kept survivor's place in the output is the number of kept survivors in
	// front of it in the list — non-max suppression, mask and emission are then ONE loop with a ballot, where a verdict bitmap, a row prefix and a second loop stood
	constexpr bool kOrdered = MCS_FAST_ORDERED && P == 16 && MCS_FAST_SEG16 && AG < 0;
	__shared__ int waveTot[kFastBS / 64];
	int listRun = 0;   // survivors listed by the trips so far (uniform)

	// XCD-aware mapping: hardware places block i on XCD i%8; give every XCD a contiguous run of cells so that the
	// overlapping cell rings / shared cache lines of neighbouring cells hit the same L2.
	const int logical = (blockIdx.x % kNumXCD) * perXcd + blockIdx.x / kNumXCD;
	if (logical >= nblocks) return;
	const PyrDesc& d = *b.desc;
	const int img = (int)__umulhi((unsigned)logical, ncellsM);   // logical / ncells (launch_fast checks the multiplier's range); this launch covers cells [cell0, cell0 + ncells) of every image
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
	const int th = ch + 6;
#ifndef MCS_FAST_STAGE_DMA
#define MCS_FAST_STAGE_DMA 0   // 1 (A/B, round 6): the tile by LDS-DMA (below) — a quarter fewer VALU instructions in this phase, bit-exact, and SLOWER: 0.345 against 0.337 ms (481 dword requests per cell, each wave's M0 set-up and the wait in front of the barrier)
#endif
#if MCS_FAST_STAGE_DMA
	// Staging by LDS-DMA (round 6): global_load_lds_dword moves a dword per lane from global memory straight into LDS at (uniform base + 4 * lane) — lane c of
	// the workgroup owns dword c of the tile in row-major order (a tile row = kTilePitch / 4 dwords of the image row starting at x0 - kTileX), so the only VALU
	// work left is c -> (row, dword) and the global offset: ~6 instructions per thread and trip, no LDS stores, no registers in between.  The kernel is bound by
	// VALU issue (0.92 busy) and a fifth of its VALU instructions were this copy.  A row's dwords past the image row's end are not requested (the rightmost
	// cell of a narrow level; they lie beyond the ring and the one dword the compass test reads past it).
	{
		constexpr int kRowDw = kTilePitch / 4;
		constexpr unsigned kRowM = (65536u + kRowDw - 1) / kRowDw;   // c / kRowDw = (c * kRowM) >> 16, exact for c < kTileRows * kRowDw <= 72 * 18
		static_assert((unsigned)(kTileRows * kRowDw) * (kRowM * kRowDw - 65536u) < 65536u, "row of a tile dword by multiplication");
		const int availDw = min(kRowDw, (L.w - (cell.x0 - Geo::kTileX) + 3) >> 2);   // dwords of a tile row that start inside the image row (level-ROI coordinates: x0 >= 22 > kTileX)
		const int ndwTile = kRowDw * th;
		for (int base = 0; base < ndwTile; base += kFastBS) {   // (uniform trip count: the LDS base of a wave's request is wave-uniform)
			const int c = base + tid;
			const unsigned ty = ((unsigned)c * kRowM) >> 16, kx = (unsigned)c - ty * (unsigned)kRowDw;
			if (c < ndwTile && (int)kx < availDw)
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (ty * (unsigned)stride + 4u * kx)),
				                                 (__attribute__((address_space(3))) void*)(tile + 4 * (base + (tid & ~63))), 4, 0, 0);
		}
	}
	// the score tile cleared as 16-byte words by one trip, the verdict bitmap as dwords
	{
		constexpr int kSc16 = (kScRows * kScPitch + 15) / 16;
		for (int i = tid; i < kSc16; i += kFastBS) reinterpret_cast<uint4*>(sc)[i] = uint4{0u, 0u, 0u, 0u};
	}
	if (!kOrdered) for (int i = tid; i < 2 * ch; i += kFastBS) keepBits[i] = 0;
	if (tid == 0) { runBase = 0; nSurv = 0; }
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's requests have landed (hipcc does not count LDS-DMA loads in front of a barrier)
	__syncthreads();
#else
#ifndef MCS_FAST_STAGE16
#define MCS_FAST_STAGE16 1   // 0 (A/B): the round-5 staging for every cell (two dwords per thread and trip)
#endif
	// Staging in 16-byte chunks (round 6): a tile row (kTilePitch bytes of the image row from x0 - kTileX) is ceil(kTilePitch / 16) chunks, the last one pulled back to end
	// with the row (its leading bytes then duplicate the chunk before it: same bytes to the same LDS addresses); a thread loads one chunk (one unaligned 16-byte load)
	// and stores it as four dwords.  The index arithmetic is paid per 16 bytes instead of per 4: this phase was a fifth of the kernel's VALU instructions, and the kernel is
	// bound by VALU issue.  Only for cells whose tile rows end inside the image row (cw >= kTilePitch - 26 for FAST: every cell of the usual 30-px grid with the 40-px
	// instance); the rest — a wide instance serving a level of narrow cells — take the dword loop below.
	const bool stage16 = MCS_FAST_STAGE16 && (cell.x0 - Geo::kTileX + kTilePitch <= L.w);
	if (stage16) {
		constexpr int kChunks = (kTilePitch + 15) / 16, kLast = kTilePitch - 16;   // kTilePitch is a multiple of 4, >= 16
		static_assert((kChunks & (kChunks - 1)) == 0 || kChunks == 3 || kChunks == 5 || kChunks == 6, "chunks per row");
		const int nch = kChunks * th;
		for (int c = tid; c < nch; c += kFastBS) {
			const unsigned ty = kChunks == 4 ? (unsigned)c >> 2 : kChunks == 8 ? (unsigned)c >> 3 : (unsigned)c / (unsigned)kChunks;   // (a constant divisor: multiply + shift)
			const unsigned k = (unsigned)c - ty * (unsigned)kChunks;
			const unsigned xo = min(16u * k, (unsigned)kLast);
			uint4 v;
			__builtin_memcpy(&v, src + (ty * (unsigned)stride + xo), 16);
			uint32_t* dstw = reinterpret_cast<uint32_t*>(&tile[ty * kTilePitch + xo]);
			dstw[0] = v.x; dstw[1] = v.y; dstw[2] = v.z; dstw[3] = v.w;
		}
		for (int i = tid; i < (kScRows * kScPitch + 15) / 16; i += kFastBS) reinterpret_cast<uint4*>(sc)[i] = uint4{0u, 0u, 0u, 0u};   // the score tile cleared as 16-byte words
		if (!kOrdered) for (int i = tid; i < 2 * ch; i += kFastBS) keepBits[i] = 0;
		if (tid == 0) { runBase = 0; nSurv = 0; }
		__syncthreads();
	} else {
	const int tw = cw + Geo::kTileX + 3;
	const int ndw = (tw + 3) >> 2;   // unaligned dword loads; the <= 3 bytes of over-read per row stay inside the image row
	// i / ndw by multiplication (CellInfo.rowM = ceil(2^16 / ndw)): ndw <= 17 and i < 17 * 66, so the error term i * (M*ndw - 2^16) < 2^16 and
	// (i * M) >> 16 is exact; 32-bit offsets keep the address arithmetic out of 64-bit multiplies.
	const unsigned rowM = (unsigned)cell.rowM;
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
	const int sh = ch + 2;
	for (int i = tid; i < sh * (kScPitch / 4); i += kFastBS) reinterpret_cast<uint32_t*>(sc)[i] = 0;   // score tile cleared as dwords
	if (!kOrdered) for (int i = tid; i < 2 * ch; i += kFastBS) keepBits[i] = 0;
	if (tid == 0) { runBase = 0; nSurv = 0; }
	__syncthreads();
	}
#endif

	const int t = d.fastThreshold;
	const int lane = tid & 63, wave = tid >> 6;
	// pass 1: cheap compass test on every pixel; survivors are compacted into an LDS list (order is irrelevant here) so that
	// pass 2 — the full 16-pixel arc score, ~10x the work — runs with all lanes busy instead of diverging inside each wave
	if constexpr ((P == 16 && MCS_FAST_SEG16) || AG >= 0) {
		// pass 1, 16-pixel ring, round 5: SIXTEEN adjacent pixels per thread (four fast_quick4 groups whose row dwords overlap: 14 LDS dwords instead of 20), so a
		// 31 x 31 cell is 62 lanes — ONE trip of ONE wave where four-pixel lanes took two trips of both waves —, and the per-trip overhead (index arithmetic, prefix sum
		// over the lanes' survivor counts, the atomic, the list writes) is paid once per 16 pixels: ~236 instead of ~470 wave-instructions per cell for this pass.
		// The survivors of a lane go out in a loop over its set bits (as many trips as the fullest lane has survivors: ~6 of 16).
		const int segs = (cw + 15) >> 4, nseg = segs * ch;   // segs <= 4 (cells up to 60 wide)
		const unsigned sM = segs == 1 ? 65536u : segs == 2 ? 32768u : segs == 3 ? 21846u : 16384u;   // s / segs = (s * sM) >> 16, exact for s < 4 * 60
		const v2s tt = {(short)t, (short)t};
		for (int base = 0; base < nseg; base += kFastBS) {
			const int sg = base + tid;
			uint32_t bits = 0;
			int code = 0;
			if (sg < nseg) {
				const int py = (int)(((unsigned)sg * sM) >> 16), sx = sg - py * segs;
				const uint8_t* rowc = &tile[(py + 3) * kTilePitch + 16 * sx + Geo::kTileX];
				const int valid = cw - 16 * sx;   // pixels of the segment inside the cell: >= 1
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					if (4 * g < valid) {   // (a group wholly outside the cell is not read: the tile's right margin holds only one dword past the ring)
						const uint32_t q = fast_quick4<kTilePitch, QR>(rowc + 4 * g, tt);   // verdicts of pixels 0, 1, 2, 3 in bits 0, 1, 16, 17
						bits |= ((q | (q >> 14)) & 0xFu) << (4 * g);
					}
				}
				if (valid < 16) bits &= (1u << valid) - 1u;
				code = (py << 6) | (16 * sx);
			}
			const int cnt = __popc(bits);
			const int incl = wave_incl_scan(cnt);
			const int total = __builtin_amdgcn_readlane(incl, 63);
			int wbase = 0;
			if constexpr (kOrdered) {
				// list space in (trip, wave, lane) order = the segments' row-major order: the waves' totals meet in LDS (one barrier per trip: a cell of up to 40 x 40 is one
				// trip) instead of racing for an atomic — pass 3 below then ranks the kept survivors by list position alone
				if (lane == 0) waveTot[wave] = total;
				__syncthreads();
				int before = 0, all = 0;
#pragma unroll
				for (int w = 0; w < kFastBS / 64; ++w) { const int tw_ = waveTot[w]; before += w < wave ? tw_ : 0; all += tw_; }
				wbase = listRun + before;
				listRun += all;
				if (base + kFastBS < nseg) __syncthreads();   // (waveTot is rewritten by the next trip; pass 3 reuses it behind the barriers in between)
			} else if (lane == 0 && total) wbase = atomicAdd(&nSurv, total);
			int pos = __builtin_amdgcn_readfirstlane(wbase) + incl - cnt;
			while (bits) {
				const int j = __builtin_ctz(bits);
				surv[pos++] = (unsigned short)(code + j);
				bits &= bits - 1u;
			}
		}
	} else if constexpr (P == 16) {
		// pass 1, 16-pixel ring: four adjacent pixels per thread (fast_quick4); the wave reserves list space for all of them with one prefix sum over the lanes'
		// survivor counts (DPP) and one atomic
		const int gpr = (cw + 3) >> 2, ngrp = gpr * ch;   // groups per row: gpr <= 15, g < 15 * 60: CellInfo.grpM = ceil(2^16 / gpr) is exact
		const unsigned gM = (unsigned)cell.grpM;
		const v2s tt = {(short)t, (short)t};
		for (int base = 0; base < ngrp; base += kFastBS) {
			const int g = base + tid;
			uint32_t bits = 0;
			int code = 0;
			if (g < ngrp) {
				const int py = (int)(((unsigned)g * gM) >> 16), gx = g - py * gpr;
				bits = fast_quick4<kTilePitch>(&tile[(py + 3) * kTilePitch + 4 * gx + Geo::kTileX], tt);
				const int left = cw - 4 * gx;   // pixels of the group inside the cell (the last group of a row may be partial): pixels 0, 1, 2, 3 are bits 0, 1, 16, 17
				if (left < 4) bits &= left == 3 ? 0x00010003u : left == 2 ? 0x00000003u : 0x00000001u;
				code = (py << 6) | (4 * gx);
			}
			const int cnt = __popc(bits);
			const int incl = wave_incl_scan(cnt);
			const int total = __builtin_amdgcn_readlane(incl, 63);
			int wbase = 0;
			if (lane == 0 && total) wbase = atomicAdd(&nSurv, total);
			wbase = __builtin_amdgcn_readfirstlane(wbase) + incl - cnt;
			if (bits & 0x00000001u) surv[wbase] = (unsigned short)code;
			if (bits & 0x00000002u) surv[wbase + (int)(bits & 1u)] = (unsigned short)(code + 1);
			if (bits & 0x00010000u) surv[wbase + __popc(bits & 0x3u)] = (unsigned short)(code + 2);
			if (bits & 0x00020000u) surv[wbase + __popc(bits & 0x10003u)] = (unsigned short)(code + 3);
		}
	} else {
		// p / cw by multiplication: cw <= 60 and p < 3600, so with M = ceil(2^18 / cw) the error term p * (M*cw - 2^18) < 3600 * 60 < 2^18 and (p * M) >> 18 is exact
		const int npx = cw * ch;
		const unsigned divM = (262144u + (unsigned)cw - 1u) / (unsigned)cw;
		for (int base = 0; base < npx; base += kFastBS) {
			const int p = base + tid;
			bool pass = false;
			int code = 0;
			if (p < npx) {
				const int py = (int)(((unsigned)p * divM) >> 18), px = p - py * cw;
				pass = small_ring_quick<P, kTilePitch>(&tile[(py + 3) * kTilePitch + px + Geo::kTileX], t);
				code = (py << 6) | px;
			}
			const unsigned long long bal = __ballot(pass);
			int wbase = 0;
			if (lane == 0 && bal) wbase = atomicAdd(&nSurv, __popcll(bal));
			wbase = __shfl(wbase, 0);
			if (pass) surv[wbase + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)code;
		}
	}
	__syncthreads();
	const int ns = kOrdered ? listRun : nSurv;
	if constexpr (P == 16) {
		for (int i = 2 * tid; i < ns; i += 2 * kFastBS) {
			const int pa = surv[i], pb = surv[i + 1 < ns ? i + 1 : i];
			const int ya = pa >> 6, xa = pa & 63, yb = pb >> 6, xb = pb & 63;
			const uint32_t s2 = fast_score2<kTilePitch>(&tile[(ya + 3) * kTilePitch + xa + Geo::kTileX], &tile[(yb + 3) * kTilePitch + xb + Geo::kTileX], t);
			sc[(ya + 1) * kScPitch + xa + 1] = (uint8_t)s2;
			if (i + 1 < ns) sc[(yb + 1) * kScPitch + xb + 1] = (uint8_t)(s2 >> 8);
		}
	} else
	for (int i = tid; i < ns; i += kFastBS) {
		const int p = surv[i];
		const int py = p >> 6, px = p & 63;
		const uint8_t* c = &tile[(py + 3) * kTilePitch + px + Geo::kTileX];
		if constexpr (AG >= 0) sc[(py + 1) * kScPitch + px + 1] = (uint8_t)agast_score<(AG < 0 ? 0 : AG), kTilePitch>(c, t);
		else sc[(py + 1) * kScPitch + px + 1] = (uint8_t)small_ring_score<P, kTilePitch>(c, t);
	}
	__syncthreads();

	uint32_t* slots = b.slots + (size_t)img * d.slotsPerImage + cell.slot;
	const short* mapX = b.maskMap + L.mapX;
	const short* mapY = b.maskMap + L.mapY;
	const uint8_t* mask = b.mask0 ? b.mask0 + (size_t)img * b.mask0Pitch : nullptr;
	if constexpr (AG >= 0) {
		// ---- AGAST: region suppression (cv::AGAST's nmsFlags loop) -------------------------------------------------------------------------------------------
		// corners = survivors with a score; their raster order (the order of cv::AGAST's keypoint list) from a bitmap + row prefix
		for (int i = tid; i < ns; i += kFastBS) {
			const int p = surv[i];
			if (sc[((p >> 6) + 1) * kScPitch + (p & 63) + 1]) atomicOr(&keepBits[p >> 5], 1u << (p & 31));
		}
		__syncthreads();
		if (wave == 0) {
			const int v = lane < ch ? __popc(keepBits[2 * lane]) + __popc(keepBits[2 * lane + 1]) : 0;
			const int incl = wave_incl_scan(v);
			rowOff[lane] = incl - v;
			if (lane == 63) runBase = incl;
		}
		__syncthreads();
		auto bit = [&](int code) { return (keepBits[code >> 5] >> (code & 31)) & 1u; };
		auto index = [&](int code) {   // corners before `code` in raster order
			const uint32_t w = keepBits[code >> 5], lowm = (1u << (code & 31)) - 1u;
			return rowOff[code >> 6] + (int)((code & 32) ? __popc(keepBits[(code >> 5) - 1]) + __popc(w & lowm) : __popc(w & lowm));
		};
		unsigned short* codes = reinterpret_cast<unsigned short*>(tile);   // the corners in raster order = cv::AGAST's keypoint list
		short* flags = reinterpret_cast<short*>(surv);                     // nmsFlags: -1 = a maximum, else the corner that dominates it
		for (int i = tid; i < ns; i += kFastBS) {
			const int p = surv[i];
			if (sc[((p >> 6) + 1) * kScPitch + (p & 63) + 1]) codes[index(p)] = (unsigned short)p;
		}
		__syncthreads();
		const int n = runBase;
		for (int i = tid; i < n; i += kFastBS) flags[i] = -1;
		__syncthreads();
		if (wave != 0) return;
		uint32_t* slotsA = b.slots + (size_t)img * d.slotsPerImage + cell.slot;
		const uint8_t* maskA = b.mask0 ? b.mask0 + (size_t)img * b.mask0Pitch : nullptr;
		// a maximum leaves through the mirror mask (AgastFeatureDetector::detect: KeyPointsFilter::runByPixelsMask after the suppression), in raster order
		int run = 0;
		auto emit = [&](bool keep, uint32_t inf) {
			const int px = inf & 63, py = (inf >> 6) & 63;
			if (keep && maskA) keep = maskA[(size_t)(b.maskMap + L.mapY)[cell.y0 + py] * b.mask0Stride + (b.maskMap + L.mapX)[cell.x0 + px]] != 0;
			const unsigned long long bal = __ballot(keep);
			if (keep) slotsA[run + __popcll(bal & ((1ull << lane) - 1ull))] =
				(uint32_t)(cell.x0 + px - kMinBorder) | ((uint32_t)(cell.y0 + py - kMinBorder) << 12) | (((inf >> 12) & 0xffu) << 24);
			run += __popcll(bal);
		};
		// corner i as one word: code | score << 12 | has-left << 20 | has-above << 21
		auto word = [&](int i) -> uint32_t {
			if (i >= n) return 0u;
			const int p = codes[i], py = p >> 6, px = p & 63;
			const uint32_t s0 = sc[(py + 1) * kScPitch + px + 1];
			return (uint32_t)p | (s0 << 12) | ((px > 0 ? bit(p - 1) : 0u) << 20) | ((py > 0 ? bit(p - 64) : 0u) << 21);
		};
		// The walk is wave-uniform (every lane executes the same statements on the same LDS words); only corners with a neighbour above or to the left do anything.
		// A step is a chain of dependent LDS round trips, so it is kept short: the index of the corner above is worked out by the corner's own lane beforehand, a hop of
		// the root chase fetches the link and the code of the same corner together (the code leads to the response if the hop was the last), and what the step itself
		// has just written is carried in registers instead of being read back.  (Held in registers altogether — readlane / compare-select on uniform indices, for
		// cells of up to 256 corners — the walk was slower: 1.05 against 0.94 ms for OAST_9_16 on the bench stream's images; the links as extra LDS arrays cost
		// occupancy: 1.79 ms.)
		auto respOf = [&](int q) { return (int)sc[((q >> 6) + 1) * kScPitch + (q & 63) + 1]; };
		auto root = [&](int& w) {   // follows the links from w to its maximum; returns that corner's response
			int f = flags[w], q = codes[w];
			while (f != -1) { w = f; f = flags[w]; q = codes[w]; }
			return respOf(q);
		};
		for (int base = 0; base < n; base += 64) {
			const uint32_t inf = word(base + lane);
			const int ab = (inf >> 21) & 1u ? index((int)(inf & 0xfffu) - 64) : 0;
			unsigned long long act = __ballot(((inf >> 20) & 3u) != 0u);
			while (act) {
				const int l = __builtin_ctzll(act);
				act &= act - 1ull;
				const int cur = base + l;
				const uint32_t ci = __builtin_amdgcn_readlane(inf, l);
				const int cr = (int)((ci >> 12) & 0xffu);
				int maxAbove = -1, respAbove = 0;   // nmsFlags[cur] after the check above, and that corner's response
				if (ci & (1u << 21)) {   // check above: the maximum of the block the corner above belongs to
					int w = __builtin_amdgcn_readlane(ab, l);
					const int rw = root(w);
					if (cr < rw) { flags[cur] = (short)w; maxAbove = w; respAbove = rw; }
					else flags[w] = (short)cur;
				}
				if (ci & (1u << 20)) {   // check left
					int tl = cur - 1;
					const int rt = root(tl);
					if (maxAbove == -1) {   // no maximum above
						if (tl != cur) {
							if (cr < rt) flags[cur] = (short)tl;
							else flags[tl] = (short)cur;
						}
					} else if (tl != maxAbove) {   // maximum above
						if (respAbove < rt) { flags[maxAbove] = (short)tl; flags[cur] = (short)tl; }
						else { flags[tl] = (short)maxAbove; flags[cur] = (short)maxAbove; }
					}
				}
			}
		}
		for (int base = 0; base < n; base += 64) {
			const int i = base + lane;
			emit(i < n && flags[i] == -1, word(i));
		}
		if (lane == 0) *countOut = run;
		return;
	}
	if constexpr (kOrdered) {
		// pass 3, ordered list: suppression + mask + emission in one loop.  A trip covers kFastBS consecutive list entries; the kept ones among them leave in list order:
		// a ballot per wave, the waves' counts through LDS (one barrier per trip; 113 survivors per cell on the bench stream: one trip)
		int run = 0;
		for (int base = 0; base < ns; base += kFastBS) {
			const int i = base + tid;
			bool keep = false;
			uint32_t rec = 0;
			if (i < ns) {
				const int p = surv[i];
				const int py = p >> 6, px = p & 63;
				const uint8_t* q = &sc[(py + 1) * kScPitch + px + 1];
				const int s0 = q[0];
				const int nb = max(max(max((int)q[-1], (int)q[1]), max((int)q[-kScPitch - 1], (int)q[-kScPitch])),
				                   max(max((int)q[-kScPitch + 1], (int)q[kScPitch - 1]), max((int)q[kScPitch], (int)q[kScPitch + 1])));
				keep = s0 > nb;   // strictly greater than all 8 neighbours (a score of 0 never is)
				if (keep && mask) {   // KeyPointsFilter::runByPixelsMask on the (nearest-neighbour) mask pyramid
					const int mx = mapX[cell.x0 + px], my = mapY[cell.y0 + py];
					keep = mask[(size_t)my * b.mask0Stride + mx] != 0;
				}
				rec = (uint32_t)(cell.x0 + px - kMinBorder) | ((uint32_t)(cell.y0 + py - kMinBorder) << 12) | ((uint32_t)s0 << 24);
			}
			const unsigned long long bal = __ballot(keep);
			if (lane == 0) waveTot[wave] = __popcll(bal);
			__syncthreads();
			int before = 0, all = 0;
#pragma unroll
			for (int w = 0; w < kFastBS / 64; ++w) { const int tw_ = waveTot[w]; before += w < wave ? tw_ : 0; all += tw_; }
			if (keep) slots[run + before + __popcll(bal & ((1ull << lane) - 1ull))] = rec;
			run += all;
			if (base + kFastBS < ns) __syncthreads();   // (waveTot is rewritten by the next trip)
		}
		if (tid == 0) *countOut = run;
		return;
	}
	// pass 3a: non-max suppression + mirror mask, only for the pixels that have a score at all (the compass survivors); the verdicts go
	// into a bitmap indexed by the pixel's code
	for (int i = tid; i < ns; i += kFastBS) {
		const int p = surv[i];
		const int py = p >> 6, px = p & 63;
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
	// the emission order (row-major inside the cell, the reference's) is an exclusive prefix sum over the kept-pixel counts of the <= 60 rows (two bitmap words
	// each; one wave does it) — two barriers for the whole cell
	if (wave == 0) {
		const int v = lane < ch ? __popc(keepBits[2 * lane]) + __popc(keepBits[2 * lane + 1]) : 0;
		const int incl = wave_incl_scan(v);
		rowOff[lane] = incl - v;
		if (lane == 63) runBase = incl;
	}
	__syncthreads();
	// every kept survivor writes its record at (kept pixels before it in row-major order): row prefix + kept bits below it in its own row
	for (int i = tid; i < ns; i += kFastBS) {
		const int p = surv[i];
		const uint32_t w = keepBits[p >> 5];
		if ((w >> (p & 31)) & 1u) {
			const int py = p >> 6, px = p & 63;
			const uint32_t below = (p & 32) ? __popc(keepBits[2 * py]) + __popc(w & ((1u << (p & 31)) - 1u)) : __popc(w & ((1u << (p & 31)) - 1u));
			const int s0 = sc[(py + 1) * kScPitch + px + 1];
			slots[rowOff[py] + (int)below] = (uint32_t)(cell.x0 + px - kMinBorder) | ((uint32_t)(cell.y0 + py - kMinBorder) << 12) | ((uint32_t)s0 << 24);
		}
	}
	if (tid == 0) *countOut = runBase;
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
	// block -> image by multiplication: M = ceil(2^32 / ncells) is exact while block * ncells < 2^32 (block * (M * ncells - 2^32) < 2^32)
	if ((unsigned long long)nblocks * (unsigned long long)ncells >= (1ull << 32)) { fprintf(stderr, "mcs: FAST launch too large (%d blocks)\n", nblocks); abort(); }
	const unsigned ncellsM = (unsigned)(((1ull << 32) + (unsigned long long)ncells - 1ull) / (unsigned long long)ncells);
#define MCS_FAST_LAUNCH(P)                                                                                                                      \
	do {                                                                                                                                        \
		if (cellMax <= 40) hipLaunchKernelGGL((k_fast_cells<40, MCS_FAST_BS, P>), grid, dim3(MCS_FAST_BS), 0, s, b, nimg, nblocks, perXcd, cell0, ncells, ncellsM);     \
		else hipLaunchKernelGGL((k_fast_cells<60, 256, P>), grid, dim3(256), 0, s, b, nimg, nblocks, perXcd, cell0, ncells, ncellsM);                   \
	} while (0)
	if (hd.agast >= 0) {
		// AGAST: the processed region of a cell is its view minus the ring's radius on every side (FAST: minus 3), so cells are up to 4 pixels larger
		const int B = hd.agast == 0 ? 1 : (hd.agast == 2 ? 2 : 3);
		const int cellMaxA = cellMax + 6 - 2 * B;
#define MCS_AGAST_LAUNCH(T)                                                                                                                                              \
		do {                                                                                                                                                                 \
			if (cellMaxA <= 44) hipLaunchKernelGGL((k_fast_cells<44, 128, AgastGeom<T>::P, T>), grid, dim3(128), 0, s, b, nimg, nblocks, perXcd, cell0, ncells, ncellsM);   \
			else hipLaunchKernelGGL((k_fast_cells<64, 256, AgastGeom<T>::P, T>), grid, dim3(256), 0, s, b, nimg, nblocks, perXcd, cell0, ncells, ncellsM);                    \
		} while (0)
		if (hd.agast == 0) MCS_AGAST_LAUNCH(0);
		else if (hd.agast == 1) MCS_AGAST_LAUNCH(1);
		else if (hd.agast == 2) MCS_AGAST_LAUNCH(2);
		else MCS_AGAST_LAUNCH(3);
#undef MCS_AGAST_LAUNCH
		return;
	}
	if (hd.fastRing == 16) MCS_FAST_LAUNCH(16);
	else if (hd.fastRing == 12) MCS_FAST_LAUNCH(12);
	else MCS_FAST_LAUNCH(8);
#undef MCS_FAST_LAUNCH
}

}  // namespace mcs

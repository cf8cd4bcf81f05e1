// mcs_blur.hip — E6: cv::boxFilter(level, level, -1, Size(5,5), normalize=true, BORDER_REFLECT_101) of every level
// (reference src/mdBRIEFextractorOct.cpp:1301): out = (sum of the 5x5 window + 12) / 25 in integers,
// window pixels outside the ROI taken from the reflect-101 frame of the UNBLURRED level (= reflected indices here).
// The blurred pyramid is a separate buffer, so FAST / orientation keep reading the unblurred one.
//
// HBM-bound: reads S, writes S bytes per image.  128x32-pixel output tile per 256-thread workgroup (enough bytes in
// flight per workgroup to cover HBM latency), the 132x36 input tile is staged in LDS with coalesced dword loads,
// horizontal 5-sums are formed once per input row (u16 in LDS), each thread then emits 4 adjacent pixels per row as one
// dword store.  All levels of all images go out in ONE launch (tile list per level is prefix-indexed).
#include "mcs_common.h"

namespace mcs {

constexpr int BT_W = 128, BT_H = 32;
constexpr int BI_W = BT_W + 4, BI_H = BT_H + 4;
constexpr int BI_PITCH = BI_W + 4;   // 136: multiple of 4

__device__ __forceinline__ int reflect101(int p, int len) {
	// single reflection is enough for |overshoot| < len; callers guarantee len >= 3 and overshoot <= 25 < len
	p = p < 0 ? -p : p;
	return p >= len ? 2 * (len - 1) - p : p;
}

__global__ __launch_bounds__(256) void k_blur(ExtractBuffers b, int tilesPerImage) {
	__shared__ __attribute__((aligned(16))) uint8_t in[BI_H][BI_PITCH];
	__shared__ __attribute__((aligned(16))) unsigned short hs[BI_H][BT_W];
	const PyrDesc& d = *b.desc;
	const int img = blockIdx.x / tilesPerImage;
	int t = blockIdx.x - img * tilesPerImage;
	int level = 0, tx = 0;
	for (; level < d.nlevels; ++level) {
		tx = (d.lv[level].w + BT_W - 1) / BT_W;
		const int nt = tx * ((d.lv[level].h + BT_H - 1) / BT_H);
		if (t < nt) break;
		t -= nt;
	}
	const LevelInfo& L = d.lv[level];
	const int ty0 = (t / tx) * BT_H, tx0 = (t % tx) * BT_W;
	int stride;
	const uint8_t* src = level_ptr(b, d, img, level, &stride);
	const int tid = threadIdx.x;
	// 33 dwords per input row, coalesced (global addresses may be unaligned; LDS rows are 4-byte aligned).  Rows are reflected per row; a dword that
	// lies inside the level's columns is one load whatever the tile — only the <= 2 dwords per row that straddle the left / right border gather their
	// bytes through reflected indices, and dwords wholly past column w + 1 (never part of a stored pixel's window) are zero.
	// All five dwords of a thread are requested before the first LDS store waits for one: a load -> store loop would pay the memory latency five times over.
	constexpr int kStage = (BI_H * (BI_W / 4) + 255) / 256;
	uint32_t sv[kStage];
#pragma unroll
	for (int u = 0; u < kStage; ++u) {
		const int i = u * 256 + tid;
		const int r = (int)(((unsigned)i * 1986u) >> 16), k = i - r * (BI_W / 4);   // i / 33 exactly for i < 32768 (33 * 1986 = 2^16 + 2)
		const int y = reflect101(min(ty0 + r - 2, L.h + 1), L.h);
		const int xs = tx0 - 2 + 4 * k;
		const uint8_t* row = src + (unsigned)y * (unsigned)stride;
		uint32_t v = 0;
		if (i < BI_H * (BI_W / 4)) {
			if (xs >= 0 && xs + 3 < L.w) __builtin_memcpy(&v, row + xs, 4);
			else if (xs < L.w + 2) {
#pragma unroll
				for (int e = 0; e < 4; ++e) v |= (uint32_t)row[reflect101(min(xs + e, L.w + 1), L.w)] << (8 * e);
			}
		}
		sv[u] = v;
	}
#pragma unroll
	for (int u = 0; u < kStage; ++u) {
		const int i = u * 256 + tid;
		const int r = (int)(((unsigned)i * 1986u) >> 16), k = i - r * (BI_W / 4);
		if (i < BI_H * (BI_W / 4)) *reinterpret_cast<uint32_t*>(&in[r][4 * k]) = sv[u];
	}
	__syncthreads();
	// horizontal 5-sums, 4 adjacent columns per thread from two aligned dword reads: v_sad_u8 sums the four bytes of a window in one instruction,
	// v_alignbyte slides the window (window k = bytes k..k+3 of the 8, the fifth byte is the sad's addend)
	for (int i = tid; i < BI_H * (BT_W / 4); i += 256) {
		const int r = i / (BT_W / 4), c = (i - r * (BT_W / 4)) * 4;
		const uint32_t a = *reinterpret_cast<const uint32_t*>(&in[r][c]), bq = *reinterpret_cast<const uint32_t*>(&in[r][c + 4]);
		const uint32_t s0 = __builtin_amdgcn_sad_u8(a, 0u, bq & 0xffu);
		const uint32_t s1 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(bq, a, 1), 0u, (bq >> 8) & 0xffu);
		const uint32_t s2 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(bq, a, 2), 0u, (bq >> 16) & 0xffu);
		const uint32_t s3 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(bq, a, 3), 0u, bq >> 24);
		*reinterpret_cast<uint2*>(&hs[r][c]) = make_uint2(s0 | (s1 << 16), s2 | (s3 << 16));
	}
	__syncthreads();
	// vertical: a thread owns 4 adjacent columns x 4 CONSECUTIVE rows and slides the 5-row window down (packed 16-bit adds: two columns per
	// instruction; a 5x5 sum is <= 6375).  (s + 12) / 25 == ((s + 12) * 5243) >> 17 for s + 12 <= 6387: 25 * 5243 = 2^17 + 3, so the quotient's error
	// 3q / 2^17 <= 0.006 never carries a remainder of at most 24/25 over the next integer.
	typedef unsigned short us2 __attribute__((ext_vector_type(2)));
	const int ox = (tid & 31) * 4, oy0 = (tid >> 5) * 4;
	const int x = tx0 + ox;
	if (x >= L.w) return;
	uint8_t* dst = b.blur + (size_t)img * d.pyrBytes + L.off;
	auto row2 = [&](int r, us2& lo, us2& hi) {
		const uint2 v = *reinterpret_cast<const uint2*>(&hs[r][ox]);
		lo = __builtin_bit_cast(us2, v.x); hi = __builtin_bit_cast(us2, v.y);
	};
	us2 rl[8], rh[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) row2(oy0 + k, rl[k], rh[k]);
	us2 sl = rl[0] + rl[1] + rl[2] + rl[3] + rl[4], sh = rh[0] + rh[1] + rh[2] + rh[3] + rh[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int y = ty0 + oy0 + k;
		if (y < L.h) {
			const uint32_t q0 = ((uint32_t)sl.x * 5243u + 12u * 5243u) >> 17, q1 = ((uint32_t)sl.y * 5243u + 12u * 5243u) >> 17;
			const uint32_t q2 = ((uint32_t)sh.x * 5243u + 12u * 5243u) >> 17, q3 = ((uint32_t)sh.y * 5243u + 12u * 5243u) >> 17;
			*reinterpret_cast<uint32_t*>(dst + (size_t)y * L.stride + x) = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);   // pitch is a multiple of 64: tail bytes stay in-pitch
		}
		if (k < 3) { sl = sl + rl[k + 5] - rl[k]; sh = sh + rh[k + 5] - rh[k]; }
	}
}

void launch_blur(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s) {
	int tiles = 0;
	for (int l = 0; l < hd.nlevels; ++l) tiles += ((hd.lv[l].w + BT_W - 1) / BT_W) * ((hd.lv[l].h + BT_H - 1) / BT_H);
	hipLaunchKernelGGL(k_blur, dim3(nimg * tiles), dim3(256), 0, s, b, tiles);
}

}  // namespace mcs

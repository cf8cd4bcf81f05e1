// mcs_blur.hip — E6: cv::boxFilter(level, level, -1, Size(5,5), normalize=true, BORDER_REFLECT_101) of every level
// (reference src/mdBRIEFextractorOct.cpp:1301): out = (sum of the 5x5 window + 12) / 25 in integers,
// window pixels outside the ROI taken from the reflect-101 frame of the UNBLURRED level (= reflected indices here).
// The blurred pyramid is a separate buffer, so FAST / orientation keep reading the unblurred one.
//
// HBM-bound: reads S, writes S bytes per image.  64x16-pixel output tile per 256-thread workgroup, the 68x20 input tile
// is staged in LDS, horizontal 5-sums are formed once per input row (u16 in LDS), each thread then emits 4 adjacent
// pixels as one dword store.  All levels of all images go out in ONE launch (tile list per level is prefix-indexed).
#include "mcs_common.h"

namespace mcs {

constexpr int BT_W = 64, BT_H = 16;
constexpr int BI_W = BT_W + 4, BI_H = BT_H + 4;

__device__ __forceinline__ int reflect101(int p, int len) {
	// single reflection is enough for |overshoot| < len; callers guarantee len >= 3 and overshoot <= 25 < len
	p = p < 0 ? -p : p;
	return p >= len ? 2 * (len - 1) - p : p;
}

__global__ __launch_bounds__(256) void k_blur(ExtractBuffers b, int tilesPerImage) {
	__shared__ uint8_t in[BI_H][BI_W + 4];
	__shared__ unsigned short hs[BI_H][BT_W];
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
	for (int i = tid; i < BI_H * BI_W; i += 256) {
		const int r = i / BI_W, c = i - r * BI_W;
		const int y = reflect101(min(ty0 + r - 2, L.h + 1), L.h), x = reflect101(min(tx0 + c - 2, L.w + 1), L.w);
		in[r][c] = src[(size_t)y * stride + x];
	}
	__syncthreads();
	for (int i = tid; i < BI_H * BT_W; i += 256) {
		const int r = i / BT_W, c = i - r * BT_W;
		hs[r][c] = (unsigned short)(in[r][c] + in[r][c + 1] + in[r][c + 2] + in[r][c + 3] + in[r][c + 4]);
	}
	__syncthreads();
	const int oy = tid / 16, ox = (tid % 16) * 4;
	const int y = ty0 + oy, x = tx0 + ox;
	if (y < L.h && x < L.w) {
		uint32_t packed = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int s = hs[oy][ox + i] + hs[oy + 1][ox + i] + hs[oy + 2][ox + i] + hs[oy + 3][ox + i] + hs[oy + 4][ox + i];
			packed |= (uint32_t)((s + 12) / 25) << (8 * i);
		}
		uint8_t* dst = b.blur + (size_t)img * d.pyrBytes + L.off;
		*reinterpret_cast<uint32_t*>(dst + (size_t)y * L.stride + x) = packed;   // pitch is a multiple of 64: tail bytes stay in-pitch
	}
}

void launch_blur(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s) {
	int tiles = 0;
	for (int l = 0; l < hd.nlevels; ++l) tiles += ((hd.lv[l].w + BT_W - 1) / BT_W) * ((hd.lv[l].h + BT_H - 1) / BT_H);
	hipLaunchKernelGGL(k_blur, dim3(nimg * tiles), dim3(256), 0, s, b, tiles);
}

}  // namespace mcs

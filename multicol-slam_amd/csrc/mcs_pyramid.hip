// mcs_pyramid.hip — E1: image pyramid.  Level l = fixed-point bilinear resize (cv::resize INTER_LINEAR, 8UC1) of the
// UNBLURRED level l-1 (reference src/mdBRIEFextractorOct.cpp:1158-1201; arithmetic per SURVEY Appendix A.1).
// The coefficient tables (xofs/ialpha, yofs/ibeta) are built once on the host in the same float/double steps as
// OpenCV; the kernel is pure integer.  No 25-px reflect-101 frame is materialised: every consumer that can leave the
// ROI (blur, distorted descriptor samples) reflects indices on the fly, which reads the same pixel values.
//
// HBM-bound streaming kernel: per level reads s(l-1) bytes, writes s(l) bytes.  Each thread produces 4 horizontally
// adjacent destination pixels and stores one dword (row pitch is a multiple of 64 B so the tail dword is in-pitch).
#include "mcs_common.h"

namespace mcs {

__global__ __launch_bounds__(256) void k_resize_level(ExtractBuffers b, int level) {
	const PyrDesc& d = *b.desc;
	const LevelInfo& L = d.lv[level];
	const LevelInfo& P = d.lv[level - 1];
	const int img = blockIdx.z;
	const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
	const int y = blockIdx.y * 4 + threadIdx.y;
	if (x4 >= L.w || y >= L.h) return;
	int sstride;
	const uint8_t* src = level_ptr(b, d, img, level - 1, &sstride);
	uint8_t* dst = b.pyr + (size_t)img * d.pyrBytes + L.off;
	const ResizeTap ty = b.taps[L.tabY + y];
	int sy0 = ty.ofs, sy1 = ty.ofs + 1;
	sy0 = sy0 < 0 ? 0 : (sy0 >= P.h ? P.h - 1 : sy0);   // clip(sy, 0, ssize.height)
	sy1 = sy1 < 0 ? 0 : (sy1 >= P.h ? P.h - 1 : sy1);
	const uint8_t* r0 = src + (size_t)sy0 * sstride;
	const uint8_t* r1 = src + (size_t)sy1 * sstride;
	uint32_t packed = 0;
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		int x = x4 + i;
		if (x < L.w) {
			const ResizeTap tx = b.taps[L.tabX + x];
			int sx = tx.ofs;
			int sx1 = sx + 1 < P.w ? sx + 1 : P.w - 1;   // a1 == 0 whenever sx+1 is out of range (dx >= xmax)
			int t0 = (int)r0[sx] * tx.a0 + (int)r0[sx1] * tx.a1;
			int t1 = (int)r1[sx] * tx.a0 + (int)r1[sx1] * tx.a1;
			int v = ((((int)ty.a0 * (t0 >> 4)) >> 16) + (((int)ty.a1 * (t1 >> 4)) >> 16) + 2) >> 2;
			packed |= (uint32_t)(v & 0xff) << (8 * i);
		}
	}
	*reinterpret_cast<uint32_t*>(dst + (size_t)y * L.stride + x4) = packed;
}

void launch_pyramid(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s) {
	for (int level = 1; level < hd.nlevels; ++level) {
		const LevelInfo& L = hd.lv[level];
		dim3 block(64, 4);
		dim3 grid((L.w + 255) / 256, (L.h + 3) / 4, nimg);
		hipLaunchKernelGGL(k_resize_level, grid, block, 0, s, b, level);
	}
}

}  // namespace mcs

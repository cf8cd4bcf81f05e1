// mcs_pyramid.hip — E1: image pyramid.  Level l = fixed-point bilinear resize (cv::resize INTER_LINEAR, 8UC1) of the
// UNBLURRED level l-1 (reference src/mdBRIEFextractorOct.cpp:1158-1201; arithmetic per SURVEY Appendix A.1).
// The coefficient tables (xofs/ialpha, yofs/ibeta) are built once on the host in the same float/double steps as
// OpenCV; the kernel is pure integer.  No 25-px reflect-101 frame is materialised: every consumer that can leave the
// ROI (blur, distorted descriptor samples) reflects indices on the fly, which reads the same pixel values.
//
// HBM-bound streaming kernel: per level reads s(l-1) bytes, writes s(l) bytes.  A 256-thread workgroup produces a
// 128x32 destination tile: the source footprint (<= ~160x41 px at scale 1.2) is staged in LDS with coalesced dword
// loads (row segments start 4-byte aligned in LDS; global addresses may be unaligned for level 0, which gfx950 global
// loads support), every thread then emits 4 adjacent destination pixels as one dword store.
#include "mcs_common.h"

namespace mcs {

constexpr int PT_W = 128, PT_H = 32;         // destination tile (4 rows per thread: enough bytes in flight per workgroup)
constexpr int PS_PITCH = 192;                // LDS row pitch (bytes): covers 128 * max scale 1.45 + slack
constexpr int PS_ROWS = 50;                  // source rows per tile: 32 * 1.45 + 2 + slack

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
	uint32_t v;
	__builtin_memcpy(&v, p, 4);
	return v;
}

__global__ __launch_bounds__(256) void k_resize_level(ExtractBuffers b, int level, int tilesX, int tilesY) {
	__shared__ __attribute__((aligned(16))) uint8_t src_t[PS_ROWS * PS_PITCH];
	const PyrDesc& d = *b.desc;
	const LevelInfo& L = d.lv[level];
	const LevelInfo& P = d.lv[level - 1];
	const int img = blockIdx.x / (tilesX * tilesY);
	const int t = blockIdx.x - img * (tilesX * tilesY);
	const int ty = t / tilesX, tx = t - ty * tilesX;
	const int x0 = tx * PT_W, y0 = ty * PT_H;
	const int x1 = min(x0 + PT_W, L.w) - 1, y1 = min(y0 + PT_H, L.h) - 1;   // last dst column / row of the tile
	int sstride;
	const uint8_t* src = level_ptr(b, d, img, level - 1, &sstride);
	const ResizeTap* tapX = b.taps + L.tabX;
	const ResizeTap* tapY = b.taps + L.tabY;
	// source footprint of the tile
	int sxa = tapX[x0].ofs, sxb = min(tapX[x1].ofs + 1, P.w - 1);
	int sya = tapY[y0].ofs, syb = tapY[y1].ofs + 1;
	sya = max(sya, 0); syb = min(syb, P.h - 1);
	const int nrows = syb - sya + 1;
	const int ncols = sxb - sxa + 1;
	const int ndw = (ncols + 3) >> 2;
	const int tid = threadIdx.x;
	if (nrows <= PS_ROWS && ndw * 4 <= PS_PITCH) {
		for (int i = tid; i < nrows * ndw; i += 256) {
			const int r = i / ndw, k = i - r * ndw;
			const uint8_t* gp = src + (size_t)(sya + r) * sstride + sxa + 4 * k;
			uint32_t v;
			if (sxa + 4 * k + 3 < P.w) v = load_u32_unaligned(gp);
			else {   // row tail: never read past the last pixel of the source row
				v = 0;
				for (int e = 0; e < 4; ++e) if (sxa + 4 * k + e < P.w) v |= (uint32_t)gp[e] << (8 * e);
			}
			*reinterpret_cast<uint32_t*>(&src_t[r * PS_PITCH + 4 * k]) = v;
		}
	}
	__syncthreads();
	const int lx = (tid & 31) * 4;
	const int x4 = x0 + lx;
	if (x4 >= L.w) return;
	const bool inLds = nrows <= PS_ROWS && ndw * 4 <= PS_PITCH;
	ResizeTap txv[4];
	int sxc[4], sx1c[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int x = min(x4 + i, L.w - 1);
		txv[i] = tapX[x];
		sxc[i] = txv[i].ofs;
		sx1c[i] = sxc[i] + 1 < P.w ? sxc[i] + 1 : P.w - 1;   // a1 == 0 whenever sx+1 is out of range (dx >= xmax)
	}
	uint8_t* dst = b.pyr + (size_t)img * d.pyrBytes + L.off;
	for (int ly = tid >> 5; ly < PT_H; ly += 8) {
		const int y = y0 + ly;
		if (y >= L.h) break;
		const ResizeTap tyv = tapY[y];
		int sy0 = tyv.ofs, sy1 = tyv.ofs + 1;
		sy0 = sy0 < 0 ? 0 : (sy0 >= P.h ? P.h - 1 : sy0);   // clip(sy, 0, ssize.height)
		sy1 = sy1 < 0 ? 0 : (sy1 >= P.h ? P.h - 1 : sy1);
		uint32_t packed = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			int p00, p01, p10, p11;
			if (inLds) {   // LDS indices are formed without ever stepping outside the array (no negative LDS base pointers)
				const int o0 = (sy0 - sya) * PS_PITCH, o1 = (sy1 - sya) * PS_PITCH;
				p00 = src_t[o0 + sxc[i] - sxa]; p01 = src_t[o0 + sx1c[i] - sxa];
				p10 = src_t[o1 + sxc[i] - sxa]; p11 = src_t[o1 + sx1c[i] - sxa];
			} else {       // generic path for scale factors whose footprint does not fit the LDS tile
				const uint8_t* r0 = src + (size_t)sy0 * sstride;
				const uint8_t* r1 = src + (size_t)sy1 * sstride;
				p00 = r0[sxc[i]]; p01 = r0[sx1c[i]]; p10 = r1[sxc[i]]; p11 = r1[sx1c[i]];
			}
			const int t0 = p00 * txv[i].a0 + p01 * txv[i].a1;
			const int t1 = p10 * txv[i].a0 + p11 * txv[i].a1;
			const int v = ((((int)tyv.a0 * (t0 >> 4)) >> 16) + (((int)tyv.a1 * (t1 >> 4)) >> 16) + 2) >> 2;
			packed |= (x4 + i < L.w) ? (uint32_t)(v & 0xff) << (8 * i) : 0u;
		}
		*reinterpret_cast<uint32_t*>(dst + (size_t)y * L.stride + x4) = packed;   // row pitch is a multiple of 64: the tail dword stays in-pitch
	}
}

void launch_pyramid(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s, int level0, int level1) {   // levels [level0, level1), level0 >= 1
	for (int level = level0 < 1 ? 1 : level0; level < hd.nlevels && level < level1; ++level) {
		const LevelInfo& L = hd.lv[level];
		const int tilesX = (L.w + PT_W - 1) / PT_W, tilesY = (L.h + PT_H - 1) / PT_H;
		hipLaunchKernelGGL(k_resize_level, dim3(nimg * tilesX * tilesY), dim3(256), 0, s, b, level, tilesX, tilesY);
	}
}

}  // namespace mcs

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
	// the thread's own column taps (4 adjacent destination columns) are requested first so that they arrive with the tile
	const int lx = (tid & 31) * 4;
	const int x4 = x0 + lx;
	ResizeTap txv[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) txv[i] = tapX[min(x4 + i, L.w - 1)];
	const bool inLds = nrows <= PS_ROWS && ndw * 4 + 4 <= PS_PITCH;   // + 4: the right neighbour of the last column is read (with weight 0) even past the footprint
	if (inLds) {
		// i / ndw by multiplication: ndw <= 47 and i < 50 * 47, so with M = ceil(2^18 / ndw) the error term i * (M*ndw - 2^18) < i * ndw < 2^18 and (i * M) >> 18 is exact
		const unsigned rowM = (262144u + (unsigned)ndw - 1u) / (unsigned)ndw;
		// four dwords per thread are requested before the first LDS store waits (a load -> store loop pays the memory latency once per trip)
		const int ndwTile = nrows * ndw;
		for (int base = 0; base < ndwTile; base += 4 * 256) {
			uint32_t sv[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const int i = base + u * 256 + tid;
				const int r = (int)(((unsigned)i * rowM) >> 18), k = i - r * ndw;
				const uint8_t* gp = src + ((unsigned)(sya + r) * (unsigned)sstride + (unsigned)(sxa + 4 * k));
				uint32_t v = 0;
				if (i < ndwTile) {
					if (sxa + 4 * k + 3 < P.w) v = load_u32_unaligned(gp);
					else   // row tail: never read past the last pixel of the source row
						for (int e = 0; e < 4; ++e) if (sxa + 4 * k + e < P.w) v |= (uint32_t)gp[e] << (8 * e);
				}
				sv[u] = v;
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const int i = base + u * 256 + tid;
				const int r = (int)(((unsigned)i * rowM) >> 18), k = i - r * ndw;
				if (i < ndwTile) *reinterpret_cast<uint32_t*>(&src_t[r * PS_PITCH + 4 * k]) = sv[u];
			}
		}
	}
	__syncthreads();
	if (x4 >= L.w) return;
	uint8_t* dst = b.pyr + (size_t)img * d.pyrBytes + L.off;
	// out = ((b0 * ((p00*a0 + p01*a1) >> 4)) >> 16) + ((b1 * ((p10*a0 + p11*a1) >> 4)) >> 16) + 2) >> 2, all in int32 like the reference's fixed-point path
	uint32_t live = 0;   // bytes of the thread's dword that are pixels of the level (the rest is row padding, stored as 0)
#pragma unroll
	for (int i = 0; i < 4; ++i) live |= (x4 + i < L.w) ? 0xffu << (8 * i) : 0u;
	auto emit = [&](int y, const ResizeTap tyv, auto&& px) {
		uint32_t packed = 0;
		int p00[4], p01[4], p10[4], p11[4];
#pragma unroll
		for (int i = 0; i < 4; ++i) px(i, p00[i], p01[i], p10[i], p11[i]);   // all sixteen reads in flight before the first multiply waits
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int t0 = p00[i] * txv[i].a0 + p01[i] * txv[i].a1;
			const int t1 = p10[i] * txv[i].a0 + p11[i] * txv[i].a1;
			const int v = ((((int)tyv.a0 * (t0 >> 4)) >> 16) + (((int)tyv.a1 * (t1 >> 4)) >> 16) + 2) >> 2;
			packed |= (uint32_t)(v & 0xff) << (8 * i);
		}
		*reinterpret_cast<uint32_t*>(dst + (unsigned)y * (unsigned)L.stride + (unsigned)x4) = packed & live;   // row pitch is a multiple of 64: the tail dword stays in-pitch
	};
	if (inLds) {
		// LDS path (every shipped scale factor): the right neighbour is always column + 1 — where that is past the source row its weight a1 is 0
		// (dx >= xmax in the reference's table), so whatever the tile holds there is multiplied away
		int col[4], col1[4];
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			col[i] = txv[i].ofs - sxa;
			col1[i] = col[i] + 1;
			asm volatile("" : "+v"(col1[i]));   // keeps the two reads byte-sized: fused into one 16-bit LDS read they land on odd addresses
		}
#pragma unroll 1   // rolled: unrolled, the four rows' reads cost ~30 registers and two resident workgroups per CU (0.29 vs 0.235 ms)
		for (int j = 0; j < PT_H / 8; ++j) {
			const int y = y0 + (tid >> 5) + 8 * j;
			if (y >= L.h) break;
			const ResizeTap tyv = tapY[y];
			const int sy0 = min(max((int)tyv.ofs, 0), P.h - 1), sy1 = min(max((int)tyv.ofs + 1, 0), P.h - 1);   // clip(sy, 0, ssize.height)
			const uint8_t* r0 = &src_t[(sy0 - sya) * PS_PITCH];
			const uint8_t* r1 = &src_t[(sy1 - sya) * PS_PITCH];
			emit(y, tyv, [&](int i, int& p00, int& p01, int& p10, int& p11) {
				p00 = r0[col[i]]; p01 = r0[col1[i]]; p10 = r1[col[i]]; p11 = r1[col1[i]];
			});
		}
	} else {
		// generic path for scale factors whose footprint does not fit the LDS tile: straight from global memory
#pragma unroll 1   // rolled: unrolled, the four rows' reads cost ~30 registers and two resident workgroups per CU (0.29 vs 0.235 ms)
		for (int j = 0; j < PT_H / 8; ++j) {
			const int y = y0 + (tid >> 5) + 8 * j;
			if (y >= L.h) break;
			const ResizeTap tyv = tapY[y];
			const int sy0 = min(max((int)tyv.ofs, 0), P.h - 1), sy1 = min(max((int)tyv.ofs + 1, 0), P.h - 1);
			const uint8_t* r0 = src + (size_t)sy0 * sstride;
			const uint8_t* r1 = src + (size_t)sy1 * sstride;
			emit(y, tyv, [&](int i, int& p00, int& p01, int& p10, int& p11) {
				const int sx = txv[i].ofs, sx1 = sx + 1 < P.w ? sx + 1 : P.w - 1;
				p00 = r0[sx]; p01 = r0[sx1]; p10 = r1[sx]; p11 = r1[sx1];
			});
		}
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

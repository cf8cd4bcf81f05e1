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
#include <type_traits>

namespace mcs {

constexpr int PT_W = 128, PT_H = 32;         // destination tile (4 rows per thread: enough bytes in flight per workgroup)
constexpr int PS_PITCH = 192;                // LDS row pitch (bytes): covers 128 * max scale 1.45 + slack
constexpr int PS_ROWS = 50;                  // source rows per tile: 32 * 1.45 + 2 + slack

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
	uint32_t v;
	__builtin_memcpy(&v, p, 4);
	return v;
}

__attribute__((amdgpu_waves_per_eu(8, 8)))   // 64 registers: eight workgroups per CU (the compiler's own choice: 67, seven)
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

// ---- round 4: column-marching resize, no LDS, no barrier ------------------------------------------------------------------------------------------
// A THREAD owns 4 adjacent destination columns of one image and streams down the SOURCE rows its block of destination rows reads: per source row one
// unaligned 8-byte load (the <= 8 source bytes its four columns touch), per column one v_perm_b32 (the two neighbours as a 16-bit pair) and one
// v_dot2_u32_u16 against the packed coefficients (a0 | a1 << 16) — the horizontal interpolation is done ONCE per source row (the tile kernel redid both rows for
// every destination row).  A destination row is emitted when its second source row arrives: two v_mul_hi_u32 per pixel against b << 16 (= (b * (T >> 4)) >> 16
// exactly), add, shift, pack.  A WAVE holds 64 consecutive (image, column group) items of one block of destination rows, so the row bookkeeping (taps, clipping,
// which rows to emit) is scalar; the row taps sit one per lane and are read with v_readlane.  Same integer arithmetic as above (reference:
// src/mdBRIEFextractorOct.cpp:1158-1201 -> cv::resize INTER_LINEAR).  Serves every level whose four-column source span fits 8 bytes
// (LevelInfo.colsOk: every scale factor up to 2); k_resize_level stays for the rest.
constexpr int kColsAhead = 3;      // source rows requested ahead of the one being interpolated (ring of 4)

__global__ __launch_bounds__(256) void k_resize_cols(ExtractBuffers b, int level, int wavesX, int rows, int ncg, int nimg, int totalWaves) {
	const PyrDesc& d = *b.desc;
	const LevelInfo& L = d.lv[level];
	const LevelInfo& P = d.lv[level - 1];
	const int w = L.w, h = L.h, sw = P.w, sh = P.h, dstride = L.stride, pyrBytes = d.pyrBytes, loff = L.off;   // read before the first store
	const ResizeTap* tapX = b.taps + L.tabX;
	const ResizeTap* tapY = b.taps + L.tabY;
	const int blk = (blockIdx.x & (kNumXCD - 1)) * (gridDim.x / kNumXCD) + (blockIdx.x / kNumXCD);   // XCD-contiguous wave order (k_blur)
	const int wave = __builtin_amdgcn_readfirstlane(blk * 4 + (threadIdx.x >> 6));
	if (wave >= totalWaves) return;
	const int rb = wave / wavesX, wx = wave - rb * wavesX;
	const int y0 = rb * rows;
	const int nrows = min(rows, h - y0);
	const int lane = threadIdx.x & 63;
	const int items = nimg * ncg;
	const int item0 = wx * 64;
	const int item = min(item0 + lane, items - 1);   // lanes past the last item repeat it: the same bytes to the same addresses
	const int img = item / ncg, cg = item - img * ncg;
	const int img0 = item0 / ncg;
	const int x = cg * 4;
	int sstride;
	const uint8_t* src = level_ptr(b, d, img0, level - 1, &sstride);
	const size_t simg = level == 1 ? b.img0Pitch : (size_t)pyrBytes;
	uint8_t* dst = b.pyr + (size_t)img0 * pyrBytes + loff;
	// column taps: the thread's 8 source bytes start at A; column i reads bytes ofs_i - A and ofs_i + 1 - A of them (the right neighbour past the row has weight 0)
	ResizeTap tx[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) tx[i] = tapX[min(x + i, w - 1)];
	const int A = min((int)tx[0].ofs, sw - 8);
	uint32_t sel[4], wt[4], liveMask = 0;
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const uint32_t c0 = (uint32_t)(tx[i].ofs - A) & 7u, c1 = (uint32_t)(min(tx[i].ofs + 1, sw - 1) - A) & 7u;
		sel[i] = c0 | 0x0c00u | (c1 << 16) | 0x0c000000u;
		wt[i] = (uint32_t)(unsigned short)tx[i].a0 | ((uint32_t)(unsigned short)tx[i].a1 << 16);
		liveMask |= (x + i < w) ? 0xffu << (8 * i) : 0u;   // bytes past the level's width are row padding, stored as 0
	}
	uint32_t voffS = (uint32_t)((size_t)(img - img0) * simg) + (uint32_t)A;
	uint32_t voffD = (uint32_t)((size_t)(img - img0) * (size_t)pyrBytes) + (uint32_t)x;
	// row taps of the block, one per lane
	uint32_t tyA, tyB;
	{
		const uint2 t = *reinterpret_cast<const uint2*>(&tapY[min(y0 + lane, h - 1)]);
		tyA = t.x; tyB = t.y;   // ofs | a0 << 16,  a1 | pad << 16
	}
	auto clipRow = [&](int r) { return min(max(r, 0), sh - 1); };
	const int rs = clipRow((int)(short)(__builtin_amdgcn_readlane(tyA, 0) & 0xffff));
	const int re = clipRow((int)(short)(__builtin_amdgcn_readlane(tyA, nrows - 1) & 0xffff) + 1);
	auto loadRow = [&](int r) -> uint2 {
		asm volatile("" : "+v"(voffS));   // opaque per call: SGPR row base + 32-bit lane offset (k_blur)
		uint2 v;
		__builtin_memcpy(&v, src + (size_t)((unsigned)min(r, re) * (unsigned)sstride) + voffS, 8);
		return v;
	};
	uint2 raw[4];
#pragma unroll
	for (int k = 0; k < kColsAhead; ++k) raw[k] = loadRow(rs + k);
	uint32_t T[2][4];
#pragma unroll
	for (int i = 0; i < 4; ++i) T[0][i] = T[1][i] = 0;
	int yi = 0;                      // next destination row of the block
	int sy = (int)(short)(__builtin_amdgcn_readlane(tyA, 0) & 0xffff);
	auto emit = [&](const uint32_t* T0, const uint32_t* T1) {
		const uint32_t b0 = __builtin_amdgcn_readlane(tyA, yi) & 0xffff0000u;          // a0 << 16
		const uint32_t b1 = __builtin_amdgcn_readlane(tyB, yi) << 16;                  // a1 << 16
		uint32_t v[4];
#pragma unroll
		for (int i = 0; i < 4; ++i) v[i] = (__umulhi(T0[i], b0) + __umulhi(T1[i], b1) + 2u) >> 2;
		const uint32_t out = (v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24)) & liveMask;
		asm volatile("" : "+v"(voffD));
		*reinterpret_cast<uint32_t*>(dst + (size_t)((unsigned)(y0 + yi) * (unsigned)dstride) + voffD) = out;   // row pitch is a multiple of 64: the tail dword stays in-pitch
	};
	// one source row: request row r + 3, interpolate row r horizontally into T[K & 1], emit the destination rows whose second source row it is
	auto body = [&](int r, auto kc) {
		constexpr int K = decltype(kc)::value;
		raw[(K + kColsAhead) % 4] = loadRow(r + kColsAhead);
		uint32_t* cur = T[K & 1];
		const uint32_t* prev = T[(K & 1) ^ 1];
		typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#pragma unroll
		for (int i = 0; i < 4; ++i)
			cur[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(raw[K].y, raw[K].x, sel[i])), __builtin_bit_cast(us2, wt[i]), 0u, false) >> 4;
		while (yi < nrows && clipRow(sy + 1) == r) {
			if (clipRow(sy) == r) emit(cur, cur);       // both source rows clipped to the same one (first / last row of the level)
			else emit(prev, cur);
			++yi;
			sy = (int)(short)(__builtin_amdgcn_readlane(tyA, min(yi, nrows - 1)) & 0xffff);
		}
	};
	int r = rs;
	for (; r + 3 <= re; r += 4) {
		body(r, std::integral_constant<int, 0>()); body(r + 1, std::integral_constant<int, 1>());
		body(r + 2, std::integral_constant<int, 2>()); body(r + 3, std::integral_constant<int, 3>());
	}
	if (r <= re) body(r, std::integral_constant<int, 0>());
	if (r + 1 <= re) body(r + 1, std::integral_constant<int, 1>());
	if (r + 2 <= re) body(r + 2, std::integral_constant<int, 2>());
}

// ---- the whole chain in one launch ------------------------------------------------------------------------------------------------------------------
// Seven dependent launches, the small levels latency-bound and each waiting for the previous one's last workgroup, were the longest stretch of the
// overlapped step (0.7 ms beside FAST for 0.22 ms of work).  k_resize_chain gives one workgroup a 64 x 16 tile of the LAST level and lets it compute, level by
// level, the region of every level below that the tile depends on: the level-0 footprint is staged in LDS once, level l is computed from the LDS copy of level
// l - 1 into the other LDS buffer, and written to memory for the part the workgroup OWNS.  Ownership is the tile grid of the last level pushed down through the
// tap tables (boundary of level l - 1 = tap offset of the boundary of level l, columns rounded down to a multiple of 4 so that every stored dword has one
// owner); the region a workgroup computes is the hull of what its next level needs and what it owns.  Same integer arithmetic per pixel as k_resize_level.
// Overlap between neighbouring workgroups' regions (about two pixels per level and side) is computed twice: ~25 % more arithmetic, no extra memory traffic.
constexpr int FT_W = 64, FT_H = 16;              // tile of the last level
constexpr int FA_PITCH = 288, FA_ROWS = 72;      // LDS buffer of the even levels (level 0 footprint the largest, 267 x 71 at 1920 x 1080): 20 736 B
constexpr int FB_PITCH = 232, FB_ROWS = 60;      // odd levels (222 x 59): 13 920 B
constexpr int FUSED_MAX_W = 256;                 // a computed region is at most 64 column groups of 4 pixels wide
constexpr int kChainLevels = 8;                  // pyramids of up to 8 levels (the level loops are unrolled: the regions stay in scalar registers)

struct ChainRegion { int xa, xb, ya, yb; int oxs, oxe, oys, oye; };   // computed region (inclusive), owned region [oxs, oxe) x [oys, oye)

// regions of levels top .. 0 for tile (tx, ty) of level `top` (host: the table the kernel reads, [tile][kChainLevels]; evaluated on the device this is a chain of
// ~30 dependent memory round trips per workgroup — measured 0.6 ms for the launch)
static void chain_regions(const PyrDesc& d, const ResizeTap* taps, int top, int tx, int ty, int tilesX, int tilesY, ChainRegion* R) {
	const LevelInfo& T = d.lv[top];
	ChainRegion r;
	r.xa = tx * FT_W; r.xb = (tx == tilesX - 1 ? T.w : r.xa + FT_W) - 1;
	r.ya = ty * FT_H; r.yb = (ty == tilesY - 1 ? T.h : r.ya + FT_H) - 1;
	r.oxs = r.xa; r.oxe = tx == tilesX - 1 ? (T.w + 3) & ~3 : r.xa + FT_W;
	r.oys = r.ya; r.oye = r.yb + 1;
	R[top] = r;
	for (int l = top; l >= 1; --l) {
		const LevelInfo& L = d.lv[l];
		const LevelInfo& P = d.lv[l - 1];
		const ResizeTap* tapX = taps + L.tabX;
		const ResizeTap* tapY = taps + L.tabY;
		const ChainRegion& c = R[l];
		ChainRegion n;
		auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
		// what level l's region reads
		n.xa = tapX[c.xa].ofs; n.xb = clampi(tapX[c.xb < L.w ? c.xb : L.w - 1].ofs + 1, 0, P.w - 1);
		n.ya = clampi(tapY[c.ya].ofs, 0, P.h - 1); n.yb = clampi(tapY[c.yb].ofs + 1, 0, P.h - 1);
		// what the workgroup owns of level l - 1
		n.oxs = tx == 0 ? 0 : (tapX[c.oxs].ofs & ~3);
		n.oxe = tx == tilesX - 1 ? (P.w + 3) & ~3 : (tapX[c.oxe].ofs & ~3);
		n.oys = ty == 0 ? 0 : clampi(tapY[c.oys].ofs, 0, P.h - 1);
		n.oye = ty == tilesY - 1 ? P.h : clampi(tapY[c.oye].ofs, 0, P.h - 1);
		if (l - 1 >= 1) {   // level 0 is the input: nothing to own, and its footprint need not start on a dword
			if (n.oxs < n.xa) n.xa = n.oxs;
			if (n.oxe - 1 > n.xb) n.xb = (n.oxe - 1 < P.w - 1) ? n.oxe - 1 : P.w - 1;
			if (n.oys < n.ya) n.ya = n.oys;
			if (n.oye - 1 > n.yb) n.yb = n.oye - 1;
			n.xa &= ~3;
		}
		R[l - 1] = n;
	}
}

// The kernel's region table, [tile][kChainLevels] x 8 ints; false if the fused kernel cannot serve this geometry (some region of some tile outside the LDS buffers)
bool pyramid_chain_table(const PyrDesc& d, const ResizeTap* taps, std::vector<int>* table) {
	const int top = d.nlevels - 1;
	if (top < 1 || top >= kChainLevels) return false;
	const int tilesX = (d.lv[top].w + FT_W - 1) / FT_W, tilesY = (d.lv[top].h + FT_H - 1) / FT_H;
	table->assign((size_t)tilesX * tilesY * kChainLevels * 8, 0);
	for (int ty = 0; ty < tilesY; ++ty)
		for (int tx = 0; tx < tilesX; ++tx) {
			ChainRegion* R = reinterpret_cast<ChainRegion*>(table->data() + (size_t)(ty * tilesX + tx) * kChainLevels * 8);
			chain_regions(d, taps, top, tx, ty, tilesX, tilesY, R);
			for (int l = 0; l <= top; ++l) {
				const int w = R[l].xb - R[l].xa + 1, h = R[l].yb - R[l].ya + 1;
				const int pitch = (l & 1) ? FB_PITCH : FA_PITCH, rows = (l & 1) ? FB_ROWS : FA_ROWS;
				if (w < 1 || h < 1 || w + 8 > pitch || h > rows) return false;   // + 8: dword-rounded width and the weight-0 right neighbour
				if (l >= 1 && (w > FUSED_MAX_W || h > FB_ROWS)) return false;
				if (l >= 1 && (R[l].oxs < R[l].xa || R[l].oys < R[l].ya || R[l].oye - 1 > R[l].yb || R[l].oxs > R[l].oxe || R[l].oys > R[l].oye)) return false;
				if (l >= 1 && ((R[l].oxe - 1 < d.lv[l].w - 1 ? R[l].oxe - 1 : d.lv[l].w - 1) > R[l].xb)) return false;
			}
		}
	return true;
}

__global__ __launch_bounds__(256) void k_resize_chain(ExtractBuffers b, int tilesX, int tilesY) {
	__shared__ __attribute__((aligned(16))) uint8_t bufA[FA_ROWS * FA_PITCH];
	__shared__ __attribute__((aligned(16))) uint8_t bufB[FB_ROWS * FB_PITCH];
	__shared__ ChainRegion R[kChainLevels];
	__shared__ ResizeTap tyl[kChainLevels - 1][FB_ROWS];   // row taps of the regions of levels 1..: fetched per row they cost a memory round trip per trip of the row loop
	const PyrDesc& d = *b.desc;
	const int top = d.nlevels - 1;
	const int img = blockIdx.x / (tilesX * tilesY);
	const int t = blockIdx.x - img * (tilesX * tilesY);
	const int tid = threadIdx.x;
	if (tid < kChainLevels * 8) reinterpret_cast<int*>(R)[tid] = reinterpret_cast<const int*>(b.taps + d.chainRegOff)[(size_t)t * (kChainLevels * 8) + tid];
	__syncthreads();
	for (int l = 1; l <= top; ++l)
		if (tid <= R[l].yb - R[l].ya) tyl[l - 1][tid] = (b.taps + d.lv[l].tabY)[R[l].ya + tid];
	// level 0 footprint -> bufA (dword loads, four per thread in flight; row tails byte-wise)
	{
		int sstride;
		const uint8_t* src = level_ptr(b, d, img, 0, &sstride);
		const LevelInfo& P = d.lv[0];
		const int sxa = R[0].xa, sya = R[0].ya, nrows = R[0].yb - R[0].ya + 1, ndw = (R[0].xb - R[0].xa + 1 + 3) >> 2;
		const unsigned rowM = (1048576u + (unsigned)ndw - 1u) / (unsigned)ndw;   // i / ndw by multiplication: i < 72 * 72, ndw <= 72: i * (M * ndw - 2^20) < i * ndw < 2^20
		const int ndwTile = nrows * ndw;
		for (int base = 0; base < ndwTile; base += 4 * 256) {
			uint32_t sv[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const int i = base + u * 256 + tid;
				const int r = (int)(((unsigned long long)(unsigned)i * rowM) >> 20), k = i - r * ndw;
				const uint8_t* gp = src + ((unsigned)(sya + r) * (unsigned)sstride + (unsigned)(sxa + 4 * k));
				uint32_t v = 0;
				if (i < ndwTile) {
					if (sxa + 4 * k + 3 < P.w) v = load_u32_unaligned(gp);
					else
						for (int e = 0; e < 4; ++e) if (sxa + 4 * k + e < P.w) v |= (uint32_t)gp[e] << (8 * e);
				}
				sv[u] = v;
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const int i = base + u * 256 + tid;
				const int r = (int)(((unsigned long long)(unsigned)i * rowM) >> 20), k = i - r * ndw;
				if (i < ndwTile) *reinterpret_cast<uint32_t*>(&bufA[r * FA_PITCH + 4 * k]) = sv[u];
			}
		}
	}
	__syncthreads();
	const int cg = tid & 63, rs = tid >> 6;   // column group (4 pixels) and row slot of the thread
	ResizeTap txn[4];   // the thread's column taps of the next level, requested a phase ahead
#pragma unroll
	for (int i = 0; i < 4; ++i) txn[i] = (b.taps + d.lv[1].tabX)[min(R[1].xa + 4 * cg + i, d.lv[1].w - 1)];
	for (int l = 1; l <= top; ++l) {
		const LevelInfo& L = d.lv[l];
		const LevelInfo& P = d.lv[l - 1];
		const ChainRegion c = R[l], sr = R[l - 1];
		const uint8_t* sbuf = (l & 1) ? bufA : bufB;   // level l - 1
		uint8_t* dbuf = (l & 1) ? bufB : bufA;
		const int sp = (l & 1) ? FA_PITCH : FB_PITCH, dp = (l & 1) ? FB_PITCH : FA_PITCH;
		const int x4 = c.xa + 4 * cg;
		ResizeTap txv[4];
#pragma unroll
		for (int i = 0; i < 4; ++i) txv[i] = txn[i];
		if (l < top) {
			const LevelInfo& N = d.lv[l + 1];
#pragma unroll
			for (int i = 0; i < 4; ++i) txn[i] = (b.taps + N.tabX)[min(R[l + 1].xa + 4 * cg + i, N.w - 1)];
		}
		if (x4 <= c.xb) {
			int col[4], col1[4];
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				col[i] = txv[i].ofs - sr.xa;
				col1[i] = col[i] + 1;
				asm volatile("" : "+v"(col1[i]));   // byte-sized LDS reads (k_resize_level)
			}
			uint32_t live = 0;
#pragma unroll
			for (int i = 0; i < 4; ++i) live |= (x4 + i < L.w) ? 0xffu << (8 * i) : 0u;
			uint8_t* dst = b.pyr + (size_t)img * d.pyrBytes + L.off;
			const bool ownX = x4 >= c.oxs && x4 < c.oxe;
#pragma unroll 1
			for (int y = c.ya + rs; y <= c.yb; y += 4) {
				const ResizeTap tyv = tyl[l - 1][y - c.ya];
				const int sy0 = min(max((int)tyv.ofs, 0), P.h - 1), sy1 = min(max((int)tyv.ofs + 1, 0), P.h - 1);
				const uint8_t* r0 = &sbuf[(sy0 - sr.ya) * sp];
				const uint8_t* r1 = &sbuf[(sy1 - sr.ya) * sp];
				int p00[4], p01[4], p10[4], p11[4];
#pragma unroll
				for (int i = 0; i < 4; ++i) { p00[i] = r0[col[i]]; p01[i] = r0[col1[i]]; p10[i] = r1[col[i]]; p11[i] = r1[col1[i]]; }
				uint32_t packed = 0;
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int t0 = p00[i] * txv[i].a0 + p01[i] * txv[i].a1;
					const int t1 = p10[i] * txv[i].a0 + p11[i] * txv[i].a1;
					const int v = ((((int)tyv.a0 * (t0 >> 4)) >> 16) + (((int)tyv.a1 * (t1 >> 4)) >> 16) + 2) >> 2;
					packed |= (uint32_t)(v & 0xff) << (8 * i);
				}
				packed &= live;
				if (l < top) *reinterpret_cast<uint32_t*>(&dbuf[(y - c.ya) * dp + 4 * cg]) = packed;
				if (ownX && y >= c.oys && y < c.oye) *reinterpret_cast<uint32_t*>(dst + (unsigned)y * (unsigned)L.stride + (unsigned)x4) = packed;
			}
		}
		__syncthreads();
	}
}

void launch_pyramid(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s, int level0, int level1) {   // levels [level0, level1), level0 >= 1
	if (level0 <= 1 && level1 >= hd.nlevels && hd.chainFits && hd.nlevels > 1) {
		const int top = hd.nlevels - 1;
		const int tilesX = (hd.lv[top].w + FT_W - 1) / FT_W, tilesY = (hd.lv[top].h + FT_H - 1) / FT_H;
		hipLaunchKernelGGL(k_resize_chain, dim3(nimg * tilesX * tilesY), dim3(256), 0, s, b, tilesX, tilesY);
		return;
	}
	static const bool tileOnly = getenv("MCS_PYR_TILES") != nullptr;   // A/B: the LDS tile kernel for every level
	for (int level = level0 < 1 ? 1 : level0; level < hd.nlevels && level < level1; ++level) {
		const LevelInfo& L = hd.lv[level];
		if (L.colsOk && !tileOnly) {
			const int ncg = (L.w + 3) / 4, wavesX = (nimg * ncg + 63) / 64;
			// rows per thread: 32 where that still gives the chip a few thousand waves, fewer on the small levels (each block re-reads ~2 source rows)
			int nrb = (L.h + 31) / 32;
			while (wavesX * nrb < 4096 && (L.h + nrb - 1) / nrb > 8) ++nrb;
			const int rows = (L.h + nrb - 1) / nrb;
			const int waves = wavesX * ((L.h + rows - 1) / rows);
			hipLaunchKernelGGL(k_resize_cols, dim3(((waves + 3) / 4 + kNumXCD - 1) / kNumXCD * kNumXCD), dim3(256), 0, s, b, level, wavesX, rows, ncg, nimg, waves);
			continue;
		}
		const int tilesX = (L.w + PT_W - 1) / PT_W, tilesY = (L.h + PT_H - 1) / PT_H;
		hipLaunchKernelGGL(k_resize_level, dim3(nimg * tilesX * tilesY), dim3(256), 0, s, b, level, tilesX, tilesY);
	}
}

}  // namespace mcs

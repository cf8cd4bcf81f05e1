// mcs_blur.hip — E6: cv::boxFilter(level, level, -1, Size(5,5), normalize=true, BORDER_REFLECT_101) of every level
// (reference src/mdBRIEFextractorOct.cpp:1301): out = (sum of the 5x5 window + 12) / 25 in integers,
// window pixels outside the ROI taken from the reflect-101 frame of the UNBLURRED level (= reflected indices here).
// The blurred pyramid is a separate buffer, so FAST / orientation keep reading the unblurred one.
//
// Round 4: no LDS, no barrier.  A THREAD owns 4 adjacent columns of one image and marches down a block of rows; per row it loads the 8 bytes
// x-2 .. x+5 (one unaligned global_load_dwordx2 — neighbouring lanes overlap by 4 bytes, served by the vector L1), forms the four horizontal 5-sums with
// v_dot4_u32_u8 against 0/1 byte masks (two per sum, no byte extraction), slides the 5-row vertical window with plain adds, divides by 25 with one 24-bit
// multiply per pixel whose quotient lands in the top byte, and stores one dword.  The left / right reflect-101 columns are two v_perm_b32 per row with
// per-thread selectors (identity in the interior), so there is no border branch.  A WAVE holds 64 consecutive (image, column group) items of ONE row block of
// one level: the batch's images sit side by side in the item space, so even the small levels fill the lanes, and everything about the row (reflection,
// source / destination row offsets, the loop) is scalar.  ~8 lane-instructions per pixel (the LDS two-pass tile kernel it replaces: 29.5).
#include "mcs_common.h"
#include <type_traits>

namespace mcs {

constexpr int kBlurRows = 32;      // rows per thread (+ 4 halo rows: 12.5 % more loads and horizontal sums)
constexpr int kBlurAhead = 4;      // rows requested ahead of the one being summed

struct BlurLevel { int waveBase, wavesX, rows, ncg; };
struct BlurPlan { int nlevels, totalWaves; BlurLevel lv[MCS_MAX_LEVELS]; };

__device__ __forceinline__ int reflect101(int p, int len) {
	// single reflection is enough for |overshoot| < len; callers guarantee len >= 3 and overshoot <= 25 < len
	p = p < 0 ? -p : p;
	return p >= len ? 2 * (len - 1) - p : p;
}

__global__ __launch_bounds__(256) void k_blur(ExtractBuffers b, BlurPlan plan, int nimg) {
	const PyrDesc& d = *b.desc;
	// XCD-contiguous: workgroup b runs on XCD b % 8 (observed, for speed only); XCD k takes the k-th eighth of the wave order, so that neighbouring row segments
	// (which share cache lines at their ends) meet in one L2
	const int blk = (blockIdx.x & (kNumXCD - 1)) * (gridDim.x / kNumXCD) + (blockIdx.x / kNumXCD);
	const int wave = __builtin_amdgcn_readfirstlane(blk * 4 + (threadIdx.x >> 6));
	if (wave >= plan.totalWaves) return;
	int level = 0;
	while (level + 1 < plan.nlevels && wave >= plan.lv[level + 1].waveBase) ++level;
	const BlurLevel pl = plan.lv[level];
	const LevelInfo& L = d.lv[level];
	const int w = L.w, h = L.h, dstride = L.stride, pyrBytes = d.pyrBytes;   // read before the first store (a reload behind a store is a vector load)
	const int wl = wave - pl.waveBase;
	const int rb = wl / pl.wavesX, wx = wl - rb * pl.wavesX;
	const int y0 = rb * pl.rows;
	const int nrows = min(pl.rows, h - y0);           // output rows of this block
	// the lane's item: (image, column group)
	const int items = nimg * pl.ncg;
	const int item0 = wx * 64;
	const int itemRaw = item0 + (threadIdx.x & 63);
	const bool live = itemRaw < items;
	const int item = live ? itemRaw : items - 1;
	const int img = item / pl.ncg, cg = item - img * pl.ncg;
	const int img0 = item0 / pl.ncg;                 // the wave's first image: per-lane offsets are relative to it (32-bit)
	const int x = cg * 4;
	int sstride;
	const uint8_t* src = level_ptr(b, d, img0, level, &sstride);
	const size_t simg = level == 0 ? b.img0Pitch : (size_t)pyrBytes;
	uint8_t* dst = b.blur + (size_t)img0 * pyrBytes + L.off;
	// the 8 bytes of a row the thread loads start at A; window position j (column x - 2 + j, reflected) is byte sel_j of them
	const int A = min(max(x - 2, 0), w - 8);
	uint32_t selLo = 0, selHi = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const int p = reflect101(min(x - 2 + j, w + 1), w);
		const uint32_t s = (uint32_t)(p - A) & 7u;      // columns past w + 1 (never part of a stored pixel's window) may alias anything inside the 8 bytes
		if (j < 4) selLo |= s << (8 * j); else selHi |= s << (8 * (j - 4));
	}
	uint32_t voffS = (uint32_t)((size_t)(img - img0) * simg) + (uint32_t)A;
	uint32_t voffD = (uint32_t)((size_t)(img - img0) * (size_t)pyrBytes) + (uint32_t)x;
	auto loadRow = [&](int it) -> uint2 {             // source row y0 - 2 + it, reflected (scalar), this lane's 8 bytes
		const int r = reflect101(y0 - 2 + it, h);
		asm volatile("" : "+v"(voffS));                 // opaque per call: scalar row base + 32-bit lane offset stay apart (one global_load with an SGPR base, no per-lane 64-bit adds)
		uint2 v;
		__builtin_memcpy(&v, src + (size_t)((unsigned)r * (unsigned)sstride) + voffS, 8);
		return v;
	};
	const int total = nrows + 4;                       // rows read: y0 - 2 .. y0 + nrows + 1 <= h + 1, one reflection is enough
	uint2 buf[5];
#pragma unroll
	for (int k = 0; k < kBlurAhead; ++k) buf[k] = loadRow(min(k, total - 1));
	uint32_t ring[5][4];
	uint32_t V[4] = {12u, 12u, 12u, 12u};            // running 5x5 sums + the rounding term of (sum + 12) / 25
	// one row: request row it + 4, sum row it horizontally, slide the vertical window, emit output row y0 + it - 4.  k = it % 5 is a compile-time ring slot.
	auto body = [&](int it, auto kc, auto firstc, auto storec) {
		constexpr int k = decltype(kc)::value;
		buf[(k + kBlurAhead) % 5] = loadRow(min(it + kBlurAhead, total - 1));
		const uint2 raw = buf[k];
		const uint32_t W0 = __builtin_amdgcn_perm(raw.y, raw.x, selLo), W1 = __builtin_amdgcn_perm(raw.y, raw.x, selHi);
		// window bytes W0 = w0..w3, W1 = w4..w7; pixel c sums w[c .. c+4]
		uint32_t S[4];
		S[0] = __builtin_amdgcn_udot4(W0, 0x01010101u, W1 & 0xffu, false);
		S[1] = __builtin_amdgcn_udot4(W0, 0x01010100u, __builtin_amdgcn_udot4(W1, 0x00000101u, 0u, false), false);
		S[2] = __builtin_amdgcn_udot4(W0, 0x01010000u, __builtin_amdgcn_udot4(W1, 0x00010101u, 0u, false), false);
		S[3] = __builtin_amdgcn_udot4(W0, 0x01000000u, __builtin_amdgcn_udot4(W1, 0x01010101u, 0u, false), false);
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			V[c] += S[c];
			if (!decltype(firstc)::value) V[c] -= ring[k][c];
			ring[k][c] = S[c];
		}
		if (decltype(storec)::value) {
			// (s + 12) / 25 == ((s + 12) * 5243) >> 17 for s + 12 <= 6387 (25 * 5243 = 2^17 + 3: the quotient's error 3q / 2^17 <= 0.006 never carries a
			// remainder of at most 24/25 over the next integer); scaled by 2^7 the quotient is the product's top byte: 6387 * 671104 < 2^32, both factors < 2^24
			const uint32_t t0 = __umul24(V[0], 671104u), t1 = __umul24(V[1], 671104u), t2 = __umul24(V[2], 671104u), t3 = __umul24(V[3], 671104u);
			const uint32_t out = __builtin_amdgcn_perm(t1, t0, 0x0c0c0703u) | __builtin_amdgcn_perm(t3, t2, 0x07030c0cu);
			asm volatile("" : "+v"(voffD));
			// lanes past the last item repeat the last item: the same bytes to the same address.  Pitch is a multiple of 64: tail bytes stay in-pitch
			*reinterpret_cast<uint32_t*>(dst + (size_t)((unsigned)(y0 + it - 4) * (unsigned)dstride) + voffD) = out;
		}
	};
	using T = std::true_type; using F = std::false_type;
	// the first five rows fill the window (total >= 5); row 4 completes output row y0
	body(0, std::integral_constant<int, 0>(), T(), F()); body(1, std::integral_constant<int, 1>(), T(), F()); body(2, std::integral_constant<int, 2>(), T(), F());
	body(3, std::integral_constant<int, 3>(), T(), F()); body(4, std::integral_constant<int, 4>(), T(), T());
	int it0 = 5;
	for (; it0 + 5 <= total; it0 += 5) {               // branch-free groups of five rows: the compiler's vmcnt bookkeeping stays exact
		body(it0, std::integral_constant<int, 0>(), F(), T()); body(it0 + 1, std::integral_constant<int, 1>(), F(), T()); body(it0 + 2, std::integral_constant<int, 2>(), F(), T());
		body(it0 + 3, std::integral_constant<int, 3>(), F(), T()); body(it0 + 4, std::integral_constant<int, 4>(), F(), T());
	}
	const int rem = total - it0;
	if (rem > 0) body(it0, std::integral_constant<int, 0>(), F(), T());
	if (rem > 1) body(it0 + 1, std::integral_constant<int, 1>(), F(), T());
	if (rem > 2) body(it0 + 2, std::integral_constant<int, 2>(), F(), T());
	if (rem > 3) body(it0 + 3, std::integral_constant<int, 3>(), F(), T());
}

void launch_blur(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s) {
	BlurPlan plan;
	plan.nlevels = hd.nlevels;
	int waves = 0;
	for (int l = 0; l < hd.nlevels; ++l) {
		BlurLevel& p = plan.lv[l];
		const int w = hd.lv[l].w, h = hd.lv[l].h;
		p.ncg = (w + 3) / 4;
		const int nrb = (h + kBlurRows - 1) / kBlurRows;
		p.rows = (h + nrb - 1) / nrb;                   // balanced row blocks
		p.wavesX = (nimg * p.ncg + 63) / 64;
		p.waveBase = waves;
		waves += p.wavesX * ((h + p.rows - 1) / p.rows);
	}
	plan.totalWaves = waves;
	hipLaunchKernelGGL(k_blur, dim3(((waves + 3) / 4 + kNumXCD - 1) / kNumXCD * kNumXCD), dim3(256), 0, s, b, plan, nimg);
}

}  // namespace mcs

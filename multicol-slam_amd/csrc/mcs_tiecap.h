// mcs_tiecap.h — the device side of the rounding-tie capture (mcs_tiefix.hip says why): for every keypoint the exact descriptor arithmetic listed, its slot, level,
// selected-key record, angle and the (2R + 1)^2 window of Sampler::at values around it go to page-locked memory, where the host recomputes the descriptor with its
// own libm.  A function, not a kernel: k_tie_capture (device-kind batches) and the tail rows of k_extract_out (host-kind batches, no extra launch) both run it.
// Workgroups of 256 threads; workgroup `wg` of `nwg` takes the entries wg, wg + nwg, ...
#pragma once
#include "mcs_common.h"

namespace mcs {

__device__ __forceinline__ void tie_capture_body(const ExtractBuffers& b, int nimg, int wavesPerImage, int maxTies, uint8_t* __restrict__ out, int wg, int nwg) {
	const PyrDesc& d = *b.desc;
	const int n = *b.tieCount;
	TieCaptureHeader* hdr = reinterpret_cast<TieCaptureHeader*>(out);
	if (wg == 0 && threadIdx.x == 0) { hdr->count = n; hdr->status = *b.status; }
	for (int t = wg; t < n && t < maxTies; t += nwg) {   // (no barrier inside: every thread of a workgroup runs the same trips)
	TieCaptureEntry* en = reinterpret_cast<TieCaptureEntry*>(out + sizeof(TieCaptureHeader) + (size_t)t * sizeof(TieCaptureEntry));
	const uint32_t gw = b.tieList[t];
	const int img = (int)(gw / (uint32_t)wavesPerImage), sl = (int)(gw - (uint32_t)img * wavesPerImage);
	int level = -1, pos = 0, total = 0;
	if (img < nimg && sl < d.kpCap)
		for (int l = 0; l < d.nlevels; ++l) { const int c = b.selCount[(size_t)img * d.nlevels + l]; if (sl >= total && sl < total + c) { level = l; pos = sl - total; } total += c; }
	if (level < 0) { if (threadIdx.x == 0) { en->gw = gw; en->level = -1; } continue; }
	const LevelInfo& L = d.lv[level];
	const uint32_t rec = b.sel[(size_t)img * d.selPerImage + L.selBase + pos];
	if (threadIdx.x == 0) { en->gw = gw; en->level = level; en->rec = rec; en->angle = b.selAngle[(size_t)img * d.selPerImage + L.selBase + pos]; }
	const int col = (int)(rec & 0xFFF) + kMinBorder, row = (int)((rec >> 12) & 0xFFF) + kMinBorder;
	int rstride = 0;
	const uint8_t* raw = level_ptr(b, d, img, level, &rstride);
	const uint8_t* blur = b.blur + (size_t)img * d.pyrBytes + L.off;
	constexpr int D = 2 * kTiePatchR + 1;
	for (int i = threadIdx.x; i < D * D; i += 256) {
		int r = row - kTiePatchR + i / D, c = col - kTiePatchR + i % D;
		int v;
		if ((unsigned)r < (unsigned)L.h && (unsigned)c < (unsigned)L.w) v = blur[(size_t)r * L.stride + c];
		else {   // Sampler::at (mcs_describe.hip): clamped to the 25-px frame, reflect-101 into the unblurred level
			r = r < -kEdge ? -kEdge : (r > L.h + kEdge - 1 ? L.h + kEdge - 1 : r);
			c = c < -kEdge ? -kEdge : (c > L.w + kEdge - 1 ? L.w + kEdge - 1 : c);
			r = r < 0 ? -r : (r >= L.h ? 2 * (L.h - 1) - r : r);
			c = c < 0 ? -c : (c >= L.w ? 2 * (L.w - 1) - c : c);
			v = raw[(size_t)r * rstride + c];
		}
		en->patch[i] = (uint8_t)v;
	}
	}
}


}  // namespace mcs

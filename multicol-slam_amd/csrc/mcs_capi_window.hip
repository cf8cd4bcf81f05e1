// mcs_capi_window.hip — C ABI of the grid-window matchers other than SearchByProjection(F, mapPoints) and of the projection they
// consume (include/mcs_c.h: mcs_window_match, mcs_world_to_cam).  Kernels: mcs_project.hip.
#include "mcs_host.h"
#include <algorithm>
#include <cstring>
#include <vector>

using namespace mcs;

namespace {
inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }

// One device buffer per context, carved into aligned pieces per call (scratch lists + staged host-kind inputs).  It persists and only grows: every call
// on a context runs on the context's stream, so a later call's copies and kernels are ordered behind the earlier call's, and growing goes through
// hipFree, which waits for the device.  (A hipMalloc / hipFree pair per call cost more than the kernels of a single multi-frame.)
struct Arena {
	mcs_ctx* c;
	std::vector<size_t> sizes;
	uint8_t* base = nullptr;
	explicit Arena(mcs_ctx* ctx) : c(ctx) {}
	size_t add(size_t bytes) { sizes.push_back(al256(std::max<size_t>(bytes, 1))); return sizes.size() - 1; }
	hipError_t alloc() {
		size_t t = 0;
		for (size_t v : sizes) t += v;
		return ctx_arena(c, t, &base);
	}
	size_t off(size_t id) const { size_t o = 0; for (size_t i = 0; i < id; ++i) o += sizes[i]; return o; }
	size_t total() const { size_t t = 0; for (size_t v : sizes) t += v; return t; }
	uint8_t* at(size_t id) const { return base + off(id); }
};
}  // namespace

namespace mcs {
void launch_window_best(const ProjArgs& a, bool skipTaken, int* outDist, hipStream_t s);
void launch_rotation_consistency(int variant, const float* angleSlot, int strideSlot, const float* anglePartner, int stridePartner, const int* accepted, int* match,
                                 int n, int swapped, int* removedOut, hipStream_t s);
}

// bestMode 0: mcs_window_match (rule decides); 1: independent best-in-window; 2: best-in-window skipping taken features.  maxDist replaces TH_HIGH
// for bestMode != 0; dist (optional) receives the best distance per probe.
static int window_common(mcs_ctx* c, const mcs_window_probes* pr, const mcs_frame_view* f, mcs_window_rule rule, double nnratio, int dim, mcs_mem_kind kind,
                         int32_t* match, int32_t* nmatches, int bestMode, int maxDist, int32_t* dist) {
	if (!c || !pr || !f || !match || !nmatches) return fail(MCS_ERR_INVALID, "null argument");
	if (rule != MCS_WINDOW_RATIO && rule != MCS_WINDOW_BEST && rule != MCS_WINDOW_INITIALIZE) return fail(MCS_ERR_INVALID, "unknown window rule");
	if (dim != 16 && dim != 32 && dim != 64) return fail(MCS_ERR_INVALID, "dim must be 16, 32 or 64");
	if (pr->n < 0 || f->n < 0 || f->n > 65536 || f->nr_cams < 1) return fail(MCS_ERR_INVALID, "bad sizes (frame features must be <= 65536)");
	if ((pr->mask == nullptr) != (f->mask == nullptr)) return fail(MCS_ERR_INVALID, "masks must be given for both sides or neither");
	if (pr->stride < dim || f->stride < dim || (pr->stride & 3) || (f->stride & 3)) return fail(MCS_ERR_INVALID, "descriptor stride must be >= dim and a multiple of 4");
	if (rule != MCS_WINDOW_INITIALIZE && bestMode != 1 && !f->assigned) return fail(MCS_ERR_INVALID, "frame->assigned is required");
	HIPCHK(hipSetDevice(c->device));
	hipStream_t s = c->stream;
	if (c->side && c->greedyPending) { HIPCHK(hipStreamWaitEvent(s, c->evGreedy, 0)); c->greedyPending = false; }
	const bool havingMasks = pr->mask != nullptr, host = kind == MCS_MEM_HOST;
	const size_t np = pr->n, nf = f->n;
	ProjArgs a{};
	a.rule = (int)rule; a.cap = kProjListK;
	a.nproj = pr->n; a.pstride = pr->stride; a.nfeat = f->n; a.fstride = f->stride; a.nrCams = f->nr_cams;
	a.ratio = nnratio; a.dim = dim; a.th = 1.0;
	a.thHigh = havingMasks ? (int)floor(1.5 * dim) : 3 * dim;   // TH_HIGH_ / TH_LOW_ (src/cORBmatcher.cpp:46-65)
	a.thLow = havingMasks ? (int)floor((double)dim) : 2 * dim;
	if (bestMode) a.thHigh = maxDist;
	Arena ar(c);
	const size_t iLists = ar.add(np * kProjListK * 8), iCounts = ar.add(np * 4), iOwner = ar.add(nf * 4), iMdist = ar.add(nf * 4), iAsg = ar.add(nf);
	size_t iX = 0, iY = 0, iR = 0, iLo = 0, iHi = 0, iPc = 0, iPd = 0, iPm = 0, iKeys = 0, iFd = 0, iFm = 0, iFc = 0, iW = 0, iH = 0, iMatch = 0, iNm = 0;
	if (host) {
		iX = ar.add(np * 8); iY = ar.add(np * 8); iR = ar.add(np * 8); iLo = ar.add(np * 4); iHi = ar.add(np * 4); iPc = ar.add(np * 4);
		iPd = ar.add(np * pr->stride); iPm = ar.add(np * pr->stride); iKeys = ar.add(nf * sizeof(mcs_keypoint)); iFd = ar.add(nf * f->stride);
		iFm = ar.add(nf * f->stride); iFc = ar.add(nf * 4); iW = ar.add((size_t)f->nr_cams * 4); iH = ar.add((size_t)f->nr_cams * 4);
		iMatch = ar.add(np * 4); iNm = ar.add(4);
	}
	const size_t iDist = ar.add(host && dist ? np * 4 : 0);
	const bool wantAcc = rule == MCS_WINDOW_INITIALIZE && pr->accepted_out != nullptr && bestMode == 0;
	const size_t iAcc = ar.add(host && wantAcc ? np * 4 : 0);
	HIPCHK(ar.alloc());
	auto done = [&](int rc) { (void)hipStreamSynchronize(s); return rc; };   // the arena is freed by its destructor after this sync
	a.lists = (unsigned long long*)ar.at(iLists); a.counts = (int*)ar.at(iCounts); a.owner = (int*)ar.at(iOwner); a.mdist = (int*)ar.at(iMdist);
	if (host) {
		PinnedUpload up;
		HIPCHK(up.begin(c, ar.base, ar.total()));
#define UP(id, src, bytes) up.put(ar.off(id), (src), (bytes))
		UP(iX, pr->x, np * 8); UP(iY, pr->y, np * 8); UP(iR, pr->radius, np * 8); UP(iLo, pr->min_level, np * 4); UP(iHi, pr->max_level, np * 4);
		UP(iPc, pr->cam, np * 4); UP(iPd, pr->desc, np * pr->stride);
		if (havingMasks) { UP(iPm, pr->mask, np * pr->stride); UP(iFm, f->mask, nf * f->stride); }
		UP(iKeys, f->keys, nf * sizeof(mcs_keypoint)); UP(iFd, f->desc, nf * f->stride); UP(iFc, f->cam, nf * 4);
		UP(iW, f->width, (size_t)f->nr_cams * 4); UP(iH, f->height, (size_t)f->nr_cams * 4);
		if (f->assigned) UP(iAsg, f->assigned, nf);
#undef UP
		if (up.flush(s) != hipSuccess) return done(fail(MCS_ERR_HIP, "H2D copy failed"));
		a.px = (const double*)ar.at(iX); a.py = (const double*)ar.at(iY); a.rad = (const double*)ar.at(iR); a.minLvl = (const int*)ar.at(iLo);
		a.maxLvl = (const int*)ar.at(iHi); a.pcam = (const int*)ar.at(iPc); a.pdesc = ar.at(iPd); a.pmask = havingMasks ? ar.at(iPm) : nullptr;
		a.keys = (const mcs_keypoint*)ar.at(iKeys); a.fdesc = ar.at(iFd); a.fmask = havingMasks ? ar.at(iFm) : nullptr; a.fcam = (const int*)ar.at(iFc);
		a.width = (const int*)ar.at(iW); a.height = (const int*)ar.at(iH); a.assigned = ar.at(iAsg);
		a.match = (int*)ar.at(iMatch); a.nmatches = (int*)ar.at(iNm);
	} else {
		a.px = pr->x; a.py = pr->y; a.rad = pr->radius; a.minLvl = pr->min_level; a.maxLvl = pr->max_level; a.pcam = pr->cam; a.pdesc = pr->desc;
		a.pmask = pr->mask; a.keys = f->keys; a.fdesc = f->desc; a.fmask = f->mask; a.fcam = f->cam; a.width = f->width; a.height = f->height;
		a.assigned = f->assigned ? f->assigned : ar.at(iAsg); a.match = match; a.nmatches = nmatches;
	}
	if (!f->assigned) { if (hipMemsetAsync(ar.at(iAsg), 0, std::max<size_t>(nf, 1), s) != hipSuccess) return done(fail(MCS_ERR_HIP, "memset failed")); }
	a.accepted = wantAcc ? (host ? (int*)ar.at(iAcc) : pr->accepted_out) : nullptr;
	int* ddist = dist ? (host ? (int*)ar.at(iDist) : dist) : nullptr;
	if (pr->n > 0) {
		if (bestMode) launch_window_best(a, bestMode == 2, ddist, s);
		else { c->tic("win_candidates"); launch_proj_candidates(a, s); c->toc("win_candidates"); c->tic("win_greedy"); launch_proj_greedy(a, s); c->toc("win_greedy"); }
	}
	else if (!host) { if (hipMemsetAsync(nmatches, 0, 4, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "memset failed")); }
	if (hipGetLastError() != hipSuccess) return done(fail(MCS_ERR_HIP, "window kernels failed to launch"));
	if (host && dist && pr->n > 0 && hipMemcpyAsync(dist, ddist, np * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "D2H"));
	if (host && wantAcc && pr->n > 0 && hipMemcpyAsync(pr->accepted_out, a.accepted, np * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "D2H"));
	if (host) {
		*nmatches = 0;
		if (pr->n > 0) {
			if (hipMemcpyAsync(match, ar.at(iMatch), np * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "D2H"));
			if (hipMemcpyAsync(nmatches, ar.at(iNm), 4, hipMemcpyDeviceToHost, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "D2H"));
			if (f->assigned && nf && hipMemcpyAsync(f->assigned, ar.at(iAsg), nf, hipMemcpyDeviceToHost, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "D2H"));
		}
	}
	return done(MCS_OK);
}

int mcs_window_match(mcs_ctx* c, const mcs_window_probes* pr, const mcs_frame_view* f, mcs_window_rule rule, double nnratio, int dim, mcs_mem_kind kind,
                     int32_t* match, int32_t* nmatches) {
	return window_common(c, pr, f, rule, nnratio, dim, kind, match, nmatches, 0, 0, nullptr);
}

int mcs_window_best(mcs_ctx* c, const mcs_window_probes* pr, const mcs_frame_view* f, int max_dist, int skip_taken, int dim, mcs_mem_kind kind,
                    int32_t* match, int32_t* dist, int32_t* nmatches) {
	if (max_dist < 0) return fail(MCS_ERR_INVALID, "max_dist must be >= 0");
	return window_common(c, pr, f, MCS_WINDOW_BEST, 1.0, dim, kind, match, nmatches, skip_taken ? 2 : 1, max_dist, dist);
}

int mcs_world_to_cam(mcs_ctx* c, const double* MtMc_inv, const mcs_ocam* cams, int nr_cams, const uint8_t* const* mirror_masks, const double* pts3,
                     const int32_t* cam, int n, mcs_mem_kind kind, double* uv, uint8_t* flags) {
	if (!c || !MtMc_inv || !cams || !pts3 || !cam || !uv || !flags) return fail(MCS_ERR_INVALID, "null argument");
	if (nr_cams < 1 || n < 0) return fail(MCS_ERR_INVALID, "bad sizes");
	HIPCHK(hipSetDevice(c->device));
	hipStream_t s = c->stream;
	const bool host = kind == MCS_MEM_HOST;
	std::vector<OcamDev> hc(nr_cams);
	std::vector<int> w(nr_cams), h(nr_cams);
	size_t maskBytes = 0;
	for (int i = 0; i < nr_cams; ++i) {
		const mcs_ocam& m = cams[i];
		if (m.p_deg < 1 || m.p_deg > MCS_MAX_POLY || m.invP_deg < 1 || m.invP_deg > MCS_MAX_POLY) return fail(MCS_ERR_INVALID, "bad polynomial degree");
		if (m.width < 1 || m.height < 1) return fail(MCS_ERR_INVALID, "bad image size");
		OcamDev& o = hc[i];
		memset(&o, 0, sizeof(o));
		o.c = m.c; o.d = m.d; o.e = m.e; o.u0 = m.u0; o.v0 = m.v0; o.invAffine = m.c - m.d * m.e;
		for (int k = 0; k < m.p_deg; ++k) o.p[k] = m.p[k];
		for (int k = 0; k < m.invP_deg; ++k) o.invP[k] = m.invP[k];
		o.p_deg = m.p_deg; o.invP_deg = m.invP_deg;
		w[i] = m.width; h[i] = m.height;
		if (mirror_masks && mirror_masks[i]) maskBytes += al256((size_t)m.width * m.height);
	}
	Arena ar(c);
	const size_t iM = ar.add((size_t)nr_cams * 128), iC = ar.add(sizeof(OcamDev) * nr_cams), iW = ar.add(4 * (size_t)nr_cams), iH = ar.add(4 * (size_t)nr_cams),
	             iMp = ar.add(sizeof(void*) * nr_cams), iMk = ar.add(host ? maskBytes : 0);
	size_t iP = 0, iPc = 0, iUv = 0, iFl = 0;
	if (host) { iP = ar.add((size_t)n * 24); iPc = ar.add((size_t)n * 4); iUv = ar.add((size_t)n * 16); iFl = ar.add((size_t)n); }
	HIPCHK(ar.alloc());
	auto done = [&](int rc) { (void)hipStreamSynchronize(s); return rc; };
#define UP(dst, src, bytes) do { if ((bytes) && hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "H2D copy failed")); } while (0)
	// the matrices / calibrations are host values for either kind (they are the camera system's state, a few hundred bytes)
	UP(ar.at(iM), MtMc_inv, (size_t)nr_cams * 128); UP(ar.at(iC), hc.data(), sizeof(OcamDev) * nr_cams); UP(ar.at(iW), w.data(), 4 * (size_t)nr_cams);
	UP(ar.at(iH), h.data(), 4 * (size_t)nr_cams);
	std::vector<const uint8_t*> mp(nr_cams, nullptr);
	if (mirror_masks) {
		size_t off = 0;
		for (int i = 0; i < nr_cams; ++i) {
			if (!mirror_masks[i]) continue;
			if (host) { uint8_t* d = ar.at(iMk) + off; UP(d, mirror_masks[i], (size_t)w[i] * h[i]); mp[i] = d; off += al256((size_t)w[i] * h[i]); }
			else mp[i] = mirror_masks[i];
		}
	}
	UP(ar.at(iMp), mp.data(), sizeof(void*) * nr_cams);
	WorldToCamArgs a{};
	a.M = (const double*)ar.at(iM); a.cams = (const OcamDev*)ar.at(iC); a.width = (const int*)ar.at(iW); a.height = (const int*)ar.at(iH);
	a.masks = mirror_masks ? (const uint8_t* const*)ar.at(iMp) : nullptr; a.n = n;
	if (host) {
		UP(ar.at(iP), pts3, (size_t)n * 24); UP(ar.at(iPc), cam, (size_t)n * 4);
		a.pts = (const double*)ar.at(iP); a.pcam = (const int*)ar.at(iPc); a.uv = (double*)ar.at(iUv); a.flags = ar.at(iFl);
	} else { a.pts = pts3; a.pcam = cam; a.uv = uv; a.flags = flags; }
#undef UP
	launch_world_to_cam(a, s);
	if (hipGetLastError() != hipSuccess) return done(fail(MCS_ERR_HIP, "k_world_to_cam failed to launch"));
	if (host && n > 0) {
		if (hipMemcpyAsync(uv, ar.at(iUv), (size_t)n * 16, hipMemcpyDeviceToHost, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "D2H"));
		if (hipMemcpyAsync(flags, ar.at(iFl), (size_t)n, hipMemcpyDeviceToHost, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "D2H"));
	}
	return done(MCS_OK);
}

int mcs_distinctive_descriptors(mcs_ctx* c, const uint8_t* desc, const uint8_t* mask, int stride, int dim, const int32_t* offsets, int npoints,
                                mcs_mem_kind kind, int32_t* best_idx) {
	if (!c || !desc || !offsets || !best_idx) return fail(MCS_ERR_INVALID, "null argument");
	if (dim != 16 && dim != 32 && dim != 64) return fail(MCS_ERR_INVALID, "dim must be 16, 32 or 64");
	if (stride < dim || (stride & 3) || npoints < 0) return fail(MCS_ERR_INVALID, "bad stride / count");
	if (npoints == 0) return MCS_OK;
	HIPCHK(hipSetDevice(c->device));
	hipStream_t s = c->stream;
	const bool host = kind == MCS_MEM_HOST;
	DistinctArgs a{};
	a.stride = stride; a.dim = dim; a.npoints = npoints;
	if (!host) {   // offsets are validated by the caller in this mode
		a.desc = desc; a.mask = mask; a.offsets = offsets; a.bestIdx = best_idx;
		launch_distinct(a, s);
		HIPCHK(hipGetLastError());
		return MCS_OK;
	}
	if (offsets[0] != 0) return fail(MCS_ERR_INVALID, "offsets[0] must be 0");
	for (int k = 0; k < npoints; ++k)
		if (offsets[k + 1] < offsets[k] || offsets[k + 1] - offsets[k] > 65535) return fail(MCS_ERR_INVALID, "offsets must be non-decreasing, <= 65535 rows per map point");
	const size_t rows = (size_t)offsets[npoints];
	Arena ar(c);
	const size_t iD = ar.add(rows * stride), iM = ar.add(mask ? rows * stride : 0), iO = ar.add(((size_t)npoints + 1) * 4), iB = ar.add((size_t)npoints * 4);
	HIPCHK(ar.alloc());
	auto done = [&](int rc) { (void)hipStreamSynchronize(s); return rc; };
#define UP(id, src, bytes) do { if ((bytes) && hipMemcpyAsync(ar.at(id), (src), (bytes), hipMemcpyHostToDevice, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "H2D copy failed")); } while (0)
	UP(iD, desc, rows * stride);
	if (mask) UP(iM, mask, rows * stride);
	UP(iO, offsets, ((size_t)npoints + 1) * 4);
#undef UP
	a.desc = ar.at(iD); a.mask = mask ? ar.at(iM) : nullptr; a.offsets = (const int*)ar.at(iO); a.bestIdx = (int*)ar.at(iB);
	launch_distinct(a, s);
	if (hipGetLastError() != hipSuccess) return done(fail(MCS_ERR_HIP, "k_distinct failed to launch"));
	if (hipMemcpyAsync(best_idx, ar.at(iB), (size_t)npoints * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "D2H"));
	return done(MCS_OK);
}

namespace mcs { void launch_selftest_recip(unsigned long long seed, int n, int* mismatches, hipStream_t s); }

int mcs_selftest_shared_reciprocal(mcs_ctx* c, uint64_t seed, int n, int32_t* mismatches) {
	if (!c || !mismatches || n < 1) return fail(MCS_ERR_INVALID, "bad argument");
	HIPCHK(hipSetDevice(c->device));
	int* d = nullptr;
	HIPCHK(hipMalloc((void**)&d, 4));
	(void)hipMemsetAsync(d, 0, 4, c->stream);
	launch_selftest_recip(seed, n, d, c->stream);
	const hipError_t e1 = hipGetLastError();
	const hipError_t e2 = hipMemcpyAsync(mismatches, d, 4, hipMemcpyDeviceToHost, c->stream);
	const hipError_t e3 = hipStreamSynchronize(c->stream);
	(void)hipFree(d);
	if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(MCS_ERR_HIP, "selftest kernel failed");
	return MCS_OK;
}

int mcs_rotation_consistency(mcs_ctx* c, int variant, const float* angle_slot, int stride_slot, const float* angle_partner, int stride_partner,
                             const int32_t* accepted, int32_t* match, int n, int n_partner, int swapped, mcs_mem_kind kind, int32_t* removed) {
	if (!c || !angle_slot || !angle_partner || !match || !removed) return fail(MCS_ERR_INVALID, "null argument");
	if (variant < 0 || variant > 3 || n < 0 || n_partner < 0 || stride_slot < 4 || stride_partner < 4 || (stride_slot & 3) || (stride_partner & 3))
		return fail(MCS_ERR_INVALID, "bad variant / sizes / strides");
	HIPCHK(hipSetDevice(c->device));
	hipStream_t s = c->stream;
	if (c->side && c->greedyPending) { HIPCHK(hipStreamWaitEvent(s, c->evGreedy, 0)); c->greedyPending = false; }   // `match` may come from a search whose greedy pass runs on the side stream
	const bool host = kind == MCS_MEM_HOST;
	if (!host) {
		launch_rotation_consistency(variant, angle_slot, stride_slot, angle_partner, stride_partner, accepted, match, n, swapped, removed, s);
		HIPCHK(hipGetLastError());
		return MCS_OK;
	}
	const size_t bs = (size_t)n * stride_slot, bp = (size_t)n_partner * stride_partner;
	Arena ar(c);
	const size_t iS = ar.add(bs), iP = ar.add(bp), iA = ar.add(accepted ? (size_t)n * 4 : 0), iM = ar.add((size_t)n * 4), iR = ar.add(4);
	HIPCHK(ar.alloc());
	auto done = [&](int rc) { (void)hipStreamSynchronize(s); return rc; };
#define UP(id, src, bytes) do { if ((bytes) && hipMemcpyAsync(ar.at(id), (src), (bytes), hipMemcpyHostToDevice, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "H2D copy failed")); } while (0)
	UP(iS, angle_slot, bs > 0 ? bs - (stride_slot - 4) : 0); UP(iP, angle_partner, bp > 0 ? bp - (stride_partner - 4) : 0);   // the last keypoint's tail may not be readable
	if (accepted) UP(iA, accepted, (size_t)n * 4);
	UP(iM, match, (size_t)n * 4);
#undef UP
	launch_rotation_consistency(variant, (const float*)ar.at(iS), stride_slot, (const float*)ar.at(iP), stride_partner, accepted ? (const int*)ar.at(iA) : nullptr,
	                            (int*)ar.at(iM), n, swapped, (int*)ar.at(iR), s);
	if (hipGetLastError() != hipSuccess) return done(fail(MCS_ERR_HIP, "k_rotation_consistency failed to launch"));
	if (n && hipMemcpyAsync(match, ar.at(iM), (size_t)n * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "D2H"));
	if (hipMemcpyAsync(removed, ar.at(iR), 4, hipMemcpyDeviceToHost, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "D2H"));
	return done(MCS_OK);
}

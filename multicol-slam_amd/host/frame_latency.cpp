// frame_latency.cpp — latency of ONE multi-frame through the C ABI from a native host: no Python, no torch, no HIP header.
//
// The reference's unit of work is one multi-frame per call (cTracking::GrabImageMulti builds ONE cMultiFrame and tracks it, src/cTracking.cpp:206-235); the
// throughput bench batches 64 of them.  This program measures the other end: per call
//     mcs_extract_batch   the rig's images of one multi-frame, HOST buffers in (page-locked staging, mcs_host_alloc) and out, synchronous
//     mcs_search_kf_kf    SearchByBoW(KF,KF) of that multi-frame against the previous one (src/cORBmatcher.cpp:885-966), host buffers, synchronous
// each timed with the host's steady clock around the call, as a tracker would see it.  `batch B` in the configuration presents B multi-frames per call instead (one
// extraction of B x ncam images, one search over B set pairs: bench.py's batch_sweep).  Input: the key / value file of rig_host (images [camera][frame][H][W],
// one mask and one mcs_ocam per camera).  Output: one JSON line (median / p90 / p99 / mean / min of the per-call times in ms, for both calls and their sum) and,
// for the caller's oracle check, the raw outputs of the LAST call: <out>.nkp / .kps / .desc / .mask / .match (int32 per query row, -1 = none).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "../../include/mcs_c.h"

#define MCSOK(x) do { int r_ = (x); if (r_ != MCS_OK) { fprintf(stderr, "%s:%d %s: %d %s\n", __FILE__, __LINE__, #x, r_, mcs_last_error()); exit(4); } } while (0)

static std::map<std::string, std::string> read_config(const char* path) {
	std::map<std::string, std::string> m;
	std::ifstream f(path);
	std::string k, v;
	while (f >> k >> v) m[k] = v;
	return m;
}
static std::vector<uint8_t> read_file(const std::string& p) {
	std::ifstream f(p, std::ios::binary);
	if (!f) { fprintf(stderr, "cannot read %s\n", p.c_str()); exit(1); }
	return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void write_file(const std::string& p, const void* d, size_t n) { std::ofstream f(p, std::ios::binary); f.write((const char*)d, (std::streamsize)n); }

struct Stats { double med, p90, p99, mean, mn; };
static Stats stats(std::vector<double> v) {
	std::sort(v.begin(), v.end());
	double s = 0;
	for (double x : v) s += x;
	auto q = [&](double f) { return v[std::min(v.size() - 1, (size_t)(f * (double)v.size()))]; };
	return {q(0.5), q(0.9), q(0.99), s / (double)v.size(), v.front()};
}

int main(int argc, char** argv) {
	if (argc < 2) { fprintf(stderr, "usage: frame_latency <config>\n"); return 1; }
	auto cfg = read_config(argv[1]);
	auto geti = [&](const char* k, int d) { return cfg.count(k) ? atoi(cfg[k].c_str()) : d; };
	const int ncam = geti("ncam", 3), W = geti("width", 754), H = geti("height", 480), nfeat = geti("nfeatures", 1000), mode = geti("mode", 2);
	const int frames = geti("frames", 8), calls = geti("calls", 300), warm = geti("warmup", 20), topk = geti("topk", 32);
	const double ratio = cfg.count("ratio") ? atof(cfg["ratio"].c_str()) : 0.9;
	const std::vector<uint8_t> images = read_file(cfg["images"]), masks = read_file(cfg["masks"]), camb = read_file(cfg["cams"]);
	const size_t plane = (size_t)W * H;
	if (images.size() < (size_t)ncam * frames * plane || masks.size() < (size_t)ncam * plane || camb.size() < sizeof(mcs_ocam) * ncam) { fprintf(stderr, "inputs too small\n"); return 1; }
	const mcs_ocam* cams = (const mcs_ocam*)camb.data();

	// `batch` multi-frames per call (default 1 = the reference's own call shape; bench.py's batch_sweep runs 1, 2, 4 ... 64): the images of the call are staged and
	// extracted by ONE mcs_extract_batch, and every multi-frame is matched against the one before it by ONE mcs_search_kf_kf over `batch` set pairs — the first of
	// them against the last multi-frame of the previous call.  Host buffers in and out, both calls synchronous.
	const int B = std::max(1, geti("batch", 1));
	mcs_ctx* ctx = nullptr;
	MCSOK(mcs_ctx_create(geti("device", 0), nullptr, &ctx));
	const bool masksOn = mode == 2;
	mcs_extractor_params prm = {nfeat, 1.2f, 8, 25, 0, 0, 32, 20, 0, 2, mode >= 1 ? 1 : 0, masksOn ? 1 : 0, 32};
	mcs_extractor* ex = nullptr;
	const int nimg = B * ncam;
	MCSOK(mcs_extractor_create(ctx, &prm, W, H, nimg, &ex));
	int cap = 0;
	MCSOK(mcs_extractor_kp_capacity(ex, &cap));
	const size_t rows = (size_t)ncam * cap;   // rows of ONE multi-frame
	// page-locked: the staged images + masks of one call (image j * ncam + c = camera c of the call's multi-frame j), and R * B + 1 multi-frame slots of outputs
	// used round-robin: call number it writes slots 1 + r B .. (r + 1) B (r = it % R), its first pair's train side is the slot in front of them — the previous
	// call's last multi-frame, in place; only when the ring wraps (every R calls) is that one multi-frame copied to slot 0
	uint8_t *inImg = nullptr, *inMask = nullptr;
	MCSOK(mcs_host_alloc(ctx, nimg * plane, (void**)&inImg));
	MCSOK(mcs_host_alloc(ctx, nimg * plane, (void**)&inMask));
	for (int j = 0; j < B; ++j) memcpy(inMask + (size_t)j * ncam * plane, masks.data(), ncam * plane);
	// "prestaged 1": the grabber delivers into page-locked memory itself — the whole pool of multi-frames lies there as [frame][camera] planes and a call reads its
	// B consecutive multi-frames in place (the staging memcpy of the default form, ~0.1 ms per multi-frame on one host core, is the cv::Mat -> staging copy of a
	// host whose frames arrive in pageable memory)
	const bool prestaged = geti("prestaged", 0) != 0;
	uint8_t* pool = nullptr;
	if (prestaged) {
		MCSOK(mcs_host_alloc(ctx, (size_t)frames * ncam * plane, (void**)&pool));
		for (int f = 0; f < frames; ++f)
			for (int c = 0; c < ncam; ++c) memcpy(pool + ((size_t)f * ncam + c) * plane, images.data() + ((size_t)c * frames + f) * plane, plane);
	}
	std::vector<mcs_ocam> camv((size_t)nimg);
	for (int i = 0; i < nimg; ++i) camv[i] = cams[i % ncam];
	// the rig's mirror masks never change: they stay on the device ("resident 0" in the config: uploaded with every call, as the reference's call shape would)
	const bool resident = geti("resident", 1) != 0;
	if (resident) MCSOK(mcs_extractor_set_masks(ex, nimg, inMask, plane, W, MCS_MEM_HOST));
	struct Out { int32_t* nkp; mcs_keypoint* kps; uint8_t *desc, *mask, *valid; double* rays; int32_t* match; } o;
	const int R = std::max(2, 64 / B);
	const size_t slots = (size_t)B * R + 1;
	MCSOK(mcs_host_alloc(ctx, sizeof(int32_t) * ncam * slots, (void**)&o.nkp));
	MCSOK(mcs_host_alloc(ctx, sizeof(mcs_keypoint) * rows * slots, (void**)&o.kps));
	MCSOK(mcs_host_alloc(ctx, 32 * rows * slots, (void**)&o.desc));
	MCSOK(mcs_host_alloc(ctx, 32 * rows * slots, (void**)&o.mask));
	MCSOK(mcs_host_alloc(ctx, rows * slots, (void**)&o.valid));
	MCSOK(mcs_host_alloc(ctx, sizeof(double) * 3 * rows * slots, (void**)&o.rays));
	MCSOK(mcs_host_alloc(ctx, sizeof(int32_t) * rows * B, (void**)&o.match));
	memset(o.valid, 0, rows * slots);
	memset(o.nkp, 0, sizeof(int32_t) * ncam * slots);
	std::vector<double> tE, tM, tS;
	int32_t* counts = nullptr;   // [0 .. B) matches, [B .. 2B) rescans per pair: page-locked like the match array, so that the search writes all three with one launch
	MCSOK(mcs_host_alloc(ctx, 2 * B * sizeof(int32_t), (void**)&counts));
	memset(counts, 0, 2 * B * sizeof(int32_t));
	int32_t *nmatch = counts, *nfb = counts + B;
	long fnext = 0;   // the stream's next multi-frame (frame index = fnext % frames)
	for (int it = 0; it < warm + calls; ++it) {
		const auto t0 = std::chrono::steady_clock::now();
		const size_t s0 = (size_t)(it % R) * B;   // the slot in front of this call's
		if (it > 0 && s0 == 0) {   // the ring wraps: the previous call's last multi-frame becomes slot 0
			memcpy(o.nkp, o.nkp + (size_t)B * R * ncam, sizeof(int32_t) * ncam);
			memcpy(o.desc, o.desc + (size_t)B * R * rows * 32, rows * 32);
			memcpy(o.mask, o.mask + (size_t)B * R * rows * 32, rows * 32);
			memcpy(o.valid, o.valid + (size_t)B * R * rows, rows);
		}
		Out c = {o.nkp + (s0 + 1) * ncam, o.kps + (s0 + 1) * rows, o.desc + (s0 + 1) * rows * 32, o.mask + (s0 + 1) * rows * 32, o.valid + (s0 + 1) * rows, o.rays + (s0 + 1) * rows * 3, o.match};
		const uint8_t* src = inImg;
		if (prestaged && (int)(fnext % frames) + B <= frames) src = pool + (size_t)(fnext % frames) * ncam * plane;
		else for (int j = 0; j < B; ++j) {
			const int f = (int)((fnext + j) % frames);
			for (int c = 0; c < ncam; ++c) memcpy(inImg + ((size_t)j * ncam + c) * plane, images.data() + ((size_t)c * frames + f) * plane, plane);   // what a grabber's cv::Mat -> staging copy costs
		}
		MCSOK(mcs_extract_batch(ex, nimg, src, plane, W, resident ? MCS_MASKS_RESIDENT : inMask, plane, W, camv.data(), MCS_MEM_HOST, c.nkp, c.kps, c.desc, c.mask, c.rays));
		// every multi-frame as ONE descriptor set: camera blocks of `cap` rows, the rows past a camera's count flagged invalid (every keypoint "has a map point")
		for (int i = 0; i < nimg; ++i) {
			uint8_t* v = c.valid + (size_t)i * cap;
			const int n = c.nkp[i];
			memset(v, 1, (size_t)n);
			memset(v + n, 0, (size_t)(cap - n));
		}
		const auto t1 = std::chrono::steady_clock::now();
		if (it > 0) {
			mcs_desc_set q, t;
			memset(&q, 0, sizeof(q)); memset(&t, 0, sizeof(t));
			q.desc = c.desc; q.mask = masksOn ? c.mask : nullptr; q.valid = c.valid; q.n = (int)rows; q.stride = 32;
			t.desc = c.desc - rows * 32; t.mask = masksOn ? c.mask - rows * 32 : nullptr; t.valid = c.valid - rows; t.n = (int)rows; t.stride = 32;
			MCSOK(mcs_search_kf_kf(ctx, B, &q, rows, &t, rows, 32, ratio, topk, MCS_MEM_HOST, o.match, nmatch, nfb));
		}
		const auto t2 = std::chrono::steady_clock::now();
		if (it >= warm) {
			tE.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
			tM.push_back(std::chrono::duration<double, std::milli>(t2 - t1).count());
			tS.push_back(std::chrono::duration<double, std::milli>(t2 - t0).count());
		}
		fnext += B;
	}
	const int firstFrame = (int)((fnext - B) % frames), lastFrame = (int)((fnext - 1) % frames);
	const size_t sl = (size_t)((warm + calls - 1) % R) * B + 1;   // first slot of the last call
	if (cfg.count("out")) {   // the LAST call: its B multi-frames (slots 1 .. B), the match arrays of its B pairs, and the counts of slot 0 (the first pair's train side)
		const std::string p = cfg["out"];
		write_file(p + ".nkp", o.nkp + sl * ncam, sizeof(int32_t) * nimg);
		write_file(p + ".kps", o.kps + sl * rows, sizeof(mcs_keypoint) * rows * B);
		write_file(p + ".desc", o.desc + sl * rows * 32, 32 * rows * B);
		write_file(p + ".mask", o.mask + sl * rows * 32, 32 * rows * B);
		write_file(p + ".match", o.match, sizeof(int32_t) * rows * B);
		write_file(p + ".nmatch", nmatch, sizeof(int32_t) * B);
		write_file(p + ".prev_nkp", o.nkp + (sl - 1) * ncam, sizeof(int32_t) * ncam);
	}
	int feats = 0, matches = 0, rescans = 0;
	for (int i = 0; i < nimg; ++i) feats += o.nkp[sl * ncam + i];
	for (int j = 0; j < B; ++j) { matches += nmatch[j]; rescans += nfb[j]; }
	const Stats e = stats(tE), m = stats(tM), s = stats(tS);
	printf("{\"masks_resident\": %s, \"batch\": %d, \"calls\": %d, \"warmup\": %d, \"cap\": %d, \"features_last\": %d, \"matches_last\": %d, \"rescans_last\": %d, \"first_frame\": %d, \"last_frame\": %d, "
	       "\"extract_ms\": {\"median\": %.4f, \"p90\": %.4f, \"p99\": %.4f, \"mean\": %.4f, \"min\": %.4f}, "
	       "\"match_ms\": {\"median\": %.4f, \"p90\": %.4f, \"p99\": %.4f, \"mean\": %.4f, \"min\": %.4f}, "
	       "\"total_ms\": {\"median\": %.4f, \"p90\": %.4f, \"p99\": %.4f, \"mean\": %.4f, \"min\": %.4f}}\n",
	       resident ? "true" : "false", B, calls, warm, cap, feats, matches, rescans, firstFrame, lastFrame, e.med, e.p90, e.p99, e.mean, e.mn, m.med, m.p90, m.p99, m.mean, m.mn, s.med, s.p90, s.p99, s.mean, s.mn);
	mcs_host_free(ctx, o.nkp); mcs_host_free(ctx, o.kps); mcs_host_free(ctx, o.desc); mcs_host_free(ctx, o.mask); mcs_host_free(ctx, o.valid); mcs_host_free(ctx, o.rays); mcs_host_free(ctx, o.match);
	mcs_host_free(ctx, inImg); mcs_host_free(ctx, inMask); mcs_host_free(ctx, counts);
	if (pool) mcs_host_free(ctx, pool);
	MCSOK(mcs_extractor_destroy(ex));
	MCSOK(mcs_ctx_destroy(ctx));
	return 0;
}

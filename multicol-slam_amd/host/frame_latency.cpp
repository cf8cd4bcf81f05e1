// frame_latency.cpp — latency of ONE multi-frame through the C ABI from a native host: no Python, no torch, no HIP header.
//
// The reference's unit of work is one multi-frame per call (cTracking::GrabImageMulti builds ONE cMultiFrame and tracks it, src/cTracking.cpp:206-235); the
// throughput bench batches 64 of them.  This program measures the other end: per call
//     mcs_extract_batch   the rig's images of one multi-frame, HOST buffers in (page-locked staging, mcs_host_alloc) and out, synchronous
//     mcs_search_kf_kf    SearchByBoW(KF,KF) of that multi-frame against the previous one (src/cORBmatcher.cpp:885-966), host buffers, synchronous
// each timed with the host's steady clock around the call, as a tracker would see it.  Input: the key / value file of rig_host (images [camera][frame][H][W],
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

	mcs_ctx* ctx = nullptr;
	MCSOK(mcs_ctx_create(geti("device", 0), nullptr, &ctx));
	const bool masksOn = mode == 2;
	mcs_extractor_params prm = {nfeat, 1.2f, 8, 25, 0, 0, 32, 20, 0, 2, mode >= 1 ? 1 : 0, masksOn ? 1 : 0, 32};
	mcs_extractor* ex = nullptr;
	MCSOK(mcs_extractor_create(ctx, &prm, W, H, ncam, &ex));
	int cap = 0;
	MCSOK(mcs_extractor_kp_capacity(ex, &cap));
	const size_t rows = (size_t)ncam * cap;
	// page-locked: the staged images + masks of one multi-frame, and two output sets (this multi-frame and the previous one, the matcher's train side)
	uint8_t *inImg = nullptr, *inMask = nullptr;
	MCSOK(mcs_host_alloc(ctx, ncam * plane, (void**)&inImg));
	MCSOK(mcs_host_alloc(ctx, ncam * plane, (void**)&inMask));
	memcpy(inMask, masks.data(), ncam * plane);
	// the rig's mirror masks never change: they stay on the device ("resident 0" in the config: uploaded with every call, as the reference's call shape would)
	const bool resident = geti("resident", 1) != 0;
	if (resident) MCSOK(mcs_extractor_set_masks(ex, ncam, inMask, plane, W, MCS_MEM_HOST));
	struct Out { int32_t* nkp; mcs_keypoint* kps; uint8_t *desc, *mask, *valid; double* rays; int32_t* match; };
	Out o[2];
	for (Out& s : o) {
		MCSOK(mcs_host_alloc(ctx, sizeof(int32_t) * ncam, (void**)&s.nkp));
		MCSOK(mcs_host_alloc(ctx, sizeof(mcs_keypoint) * rows, (void**)&s.kps));
		MCSOK(mcs_host_alloc(ctx, 32 * rows, (void**)&s.desc));
		MCSOK(mcs_host_alloc(ctx, 32 * rows, (void**)&s.mask));
		MCSOK(mcs_host_alloc(ctx, rows, (void**)&s.valid));
		MCSOK(mcs_host_alloc(ctx, sizeof(double) * 3 * rows, (void**)&s.rays));
		MCSOK(mcs_host_alloc(ctx, sizeof(int32_t) * rows, (void**)&s.match));
		memset(s.valid, 0, rows);
	}
	std::vector<double> tE, tM, tS;
	int32_t* counts = nullptr;   // [0] matches, [1] rescans: page-locked like the match array, so that the search writes all three with one launch
	MCSOK(mcs_host_alloc(ctx, 2 * sizeof(int32_t), (void**)&counts));
	counts[0] = counts[1] = 0;
	int32_t &nmatch = counts[0], &nfb = counts[1];
	for (int it = 0; it < warm + calls; ++it) {
		const int f = it % frames;
		Out& cur = o[it & 1];
		Out& prev = o[(it & 1) ^ 1];
		const auto t0 = std::chrono::steady_clock::now();
		for (int c = 0; c < ncam; ++c) memcpy(inImg + c * plane, images.data() + ((size_t)c * frames + f) * plane, plane);   // what a grabber's cv::Mat -> staging copy costs
		MCSOK(mcs_extract_batch(ex, ncam, inImg, plane, W, resident ? MCS_MASKS_RESIDENT : inMask, plane, W, cams, MCS_MEM_HOST, cur.nkp, cur.kps, cur.desc, cur.mask, cur.rays));
		// the multi-frame as ONE descriptor set: camera blocks of `cap` rows, the rows past a camera's count flagged invalid (every keypoint "has a map point")
		for (int c = 0; c < ncam; ++c) {
			memset(cur.valid + (size_t)c * cap, 1, (size_t)cur.nkp[c]);
			memset(cur.valid + (size_t)c * cap + cur.nkp[c], 0, (size_t)(cap - cur.nkp[c]));
		}
		const auto t1 = std::chrono::steady_clock::now();
		if (it > 0) {
			mcs_desc_set q, t;
			memset(&q, 0, sizeof(q)); memset(&t, 0, sizeof(t));
			q.desc = cur.desc; q.mask = masksOn ? cur.mask : nullptr; q.valid = cur.valid; q.n = (int)rows; q.stride = 32;
			t.desc = prev.desc; t.mask = masksOn ? prev.mask : nullptr; t.valid = prev.valid; t.n = (int)rows; t.stride = 32;
			MCSOK(mcs_search_kf_kf(ctx, 1, &q, 0, &t, 0, 32, ratio, topk, MCS_MEM_HOST, cur.match, &nmatch, &nfb));
		}
		const auto t2 = std::chrono::steady_clock::now();
		if (it >= warm) {
			tE.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
			tM.push_back(std::chrono::duration<double, std::milli>(t2 - t1).count());
			tS.push_back(std::chrono::duration<double, std::milli>(t2 - t0).count());
		}
	}
	const int last = (warm + calls - 1) & 1, lastFrame = (warm + calls - 1) % frames;
	if (cfg.count("out")) {
		const std::string p = cfg["out"];
		write_file(p + ".nkp", o[last].nkp, sizeof(int32_t) * ncam);
		write_file(p + ".kps", o[last].kps, sizeof(mcs_keypoint) * rows);
		write_file(p + ".desc", o[last].desc, 32 * rows);
		write_file(p + ".mask", o[last].mask, 32 * rows);
		write_file(p + ".match", o[last].match, sizeof(int32_t) * rows);
		write_file(p + ".prev_nkp", o[last ^ 1].nkp, sizeof(int32_t) * ncam);
	}
	int feats = 0;
	for (int c = 0; c < ncam; ++c) feats += o[last].nkp[c];
	const Stats e = stats(tE), m = stats(tM), s = stats(tS);
	printf("{\"masks_resident\": %s, \"calls\": %d, \"warmup\": %d, \"cap\": %d, \"features_last\": %d, \"matches_last\": %d, \"rescans_last\": %d, \"last_frame\": %d, "
	       "\"extract_ms\": {\"median\": %.4f, \"p90\": %.4f, \"p99\": %.4f, \"mean\": %.4f, \"min\": %.4f}, "
	       "\"match_ms\": {\"median\": %.4f, \"p90\": %.4f, \"p99\": %.4f, \"mean\": %.4f, \"min\": %.4f}, "
	       "\"total_ms\": {\"median\": %.4f, \"p90\": %.4f, \"p99\": %.4f, \"mean\": %.4f, \"min\": %.4f}}\n",
	       resident ? "true" : "false", calls, warm, cap, feats, nmatch, nfb, lastFrame, e.med, e.p90, e.p99, e.mean, e.mn, m.med, m.p90, m.p99, m.mean, m.mn, s.med, s.p90, s.p99, s.mean, s.mn);
	for (Out& s2 : o) { mcs_host_free(ctx, s2.nkp); mcs_host_free(ctx, s2.kps); mcs_host_free(ctx, s2.desc); mcs_host_free(ctx, s2.mask); mcs_host_free(ctx, s2.valid); mcs_host_free(ctx, s2.rays); mcs_host_free(ctx, s2.match); }
	mcs_host_free(ctx, inImg); mcs_host_free(ctx, inMask); mcs_host_free(ctx, counts);
	MCSOK(mcs_extractor_destroy(ex));
	MCSOK(mcs_ctx_destroy(ctx));
	return 0;
}

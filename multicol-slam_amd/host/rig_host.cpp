// rig_host.cpp — native host of the camera-sharded rig: ONE process, one host thread + one mcs_ctx per GPU, RCCL communicators from ncclCommInitAll.
//
// The reference runs the cameras of a multi-frame on one OpenMP thread each inside one process (#pragma omp parallel for num_threads(nrCams),
// src/cMultiFrame.cpp:128-164); this is the same shape with a GPU behind every thread.  Everything the step computes goes through the C ABI of
// libmcs_hip.so (include/mcs_c.h); the exchange is issued HERE, on the context's own stream, with the caller's communicator — the library never links RCCL:
//
//     mcs_extract_batch_strided   this rank's slab of (camera, frame) images -> descriptor | mask rows in exchange blocks      (all device memory)
//     mcs_rig_pack_headers        keypoint counts into the blocks' header rows
//     ncclAllGather               database sweeps (every rank needs every multi-frame), or
//     ncclSend / ncclRecv group   frame ring (a rank needs its own frames and one predecessor: multicol-slam_amd/rig.py RingExchange, restated below)
//     mcs_rig_rows_valid          row flags from the received headers
//     mcs_search_kf_f_sweep       every multi-frame x this rank's stored keyframes (k -> rank k mod N), or
//     mcs_search_kf_kf_ring       this rank's frames against their predecessors
//
// Input: a key / value text file (see tests/test_gpu_rig_host.py) naming raw binary inputs — images [camera][frame][H][W] u8 for the whole job, one mask per
// camera, one mcs_ocam per camera.  Output: per rank its received array, keypoints, counts and match arrays as raw files + one JSON line on stdout
// (steps, ms per step = max over ranks between two barriers, and each rank's own).  No Python, no torch.
// The step is pipelined like bench.py's: three buffer sets, the exchange of step n on a second stream beside the extraction of step n + 1, the searches deferred
// (mcs_ctx_set_async_search) and fenced before their inputs are reused.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <pthread.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcs_c.h"

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define NCCLOK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, ncclGetErrorString(r_)); exit(3); } } while (0)
#define MCSOK(x) do { int r_ = (x); if (r_ != MCS_OK) { fprintf(stderr, "%s:%d %s: %d %s\n", __FILE__, __LINE__, #x, r_, mcs_last_error()); exit(4); } } while (0)

// ---- the layout arithmetic of multicol-slam_amd/rig.py (RigLayout, RingExchange), restated -------------------------------------------------------------
struct Layout {
	int ncam, FT, world, cap, ds;
	int images_total, L, rows_img, row_stride, rows_frame;
	size_t block_bytes, send_bytes;
	Layout(int ncam_, int FT_, int world_, int cap_, int ds_ = 32) : ncam(ncam_), FT(FT_), world(world_), cap(cap_), ds(ds_) {
		images_total = ncam * FT; L = images_total / world; rows_img = cap + 1; row_stride = 2 * ds; rows_frame = ncam * cap;
		block_bytes = (size_t)rows_img * row_stride; send_bytes = (size_t)L * block_bytes;
	}
	int image_index(int cam, int frame) const { return cam * FT + frame; }
	// multi-frame `frame` of a [camera][FT][rows_img] array as a block-structured descriptor set
	mcs_desc_set frame_set(const uint8_t* G, const uint8_t* valid, int frame, bool masks) const {
		mcs_desc_set s;
		memset(&s, 0, sizeof(s));
		s.desc = G + (size_t)frame * block_bytes; s.mask = masks ? s.desc + ds : nullptr; s.valid = valid + (size_t)frame * rows_img; s.group = nullptr;
		s.n = rows_frame; s.stride = row_stride; s.block_rows = cap; s.block_pitch_rows = (int64_t)FT * rows_img;
		return s;
	}
};
struct Run { int owner, src, dst, n; };   // blocks [src, src+n) of the owner's send buffer -> blocks [dst, dst+n) of the receiver's local array
static std::vector<Run> ring_runs(const Layout& lay, int rank) {
	const int F = lay.FT / lay.world;
	auto gframe = [&](int j) { return ((rank * F - 1 + j) % lay.FT + lay.FT) % lay.FT; };
	std::vector<Run> out;
	for (int c = 0; c < lay.ncam; ++c)
		for (int j = 0; j <= F;) {
			const int x = lay.image_index(c, gframe(j)), owner = x / lay.L;
			int n = 1;
			while (j + n <= F && lay.image_index(c, gframe(j + n)) == x + n && (x + n) / lay.L == owner) ++n;
			out.push_back({owner, x - owner * lay.L, c * (F + 1) + j, n});
			j += n;
		}
	return out;
}

// The plan's digest — the same 32 bytes multicol-slam_amd/rig.py plan_digest computes (four 64-bit FNV-1a hashes over the plan's integers as little-endian int64:
// arguments, every rank's slab, every ring run or keyframe shard, every frame pair), here from the C++ restatement above.  `bench.py --dry-run` prints it for the
// plan rig.plan_check has verified; every rank of a run computes it before its first exchange and the ranks compare (rank_main).
static std::string plan_digest(int ncam, int F, int world, int cap, int D, int ds, int topk) {
	const Layout lay(ncam, F * world, world, cap, ds);
	std::vector<int64_t> v = {ncam, F, world, cap, D, ds, topk, lay.L, lay.rows_img, lay.row_stride, (int64_t)lay.block_bytes, (int64_t)lay.send_bytes};
	for (int r = 0; r < world; ++r)
		for (int x = r * lay.L; x < (r + 1) * lay.L; ++x) { v.push_back(x / lay.FT); v.push_back(x % lay.FT); }
	if (D == 0) {
		for (int r = 0; r < world; ++r) {
			for (const Run& q : ring_runs(lay, r)) { v.push_back(q.owner); v.push_back(q.src); v.push_back(q.dst); v.push_back(q.n); }
			for (int f = r * F; f < (r + 1) * F; ++f) { v.push_back(f); v.push_back(((f - 1) % lay.FT + lay.FT) % lay.FT); }
		}
	} else {
		for (int r = 0; r < world; ++r) {
			std::vector<int64_t> sh;
			for (int k = 0; k < D; ++k) if (k % world == r) sh.push_back(k);
			v.push_back((int64_t)sh.size());
			v.insert(v.end(), sh.begin(), sh.end());
		}
	}
	static const uint64_t bases[4] = {0xCBF29CE484222325ull, 0x84222325CBF29CE4ull, 0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full};
	std::string hex;
	for (uint64_t h : bases) {
		for (int64_t x : v)
			for (int b = 0; b < 8; ++b) h = (h ^ (uint64_t)(((uint64_t)x >> (8 * b)) & 0xFF)) * 0x100000001B3ull;
		for (int b = 0; b < 8; ++b) { char t[3]; snprintf(t, sizeof t, "%02x", (unsigned)((h >> (8 * b)) & 0xFF)); hex += t; }
	}
	return hex;
}
static std::vector<std::string> g_digest;   // per rank, compared before the first exchange

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
template <class T> static T* dalloc(size_t n) { void* p = nullptr; HIPOK(hipMalloc(&p, std::max<size_t>(n * sizeof(T), 16))); HIPOK(hipMemset(p, 0, std::max<size_t>(n * sizeof(T), 16))); return (T*)p; }

struct Job {
	int ncam, W, H, nfeat, mode, F, D, world, steps, warmup, topk;
	std::vector<uint8_t> images, masks, cams;
	std::string out;
};

static pthread_barrier_t g_barrier;
static std::atomic<int> g_firstExchange{0};   // ranks whose first exchange has completed (the start-up watchdog in main waits for all of them)
static std::vector<unsigned> g_xconf;
static std::vector<long> g_ties;   // rounding-tie rows recomputed on the host inside the loop, per rank   // per rank: which of the context's streams the exchange stream shares a hardware queue with (mcs_ctx_stream_conflicts)
static std::vector<double> g_ms, g_msAll;   // per rank: its own step time, and the time to the barrier behind the slowest rank

static void rank_main(const Job& J, int rank, ncclComm_t comm) {
	HIPOK(hipSetDevice(rank));
	hipStream_t stream;
	HIPOK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
	mcs_ctx* ctx = nullptr;
	MCSOK(mcs_ctx_create(rank, stream, &ctx));
	const bool masksOn = J.mode == 2;
	mcs_extractor_params prm = {J.nfeat, 1.2f, 8, 25, 0, 0, 32, 20, 0, 2, J.mode >= 1 ? 1 : 0, masksOn ? 1 : 0, 32};
	const int FT = J.F * J.world;
	mcs_extractor* probe = nullptr;
	MCSOK(mcs_extractor_create(ctx, &prm, J.W, J.H, 1, &probe));
	int cap = 0;
	MCSOK(mcs_extractor_kp_capacity(probe, &cap));
	MCSOK(mcs_extractor_destroy(probe));
	const Layout lay(J.ncam, FT, J.world, cap);
	const bool ring = J.D == 0;
	const Layout view = ring ? Layout(J.ncam, J.F + 1, 1, cap) : lay;   // what this rank holds after the exchange
	mcs_extractor* ex = nullptr;
	MCSOK(mcs_extractor_create(ctx, &prm, J.W, J.H, lay.L, &ex));

	// ---- this rank's slab: images x in [rank*L, (rank+1)*L) of the camera-major job, their masks and camera models
	const size_t px = (size_t)J.W * J.H;
	uint8_t* d_img = dalloc<uint8_t>(lay.L * px);
	uint8_t* d_msk = dalloc<uint8_t>(lay.L * px);
	std::vector<mcs_ocam> cam(lay.L);
	HIPOK(hipMemcpy(d_img, J.images.data() + (size_t)rank * lay.L * px, lay.L * px, hipMemcpyHostToDevice));
	for (int i = 0; i < lay.L; ++i) {
		const int c = (rank * lay.L + i) / FT;
		HIPOK(hipMemcpy(d_msk + i * px, J.masks.data() + c * px, px, hipMemcpyHostToDevice));
		memcpy(&cam[i], J.cams.data() + (size_t)c * sizeof(mcs_ocam), sizeof(mcs_ocam));
	}
	// ---- three buffer sets in rotation (the schedule of bench.py's Job.step): step n extracts set n % 3, then patches the rounding-tie rows of set (n - 1) % 3 on
	// the host and STARTS that set's exchange on a second stream — beside step n's kernels —; its rows are flagged and matched stream-ordered behind it, as a deferred search
	// (mcs_ctx_set_async_search: lists + greedy pass on the library's own stream beside the next extraction), so results are one step late.  The search that last
	// read a set is fenced (mcs_ctx_search_fence) before the set is extracted into again: nothing of a search's inputs is overwritten while it runs.
	constexpr int NS = 3;
	struct Set {
		uint8_t *send, *G, *valid; int32_t *nkp, *nkpAll; mcs_keypoint* kps; double* rays; int32_t *match, *nmatch, *fb;
		hipEvent_t evExtracted, evExchanged;
	};
	std::vector<int> kfs;
	for (int k = 0; k < J.D; ++k) if (k % J.world == rank) kfs.push_back(k);
	const int nkf = (int)kfs.size(), npairs = ring ? J.F : FT * std::max(nkf, 1);
	Set sets[NS];
	for (Set& b : sets) {
		b.send = dalloc<uint8_t>(lay.send_bytes);
		b.G = dalloc<uint8_t>(view.images_total * view.block_bytes);
		b.valid = dalloc<uint8_t>((size_t)view.images_total * view.rows_img);
		b.nkp = dalloc<int32_t>(lay.L);
		b.nkpAll = dalloc<int32_t>(view.images_total);
		b.kps = dalloc<mcs_keypoint>((size_t)lay.L * cap);
		b.rays = dalloc<double>((size_t)lay.L * cap * 3);
		b.match = dalloc<int32_t>((size_t)npairs * lay.rows_frame);
		b.nmatch = dalloc<int32_t>(npairs);
		b.fb = dalloc<int32_t>(npairs);
		HIPOK(hipEventCreateWithFlags(&b.evExtracted, hipEventDisableTiming));
		HIPOK(hipEventCreateWithFlags(&b.evExchanged, hipEventDisableTiming));
	}
	// the exchange's own stream: the collective of step n runs beside the extraction kernels of step n + 1.  Which hardware queue a stream gets is the runtime's
	// choice and a queue runs in order: the library probes candidates until one keeps clear of the queues the extraction runs on (mcs_ctx_transfer_stream)
	hipStream_t xstream;
	unsigned xconf = 0;
	MCSOK(mcs_ctx_transfer_stream(ctx, (void**)&xstream, &xconf));
	g_xconf[rank] = xconf;
	uint8_t* d_db = dalloc<uint8_t>((size_t)std::max(nkf, 1) * lay.rows_frame * lay.row_stride);
	uint8_t* d_dbValid = dalloc<uint8_t>((size_t)std::max(nkf, 1) * lay.rows_frame);

	const std::vector<Run> mine = ring ? ring_runs(lay, rank) : std::vector<Run>();
	std::vector<std::vector<Run> > theirs(J.world);   // what every destination needs (to find my sends, in the receiver's order)
	if (ring) for (int r = 0; r < J.world; ++r) theirs[r] = ring_runs(lay, r);

	auto extract = [&](Set& b) {
		MCSOK(mcs_extract_batch_strided(ex, lay.L, d_img, px, J.W, d_msk, px, J.W, cam.data(), b.nkp, b.kps, b.send, b.send + lay.ds, b.rays, lay.rows_img, lay.row_stride));
		MCSOK(mcs_rig_pack_headers(ctx, b.nkp, lay.L, cap, b.send, lay.row_stride));
		HIPOK(hipEventRecord(b.evExtracted, stream));
	};
	auto exchange_begin = [&](Set& b) {   // on the exchange stream, behind this set's extraction
		HIPOK(hipStreamWaitEvent(xstream, b.evExtracted, 0));
		if (!ring) NCCLOK(ncclAllGather(b.send, b.G, lay.send_bytes, ncclUint8, comm, xstream));
		else {
			NCCLOK(ncclGroupStart());
			for (const Run& r : mine) NCCLOK(ncclRecv(b.G + (size_t)r.dst * lay.block_bytes, (size_t)r.n * lay.block_bytes, ncclUint8, r.owner, comm, xstream));
			for (int dest = 0; dest < J.world; ++dest)
				for (const Run& r : theirs[dest])
					if (r.owner == rank) NCCLOK(ncclSend(b.send + (size_t)r.src * lay.block_bytes, (size_t)r.n * lay.block_bytes, ncclUint8, dest, comm, xstream));
			NCCLOK(ncclGroupEnd());
		}
		HIPOK(hipEventRecord(b.evExchanged, xstream));
	};
	auto exchange_end = [&](Set& b) {     // the context's stream continues behind the exchange: flags from the received headers
		HIPOK(hipStreamWaitEvent(stream, b.evExchanged, 0));
		MCSOK(mcs_rig_rows_valid(ctx, b.G, view.images_total, cap, lay.row_stride, b.valid, b.nkpAll));
	};
	auto match = [&](Set& b) {
		const mcs_desc_set fr = view.frame_set(b.G, b.valid, 0, masksOn);
		if (ring) MCSOK(mcs_search_kf_kf_ring(ctx, J.F + 1, 1, J.F, &fr, view.rows_img, 32, 0.9, J.topk, MCS_MEM_DEVICE, b.match, b.nmatch, b.fb));
		else if (nkf) {
			mcs_desc_set kf;
			memset(&kf, 0, sizeof(kf));
			kf.desc = d_db; kf.mask = masksOn ? d_db + lay.ds : nullptr; kf.valid = d_dbValid; kf.n = lay.rows_frame; kf.stride = lay.row_stride;
			MCSOK(mcs_search_kf_f_sweep(ctx, nkf, &kf, lay.rows_frame, FT, &fr, lay.rows_img, 32, 0.9, J.topk, MCS_MEM_DEVICE, b.match, b.nmatch, b.fb));
		}
	};

	if (!ring) {   // untimed: stored keyframe k = multi-frame k % FT of one pass (as bench.py fills its database)
		Set& b = sets[0];
		extract(b);
		HIPOK(hipStreamSynchronize(stream));
		MCSOK(mcs_extractor_fix_ties(ex, nullptr));   // rounding-tie rows of the stored keyframes: the host libm's (synchronous form, untimed pass)
		exchange_begin(b); exchange_end(b);
		for (int j = 0; j < nkf; ++j)
			for (int c = 0; c < J.ncam; ++c) {
				const size_t x = lay.image_index(c, kfs[j] % FT);
				HIPOK(hipMemcpyAsync(d_db + ((size_t)j * lay.rows_frame + (size_t)c * cap) * lay.row_stride, b.G + x * lay.block_bytes, (size_t)cap * lay.row_stride, hipMemcpyDeviceToDevice, stream));
				HIPOK(hipMemcpyAsync(d_dbValid + (size_t)j * lay.rows_frame + (size_t)c * cap, b.valid + x * lay.rows_img, cap, hipMemcpyDeviceToDevice, stream));
			}
		HIPOK(hipStreamSynchronize(stream));
	}
	// before the first exchange: this rank's digest of the whole plan, printed and compared across the ranks (a mismatch names the rank and ends the run, exit 7)
	g_digest[rank] = plan_digest(J.ncam, J.F, J.world, cap, J.D, 32, J.topk);
	fprintf(stderr, "rig_host: rank %d of %d plan digest %s\n", rank, J.world, g_digest[rank].c_str());
	pthread_barrier_wait(&g_barrier);
	for (int r = 0; r < J.world; ++r)
		if (g_digest[r] != g_digest[0]) {
			if (rank == 0) fprintf(stderr, "rig_host: PLAN DIGEST MISMATCH — rank %d holds %s, rank 0 holds %s\n", r, g_digest[r].c_str(), g_digest[0].c_str());
			pthread_barrier_wait(&g_barrier);
			_exit(7);
		}
	MCSOK(mcs_ctx_set_async_search(ctx, 1));
	// Rounding ties are enforced INSIDE the loop (include/mcs_c.h: mcs_extractor_set_tie_capture): a capture slot per buffer set; the rows of step n - 1 are patched
	// on the host — behind that batch's event only, the device already runs step n — BEFORE they leave for the other ranks and before their search is enqueued.
	MCSOK(mcs_extractor_set_tie_capture(ex, NS, 256));
	int cur = 0;
	bool pending = false;   // the set before `cur` has been extracted and nobody has exchanged / matched it yet
	long tiesPatched = 0;
	auto finish = [&](Set& p) {               // patch (host), exchange, flags, search of the set extracted one call earlier
		int listed = 0, fixed = 0;
		MCSOK(mcs_extractor_patch_ties(ex, 1, &listed, &fixed));
		tiesPatched += fixed;
		exchange_begin(p); exchange_end(p); match(p);
	};
	auto step = [&]() {
		Set& b = sets[cur];
		Set& p = sets[(cur + NS - 1) % NS];
		cur = (cur + 1) % NS;
		MCSOK(mcs_ctx_search_fence(ctx, 1));   // the search issued before the latest one read the set about to be overwritten (three sets, matching one step late)
		extract(b);                            // enqueued FIRST: when the previous extraction ends the device has this one queued, the host wait below costs it nothing
		if (pending) finish(p);                // its collective runs on the exchange stream beside this step's kernels, the search stream-ordered behind it
		pending = true;
	};
	auto drain = [&]() -> Set& {              // finish the step in flight: its patch, exchange, flags and search; returns the set whose results are complete
		Set& p = sets[(cur + NS - 1) % NS];
		if (pending) {
			int listed = 0, fixed = 0;
			MCSOK(mcs_extractor_patch_ties(ex, 0, &listed, &fixed));
			tiesPatched += fixed;
			exchange_begin(p); exchange_end(p); match(p); pending = false;
		}
		MCSOK(mcs_ctx_join(ctx));
		MCSOK(mcs_ctx_synchronize(ctx));
		HIPOK(hipStreamSynchronize(xstream));
		return p;
	};
	step();                                    // prime the pipeline: the first timed step matches what is exchanged here
	for (int i = 0; i < J.warmup; ++i) step();
	MCSOK(mcs_ctx_synchronize(ctx));
	HIPOK(hipStreamSynchronize(xstream));
	g_firstExchange.fetch_add(1);               // this rank's first exchanges have completed: the start-up watchdog (main) retires once every rank is here
	pthread_barrier_wait(&g_barrier);
	const auto t0 = std::chrono::steady_clock::now();
	for (int i = 0; i < J.steps; ++i) step();   // each: one extraction, one exchange, one matching pass
	MCSOK(mcs_ctx_synchronize(ctx));
	HIPOK(hipStreamSynchronize(xstream));
	g_ms[rank] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / std::max(J.steps, 1);
	pthread_barrier_wait(&g_barrier);
	g_msAll[rank] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / std::max(J.steps, 1);
	Set& R = drain();                          // untimed: the last extracted set through its exchange and search
	g_ties[rank] = tiesPatched;
	MCSOK(mcs_extractor_status(ex));
	uint8_t *d_G = R.G, *d_valid = R.valid; int32_t *d_nkp = R.nkp, *d_match = R.match, *d_nmatch = R.nmatch; mcs_keypoint* d_kps = R.kps;

	// ---- results of the last step, as this rank holds them
	auto dump = [&](const char* name, const void* dptr, size_t bytes) {
		std::vector<uint8_t> h(bytes);
		HIPOK(hipMemcpy(h.data(), dptr, bytes, hipMemcpyDeviceToHost));
		write_file(J.out + ".r" + std::to_string(rank) + "." + name, h.data(), bytes);
	};
	dump("G", d_G, view.images_total * view.block_bytes);
	dump("valid", d_valid, (size_t)view.images_total * view.rows_img);
	dump("kps", d_kps, (size_t)lay.L * cap * sizeof(mcs_keypoint));
	dump("nkp", d_nkp, lay.L * sizeof(int32_t));
	dump("match", d_match, (size_t)npairs * lay.rows_frame * sizeof(int32_t));
	dump("nmatch", d_nmatch, npairs * sizeof(int32_t));
	if (!ring && nkf) { dump("db", d_db, (size_t)nkf * lay.rows_frame * lay.row_stride); dump("dbvalid", d_dbValid, (size_t)nkf * lay.rows_frame); }
	if (rank == 0) {
		FILE* f = fopen((J.out + ".layout").c_str(), "w");
		fprintf(f, "cap %d\nview_frames %d\nrows_img %d\nrow_stride %d\nL %d\n", cap, view.FT, view.rows_img, view.row_stride, lay.L);
		fclose(f);
	}
	MCSOK(mcs_extractor_destroy(ex));
	MCSOK(mcs_ctx_destroy(ctx));
	for (Set& b : sets) {
		for (void* p : {(void*)b.send, (void*)b.G, (void*)b.valid, (void*)b.nkp, (void*)b.nkpAll, (void*)b.kps, (void*)b.rays, (void*)b.match, (void*)b.nmatch, (void*)b.fb}) (void)hipFree(p);
		(void)hipEventDestroy(b.evExtracted); (void)hipEventDestroy(b.evExchanged);
	}
	for (void* p : {(void*)d_img, (void*)d_msk, (void*)d_db, (void*)d_dbValid}) (void)hipFree(p);
	HIPOK(hipStreamDestroy(stream));
}

int main(int argc, char** argv) {
	if (argc < 2) { fprintf(stderr, "usage: rig_host <config> [--gpus N]\n"); return 1; }
	// (round 3 asked the runtime for 8 hardware queues here so that the exchange stream would not share one with the extraction; the exchange stream is now CHOSEN by
	// probing the default four queues — mcs_ctx_transfer_stream — which is faster: 1.59 against 1.98 ms per step at world size 1, the extra queues slow the step itself)
	auto cfg = read_config(argv[1]);
	auto geti = [&](const char* k, int def) { return cfg.count(k) ? atoi(cfg[k].c_str()) : def; };
	Job J;
	J.ncam = geti("ncam", 3); J.W = geti("width", 754); J.H = geti("height", 480); J.nfeat = geti("nfeatures", 1000); J.mode = geti("mode", 2);
	J.F = geti("frames", 2); J.D = geti("keyframes", 0); J.steps = geti("steps", 2); J.warmup = geti("warmup", 1); J.topk = geti("topk", 32);
	for (int i = 2; i < argc; ++i)
		if (!strcmp(argv[i], "--plan-only")) {   // no device needed: the digest of the plan this configuration would run (`cap N` in the configuration: rows per image)
			int world = geti("gpus", 1);
			for (int k = 2; k + 1 < argc; ++k) if (!strcmp(argv[k], "--gpus")) world = atoi(argv[k + 1]);
			if (world < 1 || (J.ncam * J.F * world) % world) { fprintf(stderr, "bad plan\n"); return 1; }
			printf("{\"plan_digest\": \"%s\", \"world\": %d}\n", plan_digest(J.ncam, J.F, world, geti("cap", J.nfeat + 24), J.D, 32, J.topk).c_str(), world);
			return 0;
		}
	int ndev = 0;
	HIPOK(hipGetDeviceCount(&ndev));
	// `gpus N` in the configuration or `--gpus N` on the command line (the command line wins): the job runs on exactly N GPUs or not at all — a run that quietly
	// used fewer would report a scaling point nobody measured
	int want = geti("gpus", ndev);
	for (int i = 2; i + 1 < argc; ++i) if (!strcmp(argv[i], "--gpus")) want = atoi(argv[i + 1]);
	if (want < 1) { fprintf(stderr, "rig_host: --gpus must be >= 1\n"); return 1; }
	if (want > ndev) { fprintf(stderr, "rig_host: %d GPUs requested but only %d visible (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?)\n", want, ndev); return 5; }
	J.world = want;
	if (J.ncam < 1 || J.F < 1 || J.steps < 0 || J.warmup < 0 || J.D < 0 || J.topk < 1) { fprintf(stderr, "bad configuration: ncam >= 1, frames >= 1, keyframes >= 0, topk >= 1\n"); return 1; }
	J.images = read_file(cfg["images"]); J.masks = read_file(cfg["masks"]); J.cams = read_file(cfg["cams"]); J.out = cfg["out"];
	const size_t need = (size_t)J.ncam * J.F * J.world * J.W * J.H;
	if (J.images.size() != need || J.masks.size() != (size_t)J.ncam * J.W * J.H || J.cams.size() != (size_t)J.ncam * sizeof(mcs_ocam)) {
		fprintf(stderr, "input sizes do not match the configuration (images %zu, want %zu)\n", J.images.size(), need);
		return 1;
	}
	std::vector<ncclComm_t> comms(J.world);
	std::vector<int> devs(J.world);
	for (int i = 0; i < J.world; ++i) devs[i] = i;
	// The first multi-rank RCCL calls of a deployment are where an environment problem shows (IPC mode, visible devices, a peer that cannot be mapped): a hang there
	// must end loudly.  The watchdog covers communicator creation and every rank's first exchange (g_firstExchange counts them); after that it retires.
	std::thread dog([&] {
		for (int i = 0; i < 600 && g_firstExchange.load() < J.world; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(100));
		if (g_firstExchange.load() >= J.world) return;
		fprintf(stderr, "rig_host: WATCHDOG — %d of %d ranks finished their first exchange within 60 s; environment:\n", g_firstExchange.load(), J.world);
		for (const char* k : {"HSA_ENABLE_IPC_MODE_LEGACY", "GPU_MAX_HW_QUEUES", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "NCCL_DEBUG", "NCCL_SOCKET_IFNAME", "NCCL_P2P_DISABLE",
		                      "NCCL_SHM_DISABLE", "NCCL_IB_DISABLE", "RCCL_MSCCL_ENABLE", "HSA_FORCE_FINE_GRAIN_PCIE"})
			fprintf(stderr, "  %s=%s\n", k, getenv(k) ? getenv(k) : "(unset)");
		fflush(stderr);
		_exit(6);
	});
	NCCLOK(ncclCommInitAll(comms.data(), J.world, devs.data()));
	pthread_barrier_init(&g_barrier, nullptr, J.world);
	g_ms.assign(J.world, 0.0); g_msAll.assign(J.world, 0.0); g_xconf.assign(J.world, 0u); g_ties.assign(J.world, 0); g_digest.assign(J.world, std::string());
	std::vector<std::thread> th;
	for (int r = 0; r < J.world; ++r) th.emplace_back(rank_main, std::cref(J), r, comms[r]);
	for (auto& t : th) t.join();
	g_firstExchange.store(J.world);
	dog.join();
	for (auto c : comms) NCCLOK(ncclCommDestroy(c));
	const double ms = *std::max_element(g_msAll.begin(), g_msAll.end());   // between the two barriers: the slowest rank
	std::string per = "[";
	for (int r = 0; r < J.world; ++r) { char t[32]; snprintf(t, sizeof t, "%s%.4f", r ? ", " : "", g_ms[r]); per += t; }
	per += "]";
	printf("{\"host\": \"rig_host (C++, one process, one thread per GPU, RCCL from ncclCommInitAll; three buffer sets, exchange on its own stream, matching one step late)\", "
	       "\"n_gpus\": %d, \"steps\": %d, \"ms_per_step\": %.4f, \"ms_per_step_ranks\": %s, "
	       "\"exchange\": \"%s\", \"exchange_stream_queue_conflicts_rank0\": %u, \"multi_frames_per_step_per_gpu\": %d, \"stored_keyframes\": %d, "
	       "\"ties_patched_in_loop\": true, \"ties_recomputed_on_the_host_rank0\": %ld, \"plan_digest_all_ranks_agree\": \"%s\"}\n",
	       J.world, J.steps, ms, per.c_str(), J.D == 0 ? "ncclSend/ncclRecv group (frame ring)" : "ncclAllGather", g_xconf[0], J.F, J.D, g_ties[0], g_digest[0].c_str());
	return 0;
}

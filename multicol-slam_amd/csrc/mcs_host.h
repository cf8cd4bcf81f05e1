// mcs_host.h — host-side internals shared by the C-ABI translation units (context, error plumbing).
#pragma once
#include "mcs_common.h"
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

std::string& mcs_err();
inline int fail(int code, const std::string& msg) { mcs_err() = msg; return code; }
#define HIPCHK(expr)                                                                                       \
	do {                                                                                                   \
		hipError_t _e = (expr);                                                                            \
		if (_e != hipSuccess) return fail(MCS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
	} while (0)

static inline int cvRound_(double v) { return (int)lrint(v); }
static inline int cvRoundf_(float v) { return (int)lrintf(v); }
static inline int cvFloor_(double v) { int i = (int)v; return i - (i > v); }
static inline short sat_short(float v) { int iv = cvRoundf_(v); return (short)(iv < -32768 ? -32768 : iv > 32767 ? 32767 : iv); }

struct Timer { hipEvent_t a = nullptr, b = nullptr; bool used = false; };

// the device's view of a page-locked host pointer, or nullptr (pageable memory, another device's allocation): host-kind calls write their outputs straight
// into page-locked arrays with one launch instead of one runtime copy per array (~15 us each for the small arrays of ONE multi-frame)
inline void* device_view(const void* host) {
	if (!host) return nullptr;
	hipPointerAttribute_t a;
	memset(&a, 0, sizeof(a));
	if (hipPointerGetAttributes(&a, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
	return a.devicePointer;
}

struct mcs_ctx {
	int device = 0;
	std::vector<mcs_extractor*> extractors;   // live extractors built on this context: mcs_ctx_destroy releases them (their buffers and stream are the context's)
	hipStream_t stream = nullptr;
	bool ownStream = false;
	bool timing = false;
	std::map<std::string, Timer> timers;
	// matcher scratch
	uint32_t* partial = nullptr; size_t partialCap = 0;
	int* partialCount = nullptr; size_t partialCountCap = 0;
	uint8_t* stage = nullptr; size_t stageCap = 0;   // host-kind staging for the matcher
	int* dscalar = nullptr;
	uint32_t* topKeys = nullptr; size_t topKeysCap = 0;   // packed [set][K][nq] top-K lists feeding the greedy kernels
	uint32_t* topKeys2 = nullptr; size_t topKeys2Cap = 0; // second list buffer of the deferred searches: the greedy pass of search n reads one while the matcher of search n + 1 fills the other
	int* topCnt = nullptr; size_t topCntCap = 0;
	uint8_t* exA = nullptr; size_t exACap = 0;            // train sets expanded to matrix-core operands (mcs_match_mfma.hip), per call
	uint8_t* exW = nullptr; size_t exWCap = 0;
	int* exRows = nullptr; size_t exRowsCap = 0;
	uint8_t* stageOut = nullptr; size_t stageOutCap = 0;
	uint8_t* pinned = nullptr; size_t pinnedCap = 0;      // page-locked host mirror of the arena's staged inputs (PinnedUpload)
	uint8_t* arena = nullptr; size_t arenaCap = 0;        // scratch + host-kind staging of the window / projection / map-point entry points (mcs_capi_window.hip)
	// Second HIP stream for the latency-bound / independent kernels (blur next to FAST+oct-tree, the greedy resolution next to the
	// following batch's extraction): they leave most CUs idle, so overlapping them with the VALU-bound kernels is free throughput.
	hipStream_t side = nullptr;    // extraction fork: resize chain + blur beside FAST + oct-tree
	hipStream_t side3 = nullptr;   // deferred searches: the greedy pass, beside the next search's lists on side2
	hipEvent_t evLists = nullptr, evGreedyBuf[2] = {nullptr, nullptr};   // lists of the latest deferred search complete / the greedy pass that read list buffer i complete
	hipStream_t upload = nullptr; unsigned uploadMask = 0; std::vector<hipStream_t> probed; std::vector<unsigned> probedMask;   // mcs_ctx_transfer_stream (mcs_copy.hip)
	hipStream_t side2 = nullptr;   // the greedy match resolution (its own stream: it must not hold up the next batch's resize chain)
	hipEvent_t evFork = nullptr, evPyr1 = nullptr, evPyr = nullptr, evBlur = nullptr, evMatch = nullptr, evGreedy = nullptr;
	hipEvent_t evDescFork = nullptr, evDescJoin = nullptr;   // the exact descriptor pass over the pre-list on `side`, beside the fast pass
	bool greedyPending = false;
	// deferred searches (mcs_ctx_set_async_search): top-K lists AND greedy pass of a device-memory search on side2, completion events in a ring
	bool asyncSearch = false;
	hipEvent_t evSearch[4] = {nullptr, nullptr, nullptr, nullptr};
	long long searchSeq = 0;
	hipStream_t lastResultStream = nullptr;   // the stream the LATEST device-memory search completed its outputs on (search_common records it); nullptr: none yet
	bool overlap() const { return side != nullptr && !timing; }   // per-kernel timing runs everything in order on the main stream

	void tic(const char* name) {
		if (!timing) return;
		Timer& t = timers[name];
		if (!t.a) { (void)hipEventCreate(&t.a); (void)hipEventCreate(&t.b); }
		(void)hipEventRecord(t.a, stream);
	}
	void toc(const char* name) {
		if (!timing) return;
		Timer& t = timers[name];
		(void)hipEventRecord(t.b, stream);
		t.used = true;
	}
};

// The context's persistent scratch buffer, at least `bytes` long.  Every call on a context runs on the context's stream, so a later call's copies and
// kernels are ordered behind the earlier call's use of it; growing goes through hipFree, which waits for the device.
inline hipError_t ctx_arena(mcs_ctx* c, size_t bytes, uint8_t** out) {
	if (c->arenaCap < bytes) {
		if (c->arena) (void)hipFree(c->arena);
		c->arena = nullptr; c->arenaCap = 0;
		const hipError_t e = hipMalloc((void**)&c->arena, bytes + bytes / 2);
		if (e != hipSuccess) return e;
		c->arenaCap = bytes + bytes / 2;
	}
	*out = c->arena;
	return hipSuccess;
}

// Host-kind inputs of one call, gathered in a page-locked mirror of the arena and sent with ONE H2D copy: a dozen small hipMemcpyAsync calls from
// pageable memory cost ~20 us of runtime overhead each, more than the kernels of a single multi-frame.  Only for calls that end with a stream
// synchronisation (the mirror is reused by the next call).
struct PinnedUpload {
	uint8_t* dev = nullptr; uint8_t* pin = nullptr; size_t lo = ~size_t(0), hi = 0;
	hipError_t begin(mcs_ctx* c, uint8_t* devBase, size_t total) {
		if (c->pinnedCap < total) {
			if (c->pinned) (void)hipHostFree(c->pinned);
			c->pinned = nullptr; c->pinnedCap = 0;
			const hipError_t e = hipHostMalloc((void**)&c->pinned, total + total / 2, hipHostMallocDefault);
			if (e != hipSuccess) return e;
			c->pinnedCap = total + total / 2;
		}
		dev = devBase; pin = c->pinned;
		return hipSuccess;
	}
	void put(size_t off, const void* src, size_t bytes) {
		if (!bytes) return;
		memcpy(pin + off, src, bytes);
		lo = off < lo ? off : lo; hi = off + bytes > hi ? off + bytes : hi;
	}
	hipError_t flush(hipStream_t s) { return hi > lo ? hipMemcpyAsync(dev + lo, pin + lo, hi - lo, hipMemcpyHostToDevice, s) : hipSuccess; }
};

// mcs_copy.hip — results leave for page-locked host memory through a copy kernel of a FEW workgroups, on the stream that completes them.
//
// A live front end gets its images from the host and hands keypoints, descriptors and matches back (src/cMultiFrame.cpp:92-216: the images are cv::Mat in host
// memory).  Measured on the default step (1.54 ms device-resident; profiles/r04):
//   * device -> host through the runtime (hipMemcpyAsync: chip-wide blit kernels here) costs 0.6 ms per step, through 32 workgroups of this kernel 0.35 ms — not
//     for the CUs they hold: every kernel running beside the copy slows down by half (FAST 0.39 -> 0.58 ms), because the stores towards the link back up in the
//     memory pipeline that everybody's loads and stores share.  PCIe needs little in flight: TWO workgroups (32 KB of requests) already write 48 GB/s, and beside them
//     the step costs 0.06 ms more.  More workgroups only lengthen the queue in front of everybody else.
//   * host -> device is the opposite: the runtime's copy (SDMA engine, no CU, no shader memory traffic) costs 0.07-0.17 ms per step, a kernel READING host memory needs
//     8+ workgroups of requests in flight to cover the link's latency and costs 0.65 ms.  So: images in by hipMemcpyAsync, results out by mcs_copy_narrow.
//   * where the copy is enqueued matters as much: HIP streams share four hardware queues and a queue runs in order.  A copy on a stream of its own lands on some other
//     stream's queue (which one is the runtime's choice: the step measured 2.0 or 3.2 ms); more hardware queues (GPU_MAX_HW_QUEUES) or priority streams slow the
//     step itself (1.54 -> 1.83 ms with 8 queues, no copy at all).  mcs_ctx_result_stream names the stream on which a step's results complete in stream order:
//     the copy goes there, behind the greedy pass, without an event and without a stream of its own.
#include "mcs_host.h"
#include <chrono>
#include "../../include/mcs_c.h"

namespace mcs {

constexpr int kCopyThreads = 256, kCopyUnroll = 4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte lanes, kCopyUnroll requests in flight per lane before the first store; non-temporal on both sides (neither side is read again by this kernel,
// and the device-side lines should not push the step's working set out of L2)
__global__ __launch_bounds__(kCopyThreads) void k_copy_narrow(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t n16, int pace) {
	const size_t stride = (size_t)gridDim.x * kCopyThreads;
	size_t i = (size_t)blockIdx.x * kCopyThreads + threadIdx.x;
	for (; i + (kCopyUnroll - 1) * stride < n16; i += kCopyUnroll * stride) {
		u32x4 v[kCopyUnroll];
#pragma unroll
		for (int u = 0; u < kCopyUnroll; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
		for (int u = 0; u < kCopyUnroll; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
		for (int p = 0; p < pace; ++p) __builtin_amdgcn_s_sleep(1);   // pacing (64 cycles a unit): keep the requests in flight below what the link drains
	}
	for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
__global__ __launch_bounds__(64) void k_copy_bytes(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t n) {
	for (size_t i = threadIdx.x; i < n; i += 64) dst[i] = src[i];
}

// ---- which hardware queue did a stream get? ---------------------------------------------------------------------------------------------------------------
// k_hold keeps the stream it is launched on busy until the host writes the flag (or ~3 ms pass); k_mark says "I ran".  A marker launched on another stream AFTER the
// hold arrives while the hold still spins unless the two streams feed the same hardware queue (packets of one queue start in order).
__global__ void k_hold(volatile int* flag) {
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
	while (*flag == 0 && __builtin_amdgcn_s_memrealtime() - t0 < 300000ull) __builtin_amdgcn_s_sleep(32);
}
__global__ void k_mark(volatile int* mark) { *mark = 1; }

}  // namespace mcs

// Bit i of *mask: `hip_stream` shares a hardware queue with the context's stream i (0 the caller's / main stream, 1 the extraction's side stream, 2 the deferred
// matcher's, 3 the greedy pass's = the result stream).  HIP streams are dealt onto four hardware queues by the runtime, and a queue runs in order: a stream that
// carries long transfers (an image upload holds its queue for the whole 1.3 ms) must not sit in front of the extraction.  Synchronises the streams involved.
extern "C" int mcs_ctx_stream_conflicts(mcs_ctx* c, void* hip_stream, unsigned* mask) {
	if (!c || !mask) return fail(MCS_ERR_INVALID, "bad argument");
	HIPCHK(hipSetDevice(c->device));
	*mask = 0;
	if (!c->side) return MCS_OK;   // nothing overlapped: one stream
	volatile int* pin = nullptr;
	HIPCHK(hipHostMalloc((void**)&pin, 128, hipHostMallocDefault));
	hipStream_t cand = (hipStream_t)hip_stream;
	hipStream_t own[4] = {c->stream, c->side, c->side2, c->side3};
	bool ok = true;
	// one probe: hold own[i], launch the marker on the candidate, watch for it while the hold spins.  The verdict rests on wall-clock time, so everything else is
	// quiesced first (work in flight on ANY of the context's streams could delay the marker: a false conflict), the launches are checked, and a "conflict" is only
	// believed when a second probe agrees (a host thread descheduled between the two launches looks the same as a shared queue).  What remains undecidable from
	// here — a host stall longer than the 3 ms hold reads as "no conflict" — costs performance, not correctness: the streams are ordered by events either way.
	auto probe = [&](int i, bool* conflict) -> bool {
		for (hipStream_t st : own) if (st && hipStreamSynchronize(st) != hipSuccess) return false;
		if (hipStreamSynchronize(cand) != hipSuccess) return false;
		pin[0] = 0; pin[16] = 0;
		hipLaunchKernelGGL(mcs::k_hold, dim3(1), dim3(1), 0, own[i], pin);
		const hipError_t e1 = hipGetLastError();
		hipLaunchKernelGGL(mcs::k_mark, dim3(1), dim3(1), 0, cand, pin + 16);
		const hipError_t e2 = hipGetLastError();
		const auto t0 = std::chrono::steady_clock::now();
		bool seen = false;
		while (e1 == hipSuccess && e2 == hipSuccess && !(seen = pin[16] != 0) && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(1500)) {}
		pin[0] = 1;   // release the hold
		const bool synced = hipStreamSynchronize(own[i]) == hipSuccess && hipStreamSynchronize(cand) == hipSuccess;
		*conflict = !seen;
		return e1 == hipSuccess && e2 == hipSuccess && synced;
	};
	for (int i = 0; i < 4 && ok; ++i) {
		if (own[i] == cand) { *mask |= 1u << i; continue; }
		bool conflict = false;
		ok = probe(i, &conflict);
		if (ok && conflict) ok = probe(i, &conflict);   // believe a conflict only twice in a row
		if (ok && conflict) *mask |= 1u << i;
	}
	(void)hipHostFree((void*)pin);   // on every path
	return ok ? MCS_OK : fail(MCS_ERR_HIP, "stream probe failed");
}

// A stream for the image uploads (hipMemcpyAsync from page-locked memory): created here, probed, and kept if it shares a hardware queue with none of the context's
// streams or only with the deferred matcher's (four queues, four streams of the context: SOME stream has to be shared with; the matcher has a step of slack).
// Owned by the context.
extern "C" int mcs_ctx_transfer_stream(mcs_ctx* c, void** hip_stream, unsigned* conflicts) {
	if (!c || !hip_stream) return fail(MCS_ERR_INVALID, "bad argument");
	HIPCHK(hipSetDevice(c->device));
	if (!c->upload) {
		// measured (default step, 1.54 ms device-resident, deferred searches): sharing with the deferred matcher's stream 1.60 ms, with the greedy pass's (the results'
		// way out) 2.17 ms, with the main stream 2.89 ms.  What bits 2 and 3 MEAN depends on the search mode: deferred — side2 carries the lists (a step of slack),
		// side3 the greedy pass and the results; in order — side2 is the greedy pass and the result stream, side3 is idle.  The choice is re-scored (not re-probed)
		// when mcs_ctx_set_async_search changes the mode.
		const bool deferred = c->asyncSearch;
		auto cost = [deferred](unsigned x) {
			const int c2 = deferred ? 1 : 4, c3 = deferred ? 4 : 0;
			return (x & 1u ? 8 : 0) + (x & 2u ? 8 : 0) + (x & 4u ? c2 : 0) + (x & 8u ? c3 : 0);
		};
		hipStream_t best = nullptr; unsigned bestMask = ~0u;
		for (size_t i = 0; i < c->probed.size(); ++i)
			if (!best || cost(c->probedMask[i]) < cost(bestMask)) { best = c->probed[i]; bestMask = c->probedMask[i]; }
		for (int t = (int)c->probed.size(); t < 8 && (!best || cost(bestMask) > 1); ++t) {
			hipStream_t s = nullptr;
			HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
			unsigned m = 0;
			if (int r = mcs_ctx_stream_conflicts(c, s, &m)) { (void)hipStreamDestroy(s); return r; }
			c->probed.push_back(s);   // kept alive until the context goes: a destroyed stream would hand its queue slot to the next candidate
			c->probedMask.push_back(m);
			if (!best || cost(m) < cost(bestMask)) { best = s; bestMask = m; }
		}
		c->upload = best; c->uploadMask = bestMask;
	}
	*hip_stream = (void*)c->upload;
	if (conflicts) *conflicts = c->uploadMask;
	return MCS_OK;
}

extern "C" int mcs_copy_narrow(mcs_ctx* c, void* dst, const void* src, size_t bytes, int workgroups, void* hip_stream) {
	if (!c || !dst || !src || workgroups < 1 || workgroups > 4096) return fail(MCS_ERR_INVALID, "bad argument");
	if (bytes == 0) return MCS_OK;
	HIPCHK(hipSetDevice(c->device));
	hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
	uint8_t* d = (uint8_t*)dst;
	const uint8_t* p = (const uint8_t*)src;
	// head up to the destination's 16-byte boundary and the tail go byte-wise (one wave); the body needs both sides 16-byte aligned
	size_t head = (16 - ((uintptr_t)d & 15)) & 15;
	if (head > bytes) head = bytes;
	if ((((uintptr_t)p + head) & 15) != 0) {   // mutually misaligned buffers: no vector body
		HIPCHK(hipMemcpyAsync(d, p, bytes, hipMemcpyDefault, s));   // the runtime's own copy
		return MCS_OK;
	}
	const size_t n16 = (bytes - head) / 16, tail = bytes - head - n16 * 16;
	if (head) hipLaunchKernelGGL(mcs::k_copy_bytes, dim3(1), dim3(64), 0, s, d, p, head);
	if (n16) {
		const size_t want = (n16 + mcs::kCopyThreads - 1) / mcs::kCopyThreads;
		const int wg = (int)(want < (size_t)workgroups ? want : (size_t)workgroups);
		static const int pace = getenv("MCS_COPY_PACE") ? atoi(getenv("MCS_COPY_PACE")) : 0;
		hipLaunchKernelGGL(mcs::k_copy_narrow, dim3(wg), dim3(mcs::kCopyThreads), 0, s, (mcs::u32x4*)(d + head), (const mcs::u32x4*)(p + head), n16, pace);
	}
	if (tail) hipLaunchKernelGGL(mcs::k_copy_bytes, dim3(1), dim3(64), 0, s, d + head + n16 * 16, p + head + n16 * 16, tail);
	HIPCHK(hipGetLastError());
	return MCS_OK;
}

// The stream on which the outputs of the latest mcs_search_* call (device memory) become complete in stream order: the greedy pass's stream.  A copy of a
// step's results to the host enqueued HERE needs no event in front of it, adds no stream (streams beyond the context's own four start sharing hardware queues
// with them, and a queue runs in order), and waits in nobody's way: the only later work on this stream is the next step's greedy pass, which has a step of slack.
extern "C" int mcs_ctx_result_stream(mcs_ctx* c, void** hip_stream) {
	if (!c || !hip_stream) return fail(MCS_ERR_INVALID, "bad argument");
	// the stream the latest search actually used (recorded by the search itself: the answer cannot go stale when mcs_ctx_set_async_search or
	// mcs_ctx_enable_timing is toggled between the search and this query); before the first search, the stream the NEXT one would use in the current mode
	*hip_stream = (void*)(c->lastResultStream ? c->lastResultStream : (c->overlap() ? (c->asyncSearch ? c->side3 : c->side2) : c->stream));
	return MCS_OK;
}

// Page-locked host memory through the C ABI, so that a host translation unit that only sees include/mcs_c.h (integration/cMultiFrame_mcs.cpp inside the
// reference's build: no HIP headers there) can stage its images and receive its results without the runtime's pageable-memory detour.
extern "C" int mcs_host_alloc(mcs_ctx* c, size_t bytes, void** out) {
	if (!c || !out || bytes == 0) return fail(MCS_ERR_INVALID, "bad argument");
	HIPCHK(hipSetDevice(c->device));
	*out = nullptr;
	HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
	return MCS_OK;
}
extern "C" int mcs_host_free(mcs_ctx* c, void* p) {
	if (!c) return fail(MCS_ERR_INVALID, "bad argument");
	if (!p) return MCS_OK;
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipHostFree(p));
	return MCS_OK;
}

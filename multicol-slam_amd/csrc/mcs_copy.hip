// mcs_copy.hip — a copy between page-locked host memory and the device that takes a FEW workgroups instead of the runtime's blit kernel.
//
// A live front end gets its images from the host and hands keypoints, descriptors and matches back (src/cMultiFrame.cpp:92-216: the images are cv::Mat
// in host memory).  This runtime executes page-locked hipMemcpyAsync as blit kernels (__amd_rocclr_copyBuffer) whose grid fills the chip: beside the
// step's own kernels they take CU slots for as long as PCIe needs (≈1.2 ms for the 69.5 MB of a default step at ≈56 GB/s).  PCIe needs little
// parallelism — rate × latency ≈ 56 GB/s × 2 µs ≈ 110 KB in flight — so a handful of workgroups with several 16-byte requests per lane saturate the
// link and leave every CU to the step's kernels (they hold a few wave slots, no LDS, 24 registers).  Page-locked host memory (hipHostMalloc /
// hipHostRegister) is addressable from the device under the same pointer; loads from it are PCIe reads, stores PCIe posted writes.
#include "mcs_host.h"
#include "../../include/mcs_c.h"

namespace mcs {

constexpr int kCopyThreads = 256, kCopyUnroll = 4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte lanes, kCopyUnroll requests in flight per lane before the first store; non-temporal on both sides (neither side is read again by this kernel,
// and the device-side lines should not push the step's working set out of L2)
__global__ __launch_bounds__(kCopyThreads) void k_copy_narrow(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t n16) {
	const size_t stride = (size_t)gridDim.x * kCopyThreads;
	size_t i = (size_t)blockIdx.x * kCopyThreads + threadIdx.x;
	for (; i + (kCopyUnroll - 1) * stride < n16; i += kCopyUnroll * stride) {
		u32x4 v[kCopyUnroll];
#pragma unroll
		for (int u = 0; u < kCopyUnroll; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
		for (int u = 0; u < kCopyUnroll; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
	}
	for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
__global__ __launch_bounds__(64) void k_copy_bytes(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t n) {
	for (size_t i = threadIdx.x; i < n; i += 64) dst[i] = src[i];
}

}  // namespace mcs

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
		hipLaunchKernelGGL(mcs::k_copy_narrow, dim3(wg), dim3(mcs::kCopyThreads), 0, s, (mcs::u32x4*)(d + head), (const mcs::u32x4*)(p + head), n16);
	}
	if (tail) hipLaunchKernelGGL(mcs::k_copy_bytes, dim3(1), dim3(64), 0, s, d + head + n16 * 16, p + head + n16 * 16, tail);
	HIPCHK(hipGetLastError());
	return MCS_OK;
}

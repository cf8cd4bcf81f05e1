// pmc_patterns.hip — known byte counts for rocprofv3's FETCH_SIZE in THIS path's access widths (tools/pmc_calibrate.sh; MI355X_MICROARCH.md: "calibrate on a known
// byte count in your own access pattern before trusting an absolute").  Build: hipcc --offload-arch=gfx950 -O2 -o tools/pmc_patterns tools/pmc_patterns.hip
//
//   k_stream16 / k_stream8 / k_stream4   every lane reads 16 / 8 / 4 consecutive bytes, a wave a contiguous run: the matcher's operand stream and the copy kernel (16),
//                                         the column-marching blur / resize kernels and FAST's tile staging (8 / 4).  Compulsory bytes = the buffer.
//   k_rows48                              per "keypoint" 43 rows of 48 bytes (three 16-byte lanes per row), rows one level pitch apart, keypoints at random places
//                                         of a 1 GiB buffer: the descriptor pass's patch requests.  What the memory side must deliver at least is the set of
//                                         DISTINCT 64-byte / 128-byte lines the rows touch — the host counts both and prints them.
//   k_rows36                              per keypoint 33 rows of 36 bytes read as nine dwords by nine lanes: the orientation disc of the oct-tree kernel's tail.
// Every kernel folds what it reads into one word per wave (so nothing is optimised away) and is launched three times; compare the printed expectations with the
// FETCH_SIZE the profiler reports per dispatch.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

namespace mcs {
__global__ void k_stream16(const uint8_t* p, size_t n, uint32_t* s) { const size_t stride = (size_t)gridDim.x * blockDim.x * 16; uint32_t acc = 0;
	for (size_t at = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; at + 16 <= n; at += stride) { const uint4 v = *reinterpret_cast<const uint4*>(p + at); acc += v.x ^ v.y ^ v.z ^ v.w; }
	if (acc == 0x12345678u) s[0] = acc; }
__global__ void k_stream8(const uint8_t* p, size_t n, uint32_t* s) { const size_t stride = (size_t)gridDim.x * blockDim.x * 8; uint32_t acc = 0;
	for (size_t at = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; at + 8 <= n; at += stride) { const uint2 v = *reinterpret_cast<const uint2*>(p + at); acc += v.x ^ v.y; }
	if (acc == 0x12345678u) s[0] = acc; }
__global__ void k_stream4(const uint8_t* p, size_t n, uint32_t* s) { const size_t stride = (size_t)gridDim.x * blockDim.x * 4; uint32_t acc = 0;
	for (size_t at = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; at + 4 <= n; at += stride) acc += *reinterpret_cast<const uint32_t*>(p + at);
	if (acc == 0x12345678u) s[0] = acc; }

// one wave per keypoint: rows of 48 bytes, three lanes of 16 bytes each (lanes beyond the last row idle)
__global__ void k_rows48(const uint8_t* p, const uint32_t* origin, int nkp, int pitch, uint32_t* sink) {
	const int kp = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (kp >= nkp) return;
	const uint8_t* o = p + (size_t)origin[kp] * 16;
	uint32_t acc = 0;
	for (int t = 0; t < 3; ++t) {
		const int i = lane + 64 * t, r = i / 3, k = i - 3 * r;
		if (r < 43) { const uint4 v = *reinterpret_cast<const uint4*>(o + (size_t)r * pitch + 16 * k); acc += v.x ^ v.y ^ v.z ^ v.w; }
	}
	if (acc == 0x12345678u) sink[0] = acc;
}
// 16 lanes per keypoint: 33 rows of nine dwords (unaligned by up to 3 bytes, as the disc's rows are)
__global__ void k_rows36(const uint8_t* p, const uint32_t* origin, int nkp, int pitch, uint32_t* sink) {
	const int kp = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4), l = threadIdx.x & 15;
	if (kp >= nkp) return;
	const uint8_t* o = p + (size_t)origin[kp] * 16 + (kp & 3);
	uint32_t acc = 0;
	if (l < 9)
		for (int r = 0; r < 33; ++r) { uint32_t v; __builtin_memcpy(&v, o + (size_t)r * pitch + 4 * l, 4); acc += v; }
	if (acc == 0x12345678u) sink[0] = acc;
}
}  // namespace mcs (tools/pmc_traffic.py lists the kernels of this namespace)
using namespace mcs;

int main() {
	const size_t n = (size_t)1 << 30;
	uint8_t* buf; uint32_t* sink;
	OK(hipMalloc(&buf, n + 65536)); OK(hipMalloc(&sink, 64));
	OK(hipMemset(buf, 1, n + 65536));
	const int nkp = 200000, pitch = 768;   // a 754-px level's row pitch
	std::vector<uint32_t> org(nkp);
	uint64_t st = 0x9E3779B97F4A7C15ull;
	auto next = [&]() { st ^= st >> 12; st ^= st << 25; st ^= st >> 27; return st * 0x2545F4914F6CDD1Dull; };
	for (auto& o : org) o = (uint32_t)(next() % ((n - (size_t)64 * pitch) / 16));
	uint32_t* dorg; OK(hipMalloc(&dorg, nkp * 4)); OK(hipMemcpy(dorg, org.data(), nkp * 4, hipMemcpyHostToDevice));
	// what the rows touch, as distinct 64-byte and 128-byte lines (keypoints are random: cross-keypoint reuse inside a 4 MiB L2 is negligible, but counted right anyway)
	auto lines = [&](int rows, int width, int skew, size_t* l64, size_t* l128) {
		std::set<uint64_t> a, b;
		for (int k = 0; k < nkp; ++k)
			for (int r = 0; r < rows; ++r) {
				const uint64_t lo = (uint64_t)org[k] * 16 + (skew ? (k & 3) : 0) + (uint64_t)r * pitch, hi = lo + width - 1;
				for (uint64_t x = lo / 64; x <= hi / 64; ++x) a.insert(x);
				for (uint64_t x = lo / 128; x <= hi / 128; ++x) b.insert(x);
			}
		*l64 = a.size(); *l128 = b.size();
	};
	size_t a48, b48, a36, b36;
	lines(43, 48, 0, &a48, &b48);
	lines(33, 36, 1, &a36, &b36);
	for (int rep = 0; rep < 3; ++rep) {
		hipLaunchKernelGGL(k_stream16, dim3(4096), dim3(256), 0, 0, buf, n, sink);
		hipLaunchKernelGGL(k_stream8, dim3(4096), dim3(256), 0, 0, buf, n, sink);
		hipLaunchKernelGGL(k_stream4, dim3(4096), dim3(256), 0, 0, buf, n, sink);
		hipLaunchKernelGGL(k_rows48, dim3((nkp + 3) / 4), dim3(256), 0, 0, buf, dorg, nkp, pitch, sink);
		hipLaunchKernelGGL(k_rows36, dim3((nkp + 15) / 16), dim3(256), 0, 0, buf, dorg, nkp, pitch, sink);
		OK(hipDeviceSynchronize());
	}
	printf("expected KiB per dispatch: k_stream16 / k_stream8 / k_stream4 %.1f (the buffer)\n", n / 1024.0);
	printf("expected KiB per dispatch: k_rows48 requested %.1f, distinct 64-byte lines %.1f, distinct 128-byte lines %.1f\n", nkp * 43.0 * 48 / 1024, a48 * 64 / 1024.0, b48 * 128 / 1024.0);
	printf("expected KiB per dispatch: k_rows36 requested %.1f, distinct 64-byte lines %.1f, distinct 128-byte lines %.1f\n", nkp * 33.0 * 36 / 1024, a36 * 64 / 1024.0, b36 * 128 / 1024.0);
	return 0;
}

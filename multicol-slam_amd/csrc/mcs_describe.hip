// mcs_describe.hip — E5 + E7a/b/c + E8 + the per-keypoint part of E9: one wave64 per selected keypoint.
// Reference: IC_Angle src/mdBRIEFextractorOct.cpp:221-248; rotatePattern :285-301; rotateAndDistortPattern :250-283;
// compute_ORB :303-354; compute_dBRIEF :356-408; compute_mdBRIEF :410-554; operator() glue :1286-1336;
// omni model src/cam_model_omni.cpp:49-67,146-161, include/cam_model_omni.h:127-145, include/misc.h:115-122;
// rays src/cMultiFrame.cpp:146-152.  cv::fastAtan2 per SURVEY Appendix A.5.
//
//   orientation   845-pixel disc of the UNBLURRED level (mcs_orient.h; computed in the tail of the oct-tree kernel, read here): int32 moments
//                 reduced with cross-lane shuffles (exact, order-free), then the float polynomial of cv::fastAtan2.
//   patch         the (2R+1)^2 blurred neighbourhood (R = 21: 43 rows of 44 bytes) almost every sample touches is staged once in LDS.
//   descriptor    lane l owns pattern pairs l, l+64, l+128, ... ; one __ballot per 64 pairs yields 8 descriptor bytes
//                 (bit k -> byte k/8, LSB first, exactly the reference's packing).
//   dBRIEF        every pattern point goes through the Scaramuzza model in FP64 (sqrt, atan, Horner); the camera's
//                 coefficients sit in SGPRs (readfirstlane) and the Horner chains have the fixed zero-padded length.
//                 The mean of the 2*8*descSize distorted points is accumulated SEQUENTIALLY from LDS in the reference's
//                 order (even lanes: x, odd lanes: y, all lanes redundantly so no divergence), because FP64 addition
//                 order decides cvRound ties.  Math pass and chain pass are separate loops (registers).
//   mdBRIEF       three patterns (angle, +20deg, -20deg); mask bit = both rotated tests agree with the main test.
// Samples inside the level come from the blurred pyramid, samples in the 25-px frame from the unblurred level with
// reflect-101 indices (the reference's frame is filled before the in-place blur), beyond the frame: clamped.
// Compiled with -ffp-contract=off: no FMA contraction anywhere, like the oracle.
//
// dBRIEF / mdBRIEF run in TWO passes:
//   k_describe_fast   every keypoint.  The 2*8*descSize*npat omni-model evaluations use a cheaper arithmetic (explicit FMAs; G(s) = rho(atan(p0 / sqrt(s))) / sqrt(s)
//                     from a per-camera table indexed by the bit pattern of s = x^2 + y^2: no square root, no division, no atan, no backward polynomial) and the
//                     pattern mean is a wave tree sum instead of the reference's sequential chain.  The results differ from the reference's by at most a bound
//                     delta known per camera (host, mcs_capi.hip: describe_fast_bound) — far below half a pixel, but cvRound(coordinate - mean) only agrees for
//                     sure when no coordinate lies within delta of a rounding tie.  Every coordinate is therefore checked against a guard band eps >= delta
//                     around the ties (|frac - 0.5| < eps); a keypoint with ANY coordinate inside the band (or out of range / NaN) is not written, its slot goes
//                     onto the fallback list.  Persistent 16-wave workgroups (table and pattern in LDS once per CU) walk the batch; the walk is software-pipelined
//                     through LDS-DMA (round 4): the next keypoint's patch and the record after it travel global -> LDS while this keypoint is described.
//   k_describe_list   the keypoints of the fallback list through describe_wave — the reference's exact arithmetic (below, unchanged), a few per 10^4.
// Integer pixel offsets that pass the guard are PROVABLY the reference's, so the descriptors stay bit-identical; mcs_extractor_set_describe() can force
// the exact pass for everything or widen the band (tests/test_gpu_describe_guard.py runs both against the oracle).
#include "mcs_common.h"
#include <type_traits>
#include "mcs_orient.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace mcs {

__constant__ __attribute__((aligned(16))) signed char c_pattern[2048];

// ---- the pattern's DISTINCT points (fast pass, round 5) ---------------------------------------------------------------------------------------------
// The learned pattern repeats itself: the first 2*8*descSize points of learned_pattern_64_ORB hold 224 / 384 / 567 distinct (x, y) for descSize 16 / 32 / 64
// (multiplicities up to 3 / 4 / 7).  The omni model is a function of the point, so the fast pass evaluates every DISTINCT point once — UPat<NB>::R rounds of
// 64 lanes: 4 / 6 / 9 instead of 4 / 8 / 16 —, forms the pattern mean as the multiplicity-weighted sum, samples the patch once per distinct point and
// leaves the sampled byte in a per-wave LDS array; a lane then gathers the bytes of its own pairs' points by index.  Layout (host: build_unique_pattern):
//   upat[t * 64 + l]   point of round t, lane l (doubles).  Order: multiplicity >= 2 first, then the padding slots (copies of point 0 with weight 0 — a real
//                      point's coordinates, so that range and guard checks see nothing new), then multiplicity 1; the first RW rounds carry explicit weights
//   uw[t * 64 + l]     multiplicity of that point as a double, rounds 0..RW-1 (beyond them every weight is 1: a plain add)
//   gidx[l * 2NB + i]  index into upat (= into the wave's vals[]) of point (i & 1) of pair (i >> 1) * 64 + l, the pairs lane l ballots on
// (Measured and dropped: every lane publishing its R bytes as ONE packed store into a per-lane slot — the packing cost registers, the kernel began to spill, and a
// spill is a vector memory access that waits for the patch prefetch in flight: 470 -> 521 us.)
template <int NB> struct UPat;
template <> struct UPat<2> { static constexpr int R = 4, RW = 1; };
template <> struct UPat<4> { static constexpr int R = 6, RW = 2; };
template <> struct UPat<8> { static constexpr int R = 9, RW = 5; };
constexpr int kUPatMax = 9 * 64, kUWMax = 5 * 64;
__device__ double2 g_upat[3][kUPatMax];
__device__ double g_uw[3][kUWMax];
__device__ unsigned short g_gidx[3][64 * 16];

template <int NB>
static bool build_unique_pattern(const signed char* pattern, int slot) {
	constexpr int R = UPat<NB>::R, RW = UPat<NB>::RW, NP = 128 * NB;
	std::vector<int> first, mult, ofPoint(NP);
	for (int i = 0; i < NP; ++i) {
		int f = -1;
		for (size_t k = 0; k < first.size(); ++k)
			if (pattern[2 * first[k]] == pattern[2 * i] && pattern[2 * first[k] + 1] == pattern[2 * i + 1]) { f = (int)k; break; }
		if (f < 0) { f = (int)first.size(); first.push_back(i); mult.push_back(0); }
		mult[f]++; ofPoint[i] = f;
	}
	const int nu = (int)first.size(), pad = R * 64 - nu;
	int heavy = 0;
	for (int m : mult) heavy += m >= 2;
	if (pad < 0 || heavy + pad > RW * 64) return false;   // the round counts above belong to the shipped pattern
	std::vector<int> order;   // distinct point per slot, -1 = padding
	for (int k = 0; k < nu; ++k) if (mult[k] >= 2) order.push_back(k);
	for (int k = 0; k < pad; ++k) order.push_back(-1);
	for (int k = 0; k < nu; ++k) if (mult[k] < 2) order.push_back(k);
	std::vector<int> slotOf(nu);
	std::vector<double2> up(kUPatMax, double2{0.0, 0.0});
	std::vector<double> uw(kUWMax, 0.0);
	for (int sidx = 0; sidx < R * 64; ++sidx) {
		const int k = order[sidx], src = first[k < 0 ? 0 : k];
		up[sidx] = double2{(double)pattern[2 * src], (double)pattern[2 * src + 1]};
		if (k >= 0) slotOf[k] = sidx;
		if (sidx < RW * 64) uw[sidx] = k < 0 ? 0.0 : (double)mult[k];
		else if (k < 0 || mult[k] != 1) return false;
	}
	std::vector<unsigned short> gi(64 * 16, 0);
	for (int l = 0; l < 64; ++l)
		for (int i = 0; i < 2 * NB; ++i) gi[l * 2 * NB + i] = (unsigned short)slotOf[ofPoint[2 * ((i >> 1) * 64 + l) + (i & 1)]];
	return hipMemcpyToSymbol(HIP_SYMBOL(g_upat), up.data(), sizeof(double2) * kUPatMax, sizeof(double2) * kUPatMax * slot) == hipSuccess &&
	       hipMemcpyToSymbol(HIP_SYMBOL(g_uw), uw.data(), sizeof(double) * kUWMax, sizeof(double) * kUWMax * slot) == hipSuccess &&
	       hipMemcpyToSymbol(HIP_SYMBOL(g_gidx), gi.data(), sizeof(unsigned short) * 64 * 16, sizeof(unsigned short) * 64 * 16 * slot) == hipSuccess;
}

bool upload_describe_tables(const signed char* pattern) {
	if (!pattern) return false;
	if (hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), pattern, 2048) != hipSuccess) return false;
	return build_unique_pattern<2>(pattern, 0) && build_unique_pattern<4>(pattern, 1) && build_unique_pattern<8>(pattern, 2);
}

// Division by a shared denominator: the compiler's own f64 division expansion (rcp, two Newton steps, q0 = n*r,
// e = fma(-d, q0, n), q = fma(e, r, q0)) with its denominator-only part computed once.  v_div_scale / v_div_fixup of the full
// expansion only act on operands near the exponent limits, so for ordinary magnitudes div_shared(a, d, recip_refined(d)) is
// bit-identical to a / d (checked on the device by mcs_selftest_shared_reciprocal, tests/test_gpu_match.py).  The one difference is
// the sign of a ZERO quotient (-0 / d gives +0 here, v_div_fixup would restore -0); the omni model adds the principal point to
// every product of these quotients, so a zero's sign never reaches u or v.
__device__ __forceinline__ double recip_refined(double d) {
	const double r0 = __builtin_amdgcn_rcp(d);
	const double r1 = __builtin_fma(r0, __builtin_fma(-d, r0, 1.0), r0);
	return __builtin_fma(r1, __builtin_fma(-d, r1, 1.0), r1);
}
__device__ __forceinline__ double div_shared(double a, double d, double r) {
	const double q0 = a * r;
	return __builtin_fma(__builtin_fma(-d, q0, a), r, q0);
}

// a / d vs div_shared on pseudo-random operands of the magnitudes k_describe sees (d in [1e-14, 1e5], |a| <= 1e4, zeros and tiny
// numerators included); counts the bit mismatches
__global__ void k_selftest_recip(unsigned long long seed, int n, int* mismatches) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	unsigned long long st = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
	auto next = [&]() { st ^= st >> 12; st ^= st << 25; st ^= st >> 27; return st * 0x2545F4914F6CDD1Dull; };
	auto unit = [&]() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); };
	const double ed = -14.0 + 19.0 * unit();
	const double d = exp10(ed) * (1.0 + unit());
	double a = (unit() * 2.0 - 1.0) * exp10(-18.0 + 22.0 * unit());
	if ((i & 127) == 0) a = 0.0;
	if ((i & 127) == 1) a = -d;
	if ((i & 127) == 2) a = d;
	const double r = recip_refined(d);
	const double q1 = div_shared(a, d, r), q2 = a / d;
	if (__double_as_longlong(q1) != __double_as_longlong(q2)) atomicAdd(mismatches, 1);
}

void launch_selftest_recip(unsigned long long seed, int n, int* mismatches, hipStream_t s) {
	hipLaunchKernelGGL(k_selftest_recip, dim3((n + 255) / 256), dim3(256), 0, s, seed, n, mismatches);
}

// wave-uniform double load: both halves go through v_readfirstlane so the value lives in SGPRs
__device__ __forceinline__ double uniform_f64(const double* p) {
	const unsigned lo = __builtin_amdgcn_readfirstlane(reinterpret_cast<const unsigned*>(p)[0]);
	const unsigned hi = __builtin_amdgcn_readfirstlane(reinterpret_cast<const unsigned*>(p)[1]);
	return __hiloint2double((int)hi, (int)lo);
}

// include/misc.h:115-122.  The coefficient arrays of OcamDev are MCS_MAX_POLY long and zero above the model's degree, and
// 0*x + 0 = +0 for finite x, so the fixed-length form is bit-identical to the reference's loop while all coefficient
// loads are independent (one memory round trip instead of one per term).
__device__ __forceinline__ double horner_d(const double* coeffs, int s, double x) {
	double c[MCS_MAX_POLY];
#pragma unroll
	for (int i = 0; i < MCS_MAX_POLY; ++i) c[i] = coeffs[i];
	double res = 0.0;
#pragma unroll
	for (int i = MCS_MAX_POLY - 1; i >= 0; i--) res = res * x + c[i];
	(void)s;
	return res;
}

__device__ __forceinline__ void world2img(const OcamDev& cam, double x, double y, double z, double& u, double& v) {
	double norm = sqrt(x * x + y * y);
	if (norm == 0.0) norm = 1e-14;
	const double theta = atan(-z / norm);
	const double rho = horner_d(cam.invP, cam.invP_deg, theta);
	const double uu = x / norm * rho;
	const double vv = y / norm * rho;
	u = uu * cam.c + vv * cam.d + cam.u0;
	v = uu * cam.e + vv + cam.v0;
}

__device__ __forceinline__ void img2world(const OcamDev& cam, double u, double v, double& xo, double& yo, double& zo) {
	const double u_t = u - cam.u0;
	const double v_t = v - cam.v0;
	double x = (u_t - cam.d * v_t) / cam.invAffine;
	double y = (-cam.e * u_t + cam.c * v_t) / cam.invAffine;
	const double X2 = x * x;
	const double Y2 = y * y;
	double z = -horner_d(cam.p, cam.p_deg, sqrt(X2 + Y2));
	double norm = sqrt(X2 + Y2 + z * z);
	xo = x / norm;
	yo = y / norm;
	zo = z / norm;
}

// Blurred neighbourhood of the keypoint staged in LDS: rows row-R..row+R, 4*ceil((2R+1)/4) bytes from col-R.  A keypoint sits >= 25 px
// inside the level, so the patch is always inside the ROI; rotated ORB offsets (pattern radius <= 21.2) and almost all distorted ones land in
// it, the rest takes the general path of Sampler::at — a scattered 64-lane byte gather from global memory per sample was address-unit-bound.
#ifndef MCS_PATCH_R
#define MCS_PATCH_R 21   // 43 rows of 44 bytes: with the 8.2 KB coordinate buffer a wave then needs 10 112 B of LDS, so 16 waves fit a CU (R = 24: 14)
#endif
constexpr int kPatchR = MCS_PATCH_R, kPatchRows = 2 * kPatchR + 1, kPatchDw = (kPatchRows + 3) / 4, kPatchPitch = 4 * kPatchDw;
constexpr int kPatchBytes = (kPatchRows * kPatchPitch + 15) / 16 * 16;
// the fast pass's patch rows are 16-byte multiples: a row is requested by (kFPitch / 16) lanes of one global_load_lds_dwordx4
constexpr int kFPitch = (kPatchRows + 15) / 16 * 16, kFPatchBytes = kPatchRows * kFPitch;

struct Sampler {
	const uint8_t* blur; int bstride;
	const uint8_t* raw; int rstride;
	int w, h;
	const uint8_t* patch; int prow, pcol;   // LDS patch and the level coordinates of its origin
	template <int PITCH = kPatchPitch>
	__device__ __forceinline__ int at(int r, int c) const {
		const unsigned pr = (unsigned)(r - prow), pc = (unsigned)(c - pcol);
		if (pr < (unsigned)kPatchRows && pc < (unsigned)kPatchRows) return patch[pr * PITCH + pc];
		if ((unsigned)r < (unsigned)h && (unsigned)c < (unsigned)w) return blur[(size_t)r * bstride + c];
		r = r < -kEdge ? -kEdge : (r > h + kEdge - 1 ? h + kEdge - 1 : r);   // clamp to the bordered buffer
		c = c < -kEdge ? -kEdge : (c > w + kEdge - 1 ? w + kEdge - 1 : c);
		r = r < 0 ? -r : (r >= h ? 2 * (h - 1) - r : r);                   // BORDER_REFLECT_101
		c = c < 0 ? -c : (c >= w ? 2 * (w - 1) - c : c);
		return raw[(size_t)r * rstride + c];
	}
	// Both samples of a test pair given as offsets from the keypoint (the patch centre).  Fast path, taken when every lane of the
	// wave has both samples inside the LDS patch (practically always): two ds_read_u8 with 32-bit LDS addresses.  The general
	// `at()` (generic pointer to LDS / blurred level / bordered raw level, three-way address select) stays for the rest.
	__device__ __forceinline__ void pair(int row, int col, int dy0, int dx0, int dy1, int dx1, int& t0, int& t1) const {
		const unsigned r0 = (unsigned)(dy0 + kPatchR), c0 = (unsigned)(dx0 + kPatchR), r1 = (unsigned)(dy1 + kPatchR), c1 = (unsigned)(dx1 + kPatchR);
		const bool inside = max(max(r0, c0), max(r1, c1)) < (unsigned)kPatchRows;
		if (!__any(!inside)) {
			t0 = patch[r0 * kPatchPitch + c0];
			t1 = patch[r1 * kPatchPitch + c1];
		} else {
			t0 = at(row + dy0, col + dx0);
			t1 = at(row + dy1, col + dx1);
		}
	}
};

#ifndef MCS_ABLATE
#define MCS_ABLATE 0   // A/B experiments only: 1 skip the sequential mean, 2 skip the omni model, 4 skip sampling
#endif
#ifndef MCS_MERGE_CHAINS
#define MCS_MERGE_CHAINS 0   // A/B only.  1 = mdBRIEF keeps all three distorted patterns in LDS and runs their six coordinate sums as ONE
                             // chain (lanes 0..5): 1024 fewer dependent adds per keypoint, but 26.5 KB LDS per wave (6 waves/CU instead
                             // of 12) and 186 VGPRs; measured 3.00 ms vs 2.93 ms per 192 images, so the separate chains stay.
#endif
// coordinate buffers per wave: [pattern][x | y][npoints] doubles
// The y array starts 2 doubles after the end of the x array: x[p] and y[p] are read in the same ds_read (even / odd lanes) and must not
// share LDS banks (an offset of exactly npoints doubles = a multiple of 256 B made every read of the sum chain a 2-way bank conflict).
__host__ __device__ constexpr int pat_doubles(int npoints) { return 2 * npoints + 2; }
__host__ __device__ constexpr int coord_bytes(int mode, int npoints) { return mode == 0 ? 0 : (mode == 2 && MCS_MERGE_CHAINS ? 3 : 1) * pat_doubles(npoints) * 8; }

// Everything a keypoint needs before its descriptor: slot -> (level, position), the output row, the blurred patch in LDS, orientation (E5), the
// keypoint record and ray (E8/E9).  Shared by the exact pass (describe_wave) and the fast pass (k_describe_fast); false = this wave has no keypoint.
constexpr int kPatchTrips = (kPatchRows * kPatchDw + 63) / 64;
struct KeyPt {
	int img, out, level, row, col;
	float angle;
	double rayx, rayy, rayz;
	Sampler sm;
};

// slot s of an image -> (level, position in the level's selection): slot = the keypoint's row in the image's output block, levels in order, a level's
// keys in their final list order.  `total` = all selected keys of the image.
__device__ __forceinline__ void find_slot(const PyrDesc& d, const int* selCount, int s, int& level, int& pos, int& total) {
	total = 0; level = -1; pos = 0;
	for (int l = 0; l < d.nlevels; ++l) {
		const int c = selCount[l];
		if (s >= total && s < total + c) { level = l; pos = s - total; }
		total += c;
	}
}

// issue the loads of the blurred (2R+1)^2 neighbourhood (kPatchDw unaligned dwords per row, rows are in-pitch even at the right edge) / write them to LDS
__device__ __forceinline__ void patch_load(const uint8_t* blur, int bstride, int row, int col, uint32_t (&pv)[kPatchTrips]) {
	const int lane = threadIdx.x & 63;
	const uint8_t* bp = blur + (size_t)(row - kPatchR) * bstride + (col - kPatchR);
#pragma unroll
	for (int t = 0; t < kPatchTrips; ++t) {
		const int i = min(lane + 64 * t, kPatchRows * kPatchDw - 1);
		const int r = i / kPatchDw, k = i - r * kPatchDw;
		__builtin_memcpy(&pv[t], bp + (size_t)r * bstride + 4 * k, 4);
	}
}
__device__ __forceinline__ void patch_store(uint8_t* patch, const uint32_t (&pv)[kPatchTrips]) {   // lane-private slots; every later read is by the same wave
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int t = 0; t < kPatchTrips; ++t) {
		const int i = min(lane + 64 * t, kPatchRows * kPatchDw - 1);   // the last trip's surplus lanes rewrite the last dword
		const int r = i / kPatchDw, k = i - r * kPatchDw;
		*reinterpret_cast<uint32_t*>(&patch[r * kPatchPitch + 4 * k]) = pv[t];
	}
}

template <bool NEED_RAY>   // the ray is also computed when the caller asked for rays
__device__ __forceinline__ bool kp_prologue(const ExtractBuffers& b, int wavesPerImage, int gw, uint8_t* patch, KeyPt& kp_) {
	const PyrDesc& d = *b.desc;
	const int lane = threadIdx.x & 63;
	// the slot is the same for the whole wave: say so (v_readfirstlane), and everything addressed through it — level table, selection counts, the
	// keypoint record, the camera model — is read with scalar loads into SGPRs instead of 64 identical vector loads
	gw = __builtin_amdgcn_readfirstlane(gw);
	const int img = gw / wavesPerImage;
	const int s = gw - img * wavesPerImage;
	const int* selCount = b.selCount + (size_t)img * d.nlevels;

	int total, level, pos;
	find_slot(d, selCount, s, level, pos, total);
	if (s == 0 && lane == 0) b.nkp[img] = total < d.kpCap ? total : d.kpCap;
	bool active = level >= 0;
	const int out = s;
	if (s == 0 && total > d.kpCap && lane == 0) atomicExch(b.status, MCS_ERR_CAPACITY);   // more keys selected than output rows (cannot happen with kpCap as sized by the host)
	if (active && out >= d.kpCap) active = false;
	if (!active) return false;   // waves are independent (no block barriers anywhere in these kernels)

	float angle = 0.f, pxf = 0.f, pyf = 0.f;
	double rayx = 0.0, rayy = 0.0, rayz = 0.0;
	int row = 0, col = 0;
	Sampler sm = {};
	{
		const LevelInfo& L = d.lv[level];
		const uint32_t rec = b.sel[(size_t)img * d.selPerImage + L.selBase + pos];
		col = (int)(rec & 0xFFF) + kMinBorder;
		row = (int)((rec >> 12) & 0xFFF) + kMinBorder;
		const float resp = (float)(rec >> 24);
		int rstride;
		const uint8_t* raw = level_ptr(b, d, img, level, &rstride);
		sm.blur = b.blur + (size_t)img * d.pyrBytes + L.off; sm.bstride = L.stride;
		sm.raw = raw; sm.rstride = rstride; sm.w = L.w; sm.h = L.h;
		// the patch loads are ISSUED here and written to LDS only after the orientation and the keypoint's ray are done: their round trip hides behind that arithmetic
		uint32_t pv[kPatchTrips];
		patch_load(sm.blur, sm.bstride, row, col, pv);
		sm.patch = patch; sm.prow = row - kPatchR; sm.pcol = col - kPatchR;
		angle = b.selAngle[(size_t)img * d.selPerImage + L.selBase + pos];   // IC_Angle: computed by the oct-tree kernel's tail (mcs_orient.h)
		// ---- keypoint record (E8): level coordinates -> image coordinates with the FLOAT scale (:1305,1331)
		pxf = (float)col; pyf = (float)row;
		if (level != 0) { pxf = pxf * L.scale; pyf = pyf * L.scale; }
		// ImgToWorld of the keypoint: the ray of src/cMultiFrame.cpp:146-152 AND the input of undistortPointsOcam (:1306-1317)
		if (b.cams && (b.rays || NEED_RAY)) img2world(b.cams[img], (double)pxf, (double)pyf, rayx, rayy, rayz);
		if (lane == 0) {
			mcs_keypoint kp;
			kp.x = pxf; kp.y = pyf; kp.size = L.kpSize; kp.angle = angle; kp.response = resp; kp.octave = level; kp.class_id = -1;
			b.kps[(size_t)img * d.kpCap + out] = kp;
			if (b.rays && b.cams) {
				double* rp = b.rays + ((size_t)img * d.kpCap + out) * 3;
				rp[0] = rayx; rp[1] = rayy; rp[2] = rayz;
			}
		}
		patch_store(patch, pv);
	}
	kp_.img = img; kp_.out = out; kp_.level = level; kp_.row = row; kp_.col = col; kp_.angle = angle;
	kp_.rayx = rayx; kp_.rayy = rayy; kp_.rayz = rayz; kp_.sm = sm;
	return true;
}

// ---- how close did a cvRound argument of the exact arithmetic come to a tie? -------------------------------------------------------------------------
// The exact passes (ORB rotation; rotateAndDistortPattern for dBRIEF / mdBRIEF) call ocml's sincos / atan where the reference calls glibc's: results that differ in
// the last place can only change cvRound(v) if v lies within ~1e-13 px of k + 1/2.  Rather than argue that this is improbable, every such argument is measured:
// the largest |v - rint(v)| of a wave's coordinates is reduced over the wave and 1/2 minus it — the distance to the nearest tie — lowers a per-extractor minimum
// (mcs_extractor_tie_stats), which bench.py prints and the -m gpu suite asserts to stay above 1e-10 px.  (Fast-pass coordinates need no entry: the guard band
// keeps them at least guard_eps from a tie by construction.)  Reference: src/mdBRIEFextractorOct.cpp:280-281, 295-296.
__device__ __forceinline__ double tie_frac(double v) { return fabs(v - rint(v)); }
// Round 5: watching is not enforcing.  A keypoint whose closest approach is inside b.tieBand is LISTED (tieList), and the host recomputes its descriptor with the
// libm the reference links before the results count as final (mcs_tiefix.hip: automatically for host outputs, mcs_extractor_fix_ties for device outputs).
__device__ __forceinline__ void tie_commit(const ExtractBuffers& b, int gw, double maxFrac) {
	unsigned long long* const tieMin = b.tieMin;
	if (!tieMin) return;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) maxFrac = fmax(maxFrac, __shfl_xor(maxFrac, o));
	const double dist = 0.5 - maxFrac;   // >= 0: |v - rint(v)| <= 1/2; exact (both within a binade of 1/2 or smaller)
	const unsigned long long bits = (unsigned long long)__double_as_longlong(dist);   // non-negative doubles order like their bit patterns
	if ((threadIdx.x & 63) != 0) return;
	// the stored minimum only ever falls: a (possibly stale) plain read that is already smaller spares the atomic — after the first few keypoints almost every wave
	if (bits < *reinterpret_cast<volatile unsigned long long*>(tieMin)) atomicMin(tieMin, bits);
	if (dist < b.tieBand && b.tieList) {   // NaN coordinates (dist = NaN) are not ties: they round the same way everywhere
		b.tieList[atomicAdd(b.tieCount, 1)] = (uint32_t)gw;
		if (b.tieTotal) atomicAdd(b.tieTotal, 1ull);
	}
}

template <int MODE, int NB>   // MODE 0 ORB, 1 dBRIEF, 2 mdBRIEF; NB = descSize/8 ballots.  lds: MODE > 0: [waves][x|y][npoints] coordinates + patch
__device__ __forceinline__ void describe_wave(const ExtractBuffers& b, int wavesPerImage, double* lds, int gw) {   // gw = image * wavesPerImage + slot
	const PyrDesc& d = *b.desc;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	constexpr int kWaveLds = coord_bytes(MODE, 128 * NB) + kPatchBytes;
	KeyPt kp_;
	if (!kp_prologue<MODE != 0>(b, wavesPerImage, gw, reinterpret_cast<uint8_t*>(lds) + (size_t)wave * kWaveLds + coord_bytes(MODE, 128 * NB), kp_)) return;
	const bool active = true;
	const int img = kp_.img, out = kp_.out, row = kp_.row, col = kp_.col;
	const float angle = kp_.angle;
	const double rayx = kp_.rayx, rayy = kp_.rayy, rayz = kp_.rayz;
	const Sampler sm = kp_.sm;
	constexpr int nballots = NB;
	uint8_t* dout = b.out_desc + ((size_t)img * b.outImgPitch + out) * b.outRowStride;
	uint8_t* mout = b.out_mask + ((size_t)img * b.outImgPitch + out) * b.outRowStride;

	if (MODE == 0) {
		const float DEG2RADf = (float)3.1415926535897932384626433832795 / 180.f;
		const double ang = (double)(angle * DEG2RADf);
		const double ax = cos(ang), ay = sin(ang);
		double tie = 0.0;
		for (int j = 0; j < nballots; ++j) {
			const int k = j * 64 + lane;
			const double x0 = c_pattern[4 * k], y0 = c_pattern[4 * k + 1], x1 = c_pattern[4 * k + 2], y1 = c_pattern[4 * k + 3];
			const double fx0 = x0 * ax - y0 * ay, fy0 = x0 * ay + y0 * ax, fx1 = x1 * ax - y1 * ay, fy1 = x1 * ay + y1 * ax;
			const int ix0 = __double2int_rn(fx0), iy0 = __double2int_rn(fy0);
			const int ix1 = __double2int_rn(fx1), iy1 = __double2int_rn(fy1);
			tie = fmax(fmax(tie, fmax(tie_frac(fx0), tie_frac(fy0))), fmax(tie_frac(fx1), tie_frac(fy1)));
			int t0, t1;
			sm.pair(row, col, iy0, ix0, iy1, ix1, t0, t1);
			const unsigned long long bits = __ballot(t0 < t1);
			if (lane == 0) {
				*reinterpret_cast<unsigned long long*>(dout + 8 * j) = bits;
				*reinterpret_cast<unsigned long long*>(mout + 8 * j) = 0ull;   // descriptorMasks = zeros (:1216)
			}
		}
		tie_commit(b, gw, tie);
		return;
	}

	// ---------------------------------------------------------------- dBRIEF / mdBRIEF
	// Waves are independent here (one keypoint per wave, private LDS slice, no block barrier).
	if (!active) return;
	constexpr int NP = 128 * NB;              // pattern points = 2*8*descSize
	constexpr int CH = NP / (2 * NB);         // chain elements folded into one point iteration (= 64)
	constexpr int YO = NP + 2;                // offset of the y array inside a pattern buffer (bank shift, see pat_doubles)
	constexpr int PD = pat_doubles(NP);
	double* buf = reinterpret_cast<double*>(reinterpret_cast<uint8_t*>(lds) + (size_t)wave * (coord_bytes(MODE, NP) + kPatchBytes));   // [pattern][x | y][NP] distorted coordinates
	const OcamDev& cam = b.cams[img];
	// The camera is the same for the whole wave: pull the backward polynomial and the affine terms into SGPRs ONCE.  (The
	// first version re-loaded every Horner coefficient through a vector global load inside a 12-trip loop per pattern point —
	// a ~200-cycle dependent load 36 864 times per keypoint; the kernel was latency-bound on it.)  Coefficients above the
	// model's degree are zero: res = 0*x + 0 stays +0 until the first real coefficient, so the padded fixed-length Horner
	// below is bit-identical to the reference's loop (include/misc.h:115-122).
	double cP[MCS_MAX_POLY];
#pragma unroll
	for (int i = 0; i < MCS_MAX_POLY; ++i) cP[i] = uniform_f64(&cam.invP[i]);
	const int cDeg = __builtin_amdgcn_readfirstlane(cam.invP_deg);
	const double cC = uniform_f64(&cam.c), cD = uniform_f64(&cam.d), cE = uniform_f64(&cam.e), cU0 = uniform_f64(&cam.u0), cV0 = uniform_f64(&cam.v0);
	auto w2i = [&](double x, double y, double z, double& u, double& v) {   // cCamModelGeneral_::WorldToImg (src/cam_model_omni.cpp:146-161)
		double norm = sqrt(x * x + y * y);
		if (norm == 0.0) norm = 1e-14;
		// The three IEEE divisions by `norm` share ONE refined reciprocal (recip_refined / div_shared above): norm is in
		// [1e-14, 1e4] and the numerators are pattern coordinates, so every quotient is bit-identical to `a / norm` — and the
		// point costs 19 FP64 instructions less.
		const double rn = recip_refined(norm);
		auto over_norm = [&](double a) { return div_shared(a, norm, rn); };
		const double theta = atan(over_norm(-z));
		double rho = 0.0;
		if (cDeg == 12) {
#pragma unroll
			for (int i = 11; i >= 0; --i) rho = rho * theta + cP[i];
		} else {
#pragma unroll
			for (int i = MCS_MAX_POLY - 1; i >= 0; --i) rho = rho * theta + cP[i];
		}
		const double uu = over_norm(x) * rho;
		const double vv = over_norm(y) * rho;
		u = uu * cC + vv * cD + cU0;
		v = uu * cE + vv + cV0;
	};
	// this lane's 2*NB pattern points (x, y as small integers), loaded once for all patterns
	// (kept packed — one dword per pair: x0,y0,x1,y1 as int8 — and widened at the point of use, to save 28 VGPRs)
	uint32_t ppk[NB];
#pragma unroll
	for (int j = 0; j < NB; ++j) ppk[j] = reinterpret_cast<const uint32_t*>(c_pattern)[j * 64 + lane];
	const double zc = -cam.p[0];              // distortPointsOcam: WorldToImg(x, y, -p1)
	double ukx = 0.0, uky = 0.0;
	if (d.undistort) {                        // undistortPointsOcam(pt*scale, scaleF = p[0]) (:1306-1317)
		const double p0 = uniform_f64(&cam.p[0]);
		ukx = -rayx / rayz * p0;
		uky = -rayy / rayz * p0;
	}
	double ang0, ang1 = 0.0, ang2 = 0.0;
	if (MODE == 1) {
		const float DEG2RADf = (float)3.1415926535897932384626433832795 / 180.f;
		ang0 = (double)(angle * DEG2RADf);
	} else {
		const float RHOf = 180.0f / 3.1415926535897932384626f;
		const double RHOd = 180.0 / 3.1415926535897932384626433832795028841971693993;
		const double rot = 20.0 / RHOd;
		ang0 = (double)(angle / RHOf);
		ang1 = ang0 + rot; ang2 = ang0 - rot;
	}
	// One pass = (optionally) the omni model of pattern `nxt` written to buffer wb, interleaved in the same straight-line
	// code with (optionally) the SEQUENTIAL coordinate sum of the pattern in buffer rb: even lanes accumulate sum(x),
	// odd lanes sum(y), p = 0..NP-1 in the reference's order (:264-276).  The sum is a 512-long dependent FP64 add chain;
	// folding 64 of its steps into each of the 2*NB point evaluations lets the scheduler hide its latency under the math.
	auto pass = [&](bool doMath, double ang, double* wb, bool doChain, const double* rb, double& sum) {
		if (doMath) {
			double ax, ay;
			sincos(ang, &ay, &ax);   // one argument reduction for both (same ocml kernels as cos() / sin())
#pragma unroll
			for (int t = 0; t < 2 * NB; ++t) {
				const int k = (t >> 1) * 64 + lane, e = t & 1;
				const double ptx = (double)(int)(signed char)(ppk[t >> 1] >> (16 * e)), pty = (double)(int)(signed char)(ppk[t >> 1] >> (16 * e + 8));
				const double xr = ptx * ax - pty * ay + ukx;
				const double yr = ptx * ay + pty * ax + uky;
				double u, v;
				if (MCS_ABLATE & 2) { u = xr; v = yr; } else w2i(xr, yr, zc, u, v);
				wb[2 * k + e] = u; wb[YO + 2 * k + e] = v;
				// the scheduler may interleave at most 4 point evaluations: all 8 of a pattern in flight cost 156 VGPRs (3 waves per SIMD); with this
				// fence and the 128-register cap below the kernel keeps 4 waves per SIMD with 2 spilled registers (2.64 -> 2.48 ms per 192 images)
				if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
			}
		}
		if (doChain && !(MCS_ABLATE & 1)) {
			const double* arr = rb + (lane & 1) * YO;
#ifndef MCS_CHAIN_UNROLL
#define MCS_CHAIN_UNROLL 16
#endif
#pragma unroll MCS_CHAIN_UNROLL
			for (int p = 0; p < NP; ++p) sum += arr[p];
		}
		(void)CH;
	};
	unsigned long long bitsMain[NB], agree[NB];
#pragma unroll
	for (int j = 0; j < NB; ++j) { bitsMain[j] = 0ull; agree[j] = ~0ull; }
	constexpr int npat = MODE == 2 ? 3 : 1;
	constexpr bool merged = MODE == 2 && MCS_MERGE_CHAINS;
	double sumAll = 0.0, tie = 0.0;
	if (merged) {
		// all three patterns first, then ONE dependent add chain: lane 2*pat + c accumulates coordinate c of pattern pat (each of the
		// six sums still runs p = 0..NP-1 in the reference's order; the other lanes repeat lane 0's work)
#pragma unroll
		for (int pat = 0; pat < npat; ++pat) pass(true, pat == 0 ? ang0 : (pat == 1 ? ang1 : ang2), buf + pat * PD, false, buf, sumAll);
		if (!(MCS_ABLATE & 1)) {
			const int cl = lane < 2 * npat ? lane : 0;
			const double* arr = buf + (cl >> 1) * PD + (cl & 1) * YO;
#pragma unroll 16
			for (int p = 0; p < NP; ++p) sumAll += arr[p];
		}
	}
#pragma unroll
	for (int pat = 0; pat < npat; ++pat) {
		double* cur = merged ? buf + pat * PD : buf;
		double sum = sumAll;
		if (!merged) {
			pass(true, pat == 0 ? ang0 : (pat == 1 ? ang1 : ang2), cur, false, cur, sum);
			pass(false, 0.0, cur, true, cur, sum);
		}
		const double mean = sum / (double)NP;
		const double meanX = __shfl(mean, merged ? 2 * pat : 0), meanY = __shfl(mean, merged ? 2 * pat + 1 : 1);
#pragma unroll
		for (int j = 0; j < NB; ++j) {
			const int k = j * 64 + lane;
			const double fx0 = cur[2 * k] - meanX, fy0 = cur[YO + 2 * k] - meanY, fx1 = cur[2 * k + 1] - meanX, fy1 = cur[YO + 2 * k + 1] - meanY;
			const int ix0 = __double2int_rn(fx0), iy0 = __double2int_rn(fy0);
			const int ix1 = __double2int_rn(fx1), iy1 = __double2int_rn(fy1);
			tie = fmax(fmax(tie, fmax(tie_frac(fx0), tie_frac(fy0))), fmax(tie_frac(fx1), tie_frac(fy1)));
			int t0 = ix0, t1 = iy1;
			if (!(MCS_ABLATE & 4)) sm.pair(row, col, iy0, ix0, iy1, ix1, t0, t1);
			const unsigned long long bits = __ballot(t0 < t1);
			if (pat == 0) bitsMain[j] = bits;
			else agree[j] &= ~(bits ^ bitsMain[j]);
		}
	}
	if (lane == 0) {
#pragma unroll
		for (int j = 0; j < NB; ++j) {
			*reinterpret_cast<unsigned long long*>(dout + 8 * j) = bitsMain[j];
			*reinterpret_cast<unsigned long long*>(mout + 8 * j) = MODE == 2 ? agree[j] : 0ull;
		}
	}
	tie_commit(b, gw, tie);
}

// Two entry points over the same body: the 128-register cap (4 waves per SIMD) pays off wherever the LDS slice of a wave lets 16 waves share a CU;
// mdBRIEF with 64-byte descriptors needs 18 KB of LDS per wave (8 waves per CU at most), there the cap would only cost 35 spilled registers.
template <int MODE, int NB>
__attribute__((amdgpu_waves_per_eu(4, 4)))
__global__ __launch_bounds__(MODE == 0 ? 256 : 64) void k_describe(ExtractBuffers b, int wavesPerImage) {
	extern __shared__ __attribute__((aligned(16))) double lds[];
	describe_wave<MODE, NB>(b, wavesPerImage, lds, blockIdx.x * (MODE == 0 ? 4 : 1) + (threadIdx.x >> 6));
}
template <int MODE, int NB>
__global__ __launch_bounds__(MODE == 0 ? 256 : 64) void k_describe_wide(ExtractBuffers b, int wavesPerImage) {
	extern __shared__ __attribute__((aligned(16))) double lds[];
	describe_wave<MODE, NB>(b, wavesPerImage, lds, blockIdx.x * (MODE == 0 ? 4 : 1) + (threadIdx.x >> 6));
}

// The exact pass over a list of keypoint slots (one wave per block, blocks stride over the list; the lists are short): the fast pass's fallback list, and the
// pre-list k_orient_b fills.
template <int MODE, int NB>
__global__ __launch_bounds__(64) void k_describe_list(ExtractBuffers b, int wavesPerImage, const int* count, const uint32_t* list) {
	extern __shared__ __attribute__((aligned(16))) double lds[];
	// A handful of single waves on the critical path of the step, each a long dependent chain: raise their issue priority over whatever else shares the
	// SIMD (the deferred matcher of the previous step runs beside them: 0.19 ms average, 0.56 ms worst for ~100 keypoints without this, 0.03 ms alone).
	__builtin_amdgcn_s_setprio(3);
	const int n = *count;
	if (blockIdx.x == 0 && threadIdx.x == 0 && b.fbStats) atomicAdd(b.fbStats, (unsigned long long)n);
	for (int i = blockIdx.x; i < n; i += gridDim.x) describe_wave<MODE, NB>(b, wavesPerImage, lds, (int)list[i]);
}

// The exact pass over a SHORT list, split for latency (round 5).  The fallback list holds a few keypoints per batch (one or two per multi-frame), and the pass
// sits on the critical path between the fast pass and the matcher: one wave per keypoint is a chain of 2 * NB * npat = 24 evaluations of the omni model per
// lane (38 us for ONE keypoint, whatever the list's length).  Here a WORKGROUP takes the keypoint: wave (pattern, part) evaluates 2 * NB / PARTS of its pattern's
// point iterations into a coordinate buffer the workgroup shares, every wave then runs its pattern's sequential coordinate sum (the reference's order, redundantly in
// the PARTS waves of a pattern), compares its own share of the pairs, and wave 0 joins the three patterns' bits.  Statement for statement the arithmetic of
// describe_wave — same sincos, same shared reciprocal, same Horner, same sum order — so the bits are the same.
template <int MODE, int NB> struct SplitGeom {
	static constexpr int NPAT = MODE == 2 ? 3 : 1, PARTS = NB >= 4 ? 4 : 2, WAVES = NPAT * PARTS, TP = 2 * NB / PARTS, JP = NB / PARTS;
	static constexpr int NP = 128 * NB, PD = pat_doubles(NP);
	static constexpr size_t lds_bytes() { return (size_t)NPAT * PD * 8 + kPatchBytes + (size_t)NPAT * NB * 8 + (size_t)WAVES * 8; }
};
template <int MODE, int NB>
__global__ __launch_bounds__((64 * SplitGeom<MODE, NB>::WAVES)) void k_describe_list_split(ExtractBuffers b, int wavesPerImage, const int* count, const uint32_t* list) {
	typedef SplitGeom<MODE, NB> G;
	constexpr int NP = G::NP, YO = NP + 2, PD = G::PD;
	extern __shared__ __attribute__((aligned(16))) double lds[];
	double* coords = lds;                                                                   // [pattern][x | y][NP]
	uint8_t* patch = reinterpret_cast<uint8_t*>(lds) + (size_t)G::NPAT * PD * 8;          // one blurred patch, written (identically) by every wave
	unsigned long long* xbits = reinterpret_cast<unsigned long long*>(patch + kPatchBytes); // [pattern][NB] the compare bits
	double* xtie = reinterpret_cast<double*>(xbits + G::NPAT * NB);                        // [wave] closest approach to a rounding tie
	__builtin_amdgcn_s_setprio(3);
	const PyrDesc& d = *b.desc;
	const int n = *count;
	if (blockIdx.x == 0 && threadIdx.x == 0 && b.fbStats) atomicAdd(b.fbStats, (unsigned long long)n);
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int pat = wave / G::PARTS, part = wave - pat * G::PARTS;
	for (int i = blockIdx.x; i < n; i += gridDim.x) {
		const int gw = (int)list[i];
		KeyPt kp_;
		const bool ok = kp_prologue<true>(b, wavesPerImage, gw, patch, kp_);   // the same for every wave of the workgroup
		double* cur = coords + pat * PD;
		if (ok) {
			const OcamDev& cam = b.cams[kp_.img];
			double cP[MCS_MAX_POLY];
#pragma unroll
			for (int k = 0; k < MCS_MAX_POLY; ++k) cP[k] = uniform_f64(&cam.invP[k]);
			const int cDeg = __builtin_amdgcn_readfirstlane(cam.invP_deg);
			const double cC = uniform_f64(&cam.c), cD = uniform_f64(&cam.d), cE = uniform_f64(&cam.e), cU0 = uniform_f64(&cam.u0), cV0 = uniform_f64(&cam.v0);
			auto w2i = [&](double x, double y, double z, double& u, double& v) {   // describe_wave's, verbatim
				double norm = sqrt(x * x + y * y);
				if (norm == 0.0) norm = 1e-14;
				const double rn = recip_refined(norm);
				auto over_norm = [&](double a) { return div_shared(a, norm, rn); };
				const double theta = atan(over_norm(-z));
				double rho = 0.0;
				if (cDeg == 12) {
#pragma unroll
					for (int k = 11; k >= 0; --k) rho = rho * theta + cP[k];
				} else {
#pragma unroll
					for (int k = MCS_MAX_POLY - 1; k >= 0; --k) rho = rho * theta + cP[k];
				}
				const double uu = over_norm(x) * rho;
				const double vv = over_norm(y) * rho;
				u = uu * cC + vv * cD + cU0;
				v = uu * cE + vv + cV0;
			};
			const double zc = -cam.p[0];
			double ukx = 0.0, uky = 0.0;
			if (d.undistort) {
				const double p0 = uniform_f64(&cam.p[0]);
				ukx = -kp_.rayx / kp_.rayz * p0;
				uky = -kp_.rayy / kp_.rayz * p0;
			}
			double ang;
			if (MODE == 1) {
				const float DEG2RADf = (float)3.1415926535897932384626433832795 / 180.f;
				ang = (double)(kp_.angle * DEG2RADf);
			} else {
				const float RHOf = 180.0f / 3.1415926535897932384626f;
				const double RHOd = 180.0 / 3.1415926535897932384626433832795028841971693993;
				const double rot = 20.0 / RHOd;
				const double ang0 = (double)(kp_.angle / RHOf);
				ang = pat == 0 ? ang0 : (pat == 1 ? ang0 + rot : ang0 - rot);
			}
			double ax, ay;
			sincos(ang, &ay, &ax);
#pragma unroll
			for (int tt = 0; tt < G::TP; ++tt) {
				const int t = part * G::TP + tt;
				const int k = (t >> 1) * 64 + lane, e = t & 1;
				const uint32_t pp = reinterpret_cast<const uint32_t*>(c_pattern)[k];
				const double ptx = (double)(int)(signed char)(pp >> (16 * e)), pty = (double)(int)(signed char)(pp >> (16 * e + 8));
				const double xr = ptx * ax - pty * ay + ukx;
				const double yr = ptx * ay + pty * ax + uky;
				double u, v;
				w2i(xr, yr, zc, u, v);
				cur[2 * k + e] = u; cur[YO + 2 * k + e] = v;
			}
		}
		__syncthreads();
		if (ok) {
			double sum = 0.0;
			const double* arr = cur + (lane & 1) * YO;   // even lanes sum(x), odd lanes sum(y), p = 0 .. NP-1 in the reference's order (:264-276)
#pragma unroll 16
			for (int p = 0; p < NP; ++p) sum += arr[p];
			const double mean = sum / (double)NP;
			const double meanX = __shfl(mean, 0), meanY = __shfl(mean, 1);
			double tie = 0.0;
#pragma unroll
			for (int jj = 0; jj < G::JP; ++jj) {
				const int j = part * G::JP + jj;
				const int k = j * 64 + lane;
				const double fx0 = cur[2 * k] - meanX, fy0 = cur[YO + 2 * k] - meanY, fx1 = cur[2 * k + 1] - meanX, fy1 = cur[YO + 2 * k + 1] - meanY;
				const int ix0 = __double2int_rn(fx0), iy0 = __double2int_rn(fy0);
				const int ix1 = __double2int_rn(fx1), iy1 = __double2int_rn(fy1);
				tie = fmax(fmax(tie, fmax(tie_frac(fx0), tie_frac(fy0))), fmax(tie_frac(fx1), tie_frac(fy1)));
				int t0, t1;
				kp_.sm.pair(kp_.row, kp_.col, iy0, ix0, iy1, ix1, t0, t1);
				const unsigned long long bits = __ballot(t0 < t1);
				if (lane == 0) xbits[pat * NB + j] = bits;
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) tie = fmax(tie, __shfl_xor(tie, o));
			if (lane == 0) xtie[wave] = tie;
		}
		__syncthreads();
		if (ok && wave == 0) {
			if (lane == 0) {
				uint8_t* dout = b.out_desc + ((size_t)kp_.img * b.outImgPitch + kp_.out) * b.outRowStride;
				uint8_t* mout = b.out_mask + ((size_t)kp_.img * b.outImgPitch + kp_.out) * b.outRowStride;
#pragma unroll
				for (int j = 0; j < NB; ++j) {
					const unsigned long long m = xbits[j];
					unsigned long long agree = 0ull;
					if (MODE == 2) agree = ~((xbits[NB + j] ^ m) | (xbits[2 * NB + j] ^ m));   // both +-20 degree tests agree with the main test (:468-475)
					*reinterpret_cast<unsigned long long*>(dout + 8 * j) = m;
					*reinterpret_cast<unsigned long long*>(mout + 8 * j) = agree;
				}
			}
			double tie = 0.0;
#pragma unroll
			for (int w = 0; w < G::WAVES; ++w) tie = fmax(tie, xtie[w]);
			tie_commit(b, gw, tie);
		}
		__syncthreads();   // the next keypoint's prologue rewrites the patch and the coordinates
	}
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The fast pass.  Same pattern rotation angles (prepared by k_orient_b), then per pattern point
//      xr, yr   = rotation + undistorted keypoint                     (FMA form)
//      s        = xr^2 + yr^2
//      G        = rho(atan(p0 / sqrt(s))) / sqrt(s)                   from the camera's table (mcs_common.h kG*): row = exponent and top mantissa bits of s,
//                                                                     read from its bit pattern; the row's Taylor polynomial in the low mantissa fraction
//                                                                     (exact) — no square root, no reciprocal, no atan, no backward polynomial
//      u, v     = affine(xr * G, yr * G)                              without the principal point: it cancels against the pattern mean
// and a wave tree sum for the mean.  None of this is the reference's rounding; it is only USED when every one of the keypoint's
// 2 * npat * 2*8*descSize coordinates (minus the mean) stays clear of the rounding ties by more than the guard band b.guardEps, which the host keeps
// above the worst-case difference between this arithmetic and the reference's (describe_fast_bound in mcs_capi.hip).  Otherwise the
// keypoint goes to the exact pass.  A point outside the table (s < 2^kGE0, s >= 2^kGE1, NaN / Inf) takes the same way out.
#ifndef MCS_FAST_FENCE
#define MCS_FAST_FENCE 4
#endif
#ifndef MCS_FAST_ABLATE
#define MCS_FAST_ABLATE 0   // A/B experiments only (tools/ab_describe.sh): 1 no sampling, 2 no omni model, 4 no guard / rounding checks
#endif
constexpr int kFastWaves = MCS_FAST_WAVES;   // waves per workgroup of the fast pass (each with its own keypoints, all of one image): they share the camera's table in LDS

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}

// Both coordinate sums of a pattern over the wave, on the VALU (the ds_bpermute form above is six dependent LDS round trips per sum): the halves of the
// wave exchange so that lanes 0..31 carry x partial sums and lanes 32..63 y partial sums (v_permlane32_swap), the row pairs are folded
// (v_permlane16_swap) and the 16 lanes of a row by four DPP rotations; the totals are read back as wave-uniform scalars.  A balanced tree of depth 6 per
// sum, like the shuffle form (describe_fast_bound counts its roundings, not its order).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
#ifndef MCS_DPP_OLD_SELF
#define MCS_DPP_OLD_SELF 0   // 1 (round 5, A/B): old = the source itself — the compiler then copies the source in front of every v_mov_b32_dpp (two extra moves per step)
#endif
	const int vl = __double2loint(v), vh = __double2hiint(v);   // a rotation inside the rows of 16: every lane has a source, so with bound_ctrl the old value is dead and no copy is made
	const int lo = MCS_DPP_OLD_SELF ? __builtin_amdgcn_update_dpp(vl, vl, CTRL, 0xF, 0xF, false) : __builtin_amdgcn_update_dpp(0, vl, CTRL, 0xF, 0xF, true);
	const int hi = MCS_DPP_OLD_SELF ? __builtin_amdgcn_update_dpp(vh, vh, CTRL, 0xF, 0xF, false) : __builtin_amdgcn_update_dpp(0, vh, CTRL, 0xF, 0xF, true);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wave_sum2_f64(double sx, double sy, double& totx, double& toty) {
	// lanes < 32: x.lo + x.hi ; lanes >= 32: y.lo + y.hi        (v_permlane32_swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second)
	const auto l = __builtin_amdgcn_permlane32_swap(__double2loint(sx), __double2loint(sy), false, false);
	const auto h = __builtin_amdgcn_permlane32_swap(__double2hiint(sx), __double2hiint(sy), false, false);
	double z = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
	// rows (16 lanes) 0 + 1 and 2 + 3                             (v_permlane16_swap: odd rows of the first operand <-> even rows of the second)
	const auto l2 = __builtin_amdgcn_permlane16_swap(__double2loint(z), __double2loint(z), false, false);
	const auto h2 = __builtin_amdgcn_permlane16_swap(__double2hiint(z), __double2hiint(z), false, false);
	z = __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
	z += dpp_f64<0x128>(z);   // row_ror:8
	z += dpp_f64<0x124>(z);   // row_ror:4
	z += dpp_f64<0x122>(z);   // row_ror:2
	z += dpp_f64<0x121>(z);   // row_ror:1
	totx = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(z), 0), __builtin_amdgcn_readlane(__double2loint(z), 0));
	toty = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(z), 32), __builtin_amdgcn_readlane(__double2loint(z), 32));
}

// one pattern point through the fast arithmetic; `bad` collects points the table does not cover.  (Requesting the row of point t + 1 before the Horner
// chain of point t — a hand-made software pipeline — was built and measured: 0.712 against 0.705 ms, not kept.)
struct FastCam { double c, d, e; };
#ifndef MCS_FAST_CLAMP
#define MCS_FAST_CLAMP 1   // clamp the row index before the LDS gather (0, A/B: an unclamped gather measured the same, 0.594 against 0.602 ms)
#endif
// `top` collects the largest row index seen (unsigned: below the table, negative, NaN and Inf all come out huge): one compare per pattern instead of one per
// point; the gather itself reads the clamped row, its value is never used for such a point (the keypoint goes to the exact pass).
template <bool CLAMP, class Tab>
__device__ __forceinline__ void fast_w2i(const FastCam& C, Tab tab, double xr, double yr, double& u, double& v, unsigned& top) {
	const double s = __builtin_fma(xr, xr, yr * yr);
	const unsigned hi = (unsigned)__double2hiint(s), lo = (unsigned)__double2loint(s);
	const unsigned idx = (hi >> (20 - kGM)) - (unsigned)((1023 + kGE0) << kGM);            // (exponent - kGE0) * 2^kGM + top kGM mantissa bits
	top = max(top, idx);
	unsigned row = CLAMP || MCS_FAST_CLAMP ? (idx < (unsigned)kGRows ? idx : (unsigned)(kGRows - 1)) : (idx & 0xFFFFFFu);   // 24 bits: v_mad_u32_u24 forms the byte offset
	if (MCS_FAST_ABLATE & 16) row = 0;   // A/B: every lane reads the same row (LDS broadcast: no bank conflicts, no gather)
	const double frac = __hiloint2double((int)((hi & ((1u << (20 - kGM)) - 1u)) | 0x3FF00000u), (int)lo);   // 1 + the mantissa bits below the bin index
	const double tau = frac - (1.0 + 1.0 / (double)(2 << kGM));                           // exact
#if MCS_G_PACKED
	// The row (mcs_common.h kGDev*): 48 bytes in three 16-byte slots — [g0 g1] [g2 g3] as doubles, [g4 g5 g6 0] as floats.  From LDS that is three ds_read_b128
	// (12 array cycles per wave; the 56-byte rows of doubles went as three ds_read2_b64 + one ds_read_b64: 26, and collided in rows r, r + 16), and rows 16 bytes
	// aligned at a stride of three slots collide only for r = r' mod 16 within a 16-lane group.  The tail g4 + g5 tau + g6 tau^2 is evaluated in float (its share
	// of G is below tau^4 = 2^-24 |G|: GTabInfo.f32U bounds what the float roundings add), converted once and carried on in double.
	typedef double f64x2 __attribute__((ext_vector_type(2)));
	typedef float f32x4 __attribute__((ext_vector_type(4)));
	const auto rowp = reinterpret_cast<const uint8_t*>(tab) + row * (unsigned)kGDevRowBytes;
	const f64x2 g01 = *reinterpret_cast<const f64x2*>(rowp), g23 = *reinterpret_cast<const f64x2*>(rowp + 16);
	const f32x4 g46 = *reinterpret_cast<const f32x4*>(rowp + 32);
	const float tf = (float)tau;
	const float tl = __builtin_fmaf(__builtin_fmaf(g46.z, tf, g46.y), tf, g46.x);
	double G = __builtin_fma((double)tl, tau, g23.y);
	G = __builtin_fma(G, tau, g23.x);
	G = __builtin_fma(G, tau, g01.y);
	G = __builtin_fma(G, tau, g01.x);
#else
	const auto g = tab + row * kGRow;
	double gc[kGRow];
#pragma unroll
	for (int i = 0; i < kGRow; ++i) gc[i] = g[i];
	double G = gc[kGDeg];
#pragma unroll
	for (int i = kGDeg - 1; i >= 0; --i) G = __builtin_fma(G, tau, gc[i]);
#endif
	const double uu = xr * G, vv = yr * G;
	u = __builtin_fma(uu, C.c, vv * C.d);
	v = __builtin_fma(uu, C.e, vv);
}

// ---- what the fast pass needs per keypoint besides the patch --------------------------------------------------------------------------------------
// The fast pass runs 4 waves per SIMD (registers); work that is only 33 lanes wide (IC_Angle: now in the oct-tree kernel's tail, mcs_orient.h) or identical
// in all 64 lanes (ImgToWorld of the keypoint, the sincos of the pattern angles) is prepared ahead of it:
//   k_orient_b   one THREAD per output row: slot -> (level, position in the level's selection), the keypoint record (E8: level coordinates -> image coordinates
//                with the FLOAT scale, :1305,1331), ImgToWorld (ray, E9), the undistorted keypoint, the pattern angles and their sin / cos; it also decides which
//                keypoints the fast pass cannot serve at all (camera beyond the band, non-finite undistorted position) and puts them on the exact
//                pass's pre-list, which runs BESIDE the fast pass
// The arithmetic is the exact pass's, statement for statement (the exact pass still does all of it itself and writes the same values).  The scratch is one
// array per field (KpAuxSoA): a thread-per-keypoint kernel then writes whole cache lines (120-byte records cost 2.3x their size in HBM writes).
template <int MODE>
__global__ __launch_bounds__(256) void k_orient_b(ExtractBuffers b, int wavesPerImage, int nslots) {
	const int gw = blockIdx.x * 256 + threadIdx.x;
	if (gw >= nslots) return;
	KpAuxSoA A; A.carve(b.aux, nslots);
	const PyrDesc& d = *b.desc;
	const int img = gw / wavesPerImage, s = gw - img * wavesPerImage;
	int total, level, pos;
	find_slot(d, b.selCount + (size_t)img * d.nlevels, s, level, pos, total);
	if (s == 0) {
		b.nkp[img] = total < d.kpCap ? total : d.kpCap;
		if (total > d.kpCap) atomicExch(b.status, MCS_ERR_CAPACITY);
	}
	if (level < 0 || s >= d.kpCap) { A.lvl[gw] = -1; return; }
	const LevelInfo& L = d.lv[level];
	const size_t si = (size_t)img * d.selPerImage + L.selBase + pos;
	const uint32_t rec = b.sel[si];
	const float angle = b.selAngle[si];
	const int col = (int)(rec & 0xFFF) + kMinBorder, row = (int)((rec >> 12) & 0xFFF) + kMinBorder;
	float pxf = (float)col, pyf = (float)row;
	if (level != 0) { pxf = pxf * L.scale; pyf = pyf * L.scale; }
	mcs_keypoint kp;
	kp.x = pxf; kp.y = pyf; kp.size = L.kpSize; kp.angle = angle; kp.response = (float)(rec >> 24); kp.octave = level; kp.class_id = -1;
	b.kps[(size_t)img * d.kpCap + s] = kp;
	A.rc[gw] = row | (col << 16);
	A.poff[gw] = (unsigned)(L.off + (size_t)(row - kPatchR) * L.stride + (col - kPatchR));   // the fast pass requests the next keypoint's patch from this alone
	const int lvlWord = level | (L.stride << 16);                                            // (level coordinates < 4096: the stride fits 15 bits)
	if constexpr (MODE == 0) {
		// ORB (round 6): what is identical in all 64 lanes of the descriptor wave — the ray of the keypoint and the rotation's cos / sin, ~350 of the ~700
		// instructions the wave spent per keypoint — is done here by ONE thread; k_describe_orb only stages the patch and forms the 256 bits.  Same statements as
		// describe_wave<0> (cameras are optional in ORB mode: no ray without them).
		if (b.cams && b.rays) {
			double rx, ry, rz;
			img2world(b.cams[img], (double)pxf, (double)pyf, rx, ry, rz);
			double* rp = b.rays + ((size_t)img * d.kpCap + s) * 3;
			rp[0] = rx; rp[1] = ry; rp[2] = rz;
		}
		const float DEG2RADf = (float)3.1415926535897932384626433832795 / 180.f;
		const double a0 = (double)(angle * DEG2RADf);
		double* D = A.d8 + gw;
		D[(size_t)2 * nslots] = cos(a0); D[(size_t)3 * nslots] = sin(a0);
		A.lvl[gw] = lvlWord;
		return;
	}
	const OcamDev& cam = b.cams[img];
	double rayx, rayy, rayz;
	img2world(cam, (double)pxf, (double)pyf, rayx, rayy, rayz);
	if (b.rays) {
		double* rp = b.rays + ((size_t)img * d.kpCap + s) * 3;
		rp[0] = rayx; rp[1] = rayy; rp[2] = rayz;
	}
	double ukx = 0.0, uky = 0.0;
	if (d.undistort) {
		const double p0 = cam.p[0];
		ukx = -rayx / rayz * p0;
		uky = -rayy / rayz * p0;
	}
	double* D8 = A.d8 + gw;
	const size_t S = (size_t)nslots;
	D8[0] = ukx; D8[S] = uky;
	// a camera beyond the guard band, or a keypoint whose undistorted position is not finite (a ray in the image plane): not for the fast pass
	const double n2 = ukx * ukx + uky * uky;
	if (cam.fastOk == 0 || !(n2 < 1.0e300)) {
		A.lvl[gw] = lvlWord | kAuxExact;
		b.preList[atomicAdd(b.preCount, 1)] = (uint32_t)gw;
		return;   // the exact pass computes its own angles
	}
	A.lvl[gw] = lvlWord;
	double ang[3] = {0.0, 0.0, 0.0};
	if (MODE == 1) {
		const float DEG2RADf = (float)3.1415926535897932384626433832795 / 180.f;
		ang[0] = (double)(angle * DEG2RADf);
	} else {
		const float RHOf = 180.0f / 3.1415926535897932384626f;
		const double RHOd = 180.0 / 3.1415926535897932384626433832795028841971693993;
		const double rot = 20.0 / RHOd;
		ang[0] = (double)(angle / RHOf);
		ang[1] = ang[0] + rot; ang[2] = ang[0] - rot;
	}
#pragma unroll
	for (int k = 0; k < (MODE == 2 ? 3 : 1); ++k) {
		double sn, cs;
		sincos(ang[k], &sn, &cs);
		D8[(2 + 2 * k) * S] = cs; D8[(3 + 2 * k) * S] = sn;
	}
}

#ifndef MCS_FAST_WAVES_PER_EU
#define MCS_FAST_WAVES_PER_EU 4
#endif
#ifndef MCS_FAST_BLOCKS
#define MCS_FAST_BLOCKS (4096 / MCS_FAST_WAVES)   // 256 CUs x 16 resident waves (registers: 4 waves per SIMD)
#endif
constexpr int kFastBlocks = MCS_FAST_BLOCKS;
static_assert(kSlotAlign % kFastWaves == 0, "a group's keypoint slots must belong to one image");

// The fast pass's sampler: the LDS patch serves practically every sample; the general path (blurred level / bordered raw level) looks its level up only
// when it is taken, so that none of it occupies registers in the hot loop.
struct LazySampler {
	const ExtractBuffers* b; int img, level;
	const uint8_t* patch;
	// `reach` collects offset + 4096 of every sample that takes the general path: the caller sends the keypoint to the exact pass unless all stay below 8192.
	// (A sample inside the patch is within +-21: nothing to collect on the fast path — and an offset outside [-4096, 4096), NaN included, can never pass for inside.)
	// one sample at offset (dy, dx) from the keypoint through the general path (blurred level / bordered raw level; the patch too, for a lane whose own point is inside)
	__device__ __forceinline__ int sample(int dy, int dx, int row, int col) const {
		const PyrDesc& d = *b->desc;
		const LevelInfo& L = d.lv[level];
		Sampler sm;
		int rstride;
		sm.raw = level_ptr(*b, d, img, level, &rstride);
		sm.rstride = rstride;
		sm.blur = b->blur + (size_t)img * d.pyrBytes + L.off; sm.bstride = L.stride;
		sm.w = L.w; sm.h = L.h;
		sm.patch = patch; sm.prow = row - kPatchR; sm.pcol = col - kPatchR;
		return sm.at<kFPitch>(row + dy, col + dx);
	}
	__device__ __forceinline__ void pair(int row, int col, int dy0, int dx0, int dy1, int dx1, int& t0, int& t1, unsigned& reach) const {
		const unsigned r0 = (unsigned)(dy0 + kPatchR), c0 = (unsigned)(dx0 + kPatchR), r1 = (unsigned)(dy1 + kPatchR), c1 = (unsigned)(dx1 + kPatchR);
		const bool inside = max(max(r0, c0), max(r1, c1)) < (unsigned)kPatchRows;
		if (!__any(!inside)) {
			// explicitly LDS: left as loads through the generic `patch` pointer, the compiler merges this load with the general path's (below) into ONE flat load
			// behind the branch — slower than ds_read_u8, and every flat load waits on vmcnt(0), i.e. for the patch requested for the NEXT keypoint
			typedef const __attribute__((address_space(3))) uint8_t lds_byte;
			lds_byte* const pl = (lds_byte*)patch;
			t0 = pl[r0 * kFPitch + c0];
			t1 = pl[r1 * kFPitch + c1];
		} else {
			reach |= (unsigned)(dy0 + 4096) | (unsigned)(dx0 + 4096) | (unsigned)(dy1 + 4096) | (unsigned)(dx1 + 4096);
			const PyrDesc& d = *b->desc;
			const LevelInfo& L = d.lv[level];
			Sampler sm;
			int rstride;
			sm.raw = level_ptr(*b, d, img, level, &rstride);
			sm.rstride = rstride;
			sm.blur = b->blur + (size_t)img * d.pyrBytes + L.off; sm.bstride = L.stride;
			sm.w = L.w; sm.h = L.h;
			sm.patch = patch; sm.prow = row - kPatchR; sm.pcol = col - kPatchR;
			t0 = sm.at<kFPitch>(row + dy0, col + dx0);
			t1 = sm.at<kFPitch>(row + dy1, col + dx1);
			asm volatile("" : "+v"(t0), "+v"(t1));   // (keeps the two paths' loads apart, see above)
		}
	}
};

// One keypoint of the fast pass by one wave: the patch is in LDS, (ukx, uky) and the pattern angles' cos / sin come from k_orient_b.  Returns false if the
// keypoint has to take the exact pass (a coordinate in the guard band, a point outside the table, out of range).
// patD: this lane's pattern points as doubles in LDS — point t at patD[64 t] = (x, y): one ds_read_b128 where unpacking the packed bytes took four VALU instructions
// per point and pattern (hoisted out of the loops the 4 NB doubles would cost 8 NB registers)
// ahead(): the walk's requests for the keypoints to come, issued right behind this keypoint's first LDS read (k_describe_fast says why there).
template <int MODE, int NB, class Ahead>
__device__ __forceinline__ bool fast_keypoint(const ExtractBuffers& b, const FastCam& C, const double* tabLds, const LazySampler& sm, int row, int col,
                                              double ukx, double uky, const double (&axc)[3], const double (&ays)[3], const double2* patD,
                                              unsigned long long (&bitsMain)[NB], unsigned long long (&agree)[NB], Ahead&& ahead) {
	constexpr int NP = 128 * NB;
	// The rounding and its guard in fixed point: y = coordinate - mean + 1.5 * 2^20 + 0.5 lies in [2^20, 2^21) where one unit of the high word is one pixel and
	// the low word is the fraction in units of 2^-32.  floor(y) is the rounded offset (ties are excluded by the guard), the fraction within guardUnits of 0 /
	// 2^32 means the coordinate is within the band of a tie, and high word - hiword(1.5 * 2^20 - 4096) is offset + 4096, in [0, 8192) iff |offset| <= 4096
	// (NaN, Inf and anything outside the binade give a huge value).  Costs two roundings of 2^-33 (describe_fast_bound adds them).
	const double kFix = 1572864.5;                 // 1.5 * 2^20 + 0.5
	const unsigned kHiBase = 0x4137F000u;          // high word of 1.5 * 2^20 - 4096
	const unsigned guardUnits = (unsigned)__builtin_ceil(b.guardEps * 4294967296.0);
#pragma unroll
	for (int j = 0; j < NB; ++j) { bitsMain[j] = 0ull; agree[j] = ~0ull; }
	constexpr int npat = MODE == 2 ? 3 : 1;
#pragma unroll
	for (int pat = 0; pat < npat; ++pat) {
		const double ax = axc[pat], ay = ays[pat];
		double u[2 * NB], v[2 * NB];
		double sumx = 0.0, sumy = 0.0;
		bool bad = false;
		unsigned top = 0;
		// opaque per pattern — the points are read again for every pattern, not held across them — as an LDS byte offset: an opaque generic pointer would make
		// these flat loads, which count on vmcnt too and would wait for the patch just requested
		typedef double f64x2 __attribute__((ext_vector_type(2)));
		typedef const __attribute__((address_space(3))) f64x2 lds_f64x2;
		unsigned pdo = (unsigned)(uintptr_t)(lds_f64x2*)patD;
		asm volatile("" : "+v"(pdo));
		lds_f64x2* const pd = (lds_f64x2*)(uintptr_t)pdo;
		auto rotate = [&](int t, double& xr, double& yr) {
			const f64x2 p = pd[64 * t];
			if (pat == 0 && t == 0) ahead();
			xr = __builtin_fma(p.x, ax, __builtin_fma(-p.y, ay, ukx));
			yr = __builtin_fma(p.x, ay, __builtin_fma(p.y, ax, uky));
		};
		if (MCS_FAST_ABLATE & 2) {
#pragma unroll
			for (int t = 0; t < 2 * NB; ++t) { rotate(t, u[t], v[t]); sumx += u[t]; sumy += v[t]; }
		} else {
#pragma unroll
			for (int t = 0; t < 2 * NB; ++t) {
				double xr, yr;
				rotate(t, xr, yr);
				fast_w2i<false>(C, tabLds, xr, yr, u[t], v[t], top);
				sumx += u[t]; sumy += v[t];
				if ((t & (MCS_FAST_FENCE - 1)) == MCS_FAST_FENCE - 1) __builtin_amdgcn_sched_barrier(0);   // at most MCS_FAST_FENCE point evaluations in flight (registers)
			}
		}
		double totx, toty;
		wave_sum2_f64(sumx, sumy, totx, toty);
		const double meanX = totx * (1.0 / (double)NP), meanY = toty * (1.0 / (double)NP);
		bad |= top >= (unsigned)kGRows;   // a point below / above the table, NaN, Inf
		bad |= !(fabs(meanX) < 16384.0) || !(fabs(meanY) < 16384.0);
		// the guard's half-width is folded into the addend: the low word of y then reads (fraction + g) mod 2^32, and "within the band of a tie" is low word < 2g —
		// no integer add per coordinate.  (Where the fraction + g wraps, the carry lands in the high word: such a coordinate is in the band, its keypoint leaves
		// for the exact pass and the offset is never used.)  The smallest low word of the pattern is kept (v_min3_u32) and compared once.
		const double gAdd = (double)guardUnits * (1.0 / 4294967296.0);
		const double cmx = (kFix - meanX) + gAdd, cmy = (kFix - meanY) + gAdd;   // exact: one unit of the last place of kFix - mean is 2^-32 (or more)
		unsigned reach = 0, minlo = 0xFFFFFFFFu;
		// rounding, guard and the pair's test in one sweep (the samples of a keypoint that turns out to need the exact pass are wasted, nothing else: its
		// bits are not written)
#pragma unroll
		for (int j = 0; j < NB; ++j) {
			int ix[2], iy[2];
#pragma unroll
			for (int e = 0; e < 2; ++e) {
				const double yx = u[2 * j + e] + cmx, yy = v[2 * j + e] + cmy;
				const unsigned hx = (unsigned)__double2hiint(yx) - kHiBase, hy = (unsigned)__double2hiint(yy) - kHiBase;
				if (!(MCS_FAST_ABLATE & 4)) minlo = min(minlo, min((unsigned)__double2loint(yx), (unsigned)__double2loint(yy)));
				ix[e] = (int)hx - 4096; iy[e] = (int)hy - 4096;
			}
			int t0 = ix[0], t1 = iy[1];
			if (!(MCS_FAST_ABLATE & 1)) sm.pair(row, col, iy[0], ix[0], iy[1], ix[1], t0, t1, reach);
			const unsigned long long bits = __ballot(t0 < t1);
			if (pat == 0) bitsMain[j] = bits;
			else agree[j] &= ~(bits ^ bitsMain[j]);
		}
		if (!(MCS_FAST_ABLATE & 4)) bad |= minlo < 2u * guardUnits;
		if (!(MCS_FAST_ABLATE & 4)) bad |= reach >= 8192u;
		if (__any(bad)) return false;
	}
	return true;
}

#ifndef MCS_FAST_UNIQ
#define MCS_FAST_UNIQ 1   // 1: the pattern's distinct points (round 5, UPat above); 0: every point of the pattern (round 4's fast_keypoint), kept for A/B
#endif
// The LDS tables of the distinct-point form as one workgroup sees them, and this wave's byte array
struct UTabs {
	const double2* upat; const double* uw; const uint32_t* gidx;   // already offset by the lane (upat + lane, uw + lane, gidx + lane * NB)
	uint8_t* vals;                                                 // this wave's R * 64 sampled bytes
};
__host__ __device__ constexpr size_t upat_bytes(int nb) { return (size_t)(nb == 2 ? 4 : nb == 4 ? 6 : 9) * 64 * sizeof(double2); }
__host__ __device__ constexpr size_t uw_bytes(int nb) { return (size_t)(nb == 2 ? 1 : nb == 4 ? 2 : 5) * 64 * sizeof(double); }
__host__ __device__ constexpr size_t gidx_bytes(int nb) { return (size_t)64 * 2 * nb * sizeof(unsigned short); }
__host__ __device__ constexpr size_t uvals_bytes(int nb) { return (size_t)(nb == 2 ? 4 : nb == 4 ? 6 : 9) * 64; }

// One keypoint of the fast pass by one wave, distinct-point form.  Per pattern: R rounds of the omni model (lane l, round t: distinct point t * 64 + l), the
// weighted wave sum for the mean, rounding + guard per DISTINCT point in the fixed-point form of fast_keypoint, ONE patch byte per distinct point written to
// vals[], then every lane gathers the 2 NB bytes of its own pairs.  LDS operations of one wave execute in order, so the gather sees the bytes written just
// before it by the other lanes (and the next pattern's bytes land behind this pattern's gather); nothing here needs a barrier.
// The whole keypoint is ONE basic block: the measured bound of this kernel is not instruction issue but the dependent chains of a pattern (the wave sum's 22
// cross-lane steps, then rounding -> patch byte -> vals -> gather -> ballot: four LDS round trips) with only four waves per SIMD to cover them — round 4's form
// (and the first distinct-point build: 520 us either way) left the kernel after every pattern (`bad`) and branched on "a sample outside the patch", so nothing of
// pattern p + 1 could be issued under pattern p's chains.  Now every verdict is collected and taken ONCE at the end; a keypoint with a sample outside the staged
// 43 x 43 patch (the reads are clamped into it, the bits discarded) goes to the exact pass like one in the guard band — the sampler's general path is gone from
// this kernel.
template <int MODE, int NB, class Ahead>
__device__ __forceinline__ bool fast_keypoint_u(const ExtractBuffers& b, const FastCam& C, const double* tabLds, const uint8_t* patch,
                                                double ukx, double uky, const double (&axc)[3], const double (&ays)[3], const UTabs& T,
                                                unsigned long long (&bitsMain)[NB], unsigned long long (&agree)[NB], Ahead&& ahead) {
	constexpr int NP = 128 * NB, R = UPat<NB>::R, RW = UPat<NB>::RW;
	const double kFix = 1572864.5;                                       // 1.5 * 2^20 + 0.5 (fast_keypoint)
	const unsigned kHiPatch = 0x4137F000u + 4096u - (unsigned)kPatchR;   // high word of y minus this = offset + kPatchR: the patch row / column
	const unsigned guardUnits = (unsigned)__builtin_ceil(b.guardEps * 4294967296.0);
	const double gAdd = (double)guardUnits * (1.0 / 4294967296.0);       // the guard's half-width folded into the addend (fast_keypoint)
#pragma unroll
	for (int j = 0; j < NB; ++j) { bitsMain[j] = 0ull; agree[j] = ~0ull; }
	constexpr int npat = MODE == 2 ? 3 : 1;
	typedef double f64x2 __attribute__((ext_vector_type(2)));
	typedef const __attribute__((address_space(3))) f64x2 lds_f64x2;
	typedef const __attribute__((address_space(3))) double lds_f64;
	typedef const __attribute__((address_space(3))) uint32_t lds_u32;
	typedef __attribute__((address_space(3))) uint8_t lds_u8;
	typedef const __attribute__((address_space(3))) uint8_t lds_cu8;
	// opaque (as LDS byte offsets: see fast_keypoint): the tables are read again for every keypoint, nothing derived from them is held across the walk
	unsigned pdo = (unsigned)(uintptr_t)(lds_f64x2*)T.upat, pwo = (unsigned)(uintptr_t)(lds_f64*)T.uw, pgo = (unsigned)(uintptr_t)(lds_u32*)T.gidx;
	unsigned pvo = (unsigned)(uintptr_t)(lds_u8*)T.vals, plo = (unsigned)(uintptr_t)(lds_cu8*)patch;
	int lane = threadIdx.x & 63;
	asm volatile("" : "+v"(pdo), "+v"(pwo), "+v"(pgo), "+s"(pvo), "+s"(plo), "+v"(lane));
	lds_f64x2* const pd = (lds_f64x2*)(uintptr_t)pdo;
	lds_f64* const pw = (lds_f64*)(uintptr_t)pwo;
	lds_u32* const pg = (lds_u32*)(uintptr_t)pgo;
	lds_u8* const pv = (lds_u8*)(uintptr_t)pvo;
	lds_cu8* const pvr = (lds_cu8*)(uintptr_t)pvo;
	lds_cu8* const pl = (lds_cu8*)(uintptr_t)plo;
	unsigned top = 0, minlo = 0xFFFFFFFFu, maxrc = 0;   // largest table row, smallest guard word, largest patch row / column over ALL patterns of the keypoint
	bool bad = false;
#pragma unroll
	for (int pat = 0; pat < npat; ++pat) {
		const double ax = axc[pat], ay = ays[pat];
		double u[R], v[R];
		double sumx = 0.0, sumy = 0.0;
#pragma unroll
		for (int t = 0; t < R; ++t) {
			const f64x2 p = pd[64 * t];
			if (pat == 0 && t == 0) ahead();
			const double xr = __builtin_fma(p.x, ax, __builtin_fma(-p.y, ay, ukx));
			const double yr = __builtin_fma(p.x, ay, __builtin_fma(p.y, ax, uky));
			if (MCS_FAST_ABLATE & 2) { u[t] = xr; v[t] = yr; } else fast_w2i<false>(C, tabLds, xr, yr, u[t], v[t], top);
			if (t < RW) { const double w = pw[64 * t]; sumx = __builtin_fma(w, u[t], sumx); sumy = __builtin_fma(w, v[t], sumy); }
			else { sumx += u[t]; sumy += v[t]; }
			constexpr int kFence = R > 6 ? 2 : MCS_FAST_FENCE;   // at most kFence point evaluations in flight (registers; nine rounds hold 36 for u, v alone)
			if ((t & (kFence - 1)) == kFence - 1) __builtin_amdgcn_sched_barrier(0);
		}
		double totx, toty;
		wave_sum2_f64(sumx, sumy, totx, toty);
		const double meanX = totx * (1.0 / (double)NP), meanY = toty * (1.0 / (double)NP);
		bad |= !(fabs(meanX) < 16384.0) || !(fabs(meanY) < 16384.0);
		const double cmx = (kFix - meanX) + gAdd, cmy = (kFix - meanY) + gAdd;
		if (R > 6) __builtin_amdgcn_sched_barrier(0);
		int val[R];
#pragma unroll
		for (int t = 0; t < R; ++t) {
			const double yx = u[t] + cmx, yy = v[t] + cmy;
			const unsigned pc = (unsigned)__double2hiint(yx) - kHiPatch, pr = (unsigned)__double2hiint(yy) - kHiPatch;   // unsigned: left of / above the patch, NaN: huge
			if (!(MCS_FAST_ABLATE & 4)) minlo = min(minlo, min((unsigned)__double2loint(yx), (unsigned)__double2loint(yy)));
			maxrc = max(maxrc, max(pr, pc));
#ifndef MCS_PATCH_OFF_MAD
#define MCS_PATCH_OFF_MAD 0   // 1 (round 5, A/B): pr * kFPitch + pc as written — pr is an arbitrary 32-bit value, so the compiler takes v_mad_u64_u32 for it: full rate (tools/valu_rate2), but its 64-bit results cost register pairs, and the block sat at the edge of the register file (1 spill; with the packed rows 28)
#endif
			// row * 48 + column by two shift-adds on 32-bit registers (a wrapped product of a row far outside the patch is clamped like any other: any byte will do, the keypoint leaves below)
			static_assert(kFPitch == 48 || MCS_PATCH_OFF_MAD, "the shift-add form is for a pitch of 48");
			unsigned off16 = (pr << 4) + pc;
			if (!MCS_PATCH_OFF_MAD) asm volatile("" : "+v"(off16));   // (opaque: the compiler otherwise folds the two shift-adds back into the multiply-add)
			const unsigned offRaw = MCS_PATCH_OFF_MAD ? pr * (unsigned)kFPitch + pc : (pr << 5) + off16;
			const unsigned off = min(offRaw, (unsigned)(kFPatchBytes - 1));   // (a sample outside the patch: any byte of it, the keypoint leaves below)
			val[t] = (MCS_FAST_ABLATE & 1) ? (int)off : (int)pl[off];
		}
#pragma unroll
		for (int t = 0; t < R; ++t) pv[64 * t + lane] = (uint8_t)val[t];
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (compiler ordering only: the bytes above are written before the reads below are issued)
#pragma unroll
		for (int j = 0; j < NB; ++j) {   // the lane's own pairs: the two bytes of each by the points' indices
			const uint32_t g = pg[j];
			const int t0 = pvr[g & 0xFFFFu], t1 = pvr[g >> 16];
			const unsigned long long bits = __ballot(t0 < t1);
			if (pat == 0) bitsMain[j] = bits;
			else agree[j] &= ~(bits ^ bitsMain[j]);
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		// 64-byte descriptors (nine rounds): the one-block form spills (36 registers for u, v alone), and a spill is a vector memory access that waits for the
		// patch prefetch in flight — so there the kernel is left after every pattern as in round 4, which splits the block and bounds what is live
		if (R > 6) { if (__any(bad | (top >= (unsigned)kGRows) | (minlo < 2u * guardUnits) | (maxrc >= (unsigned)kPatchRows))) return false; }
	}
	bad |= top >= (unsigned)kGRows;                              // a point below / above the table, NaN, Inf
	if (!(MCS_FAST_ABLATE & 4)) bad |= minlo < 2u * guardUnits;  // a coordinate in the guard band of a rounding tie
	bad |= maxrc >= (unsigned)kPatchRows;                        // a sample outside the staged patch
	return !__any(bad);
}

#ifndef MCS_FAST_DMA
#define MCS_FAST_DMA 1   // 0: the round-3 walk (slot record by loads, patch through registers), for A/B
#endif
constexpr int kFastPatchBufs = MCS_FAST_DMA ? 2 : 1;
constexpr int kMailDwords = 32, kMailBytes = MCS_FAST_DMA ? 2 * kMailDwords * 4 : 0;   // per wave
constexpr int kMailCam = 19, kMailUsed = 26;   // dwords 0..2 lvl, rc, poff; 3..18 the eight doubles of KpAuxSoA::d8; 19 tabIdx; 20..25 cam.c, cam.d, cam.e
#if MCS_FAST_UNIQ
__host__ __device__ constexpr size_t fast_wave_lds(int nb) { return (size_t)kFastPatchBufs * kFPatchBytes + kMailBytes + uvals_bytes(nb); }   // patch buffer(s), mailboxes, the sampled bytes
__host__ __device__ constexpr size_t fast_pat_bytes(int nb) { return upat_bytes(nb) + uw_bytes(nb) + gidx_bytes(nb); }   // distinct points, weights, gather indices
#else
__host__ __device__ constexpr size_t fast_wave_lds(int) { return (size_t)kFastPatchBufs * kFPatchBytes + kMailBytes; }
__host__ __device__ constexpr size_t fast_pat_bytes(int nb) { return (size_t)2 * nb * 64 * sizeof(double2); }   // the pattern points as doubles, [point][lane]
#endif

// The requests are written as inline assembly, not __builtin_amdgcn_global_load_lds: knowing an LDS-DMA write is in flight, the compiler puts an
// s_waitcnt vmcnt(0) in front of the first LDS read it cannot prove disjoint from the destination — here every table, pattern and patch read of the keypoint
// being described, i.e. it would wait for the request right after issuing it.  The ordering that is needed is the explicit wait at the top of a trip.  (Requests the
// compiler does not know about can only make ITS vmcnt waits wait longer, never shorter: loads complete in order and it counts too few outstanding.)
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ void dma16(const void* g, unsigned ldsBase) {   // lane l: 16 bytes at g -> LDS byte ldsBase + 16 l
	asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(ldsBase) : "memory", "m0");
}
__device__ __forceinline__ void dma4(const void* g, unsigned ldsBase) {    // lane l: 4 bytes at g -> LDS byte ldsBase + 4 l
	asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(ldsBase) : "memory", "m0");
}
// 16 bytes per lane: kFPitch / 16 lanes fetch a row.  The bytes past the 43rd column are never sampled; they stay inside the level's pitch or are the first
// bytes of the next row, which exists: a keypoint sits >= 25 px inside its level, the patch's last row is at least 4 rows above the level's last.
__device__ __forceinline__ void patch_request(const uint8_t* origin, int bstride, uint8_t* patch) {
	int lane = threadIdx.x & 63;
	asm volatile("" : "+v"(lane));
	constexpr int perRow = kFPitch / 16, total = kPatchRows * perRow;
#pragma unroll
	for (int t = 0; t * 64 < total; ++t) {
		const int i = lane + 64 * t;
		const int r = i / perRow, k = i - r * perRow;
		if (64 * (t + 1) <= total || i < total) dma16(origin + (size_t)r * bstride + 16 * k, __builtin_amdgcn_readfirstlane(lds_addr(patch) + 1024 * t));
	}
}
// the record of slot gw (a keypoint of image img) -> mailbox
__device__ __forceinline__ void record_request(const KpAuxSoA& A, const OcamDev* cams, int gw, int img, uint32_t* mail) {
	int lane = threadIdx.x & 63;
	asm volatile("" : "+v"(lane));
	const size_t S = (size_t)A.slots;
	const uint32_t* src;
	if (lane < 3) src = reinterpret_cast<const uint32_t*>(A.lvl) + (size_t)lane * S + gw;            // lvl, rc, poff: three arrays of S words one after the other
	else if (lane < kMailCam) src = reinterpret_cast<const uint32_t*>(A.d8 + (size_t)((lane - 3) >> 1) * S + gw) + ((lane - 3) & 1);
	else if (lane == kMailCam) src = reinterpret_cast<const uint32_t*>(&cams[img].tabIdx);
	else src = reinterpret_cast<const uint32_t*>(&cams[img].c) + (lane - kMailCam - 1);              // c, d, e: the first three doubles of OcamDev
	if (lane < kMailUsed) dma4(src, __builtin_amdgcn_readfirstlane(lds_addr(mail)));
}
static_assert(offsetof(OcamDev, c) == 0 && offsetof(OcamDev, d) == 8 && offsetof(OcamDev, e) == 16, "record_request reads c, d, e as six consecutive words");

struct FastKp {   // a mailbox read back: wave-uniform, in SGPRs
	int lvl, rc; unsigned poff; int tabIdx;
	double d8[8]; FastCam C;
	__device__ __forceinline__ bool usable() const { return lvl >= 0 && !(lvl & kAuxExact); }   // a keypoint, and not already on the exact pass's pre-list
};
__device__ __forceinline__ int rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ double rfl2(uint32_t lo, uint32_t hi) { return __hiloint2double(rfl(hi), rfl(lo)); }
__device__ __forceinline__ FastKp mail_read(const uint32_t* mail) {
	uint4 q[7];
#pragma unroll
	for (int i = 0; i < 7; ++i) q[i] = reinterpret_cast<const uint4*>(mail)[i];   // every lane reads the same words (LDS broadcast)
	const uint32_t* w = reinterpret_cast<const uint32_t*>(q);
	FastKp r;
	r.lvl = rfl(w[0]); r.rc = rfl(w[1]); r.poff = (unsigned)rfl(w[2]); r.tabIdx = rfl(w[kMailCam]);
#pragma unroll
	for (int i = 0; i < 8; ++i) r.d8[i] = rfl2(w[3 + 2 * i], w[4 + 2 * i]);
	r.C.c = rfl2(w[20], w[21]); r.C.d = rfl2(w[22], w[23]); r.C.e = rfl2(w[24], w[25]);
	return r;
}

template <int MODE, int NB>
__attribute__((amdgpu_waves_per_eu(MCS_FAST_WAVES_PER_EU, MCS_FAST_WAVES_PER_EU)))
__global__ __launch_bounds__(64 * kFastWaves) void k_describe_fast(ExtractBuffers b, int wavesPerImage, int nslots, int groupsPerBlock) {
	extern __shared__ __attribute__((aligned(16))) double lds[];   // the camera's G table (shared, at offset 0: its reads then need no address arithmetic), then per wave: patch buffer(s), mailboxes
	static_assert(MODE == 1 || MODE == 2, "the fast pass is for the distorted patterns");
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	constexpr size_t kFastWaveLds = fast_wave_lds(NB);
	double2* const patLds = reinterpret_cast<double2*>(reinterpret_cast<uint8_t*>(lds) + kGDevDoubles * sizeof(double));
	uint8_t* const waveLds = reinterpret_cast<uint8_t*>(patLds) + fast_pat_bytes(NB) + (size_t)wave * kFastWaveLds;
	uint32_t* const mailBase = reinterpret_cast<uint32_t*>(waveLds + kFastPatchBufs * kFPatchBytes);
#if MCS_FAST_UNIQ
	// the pattern's distinct points, their weights and the lanes' gather indices (host-built, upload_describe_tables; visible after the barrier behind the first table load)
	constexpr int kSlot = NB == 2 ? 0 : NB == 4 ? 1 : 2;
	double* const uwLds = reinterpret_cast<double*>(reinterpret_cast<uint8_t*>(patLds) + upat_bytes(NB));
	uint32_t* const gidxLds = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(uwLds) + uw_bytes(NB));
	for (int i = threadIdx.x; i < UPat<NB>::R * 64; i += 64 * kFastWaves) patLds[i] = g_upat[kSlot][i];
	for (int i = threadIdx.x; i < UPat<NB>::RW * 64; i += 64 * kFastWaves) uwLds[i] = g_uw[kSlot][i];
	for (int i = threadIdx.x; i < 64 * NB; i += 64 * kFastWaves) gidxLds[i] = reinterpret_cast<const uint32_t*>(g_gidx[kSlot])[i];
#else
	// the pattern points as doubles (visible after the barrier behind the first table load)
	for (int i = threadIdx.x; i < 2 * NB * 64; i += 64 * kFastWaves) {
		const int t = i >> 6, ln = i & 63;
		const signed char* pp = c_pattern + ((t >> 1) * 64 + ln) * 4 + 2 * (t & 1);
		patLds[i] = double2{(double)pp[0], (double)pp[1]};
	}
#endif
	double* const tabLds = lds;
	const PyrDesc& d = *b.desc;
	KpAuxSoA A; A.carve(b.aux, nslots);
	const int ngroups = nslots / kFastWaves;
	// Trip k of the walk: the resident workgroups together cover the groups [k * gridDim.x, (k + 1) * gridDim.x) — 4096 consecutive keypoints, a handful of
	// images whose blurred levels then sit in L2 / Infinity Cache for everybody (a contiguous range of groups PER workgroup had every workgroup in a different
	// image: 711 instead of 365 MB fetched from HBM per launch).  Within a trip an XCD (blockIdx.x % 8: its own L2) takes a contiguous eighth.
	const int nb = (int)gridDim.x, perXcd = (nb + kNumXCD - 1) / kNumXCD;
	const int logical = nb % kNumXCD == 0 ? ((int)blockIdx.x % kNumXCD) * perXcd + (int)blockIdx.x / kNumXCD : (int)blockIdx.x;   // a bijection on [0, nb) either way
	const size_t S = (size_t)nslots;
	const size_t pyrBytes = d.pyrBytes;
	int curTab = -1;
#if MCS_FAST_DMA
	auto slot_of = [&](int kk) { const int g = kk * nb + logical; return kk < groupsPerBlock && g < ngroups ? g * kFastWaves + wave : -1; };
	auto ask_record = [&](int kk) { const int gw = slot_of(kk); if (gw >= 0) record_request(A, b.cams, gw, gw / wavesPerImage, mailBase + (kk & 1) * kMailDwords); };
	auto ask_patch = [&](const FastKp& r, int kk) {
		const int gw = slot_of(kk);
		if (gw < 0 || !r.usable()) return;
		patch_request(b.blur + (size_t)(gw / wavesPerImage) * pyrBytes + r.poff, (int)((unsigned)r.lvl >> 16), waveLds + (kk & 1) * kFPatchBytes);
	};
	// prologue: records 0 and 1, patch 0
	ask_record(0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	FastKp nxt = mail_read(mailBase);
	if (slot_of(0) < 0) nxt.lvl = -1;
	ask_patch(nxt, 0);
	ask_record(1);
	// The walk's ONE wait for vector memory sits at the END of a trip, in front of the trip's output stores (round 6): what it waits for — the next keypoint's patch
	// and the record after it — was requested a whole keypoint earlier and has long landed, and the stores then have the next trip to complete.  (At the top of the
	// trip, where it stood, it also waited for the stores just issued: on this chip stores count in vmcnt like loads, a store's round trip per keypoint.)
#ifndef MCS_FAST_WAIT_TOP
#define MCS_FAST_WAIT_TOP 1   // 1 (default): the wait at the top of the trip; 0: at its end, in front of the output stores — measured 0.506 against 0.498 ms, not kept
#endif
	FastKp cur = nxt;
	if (!MCS_FAST_WAIT_TOP) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		nxt = mail_read(mailBase + kMailDwords);
		if (slot_of(1) < 0) nxt.lvl = -1;
	}
	auto advance = [&](int k) {   // end of trip k: patch k + 1 and record k + 2 have landed
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		cur = nxt;
		nxt = mail_read(mailBase + (k & 1) * kMailDwords);   // record k + 2 went into the mailbox record k was read from
		if (slot_of(k + 2) < 0) nxt.lvl = -1;
	};
#endif
#pragma unroll 1
	for (int k = 0; k < groupsPerBlock; ++k) {
		const int g = k * nb + logical;
		if (g >= ngroups) break;   // uniform over the workgroup
		const int base = g * kFastWaves;
		const int bimg = base / wavesPerImage;   // every slot of a group is a keypoint of the same image
		const int gwu = base + wave;
		int lane = threadIdx.x & 63;
		asm volatile("" : "+v"(lane));   // opaque per trip: nothing derived from the lane id is worth holding in registers across the walk
#if MCS_FAST_DMA
#if MCS_FAST_WAIT_TOP
		advance(k - 1);   // A/B: the wait at the top of the trip (round 5)
#endif
		const FastKp me = cur;   // (its patch landed before the previous trip's stores went out: advance())
		// The requests go out behind this keypoint's first LDS read (inside fast_keypoint), not here: the compiler keeps an s_waitcnt vmcnt(0) in front of that
		// read (pending flat accesses of the sampler's general path, as its bookkeeping sees the loop) — issued before it, they would be waited for at once.
		auto ahead = [&]() {
			ask_patch(nxt, k + 1);   // the other patch buffer: its keypoint (k - 1) is finished
			ask_record(k + 2);       // the mailbox record k was read from a trip ago
		};
		const int tabIdx = me.tabIdx;
		uint8_t* const patchLds = waveLds + (k & 1) * kFPatchBytes;
#else
		const OcamDev& cam = b.cams[bimg];
		const int tabIdx = cam.tabIdx;
		uint8_t* const patchLds = waveLds;
#endif
		if (tabIdx != curTab) {   // uniform over the workgroup: every wave walks the same groups
			if (curTab >= 0) __syncthreads();   // nobody reads the old table any more
			if (!(MCS_FAST_ABLATE & 8)) {
				const double2* gt = reinterpret_cast<const double2*>(b.gTab + (size_t)tabIdx * kGDevDoubles);
				constexpr int n2 = kGDevDoubles / 2, trips = (n2 + 64 * kFastWaves - 1) / (64 * kFastWaves);
				double2 tv[trips];
#pragma unroll
				for (int t = 0; t < trips; ++t) { const int i = t * 64 * kFastWaves + (int)threadIdx.x; tv[t] = i < n2 ? gt[i] : double2{0.0, 0.0}; }
#pragma unroll
				for (int t = 0; t < trips; ++t) { const int i = t * 64 * kFastWaves + (int)threadIdx.x; if (i < n2) reinterpret_cast<double2*>(tabLds)[i] = tv[t]; }
			}
			__syncthreads();
			curTab = tabIdx;
		}
#if MCS_FAST_DMA
		if (!me.usable()) { ahead(); if (!MCS_FAST_WAIT_TOP) advance(k); continue; }   // nothing here, or already on the exact pass's pre-list
		const int level = me.lvl & 0xFF;
		const int row = me.rc & 0xFFFF, col = (int)((unsigned)me.rc >> 16);
		const FastCam C = me.C;
		const double ukx = me.d8[0], uky = me.d8[1];
		double axc[3], ays[3];
#pragma unroll
		for (int q = 0; q < 3; ++q) { axc[q] = me.d8[2 + 2 * q]; ays[q] = me.d8[3 + 2 * q]; }
#else
		// this wave's keypoint as the orientation kernels left it (wave-uniform: scalar loads)
		const int gws = __builtin_amdgcn_readfirstlane(gwu);
		const int lvlRaw = A.lvl[gws];
		if (lvlRaw < 0 || (lvlRaw & kAuxExact)) continue;   // nothing here, or already on the exact pass's pre-list
		const int level = lvlRaw & 0xFF;
		const int rc = A.rc[gws];
		const int row = rc & 0xFFFF, col = (int)((unsigned)rc >> 16);
		{
			const LevelInfo& L = d.lv[level];
			int ln = threadIdx.x & 63;
			const uint8_t* bp = b.blur + (size_t)bimg * pyrBytes + L.off + (size_t)(row - kPatchR) * L.stride + (col - kPatchR);
			uint32_t pv[kPatchTrips];
#pragma unroll
			for (int t = 0; t < kPatchTrips; ++t) { const int i = min(ln + 64 * t, kPatchRows * kPatchDw - 1); const int r = i / kPatchDw; __builtin_memcpy(&pv[t], bp + (size_t)r * L.stride + 4 * (i - r * kPatchDw), 4); }
#pragma unroll
			for (int t = 0; t < kPatchTrips; ++t) { const int i = min(ln + 64 * t, kPatchRows * kPatchDw - 1); const int r = i / kPatchDw; *reinterpret_cast<uint32_t*>(&patchLds[r * kFPitch + 4 * (i - r * kPatchDw)]) = pv[t]; }
		}
		FastCam C;
		C.c = cam.c; C.d = cam.d; C.e = cam.e;
		const double* D8 = A.d8 + gws;
		const double ukx = D8[0], uky = D8[S];
		double axc[3], ays[3];
#pragma unroll
		for (int q = 0; q < 3; ++q) { axc[q] = D8[(2 + 2 * q) * S]; ays[q] = D8[(3 + 2 * q) * S]; }
#endif
		LazySampler sm;
		sm.b = &b; sm.img = bimg; sm.level = level; sm.patch = patchLds;
		(void)sm; (void)row; (void)col;
		unsigned long long bitsMain[NB], agree[NB];
#if MCS_FAST_UNIQ
		UTabs T;
		T.upat = patLds + lane; T.uw = uwLds + lane; T.gidx = gidxLds + lane * NB; T.vals = waveLds + kFastPatchBufs * kFPatchBytes + kMailBytes;
#if MCS_FAST_DMA
		const bool ok = fast_keypoint_u<MODE, NB>(b, C, tabLds, patchLds, ukx, uky, axc, ays, T, bitsMain, agree, ahead);
#else
		const bool ok = fast_keypoint_u<MODE, NB>(b, C, tabLds, patchLds, ukx, uky, axc, ays, T, bitsMain, agree, [] {});
#endif
#elif MCS_FAST_DMA
		const bool ok = fast_keypoint<MODE, NB>(b, C, tabLds, sm, row, col, ukx, uky, axc, ays, patLds + lane, bitsMain, agree, ahead);
#else
		const bool ok = fast_keypoint<MODE, NB>(b, C, tabLds, sm, row, col, ukx, uky, axc, ays, patLds + lane, bitsMain, agree, [] {});
#endif
#if MCS_FAST_DMA
		if (!MCS_FAST_WAIT_TOP) advance(k);
#endif
		if (lane == 0) {
			if (!ok) { const int at = atomicAdd(b.fbCount, 1); b.fbList[at] = (uint32_t)gwu; }
			else {
				const int out = gwu - bimg * wavesPerImage;
				uint8_t* dout = b.out_desc + ((size_t)bimg * b.outImgPitch + out) * b.outRowStride;
				uint8_t* mout = b.out_mask + ((size_t)bimg * b.outImgPitch + out) * b.outRowStride;
#pragma unroll
				for (int j = 0; j < NB; ++j) {
					*reinterpret_cast<unsigned long long*>(dout + 8 * j) = bitsMain[j];
					*reinterpret_cast<unsigned long long*>(mout + 8 * j) = MODE == 2 ? agree[j] : 0ull;
				}
			}
		}
	}
	(void)S; (void)d;
}

// self-test of the fast arithmetic: n pseudo-random pattern points around random keypoints of camera `cam` through fast_w2i and through the exact
// world2img; maxDiff[0] = the largest |u_fast - u_exact| or |v_fast - v_exact| seen (as the bits of a non-negative double, atomicMax).  Points the table
// does not cover (the kernel sends their keypoints to the exact pass) are skipped.
__global__ void k_selftest_fast_model(const OcamDev* camp, const double* tab, unsigned long long seed, int n, int width, int height, unsigned long long* maxDiff) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const OcamDev& cam = *camp;
	unsigned long long st = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
	auto next = [&]() { st ^= st >> 12; st ^= st << 25; st ^= st >> 27; return st * 0x2545F4914F6CDD1Dull; };
	auto unit = [&]() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); };
	// a keypoint somewhere in the image, undistorted like the kernel does, plus a pattern offset of up to +-32 px
	const double px = unit() * width, py = unit() * height;
	double rx, ry, rz;
	img2world(cam, px, py, rx, ry, rz);
	const double p0 = cam.p[0];
	const double xr = -rx / rz * p0 + (unit() * 64.0 - 32.0), yr = -ry / rz * p0 + (unit() * 64.0 - 32.0);
	double ue, ve;
	world2img(cam, xr, yr, -p0, ue, ve);
	FastCam C;
	C.c = cam.c; C.d = cam.d; C.e = cam.e;
	double uf, vf;
	unsigned top = 0;
	fast_w2i<true>(C, tab, xr, yr, uf, vf, top);
	if (top >= (unsigned)kGRows) return;
	double diff = fmax(fabs(uf + cam.u0 - ue), fabs(vf + cam.v0 - ve));   // one extra rounding here (the kernel never adds the principal point)
	if (!(diff == diff)) diff = 1e300;   // NaN on either side counts as a failure
	atomicMax(maxDiff, (unsigned long long)__double_as_longlong(diff));
}

void launch_selftest_fast_model(const OcamDev* cam, const double* tab, unsigned long long seed, int n, int width, int height, unsigned long long* maxDiff, hipStream_t s) {
	hipLaunchKernelGGL(k_selftest_fast_model, dim3((n + 255) / 256), dim3(256), 0, s, cam, tab, seed, n, width, height, maxDiff);
}

// ORB descriptors of a batch prepared by k_orient_b<0>: a wave per output row stages the blurred patch and forms NB x 64 bits (computeOrbDescriptor,
// src/mdBRIEFextractorOct.cpp:1203-1242 via describe_wave<0>'s statements: double rotation, cvRound, tie watch)
template <int NB>
__global__ __launch_bounds__(256) void k_describe_orb(ExtractBuffers b, int wavesPerImage, int nslots) {
	extern __shared__ double orb_lds[];
	const PyrDesc& d = *b.desc;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave);
	if (gw >= nslots) return;
	KpAuxSoA A; A.carve(b.aux, nslots);
	const int lw = __builtin_amdgcn_readfirstlane(A.lvl[gw]);
	if (lw < 0) return;
	const int level = lw & 0xFF;
	const int rc = __builtin_amdgcn_readfirstlane(A.rc[gw]);
	const int row = rc & 0xFFFF, col = (int)((unsigned)rc >> 16);
	const int img = gw / wavesPerImage, out = gw - img * wavesPerImage;
	const double ax = uniform_f64(A.d8 + (size_t)2 * nslots + gw), ay = uniform_f64(A.d8 + (size_t)3 * nslots + gw);
	const LevelInfo& L = d.lv[level];
	uint8_t* patch = reinterpret_cast<uint8_t*>(orb_lds) + (size_t)wave * kPatchBytes;
	Sampler sm = {};
	int rstride;
	sm.raw = level_ptr(b, d, img, level, &rstride); sm.rstride = rstride; sm.w = L.w; sm.h = L.h;
	sm.blur = b.blur + (size_t)img * d.pyrBytes + L.off; sm.bstride = L.stride;
	uint32_t pv[kPatchTrips];
	patch_load(sm.blur, sm.bstride, row, col, pv);
	patch_store(patch, pv);
	sm.patch = patch; sm.prow = row - kPatchR; sm.pcol = col - kPatchR;
	uint8_t* dout = b.out_desc + ((size_t)img * b.outImgPitch + out) * b.outRowStride;
	uint8_t* mout = b.out_mask + ((size_t)img * b.outImgPitch + out) * b.outRowStride;
	double tie = 0.0;
#pragma unroll
	for (int j = 0; j < NB; ++j) {
		const int k = j * 64 + lane;
		const double x0 = c_pattern[4 * k], y0 = c_pattern[4 * k + 1], x1 = c_pattern[4 * k + 2], y1 = c_pattern[4 * k + 3];
		const double fx0 = x0 * ax - y0 * ay, fy0 = x0 * ay + y0 * ax, fx1 = x1 * ax - y1 * ay, fy1 = x1 * ay + y1 * ax;
		const int ix0 = __double2int_rn(fx0), iy0 = __double2int_rn(fy0);
		const int ix1 = __double2int_rn(fx1), iy1 = __double2int_rn(fy1);
		tie = fmax(fmax(tie, fmax(tie_frac(fx0), tie_frac(fy0))), fmax(tie_frac(fx1), tie_frac(fy1)));
		int t0, t1;
		sm.pair(row, col, iy0, ix0, iy1, ix1, t0, t1);
		const unsigned long long bits = __ballot(t0 < t1);
		if (lane == 0) {
			*reinterpret_cast<unsigned long long*>(dout + 8 * j) = bits;
			*reinterpret_cast<unsigned long long*>(mout + 8 * j) = 0ull;   // descriptorMasks = zeros (:1216)
		}
	}
	tie_commit(b, gw, tie);
}

template <int MODE, int NB>
static void launch_fast_passes(const ExtractBuffers& b, int nimg, int wavesPerImage, size_t listLds, hipStream_t s) {
	const int nslots = nimg * wavesPerImage, ngroups = nslots / kFastWaves, lblocks = std::min(nslots, 2048);
	// one workgroup of kFastWaves = 16 waves per CU (registers: 4 waves per SIMD; LDS: table + pattern + 16 x (two patch buffers + mailboxes) = 139 KB): one resident
	// generation of workgroups walks the whole batch
	const int groupsPerBlock = std::max(1, (ngroups + kFastBlocks - 1) / kFastBlocks);
	int fblocks = (ngroups + groupsPerBlock - 1) / groupsPerBlock;
	if (fblocks >= kNumXCD) fblocks = (fblocks + kNumXCD - 1) / kNumXCD * kNumXCD;   // whole XCD rounds: the kernel's block -> group mapping is XCD-contiguous
	const size_t fLds = (size_t)kFastWaves * fast_wave_lds(NB) + kGDevDoubles * sizeof(double) + fast_pat_bytes(NB);
	// PRECONDITION: fbCount and preCount — neighbours — were cleared by k_octree's first workgroup, i.e. every launch_describe follows a launch_octree of the same
	// batch on the same stream, and the previous batch's side-stream pre-list kernel has been joined (the evDescJoin wait below); extract_impl in mcs_capi.hip is
	// the only caller and keeps that order
	hipLaunchKernelGGL((k_orient_b<MODE>), dim3((nslots + 255) / 256), dim3(256), 0, s, b, wavesPerImage, nslots);
	// the pre-list (about one keypoint in a hundred: the ones next to the optical axis) through the exact pass BESIDE the fast pass
	hipStream_t ps = b.sideStream ? b.sideStream : s;
	if (b.sideStream) { (void)hipEventRecord(b.evDescFork, s); (void)hipStreamWaitEvent(ps, b.evDescFork, 0); }
	hipLaunchKernelGGL((k_describe_list<MODE, NB>), dim3(lblocks), dim3(64), listLds, ps, b, wavesPerImage, b.preCount, b.preList);
	if (b.sideStream) (void)hipEventRecord(b.evDescJoin, ps);
	// more than the default 64 KB of dynamic LDS per workgroup: the attribute belongs to the function ON THE CURRENT DEVICE, and one process may drive several
	// devices from several threads (host/rig_host.cpp), so it is set before every launch (as launch_spec in mcs_greedy.hip does), not once per process
	if (fLds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_describe_fast<MODE, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fLds);
	if (b.evFastA) (void)hipEventRecord(b.evFastA, s);
	hipLaunchKernelGGL((k_describe_fast<MODE, NB>), dim3(fblocks), dim3(64 * kFastWaves), fLds, s, b, wavesPerImage, nslots, groupsPerBlock);
	if (b.evFastB) (void)hipEventRecord(b.evFastB, s);
	// the fallbacks — a few keypoints on the critical path — by a workgroup each (k_describe_list_split); MCS_LIST_SPLIT=0: one wave each, for A/B
	static const bool split = !(getenv("MCS_LIST_SPLIT") && atoi(getenv("MCS_LIST_SPLIT")) == 0);
	typedef SplitGeom<MODE, NB> SG;
	if (split) hipLaunchKernelGGL((k_describe_list_split<MODE, NB>), dim3(std::min(nslots, 512)), dim3(64 * SG::WAVES), SG::lds_bytes(), s, b, wavesPerImage, b.fbCount, b.fbList);
	else hipLaunchKernelGGL((k_describe_list<MODE, NB>), dim3(lblocks), dim3(64), listLds, s, b, wavesPerImage, b.fbCount, b.fbList);
	if (b.sideStream) (void)hipStreamWaitEvent(s, b.evDescJoin, 0);
}

template <int MODE>
static void launch_mode(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s) {
	// ORB: 4 keypoints (waves) per 256-thread block.  dBRIEF/mdBRIEF exact pass: one wave per block with a private 2*NB KiB LDS slice.
	const int wpb = MODE == 0 ? 4 : 1;
	const int wavesPerImage = (hd.kpCap + kSlotAlign - 1) / kSlotAlign * kSlotAlign;   // one slot per output row; a multiple of kSlotAlign in every mode
	const int blocks = nimg * wavesPerImage / wpb;
	const size_t ldsBytes = (size_t)wpb * (coord_bytes(MODE, hd.npoints) + kPatchBytes);   // coordinates + blurred patch per wave
	const int nb = hd.descSize / 8;
	if constexpr (MODE != 0) {
		if (b.describeMode == 0) {   // fast pass + exact pass over the two lists
			if (nb == 2) launch_fast_passes<MODE, 2>(b, nimg, wavesPerImage, ldsBytes, s);
			else if (nb == 4) launch_fast_passes<MODE, 4>(b, nimg, wavesPerImage, ldsBytes, s);
			else launch_fast_passes<MODE, 8>(b, nimg, wavesPerImage, ldsBytes, s);
			return;
		}
	}
	if constexpr (MODE == 0) {
		static const bool orbSplit = !(getenv("MCS_ORB_SPLIT") && atoi(getenv("MCS_ORB_SPLIT")) == 0);   // 0: the one-kernel form (A/B, tests)
		// (a small batch — one multi-frame per call — is launch-bound: the one-kernel form saves a launch there and its redundant arithmetic costs nothing on an idle chip)
		static const bool orbForce = getenv("MCS_ORB_SPLIT") != nullptr;   // "1": the split form for every batch size (tests)
		if (orbSplit && b.aux && (orbForce || nimg * wavesPerImage >= 8192)) {
			const int nslots = nimg * wavesPerImage;
			hipLaunchKernelGGL((k_orient_b<0>), dim3((nslots + 255) / 256), dim3(256), 0, s, b, wavesPerImage, nslots);
			if (b.evFastA) (void)hipEventRecord(b.evFastA, s);
			if (nb == 2) hipLaunchKernelGGL((k_describe_orb<2>), dim3(blocks), dim3(256), ldsBytes, s, b, wavesPerImage, nslots);
			else if (nb == 4) hipLaunchKernelGGL((k_describe_orb<4>), dim3(blocks), dim3(256), ldsBytes, s, b, wavesPerImage, nslots);
			else hipLaunchKernelGGL((k_describe_orb<8>), dim3(blocks), dim3(256), ldsBytes, s, b, wavesPerImage, nslots);
			if (b.evFastB) (void)hipEventRecord(b.evFastB, s);
			return;
		}
	}
	if (b.evFastA) (void)hipEventRecord(b.evFastA, s);
	if (nb == 2) hipLaunchKernelGGL((k_describe<MODE, 2>), dim3(blocks), dim3(64 * wpb), ldsBytes, s, b, wavesPerImage);
	else if (nb == 4) hipLaunchKernelGGL((k_describe<MODE, 4>), dim3(blocks), dim3(64 * wpb), ldsBytes, s, b, wavesPerImage);
	else if constexpr (MODE == 2) hipLaunchKernelGGL((k_describe_wide<MODE, 8>), dim3(blocks), dim3(64 * wpb), ldsBytes, s, b, wavesPerImage);
	else hipLaunchKernelGGL((k_describe<MODE, 8>), dim3(blocks), dim3(64 * wpb), ldsBytes, s, b, wavesPerImage);
	if (b.evFastB) (void)hipEventRecord(b.evFastB, s);
}

size_t describe_aux_bytes() { return KpAuxSoA::bytes_per_slot(); }

void launch_describe(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s) {
	if (hd.mode == 0) launch_mode<0>(b, hd, nimg, s);
	else if (hd.mode == 1) launch_mode<1>(b, hd, nimg, s);
	else launch_mode<2>(b, hd, nimg, s);
}

}  // namespace mcs

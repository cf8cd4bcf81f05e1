// mcs_tiefix.hip — the descriptor of ONE keypoint by the reference's exact arithmetic ON THE HOST, with the host's libm.
//
// Why it exists.  The exact descriptor passes (mcs_describe.hip: ORB rotation, rotateAndDistortPattern for dBRIEF / mdBRIEF) are the reference's statements one by
// one; +, -, *, /, sqrt are IEEE on both sides, but cos / sin / atan come from ocml on the device and from glibc in the reference, and two faithful libms may differ
// in the last place.  That can only change cvRound(v) when v lies within ~1e-13 px of k + 1/2.  The device therefore LISTS every keypoint whose exact arithmetic
// brought a cvRound argument within `tieBand` of a tie (ExtractBuffers.tieList; default band 1e-9 px for the distorted modes — ~300x the worst chain of two-ulp
// libm differences through atan, the 12-term backward polynomial and the mean —, 1e-12 px for the plain ORB rotation, whose coordinates are |x|,|y| <= 15 times a
// cosine / sine: 2 * 15 * 2 ulp = 1.3e-14 px), and the host recomputes exactly those descriptors here, calling the libm the reference itself would call.
// The blurred / unblurred level of the keypoint is fetched from the device for it.  How often: in the default mdBRIEF mode only the fast pass's fallbacks run the exact
// arithmetic, and those are by construction the keypoints with a coordinate within the guard band (6e-8 px) of a tie — 1e-9 / 6e-8 = 1.7 % of them land in this band: about
// one keypoint per 64-multi-frame batch (measured on the bench stream: one per step).
// Reference: src/mdBRIEFextractorOct.cpp:250-283 (rotateAndDistortPattern), :285-301 (rotatePattern), :303-554 (compute_ORB / _dBRIEF / _mdBRIEF),
// src/cam_model_omni.cpp:49-67, 146-161, include/cam_model_omni.h:127-145.  This is product code (the C ABI's own host side), not the test oracle: nothing here
// includes, links or calls oracle/.
#include "mcs_common.h"
#include "mcs_tiecap.h"

#include <cmath>
#include <cstring>

namespace mcs {


namespace {

inline int cvRoundH(double v) { return (int)lrint(v); }   // cv::cvRound: round half to even (default rounding mode)

// Sampler::at of mcs_describe.hip: inside the level the blurred pyramid; in the 25-px frame the unblurred level with reflect-101 indices; beyond it clamped
inline int sample(const HostLevel& L, int r, int c) {
	if (L.patch) {   // the device evaluated Sampler::at for the window (same three cases as below)
		const unsigned pr = (unsigned)(r - L.prow), pc = (unsigned)(c - L.pcol);
		if (pr < (unsigned)L.pdim && pc < (unsigned)L.pdim) return L.patch[(size_t)pr * L.pdim + pc];
		if (L.miss) *L.miss = true;
		return 0;
	}
	if ((unsigned)r < (unsigned)L.h && (unsigned)c < (unsigned)L.w) return L.blur[(size_t)r * L.w + c];
	r = r < -kEdge ? -kEdge : (r > L.h + kEdge - 1 ? L.h + kEdge - 1 : r);
	c = c < -kEdge ? -kEdge : (c > L.w + kEdge - 1 ? L.w + kEdge - 1 : c);
	r = r < 0 ? -r : (r >= L.h ? 2 * (L.h - 1) - r : r);
	c = c < 0 ? -c : (c >= L.w ? 2 * (L.w - 1) - c : c);
	return L.raw[(size_t)r * L.w + c];
}

inline double hornerH(const double* coeffs, double x) {   // include/misc.h:115-122 on the zero-padded coefficient array (0 * x + 0 = +0 up to the first real one)
	double res = 0.0;
	for (int i = MCS_MAX_POLY - 1; i >= 0; i--) res = res * x + coeffs[i];
	return res;
}

inline void world2imgH(const OcamDev& cam, double x, double y, double z, double& u, double& v) {   // src/cam_model_omni.cpp:146-161
	double norm = std::sqrt(x * x + y * y);
	if (norm == 0.0) norm = 1e-14;
	const double theta = std::atan(-z / norm);
	const double rho = hornerH(cam.invP, theta);
	const double uu = x / norm * rho;
	const double vv = y / norm * rho;
	u = uu * cam.c + vv * cam.d + cam.u0;
	v = uu * cam.e + vv + cam.v0;
}

inline void img2worldH(const OcamDev& cam, double u, double v, double& xo, double& yo, double& zo) {   // src/cam_model_omni.cpp:49-67
	const double u_t = u - cam.u0;
	const double v_t = v - cam.v0;
	const double x = (u_t - cam.d * v_t) / cam.invAffine;
	const double y = (-cam.e * u_t + cam.c * v_t) / cam.invAffine;
	const double X2 = x * x;
	const double Y2 = y * y;
	const double z = -hornerH(cam.p, std::sqrt(X2 + Y2));
	const double norm = std::sqrt(X2 + Y2 + z * z);
	xo = x / norm;
	yo = y / norm;
	zo = z / norm;
}

}  // namespace

// desc / mask: descSize bytes each.  (row, col): the keypoint in its level's ROI; levelScale = (float)mvScaleFactor[level]; angle in degrees (IC_Angle).
void describe_host(int mode, int descSize, const signed char* pattern, const OcamDev* cam, int undistort, int level, float levelScale, int row, int col,
                   float angle, const HostLevel& L, uint8_t* desc, uint8_t* mask) {
	const int npairs = 8 * descSize, NP = 2 * npairs;
	memset(desc, 0, descSize);
	memset(mask, 0, descSize);
	if (mode == 0) {   // compute_ORB :303-354 with rotatePattern :285-301
		const float DEG2RADf = (float)3.1415926535897932384626433832795 / 180.f;
		const double ang = (double)(angle * DEG2RADf);
		const double ax = std::cos(ang), ay = std::sin(ang);
		for (int k = 0; k < npairs; ++k) {
			const double x0 = pattern[4 * k], y0 = pattern[4 * k + 1], x1 = pattern[4 * k + 2], y1 = pattern[4 * k + 3];
			const int ix0 = cvRoundH(x0 * ax - y0 * ay), iy0 = cvRoundH(x0 * ay + y0 * ax);
			const int ix1 = cvRoundH(x1 * ax - y1 * ay), iy1 = cvRoundH(x1 * ay + y1 * ax);
			if (sample(L, row + iy0, col + ix0) < sample(L, row + iy1, col + ix1)) desc[k >> 3] |= (uint8_t)(1u << (k & 7));
		}
		return;   // descriptorMasks = zeros (:1216)
	}
	// E8: level coordinates -> image coordinates with the FLOAT scale (:1305, 1331); undistortPointsOcam(pt * scale, scaleF = p[0]) (:1306-1317)
	float pxf = (float)col, pyf = (float)row;
	if (level != 0) { pxf = pxf * levelScale; pyf = pyf * levelScale; }
	double ukx = 0.0, uky = 0.0;
	if (undistort) {
		double rx, ry, rz;
		img2worldH(*cam, (double)pxf, (double)pyf, rx, ry, rz);
		ukx = -rx / rz * cam->p[0];
		uky = -ry / rz * cam->p[0];
	}
	const double zc = -cam->p[0];   // distortPointsOcam: WorldToImg(x, y, -p1)
	double ang[3] = {0.0, 0.0, 0.0};
	if (mode == 1) {
		const float DEG2RADf = (float)3.1415926535897932384626433832795 / 180.f;
		ang[0] = (double)(angle * DEG2RADf);
	} else {
		const float RHOf = 180.0f / 3.1415926535897932384626f;
		const double RHOd = 180.0 / 3.1415926535897932384626433832795028841971693993;
		const double rot = 20.0 / RHOd;
		ang[0] = (double)(angle / RHOf);
		ang[1] = ang[0] + rot; ang[2] = ang[0] - rot;
	}
	const int npat = mode == 2 ? 3 : 1;
	std::vector<double> xs(NP), ys(NP);
	std::vector<uint8_t> mainBits(npairs), agree(npairs, 1);
	for (int pat = 0; pat < npat; ++pat) {
		const double ax = std::cos(ang[pat]), ay = std::sin(ang[pat]);
		double sumX = 0.0, sumY = 0.0;
		for (int p = 0; p < NP; ++p) {   // rotateAndDistortPattern :250-283
			const double ptx = (double)pattern[2 * p], pty = (double)pattern[2 * p + 1];
			const double xr = ptx * ax - pty * ay + ukx;
			const double yr = ptx * ay + pty * ax + uky;
			world2imgH(*cam, xr, yr, zc, xs[p], ys[p]);
			sumX += xs[p];
			sumY += ys[p];
		}
		const double meanX = sumX / (double)NP, meanY = sumY / (double)NP;
		for (int k = 0; k < npairs; ++k) {
			const int ix0 = cvRoundH(xs[2 * k] - meanX), iy0 = cvRoundH(ys[2 * k] - meanY);
			const int ix1 = cvRoundH(xs[2 * k + 1] - meanX), iy1 = cvRoundH(ys[2 * k + 1] - meanY);
			const uint8_t bit = sample(L, row + iy0, col + ix0) < sample(L, row + iy1, col + ix1) ? 1 : 0;
			if (pat == 0) mainBits[k] = bit;
			else if (bit != mainBits[k]) agree[k] = 0;
		}
	}
	for (int k = 0; k < npairs; ++k) {
		if (mainBits[k]) desc[k >> 3] |= (uint8_t)(1u << (k & 7));
		if (mode == 2 && agree[k]) mask[k >> 3] |= (uint8_t)(1u << (k & 7));   // mask bit = both +-20 degree tests agree with the main test (:468-475)
	}
}

// ---- the pipelined form (round 6): device-kind batches are consumed on-stream, one step late, while the extractor's pyramid buffers already hold the NEXT
// batch — so what the host needs of a listed keypoint is captured right behind the descriptor kernels into page-locked memory: its slot, level, position,
// angle and the (2R + 1)^2 window of Sampler::at values around it (R = kTiePatchR: the pattern's radius is 15 * sqrt(2) = 21.2 px before distortion, and the
// omni model compresses away from the optical axis).  mcs_extractor_patch_ties (mcs_capi.hip) reads it behind the batch's event.
__global__ __launch_bounds__(256) void k_tie_capture(ExtractBuffers b, int nimg, int wavesPerImage, int maxTies, uint8_t* __restrict__ out) {
	tie_capture_body(b, nimg, wavesPerImage, maxTies, out, blockIdx.x, gridDim.x);
}

void launch_tie_capture(const ExtractBuffers& b, const PyrDesc& hd, int nimg, int maxTies, uint8_t* devOut, hipStream_t s) {
	const int wavesPerImage = (hd.kpCap + kSlotAlign - 1) / kSlotAlign * kSlotAlign;
	hipLaunchKernelGGL(k_tie_capture, dim3(maxTies < 64 ? maxTies : 64), dim3(256), 0, s, b, nimg, wavesPerImage, maxTies, devOut);
}

}  // namespace mcs

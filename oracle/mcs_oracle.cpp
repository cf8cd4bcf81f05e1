/*
 * mcs_oracle.cpp — CPU ORACLE: restatement of the reference hot path.   TEST INFRASTRUCTURE ONLY.
 *
 * Follows (file:line relative to /root/reference):
 *   src/mdBRIEFextractorOct.cpp:134-203 (ctor tables) 221-248 (IC_Angle) 250-301 (pattern rotation)
 *   303-554 (compute_ORB / compute_dBRIEF / compute_mdBRIEF) 569-861 (oct-tree) 863-976 (cell FAST)
 *   1158-1201 (pyramid) 1203-1337 (descriptors + operator())
 *   src/cam_model_omni.cpp:49-67,146-161,163-220   include/cam_model_omni.h:127-145   include/misc.h:33-49,115-122
 *   src/cMultiFrame.cpp:146-152,342-353   src/cORBmatcher.cpp:46-65,179-323,885-1155,2438-2474   src/misc.cpp:53-69
 * plus the OpenCV 3.x generic-C++ primitives restated in SURVEY.md Appendix A (OpenCV is not vendored by the reference and not
 * installed here).  PINNING (see mcs_oracle.h): extractor, camera model / system, cMultiFrame, the cORBmatcher searches, ComputeBoW and
 * ComputeDistinctiveDescriptors are pinned bit for bit against the reference's own sources compiled unmodified (oracle/_ref,
 * tests/test_oracle_vs_ref*.py); only the OpenCV primitives and orc_window_best are UNPINNED.
 *
 * Documented deviations where the reference has undefined / non-deterministic behaviour:
 *   (1) oct-tree sort ties (pair<int,Node*> compares heap addresses, :782) -> tie broken by node creation
 *       sequence number (later-created sorts higher) = what the reference does under an allocator that hands out increasing
 *       addresses (oracle/_ref runs it on such an arena and then agrees exactly).
 *   (2) descriptor samples that fall outside the 25-px bordered level buffer (out-of-bounds read in the
 *       reference) are clamped to the buffer.
 *   (3) DBoW2 transform: a feature whose path reaches a leaf ABOVE the FeatureVector level leaves `nid` uninitialised in the reference
 *       (TemplatedVocabulary.h:1147-1158, 1229-1251: in practice the previous feature's node) -> 0 (root) here.  Never happens with the
 *       shipped vocabulary at levelsup = 4 (its shallowest leaf is at depth 3, the FeatureVector level is 2).
 * Build: g++ -O3 -march=native -ffp-contract=off -fopenmp (reference flags CMakeLists.txt:37-38 + no FMA contraction).
 */
#include "mcs_oracle.h"
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <malloc.h>
#include <list>
#include <map>
#include <set>
#include <utility>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

using std::ptrdiff_t;
namespace {

const int PATCH_SIZE = 32;        // mdBRIEFextractorOct.cpp:84
const int HALF_PATCH_SIZE = 16;   // :85
const int EDGE_THRESHOLD = 25;    // :86
const double CV_PI_D = 3.1415926535897932384626433832795;
const float DEG2RADf = static_cast<float>(CV_PI_D) / 180.f;  // :83
// include/misc.h:33-41
const double M_PID_ = 3.1415926535897932384626433832795028841971693993;
const float M_PIf_ = 3.1415926535897932384626f;
const double RHOd = 180.0 / M_PID_;
const float RHOf = 180.0f / M_PIf_;

static const signed char kPattern[2048] = {
#include "learned_pattern_64_orb.inc"
};

// ---- A.0 scalars ----
inline int cvRound_(double v) { return (int)lrint(v); }      // round-half-even in the default FP env
inline int cvRoundf_(float v) { return (int)lrintf(v); }
inline int cvFloor_(double v) { int i = (int)v; return i - (i > v); }
inline int cvCeil_(double v) { int i = (int)v; return i + (i < v); }
inline short sat_short(float v) { int iv = cvRoundf_(v); return (short)(iv < -32768 ? -32768 : iv > 32767 ? 32767 : iv); }

struct Img {  // bordered level buffer; (0,0) of the ROI is at buf[border*stride+border]
	int w = 0, h = 0, stride = 0, border = 0;
	std::vector<uint8_t> buf;
	void alloc(int w_, int h_, int b) { w = w_; h = h_; border = b; stride = w + 2 * b; buf.assign((size_t)stride * (h + 2 * b), 0); }
	uint8_t* roi() { return buf.data() + (size_t)border * stride + border; }
	const uint8_t* roi() const { return buf.data() + (size_t)border * stride + border; }
};

}  // namespace

extern "C" {

int orc_cvRound(double v) { return cvRound_(v); }

// ---------------------------------------------------------------- E0
void orc_features_per_level(int nfeatures, float scaleFactor_, int nlevels, int* out) {
	double scaleFactor = scaleFactor_;  // member is double, ctor arg float (h:340, cpp:147)
	double factor = (1.0 / scaleFactor);
	double nDesired = nfeatures * (1 - factor) / (1 - pow(factor, nlevels));
	int sum = 0;
	for (int level = 0; level < nlevels - 1; level++) {
		out[level] = cvRound_(nDesired);
		sum += out[level];
		nDesired *= factor;
	}
	out[nlevels - 1] = std::max(nfeatures - sum, 0);
}

void orc_umax(int* umax) {  // cpp:187-202
	int v, v0, vmax = cvFloor_(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
	int vmin = cvCeil_(HALF_PATCH_SIZE * sqrt(2.f) / 2);
	const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
	for (v = 0; v <= HALF_PATCH_SIZE; ++v) umax[v] = 0;
	for (v = 0; v <= vmax; ++v) umax[v] = cvRound_(sqrt(hp2 - v * v));
	for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
		while (umax[v0] == umax[v0 + 1]) ++v0;
		umax[v] = v0;
		++v0;
	}
}

static void scale_tables(float sf_, int nlevels, std::vector<double>& sc, std::vector<double>& inv) {
	double scaleFactor = sf_;
	sc.assign(nlevels, 1.0);
	inv.assign(nlevels, 1.0);
	for (int i = 1; i < nlevels; i++) sc[i] = sc[i - 1] * scaleFactor;
	double invScaleFactor = 1.0 / scaleFactor;
	for (int i = 1; i < nlevels; i++) inv[i] = inv[i - 1] * invScaleFactor;
}

void orc_level_sizes(int W, int H, float sf, int nlevels, int* w, int* h) {  // cpp:1164-1165
	std::vector<double> sc, inv;
	scale_tables(sf, nlevels, sc, inv);
	for (int l = 0; l < nlevels; ++l) {
		w[l] = cvRound_((double)W * inv[l]);
		h[l] = cvRound_((double)H * inv[l]);
	}
}

int orc_pattern(int descSize, int* xy) {
	int npoints = 2 * 8 * descSize;  // cpp:181
	if (npoints * 2 > 2048) return -1;
	for (int i = 0; i < 2 * npoints; ++i) xy[i] = kPattern[i];
	return npoints;
}

// ---------------------------------------------------------------- A.1 resize INTER_LINEAR (8UC1)
void orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
	double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
	double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
	std::vector<int> xofs(dw), yofs(dh);
	std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
	int xmax = dw;
	for (int dx = 0; dx < dw; dx++) {
		float fx = (float)((dx + 0.5) * scale_x - 0.5);
		int sx = cvFloor_(fx);
		fx -= sx;
		if (sx < 0) { fx = 0; sx = 0; }
		if (sx + 1 >= sw) {
			xmax = std::min(xmax, dx);
			if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
		}
		xofs[dx] = sx;
		ialpha[2 * dx] = sat_short((1.f - fx) * 2048);
		ialpha[2 * dx + 1] = sat_short(fx * 2048);
	}
	for (int dy = 0; dy < dh; dy++) {
		float fy = (float)((dy + 0.5) * scale_y - 0.5);
		int sy = cvFloor_(fy);
		fy -= sy;
		yofs[dy] = sy;
		ibeta[2 * dy] = sat_short((1.f - fy) * 2048);
		ibeta[2 * dy + 1] = sat_short(fy * 2048);
	}
	std::vector<int> T0(dw), T1(dw);
	for (int dy = 0; dy < dh; dy++) {
		int sy0 = yofs[dy];
		for (int k = 0; k < 2; k++) {
			int sy = sy0 + k;
			sy = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0;  // clip(sy, 0, sh)
			const uint8_t* S = src + (size_t)sy * sstride;
			int* D = k == 0 ? T0.data() : T1.data();
			int dx = 0;
			for (; dx < xmax; dx++) D[dx] = S[xofs[dx]] * ialpha[2 * dx] + S[xofs[dx] + 1] * ialpha[2 * dx + 1];
			for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * 2048;
		}
		short b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
		uint8_t* D = dst + (size_t)dy * dstride;
		for (int dx = 0; dx < dw; dx++)
			D[dx] = (uint8_t)((((b0 * (T0[dx] >> 4)) >> 16) + ((b1 * (T1[dx] >> 4)) >> 16) + 2) >> 2);
	}
}

// ---------------------------------------------------------------- A.2 resize INTER_NEAREST
void orc_resize_nearest(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
	double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
	double ifx = 1. / inv_scale_x, ify = 1. / inv_scale_y;
	std::vector<int> x_ofs(dw);
	for (int x = 0; x < dw; x++) x_ofs[x] = std::min(cvFloor_(x * ifx), sw - 1);
	for (int y = 0; y < dh; y++) {
		int sy = std::min(cvFloor_(y * ify), sh - 1);
		for (int x = 0; x < dw; x++) dst[(size_t)y * dstride + x] = src[(size_t)sy * sstride + x_ofs[x]];
	}
}

static inline int reflect101(int p, int len) {
	if (len == 1) return 0;
	while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * (len - 1) - p; }
	return p;
}

// buf is (w+2b) x (h+2b) with the ROI already filled; fill the frame with BORDER_REFLECT_101
void orc_border_reflect101(uint8_t* buf, int w, int h, int stride, int b) {
	uint8_t* roi = buf + (size_t)b * stride + b;
	for (int y = -b; y < h + b; ++y) {
		int sy = reflect101(y, h);
		for (int x = -b; x < w + b; ++x) {
			if (x >= 0 && x < w && y >= 0 && y < h) continue;
			int sx = reflect101(x, w);
			roi[(ptrdiff_t)y * stride + x] = roi[(ptrdiff_t)sy * stride + sx];
		}
	}
}

// ---------------------------------------------------------------- A.3 FAST: FAST_t<patternSize> and cornerScore<patternSize>, patternSize 16 / 12 / 8
// (FastFeatureDetector TYPE_9_16 = 2, TYPE_7_12 = 1, TYPE_5_8 = 0; reference: src/mdBRIEFextractorOct.cpp:869-872, 912-914).  Restated from OpenCV 3.x
// modules/features2d/src/fast.cpp / fast_score.cpp, generic C++ path.  Two properties of that code matter for the two small rings and are kept literally:
//   * every pattern size skips a 3-pixel border (rows 3 .. rows-4, columns 3 .. cols-4), although the rings of TYPE_7_12 / TYPE_5_8 have radius 2 / 1;
//   * the quick rejection test always reads ring entries 0|8, 2|10, 4|12, 6|14, 1|9, 3|11, 5|13, 7|15 of the 25-entry offset table, which for the small
//     rings has wrapped around (pixel[k] = pixel[k - patternSize] for k >= patternSize).  For patternSize 8 the test therefore demands ALL 8 ring pixels
//     darker (or all brighter) than the centre, for patternSize 12 the pairs are (0,8) (2,10) (4,0) (6,2) (1,9) (3,11) (5,1) (7,3) — both stricter than the
//     5-of-8 / 7-of-12 segment criterion their names suggest.
static const int kCircle16[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                     {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
static const int kCircle12[12][2] = {{0, 2}, {1, 2}, {2, 1}, {2, 0}, {2, -1}, {1, -2}, {0, -2}, {-1, -2}, {-2, -1}, {-2, 0}, {-2, 1}, {-1, 2}};
static const int kCircle8[8][2] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}, {-1, 0}, {-1, 1}};

static inline void make_offsets(int pixel[25], int stride, int patternSize) {   // makeOffsets
	const int(*offsets)[2] = patternSize == 16 ? kCircle16 : patternSize == 12 ? kCircle12 : kCircle8;
	int k = 0;
	for (; k < patternSize; k++) pixel[k] = offsets[k][0] + offsets[k][1] * stride;
	for (; k < 25; k++) pixel[k] = pixel[k - patternSize];
}

static int corner_score16(const uint8_t* ptr, const int pixel[25], int threshold) {  // cornerScore<16>
	const int K = 8, N = K * 3 + 1;
	int k, v = ptr[0];
	short d[N];
	for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
	int a0 = threshold;
	for (k = 0; k < 16; k += 2) {
		int a = std::min((int)d[k + 1], (int)d[k + 2]);
		a = std::min(a, (int)d[k + 3]);
		if (a <= a0) continue;
		a = std::min(a, (int)d[k + 4]);
		a = std::min(a, (int)d[k + 5]);
		a = std::min(a, (int)d[k + 6]);
		a = std::min(a, (int)d[k + 7]);
		a = std::min(a, (int)d[k + 8]);
		a0 = std::max(a0, std::min(a, (int)d[k]));
		a0 = std::max(a0, std::min(a, (int)d[k + 9]));
	}
	int b0 = -a0;
	for (k = 0; k < 16; k += 2) {
		int b = std::max((int)d[k + 1], (int)d[k + 2]);
		b = std::max(b, (int)d[k + 3]);
		b = std::max(b, (int)d[k + 4]);
		b = std::max(b, (int)d[k + 5]);
		if (b >= b0) continue;
		b = std::max(b, (int)d[k + 6]);
		b = std::max(b, (int)d[k + 7]);
		b = std::max(b, (int)d[k + 8]);
		b0 = std::min(b0, std::max(b, (int)d[k]));
		b0 = std::min(b0, std::max(b, (int)d[k + 9]));
	}
	threshold = -b0 - 1;
	return threshold;
}

static int corner_score12(const uint8_t* ptr, const int pixel[25], int threshold) {  // cornerScore<12>
	const int K = 6, N = K * 3 + 1;
	int k, v = ptr[0];
	short d[N + 4];
	for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
	int a0 = threshold;
	for (k = 0; k < 12; k += 2) {
		int a = std::min((int)d[k + 1], (int)d[k + 2]);
		if (a <= a0) continue;
		a = std::min(a, (int)d[k + 3]);
		a = std::min(a, (int)d[k + 4]);
		a = std::min(a, (int)d[k + 5]);
		a = std::min(a, (int)d[k + 6]);
		a0 = std::max(a0, std::min(a, (int)d[k]));
		a0 = std::max(a0, std::min(a, (int)d[k + 7]));
	}
	int b0 = -a0;
	for (k = 0; k < 12; k += 2) {
		int b = std::max((int)d[k + 1], (int)d[k + 2]);
		b = std::max(b, (int)d[k + 3]);
		b = std::max(b, (int)d[k + 4]);
		if (b >= b0) continue;
		b = std::max(b, (int)d[k + 5]);
		b = std::max(b, (int)d[k + 6]);
		b0 = std::min(b0, std::max(b, (int)d[k]));
		b0 = std::min(b0, std::max(b, (int)d[k + 7]));
	}
	threshold = -b0 - 1;
	return threshold;
}

static int corner_score8(const uint8_t* ptr, const int pixel[25], int threshold) {  // cornerScore<8>
	const int K = 4, N = K * 3 + 1;
	int k, v = ptr[0];
	short d[N];
	for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
	int a0 = threshold;
	for (k = 0; k < 8; k += 2) {
		int a = std::min((int)d[k + 1], (int)d[k + 2]);
		if (a <= a0) continue;
		a = std::min(a, (int)d[k + 3]);
		a = std::min(a, (int)d[k + 4]);
		a0 = std::max(a0, std::min(a, (int)d[k]));
		a0 = std::max(a0, std::min(a, (int)d[k + 5]));
	}
	int b0 = -a0;
	for (k = 0; k < 8; k += 2) {
		int b = std::max((int)d[k + 1], (int)d[k + 2]);
		b = std::max(b, (int)d[k + 3]);
		if (b >= b0) continue;
		b = std::max(b, (int)d[k + 4]);
		b0 = std::min(b0, std::max(b, (int)d[k]));
		b0 = std::min(b0, std::max(b, (int)d[k + 5]));
	}
	threshold = -b0 - 1;
	return threshold;
}

static inline int corner_score(int patternSize, const uint8_t* ptr, const int pixel[25], int threshold) {
	return patternSize == 16 ? corner_score16(ptr, pixel, threshold) : patternSize == 12 ? corner_score12(ptr, pixel, threshold) : corner_score8(ptr, pixel, threshold);
}

int orc_fast_score(const uint8_t* center, int stride, int threshold) {
	int pixel[25];
	make_offsets(pixel, stride, 16);
	return corner_score16(center, pixel, threshold);
}

int orc_fast_score_type(int type, const uint8_t* center, int stride, int threshold) {
	const int patternSize = type == 2 ? 16 : type == 1 ? 12 : 8;
	int pixel[25];
	make_offsets(pixel, stride, patternSize);
	return corner_score(patternSize, center, pixel, threshold);
}

// FAST_t<patternSize>(img, kps, threshold, nonmax=true) followed by KeyPointsFilter::runByPixelsMask
static int fast_t(int patternSize, const uint8_t* img, int w, int h, int stride, const uint8_t* mask, int mstride, int threshold, orc_keypoint* out, int cap) {
	const int K = patternSize / 2, N = patternSize + K + 1;
	int pixel[25];
	make_offsets(pixel, stride, patternSize);
	threshold = std::min(std::max(threshold, 0), 255);
	uint8_t threshold_tab[512];
	for (int i = -255; i <= 255; i++) threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

	std::vector<uint8_t> bufv((size_t)w * 3, 0);
	uint8_t* buf[3] = {bufv.data(), bufv.data() + w, bufv.data() + 2 * w};
	std::vector<int> cpv((size_t)(w + 1) * 3, 0);
	int* cpbuf[3] = {cpv.data() + 1, cpv.data() + (w + 1) + 1, cpv.data() + 2 * (w + 1) + 1};
	int nout = 0;

	for (int i = 3; i < h - 2; i++) {
		const uint8_t* ptr = img + (size_t)i * stride + 3;
		uint8_t* curr = buf[(i - 3) % 3];
		int* cornerpos = cpbuf[(i - 3) % 3];
		memset(curr, 0, w);
		int ncorners = 0;
		if (i < h - 3) {
			for (int j = 3; j < w - 3; j++, ptr++) {
				int v = ptr[0];
				const uint8_t* tab = &threshold_tab[0] - v + 255;
				int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
				if (d == 0) continue;
				d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
				d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
				d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
				if (d == 0) continue;
				d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
				d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
				d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
				d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
				if (d & 1) {
					int vt = v - threshold, count = 0;
					for (int k = 0; k < N; k++) {
						int x = ptr[pixel[k]];
						if (x < vt) {
							if (++count > K) {
								cornerpos[ncorners++] = j;
								curr[j] = (uint8_t)corner_score(patternSize, ptr, pixel, threshold);
								break;
							}
						} else
							count = 0;
					}
				}
				if (d & 2) {
					int vt = v + threshold, count = 0;
					for (int k = 0; k < N; k++) {
						int x = ptr[pixel[k]];
						if (x > vt) {
							if (++count > K) {
								cornerpos[ncorners++] = j;
								curr[j] = (uint8_t)corner_score(patternSize, ptr, pixel, threshold);
								break;
							}
						} else
							count = 0;
					}
				}
			}
		}
		cornerpos[-1] = ncorners;
		if (i == 3) continue;
		const uint8_t* prev = buf[(i - 4 + 3) % 3];
		const uint8_t* pprev = buf[(i - 5 + 3) % 3];
		cornerpos = cpbuf[(i - 4 + 3) % 3];
		ncorners = cornerpos[-1];
		for (int k = 0; k < ncorners; k++) {
			int j = cornerpos[k];
			int score = prev[j];
			if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
			    score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
				float fx = (float)j, fy = (float)(i - 1);
				if (mask && mask[(size_t)(int)(fy + 0.5f) * mstride + (int)(fx + 0.5f)] == 0) continue;  // runByPixelsMask
				if (nout < cap) {
					orc_keypoint kp = {fx, fy, 7.f, -1.f, (float)score, 0, -1};
					out[nout] = kp;
				}
				nout++;
			}
		}
	}
	return nout;
}

int orc_fast9_16(const uint8_t* img, int w, int h, int stride, const uint8_t* mask, int mstride, int threshold, orc_keypoint* out, int cap) {
	return fast_t(16, img, w, h, stride, mask, mstride, threshold, out, cap);
}

// FastFeatureDetector::create(threshold, true, type)->detect(img, kps, mask): type 0 = TYPE_5_8, 1 = TYPE_7_12, 2 = TYPE_9_16
int orc_fast_type(int type, const uint8_t* img, int w, int h, int stride, const uint8_t* mask, int mstride, int threshold, orc_keypoint* out, int cap) {
	if (type < 0 || type > 2) return -1;
	return fast_t(type == 2 ? 16 : type == 1 ? 12 : 8, img, w, h, stride, mask, mstride, threshold, out, cap);
}

// ---------------------------------------------------------------- AGAST (AgastFeatureDetector, reference src/mdBRIEFextractorOct.cpp:869-870, 912-914)
// OpenCV 3.x modules/features2d/src/agast.cpp + agast_score.cpp (NOT in the reference tree: "parity unpinned" against a real OpenCV; pinned by hand-derived
// known answers and an independent literal Python transcription, tests/test_oracle_agast.py).  What OpenCV's AGAST(img, kps, threshold, true, type) does:
//   1. detection: AGAST_5_8 / AGAST_7_12d / AGAST_7_12s / OAST_9_16 walk machine-generated binary decision trees (Mair et al., ECCV 2010) that DECIDE the
//      accelerated segment test — "N contiguous pixels of the P-pixel ring are all brighter than v + t, or all darker than v - t" (strict), N / P = 5 / 8,
//      7 / 12 (diamond, radius 3), 7 / 12 (square, radius 2), 9 / 16 — in as few pixel reads as possible; the trees are an evaluation order of the
//      predicate, not a different predicate, so the predicate is what is stated here.  Scan: rows y = B .. rows - 1 - B, columns x = B .. cols - 1 - B in
//      raster order with B = 1 / 3 / 2 / 3 (the trees' xsizeB / ysizeB bounds = the ring radius); a non-continuous view is cloned first (img.isContinuous()),
//      so a cell view of a pyramid level behaves like an image of its own;
//   2. response = agast_cornerScore<type>: bisection of the threshold b in [t, 255] with the same predicate ("bmin = b_test if it is still a corner, else
//      bmax = b_test, until bmin >= bmax - 1; return bmin") = the largest b <= 254 for which the pixel is still a corner (agast_score_bisect below; the
//      closed form max(A, B) - 1 with A = max over arcs of min(v - I), B = max over arcs of min(I - v) is checked against it by the tests);
//   3. non-maximum suppression: NOT the 3x3 test of FAST — corners are merged into 4-connected regions (the corner directly above, the corner directly to
//      the left) with a forest of "is dominated by" links (nmsFlags), one survivor per region: the first-seen maximum, except that a later corner with an
//      EQUAL response takes over (`response < response` is the only comparison);
//   4. AgastFeatureDetector::detect then applies KeyPointsFilter::runByPixelsMask.  KeyPoint(x, y, size 7, angle -1, response, octave 0, class_id -1).
struct AgastRing { int P, N, B; int dx[16], dy[16]; };
static const AgastRing& agast_ring(int type) {   // makeAgastOffsets, in ring order
	static const AgastRing rings[4] = {
		{8, 5, 1, {-1, -1, 0, 1, 1, 1, 0, -1}, {0, 1, 1, 1, 0, -1, -1, -1}},
		{12, 7, 3, {-3, -2, -1, 0, 1, 2, 3, 2, 1, 0, -1, -2}, {0, 1, 2, 3, 2, 1, 0, -1, -2, -3, -2, -1}},
		{12, 7, 2, {-2, -2, -1, 0, 1, 2, 2, 2, 1, 0, -1, -2}, {0, 1, 2, 2, 2, 1, 0, -1, -2, -2, -2, -1}},
		{16, 9, 3, {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3}, {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1}},
	};
	return rings[type];
}
static bool agast_is_corner(const AgastRing& R, const uint8_t* c, int stride, int b) {
	const int cb = c[0] + b, c_b = c[0] - b;
	for (int k = 0; k < R.P; ++k) {
		bool br = true, dk = true;
		for (int j = 0; j < R.N && (br || dk); ++j) {
			const int I = c[(ptrdiff_t)R.dy[(k + j) % R.P] * stride + R.dx[(k + j) % R.P]];
			br = br && I > cb;
			dk = dk && I < c_b;
		}
		if (br || dk) return true;
	}
	return false;
}
static int agast_score_bisect(const AgastRing& R, const uint8_t* c, int stride, int threshold) {
	int bmin = threshold, bmax = 255, b_test = (bmax + bmin) / 2;
	while (true) {
		if (agast_is_corner(R, c, stride, b_test)) bmin = b_test; else bmax = b_test;
		if (bmin == bmax - 1 || bmin == bmax) return bmin;
		b_test = (bmin + bmax) / 2;
	}
}
int orc_agast_score_type(int type, const uint8_t* center, int stride, int threshold) {
	if (type < 0 || type > 3) return -1;
	return agast_score_bisect(agast_ring(type), center, stride, threshold);
}
struct AgastK { float x, y, response; };
static std::vector<AgastK> agast_corners(const AgastRing& R, const uint8_t* img, int w, int h, int stride, int threshold) {
	std::vector<AgastK> kpts;
	for (int y = R.B; y < h - R.B; ++y)
		for (int x = R.B; x < w - R.B; ++x) {
			const uint8_t* c = img + (size_t)y * stride + x;
			if (agast_is_corner(R, c, stride, threshold)) kpts.push_back(AgastK{(float)x, (float)y, (float)agast_score_bisect(R, c, stride, threshold)});
		}
	return kpts;
}
// steps 1 + 2 only (AGAST(..., nonmax_suppression = false) with the responses filled in): every corner in raster order — for the tests
int orc_agast_corners(int type, const uint8_t* img, int w, int h, int stride, int threshold, orc_keypoint* out, int cap) {
	if (type < 0 || type > 3) return -1;
	const std::vector<AgastK> kpts = agast_corners(agast_ring(type), img, w, h, stride, threshold);
	for (size_t i = 0; i < kpts.size() && (int)i < cap; ++i) { orc_keypoint kp = {kpts[i].x, kpts[i].y, 7.f, -1.f, kpts[i].response, 0, -1}; out[i] = kp; }
	return (int)kpts.size();
}
// AgastFeatureDetector::create(threshold, true, type)->detect(img, kps, mask): type 0 = AGAST_5_8, 1 = AGAST_7_12d, 2 = AGAST_7_12s, 3 = OAST_9_16
int orc_agast_type(int type, const uint8_t* img, int w, int h, int stride, const uint8_t* mask, int mstride, int threshold, orc_keypoint* out, int cap) {
	if (type < 0 || type > 3) return -1;
	typedef AgastK K;
	const std::vector<K> kpts = agast_corners(agast_ring(type), img, w, h, stride, threshold);
	// the suppression loop of AGAST(), statement by statement
	const size_t num_Corners = kpts.size();
	size_t lastRow = 0, next_lastRow = 0, lastRowCorner_ind = 0, next_lastRowCorner_ind = 0;
	std::vector<int> nmsFlags(num_Corners, -1);
	for (size_t curr_idx = 0; curr_idx < num_Corners; curr_idx++) {
		const K& cur = kpts[curr_idx];
		// check above
		if ((float)(lastRow + 1) < cur.y) { lastRow = next_lastRow; lastRowCorner_ind = next_lastRowCorner_ind; }
		if ((float)next_lastRow != cur.y) { next_lastRow = (size_t)cur.y; next_lastRowCorner_ind = curr_idx; }
		if ((float)(lastRow + 1) == cur.y) {
			while ((kpts[lastRowCorner_ind].x < cur.x) && (kpts[lastRowCorner_ind].y == (float)lastRow)) lastRowCorner_ind++;   // the corner above the current one
			if ((kpts[lastRowCorner_ind].x == cur.x) && (lastRowCorner_ind != curr_idx)) {
				size_t wi = lastRowCorner_ind;
				while (nmsFlags[wi] != -1) wi = (size_t)nmsFlags[wi];   // the maximum of that block
				if (kpts[curr_idx].response < kpts[wi].response) nmsFlags[curr_idx] = (int)wi;
				else nmsFlags[wi] = (int)curr_idx;
			}
		}
		// check left
		int t = (int)curr_idx - 1;
		if ((curr_idx != 0) && (kpts[t].y == cur.y) && (kpts[t].x + 1 == cur.x)) {
			const int currCornerMaxAbove_ind = nmsFlags[curr_idx];
			while (nmsFlags[t] != -1) t = nmsFlags[t];   // the maximum of that area
			if (currCornerMaxAbove_ind == -1) {   // no maximum above
				if ((size_t)t != curr_idx) {
					if (kpts[curr_idx].response < kpts[t].response) nmsFlags[curr_idx] = t;
					else nmsFlags[t] = (int)curr_idx;
				}
			} else if (t != currCornerMaxAbove_ind) {   // maximum above
				if (kpts[currCornerMaxAbove_ind].response < kpts[t].response) { nmsFlags[currCornerMaxAbove_ind] = t; nmsFlags[curr_idx] = t; }
				else { nmsFlags[t] = currCornerMaxAbove_ind; nmsFlags[curr_idx] = currCornerMaxAbove_ind; }
			}
		}
	}
	int nout = 0;
	for (size_t i = 0; i < num_Corners; ++i) {
		if (nmsFlags[i] != -1) continue;
		const float fx = kpts[i].x, fy = kpts[i].y;
		if (mask && mask[(size_t)(int)(fy + 0.5f) * mstride + (int)(fx + 0.5f)] == 0) continue;  // runByPixelsMask
		if (nout < cap) { orc_keypoint kp = {fx, fy, 7.f, -1.f, kpts[i].response, 0, -1}; out[nout] = kp; }
		nout++;
	}
	return nout;
}

// ---------------------------------------------------------------- A.4 5x5 normalised box filter, in place on a ROI
void orc_box5_inplace(uint8_t* roi, int w, int h, int stride) {
	std::vector<uint8_t> out((size_t)w * h);
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x) {
			int s = 0;
			for (int dy = -2; dy <= 2; ++dy) {
				const uint8_t* r = roi + (ptrdiff_t)(y + dy) * stride + x;
				s += r[-2] + r[-1] + r[0] + r[1] + r[2];
			}
			out[(size_t)y * w + x] = (uint8_t)cvRound_(s * (1. / 25));  // saturate_cast<uchar>(sum*scale)
		}
	for (int y = 0; y < h; ++y) memcpy(roi + (size_t)y * stride, out.data() + (size_t)y * w, w);
}

// ---------------------------------------------------------------- A.5 fastAtan2 (degrees)
float orc_fastAtan2(float y, float x) {
	const float K = (float)(180 / CV_PI_D);
	const float p1 = 0.9997878412794807f * K, p3 = -0.3258083974640975f * K, p5 = 0.1555786518463281f * K,
	            p7 = -0.04432655554792128f * K;
	float ax = std::abs(x), ay = std::abs(y);
	float a, c, c2;
	if (ax >= ay) {
		c = ay / (ax + (float)DBL_EPSILON);
		c2 = c * c;
		a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
	} else {
		c = ax / (ay + (float)DBL_EPSILON);
		c2 = c * c;
		a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
	}
	if (x < 0) a = 180.f - a;
	if (y < 0) a = 360.f - a;
	return a;
}

// ---------------------------------------------------------------- E5 IC_Angle (cpp:221-248)
static int g_umax[HALF_PATCH_SIZE + 1];
static bool g_umax_init = false;
static const int* umax_table() {
	if (!g_umax_init) { orc_umax(g_umax); g_umax_init = true; }
	return g_umax;
}

float orc_ic_angle(const uint8_t* img, int stride, float ptx, float pty) {
	const int* u_max = umax_table();
	int m_01 = 0, m_10 = 0;
	const uint8_t* center = img + (ptrdiff_t)cvRoundf_(pty) * stride + cvRoundf_(ptx);
	for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
	for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
		int v_sum = 0;
		int d = u_max[v];
		for (int u = -d; u <= d; ++u) {
			int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
			v_sum += (val_plus - val_minus);
			m_10 += u * (val_plus + val_minus);
		}
		m_01 += v * v_sum;
	}
	return orc_fastAtan2((float)m_01, (float)m_10);
}

// ---------------------------------------------------------------- omni camera model
static inline double horner(const double* coeffs, int s, double x) {  // misc.h:115-122
	double res = 0.0;
	for (int i = s - 1; i >= 0; i--) res = res * x + coeffs[i];
	return res;
}

void orc_world2img(const orc_ocam* cam, double x, double y, double z, double* u, double* v) {  // cam_model_omni.cpp:146-161
	double norm = sqrt(x * x + y * y);
	if (norm == 0.0) norm = 1e-14;
	const double theta = atan(-z / norm);
	const double rho = horner(cam->invP, cam->invP_deg, theta);
	const double uu = x / norm * rho;
	const double vv = y / norm * rho;
	*u = uu * cam->c + vv * cam->d + cam->u0;
	*v = uu * cam->e + vv + cam->v0;
}

void orc_img2world(const orc_ocam* cam, double u, double v, double* x_, double* y_, double* z_) {  // :49-67
	const double invAffine = cam->c - cam->d * cam->e;  // cam_model_omni.h:102
	const double u_t = u - cam->u0;
	const double v_t = v - cam->v0;
	double x = (u_t - cam->d * v_t) / invAffine;
	double y = (-cam->e * u_t + cam->c * v_t) / invAffine;
	const double X2 = x * x;
	const double Y2 = y * y;
	double z = -horner(cam->p, cam->p_deg, sqrt(X2 + Y2));
	double norm = sqrt(X2 + Y2 + z * z);
	*x_ = x / norm;
	*y_ = y / norm;
	*z_ = z / norm;
}

static inline void undistortPointsOcam(const orc_ocam* cam, double ptx, double pty, double scaleF, double* ox, double* oy) {
	double x, y, z;  // cam_model_omni.h:127-138
	orc_img2world(cam, ptx, pty, &x, &y, &z);
	*ox = -x / z * scaleF;
	*oy = -y / z * scaleF;
}

void orc_mirror_mask(const orc_ocam* cam, uint8_t* mask) {  // CreateMirrorMask level 0, cam_model_omni.cpp:163-220
	int w = cam->width, h = cam->height;
	float u0 = (float)cam->v0;  // sic: names swapped in the reference (:187-188)
	float v0 = (float)cam->u0;
	const float offset0 = 22.0f;
	for (int i = 0; i < h; ++i)
		for (int j = 0; j < w; ++j) {
			float ans = sqrtf((float)pow(i - u0, 2) + (float)pow(j - v0, 2));
			mask[(size_t)i * w + j] = (ans < (u0 + offset0)) ? 255 : 0;
		}
}

// ---------------------------------------------------------------- E3/E4 oct-tree
using std::ptrdiff_t;
namespace {
struct Pt2i { int x, y; };
struct Node {
	std::vector<orc_keypoint> vKeys;
	Pt2i UL, UR, BL, BR;
	std::list<Node>::iterator lit;
	bool bNoMore = false;
	long seq = 0;  // creation sequence (deterministic stand-in for the heap address, deviation (1))
	void DivideNode(Node& n1, Node& n2, Node& n3, Node& n4) const {  // cpp:569-629
		const int halfX = (int)ceil(static_cast<double>(UR.x - UL.x) / 2.0);
		const int halfY = (int)ceil(static_cast<double>(BR.y - UL.y) / 2.0);
		n1.UL = UL;
		n1.UR = Pt2i{UL.x + halfX, UL.y};
		n1.BL = Pt2i{UL.x, UL.y + halfY};
		n1.BR = Pt2i{UL.x + halfX, UL.y + halfY};
		n2.UL = n1.UR;
		n2.UR = UR;
		n2.BL = n1.BR;
		n2.BR = Pt2i{UR.x, UL.y + halfY};
		n3.UL = n1.BL;
		n3.UR = n1.BR;
		n3.BL = BL;
		n3.BR = Pt2i{n1.BR.x, BL.y};
		n4.UL = n3.UR;
		n4.UR = n2.BR;
		n4.BL = n3.BR;
		n4.BR = BR;
		for (size_t i = 0; i < vKeys.size(); i++) {
			const orc_keypoint& kp = vKeys[i];
			if (kp.x < n1.UR.x) {
				if (kp.y < n1.BR.y) n1.vKeys.push_back(kp);
				else n3.vKeys.push_back(kp);
			} else if (kp.y < n1.BR.y)
				n2.vKeys.push_back(kp);
			else
				n4.vKeys.push_back(kp);
		}
		if (n1.vKeys.size() == 1) n1.bNoMore = true;
		if (n2.vKeys.size() == 1) n2.bNoMore = true;
		if (n3.vKeys.size() == 1) n3.bNoMore = true;
		if (n4.vKeys.size() == 1) n4.bNoMore = true;
	}
};
typedef std::pair<std::pair<int, long>, Node*> SizeNode;  // ((size, seq), node)
}  // namespace

static std::vector<orc_keypoint> DistributeOctTree(const std::vector<orc_keypoint>& vToDistributeKeys, int minX, int maxX,
                                                   int minY, int maxY, int N) {  // cpp:631-861
	std::vector<orc_keypoint> vResultKeys;
	const int nIni = cvRound_(static_cast<double>(maxX - minX) / (maxY - minY));
	if (nIni < 1) return vResultKeys;  // reference divides by zero here; treat as "no keypoints"
	const double hX = static_cast<double>(maxX - minX) / nIni;
	std::list<Node> lNodes;
	std::vector<Node*> vpIniNodes(nIni);
	long seq = 0;
	for (int i = 0; i < nIni; i++) {
		Node ni;
		ni.UL = Pt2i{(int)(hX * static_cast<double>(i)), 0};
		ni.UR = Pt2i{(int)(hX * static_cast<double>(i + 1)), 0};
		ni.BL = Pt2i{ni.UL.x, maxY - minY};
		ni.BR = Pt2i{ni.UR.x, maxY - minY};
		ni.seq = seq++;
		lNodes.push_back(ni);
		vpIniNodes[i] = &lNodes.back();
	}
	for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
		const orc_keypoint& kp = vToDistributeKeys[i];
		size_t idx = (size_t)(kp.x / hX);
		if (idx >= (size_t)nIni) idx = nIni - 1;  // cannot happen for x < maxX-minX; guards UB
		vpIniNodes[idx]->vKeys.push_back(kp);
	}
	std::list<Node>::iterator lit = lNodes.begin();
	while (lit != lNodes.end()) {
		if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
		else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
		else lit++;
	}
	bool bFinish = false;
	std::vector<SizeNode> vSizeAndPointerToNode;
	auto push_child = [&](Node& n, bool countExpand, int& nToExpand) {
		if (n.vKeys.size() > 0) {
			n.seq = seq++;
			lNodes.push_front(n);
			lNodes.front().lit = lNodes.begin();
			if (n.vKeys.size() > 1) {
				if (countExpand) nToExpand++;
				vSizeAndPointerToNode.push_back(std::make_pair(std::make_pair((int)n.vKeys.size(), lNodes.front().seq), &lNodes.front()));
			}
		}
	};
	while (!bFinish) {
		int prevSize = (int)lNodes.size();
		lit = lNodes.begin();
		int nToExpand = 0;
		vSizeAndPointerToNode.clear();
		while (lit != lNodes.end()) {
			if (lit->bNoMore) { lit++; continue; }
			Node n1, n2, n3, n4;
			lit->DivideNode(n1, n2, n3, n4);
			push_child(n1, true, nToExpand);
			push_child(n2, true, nToExpand);
			push_child(n3, true, nToExpand);
			push_child(n4, true, nToExpand);
			lit = lNodes.erase(lit);
		}
		if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
			bFinish = true;
		} else if (((int)lNodes.size() + nToExpand * 3) > N) {
			while (!bFinish) {
				prevSize = (int)lNodes.size();
				std::vector<SizeNode> vPrev = vSizeAndPointerToNode;
				vSizeAndPointerToNode.clear();
				std::sort(vPrev.begin(), vPrev.end(),
				          [](const SizeNode& a, const SizeNode& b) { return a.first < b.first; });
				for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
					Node n1, n2, n3, n4;
					vPrev[j].second->DivideNode(n1, n2, n3, n4);
					int dummy = 0;
					push_child(n1, false, dummy);
					push_child(n2, false, dummy);
					push_child(n3, false, dummy);
					push_child(n4, false, dummy);
					lNodes.erase(vPrev[j].second->lit);
					if ((int)lNodes.size() >= N) break;
				}
				if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
			}
		}
	}
	vResultKeys.reserve(lNodes.size());
	for (std::list<Node>::iterator it = lNodes.begin(); it != lNodes.end(); it++) {
		std::vector<orc_keypoint>& vNodeKeys = it->vKeys;
		orc_keypoint* pKP = &vNodeKeys[0];
		float maxResponse = pKP->response;
		for (size_t k = 1; k < vNodeKeys.size(); k++) {
			if (vNodeKeys[k].response > maxResponse) {
				pKP = &vNodeKeys[k];
				maxResponse = vNodeKeys[k].response;
			}
		}
		vResultKeys.push_back(*pKP);
	}
	return vResultKeys;
}

int orc_distribute_octtree(const orc_keypoint* in, int n, int minX, int maxX, int minY, int maxY, int N, orc_keypoint* out,
                           int cap) {
	std::vector<orc_keypoint> v(in, in + n);
	// root iterators are needed for phase-B erase of root nodes: set them
	std::vector<orc_keypoint> r = DistributeOctTree(v, minX, maxX, minY, maxY, N);
	int m = (int)r.size();
	for (int i = 0; i < m && i < cap; ++i) out[i] = r[i];
	return m;
}

// ---------------------------------------------------------------- extractor
struct orc_extractor {
	orc_params p;
	std::vector<double> mvScaleFactor, mvInvScaleFactor;
	std::vector<int> mnFeaturesPerLevel;
	std::vector<int> pattern;  // x,y interleaved, 2*npoints ints
	int npoints;
	std::vector<Img> pyr, maskpyr, pyr_unblurred;
	bool has_mask = false;
	std::vector<std::vector<orc_keypoint> > candidates, selected;
	std::vector<char> blurred;
};

orc_extractor* orc_extractor_create(const orc_params* p) {
	if (p->nlevels < 1 || p->descSize < 1 || 2 * 2 * 8 * p->descSize > 2048) return nullptr;
	if (p->fastAgastType < 0 || p->fastAgastType > (p->useAgast ? 3 : 2)) return nullptr;  // FAST: TYPE_5_8 / 7_12 / 9_16; AGAST: AGAST_5_8 / 7_12d / 7_12s / OAST_9_16
	orc_extractor* e = new orc_extractor;
	e->p = *p;
	scale_tables(p->scaleFactor, p->nlevels, e->mvScaleFactor, e->mvInvScaleFactor);
	e->mnFeaturesPerLevel.resize(p->nlevels);
	orc_features_per_level(p->nfeatures, p->scaleFactor, p->nlevels, e->mnFeaturesPerLevel.data());
	e->npoints = 2 * 8 * p->descSize;
	e->pattern.resize(2 * e->npoints);
	orc_pattern(p->descSize, e->pattern.data());
	return e;
}
void orc_extractor_destroy(orc_extractor* e) { delete e; }

static void ComputePyramid(orc_extractor* e, const uint8_t* image, int W, int H, int stride, const uint8_t* mask, int mstride) {
	int nl = e->p.nlevels;
	e->pyr.assign(nl, Img());
	e->maskpyr.assign(nl, Img());
	e->has_mask = mask != nullptr;
	for (int level = 0; level < nl; ++level) {
		double scale = e->mvInvScaleFactor[level];
		int sw = cvRound_((double)W * scale), sh = cvRound_((double)H * scale);
		Img& L = e->pyr[level];
		L.alloc(sw, sh, EDGE_THRESHOLD);
		Img& M = e->maskpyr[level];
		if (mask) M.alloc(sw, sh, EDGE_THRESHOLD);  // frame stays 0 = BORDER_CONSTANT
		if (level != 0) {
			const Img& P = e->pyr[level - 1];
			orc_resize_linear(P.roi(), P.w, P.h, P.stride, L.roi(), sw, sh, L.stride);
			if (mask) {
				const Img& PM = e->maskpyr[level - 1];
				orc_resize_nearest(PM.roi(), PM.w, PM.h, PM.stride, M.roi(), sw, sh, M.stride);
			}
		} else {
			for (int y = 0; y < H; ++y) memcpy(L.roi() + (size_t)y * L.stride, image + (size_t)y * stride, W);
			if (mask)
				for (int y = 0; y < H; ++y) memcpy(M.roi() + (size_t)y * M.stride, mask + (size_t)y * mstride, W);
		}
		orc_border_reflect101(L.buf.data(), sw, sh, L.stride, EDGE_THRESHOLD);
	}
}

static void ComputeKeyPointsOctTree(orc_extractor* e, std::vector<std::vector<orc_keypoint> >& allKeypoints) {  // cpp:863-976
	int nl = e->p.nlevels;
	allKeypoints.assign(nl, std::vector<orc_keypoint>());
	e->candidates.assign(nl, std::vector<orc_keypoint>());
	const double W = 30.0;
	std::vector<orc_keypoint> cell(4096);
	for (int level = 0; level < nl; ++level) {
		Img& L = e->pyr[level];
		Img& M = e->maskpyr[level];
		const int minBorderX = EDGE_THRESHOLD - 3;
		const int minBorderY = minBorderX;
		const int maxBorderX = L.w - EDGE_THRESHOLD + 3;
		const int maxBorderY = L.h - EDGE_THRESHOLD + 3;
		std::vector<orc_keypoint> vToDistributeKeys;
		const double width = (maxBorderX - minBorderX);
		const double height = (maxBorderY - minBorderY);
		const int nCols = (int)(width / W);
		const int nRows = (int)(height / W);
		if (nCols < 1 || nRows < 1) continue;  // reference would divide by zero; level too small -> no keypoints
		const int wCell = (int)ceil(width / nCols);
		const int hCell = (int)ceil(height / nRows);
		for (int i = 0; i < nRows; i++) {
			const double iniY = minBorderY + i * hCell;
			double maxY = iniY + hCell + 6;
			if (iniY >= maxBorderY - 3) continue;
			if (maxY > maxBorderY) maxY = maxBorderY;
			for (int j = 0; j < nCols; j++) {
				const double iniX = minBorderX + j * wCell;
				double maxX = iniX + wCell + 6;
				if (iniX >= maxBorderX - 6) continue;
				if (maxX > maxBorderX) maxX = maxBorderX;
				int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
				const uint8_t* view = L.roi() + (ptrdiff_t)y0 * L.stride + x0;
				const uint8_t* mview = e->has_mask ? M.roi() + (ptrdiff_t)y0 * M.stride + x0 : nullptr;
				int n = e->p.useAgast ? orc_agast_type(e->p.fastAgastType, view, x1 - x0, y1 - y0, L.stride, mview, M.stride, e->p.fastThreshold, cell.data(), (int)cell.size())
				                      : orc_fast_type(e->p.fastAgastType, view, x1 - x0, y1 - y0, L.stride, mview, M.stride, e->p.fastThreshold, cell.data(), (int)cell.size());
				for (int k = 0; k < n; ++k) {
					orc_keypoint kp = cell[k];
					kp.x += j * wCell;
					kp.y += i * hCell;
					vToDistributeKeys.push_back(kp);
				}
			}
		}
		e->candidates[level] = vToDistributeKeys;
		std::vector<orc_keypoint>& keypoints = allKeypoints[level];
		keypoints = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
		                              e->mnFeaturesPerLevel[level]);
		const int scaledPatchSize = (int)(PATCH_SIZE * e->mvScaleFactor[level]);
		for (size_t i = 0; i < keypoints.size(); ++i) {
			keypoints[i].x += minBorderX;
			keypoints[i].y += minBorderY;
			keypoints[i].octave = level;
			keypoints[i].size = (float)scaledPatchSize;
		}
	}
	for (int level = 0; level < nl; ++level) {  // computeOrientation, cpp:557-566,974-975
		Img& L = e->pyr[level];
		for (auto& kp : allKeypoints[level]) kp.angle = orc_ic_angle(L.roi(), L.stride, kp.x, kp.y);
	}
}

static inline uint8_t sample(const Img& L, int row, int col) {  // image.ptr<uchar>(row)[col] on the ROI; deviation (2): clamp
	int r = row + L.border, c = col + L.border;
	int bh = L.h + 2 * L.border, bw = L.w + 2 * L.border;
	r = r < 0 ? 0 : (r >= bh ? bh - 1 : r);
	c = c < 0 ? 0 : (c >= bw ? bw - 1 : c);
	return L.buf[(size_t)r * L.stride + c];
}

static void rotatePattern(const int* pin, int npoints, int* pout, double ax, double ay) {  // cpp:285-301
	for (int p = 0; p < npoints; ++p) {
		pout[2 * p] = cvRound_(pin[2 * p] * ax - pin[2 * p + 1] * ay);
		pout[2 * p + 1] = cvRound_(pin[2 * p] * ay + pin[2 * p + 1] * ax);
	}
}

static void rotateAndDistortPattern(double ukx, double uky, const int* pin, int npoints, int* pout, const orc_ocam* cam, double ax,
                                    double ay) {  // cpp:250-283
	const double npointsd = static_cast<double>(npoints);
	std::vector<double> xcoords(npoints), ycoords(npoints);
	double sumX = 0.0, sumY = 0.0;
	const double p1 = cam->p[0];
	for (int p = 0; p < npoints; ++p) {
		double xr = pin[2 * p] * ax - pin[2 * p + 1] * ay + ukx;
		double yr = pin[2 * p] * ay + pin[2 * p + 1] * ax + uky;
		orc_world2img(cam, xr, yr, -p1, &xcoords[p], &ycoords[p]);  // distortPointsOcam, cam_model_omni.h:140-145
		sumX += xcoords[p];
		sumY += ycoords[p];
	}
	double meanX = sumX / npointsd;
	double meanY = sumY / npointsd;
	for (int p = 0; p < npoints; ++p) {
		pout[2 * p] = cvRound_(xcoords[p] - meanX);
		pout[2 * p + 1] = cvRound_(ycoords[p] - meanY);
	}
}

static void sample_bits(const Img& L, const orc_keypoint& kp, const int* pat, int descsize, uint8_t* out) {
	int row = cvRoundf_(kp.y), col = cvRoundf_(kp.x);
	for (int i = 0; i < descsize; ++i, pat += 32) {
		int val = 0;
		for (int b = 0; b < 8; ++b) {
			int t0 = sample(L, row + pat[4 * b + 1], col + pat[4 * b]);
			int t1 = sample(L, row + pat[4 * b + 3], col + pat[4 * b + 2]);
			val |= (t0 < t1) << b;
		}
		out[i] = (uint8_t)val;
	}
}

static void compute_ORB(const Img& L, const orc_keypoint& kp, const orc_extractor* e, uint8_t* desc) {  // cpp:303-354
	std::vector<int> rot(2 * e->npoints);
	double angle = static_cast<double>(kp.angle * DEG2RADf);
	rotatePattern(e->pattern.data(), e->npoints, rot.data(), cos(angle), sin(angle));
	sample_bits(L, kp, rot.data(), e->p.descSize, desc);
}

static void compute_dBRIEF(const Img& L, const orc_keypoint& kp, double ukx, double uky, const orc_extractor* e, const orc_ocam* cam,
                           uint8_t* desc) {  // cpp:356-408
	std::vector<int> rot(2 * e->npoints);
	double angle = static_cast<double>(kp.angle * DEG2RADf);
	rotateAndDistortPattern(ukx, uky, e->pattern.data(), e->npoints, rot.data(), cam, cos(angle), sin(angle));
	sample_bits(L, kp, rot.data(), e->p.descSize, desc);
}

static void compute_mdBRIEF(const Img& L, const orc_keypoint& kp, double ukx, double uky, const orc_extractor* e,
                            const orc_ocam* cam, uint8_t* desc, uint8_t* dmask) {  // cpp:410-554
	int np = e->npoints, ds = e->p.descSize;
	std::vector<int> pat(2 * np), m1(2 * np), m2(2 * np);
	double rot = 20.0 / RHOd;
	double angle = static_cast<double>(kp.angle / RHOf);
	double angle1 = angle + rot;
	double angle2 = angle - rot;
	rotateAndDistortPattern(ukx, uky, e->pattern.data(), np, pat.data(), cam, cos(angle), sin(angle));
	rotateAndDistortPattern(ukx, uky, e->pattern.data(), np, m1.data(), cam, cos(angle1), sin(angle1));
	rotateAndDistortPattern(ukx, uky, e->pattern.data(), np, m2.data(), cam, cos(angle2), sin(angle2));
	std::vector<uint8_t> d1(ds), d2(ds);
	sample_bits(L, kp, pat.data(), ds, desc);
	sample_bits(L, kp, m1.data(), ds, d1.data());
	sample_bits(L, kp, m2.data(), ds, d2.data());
	// mask bit = 1 iff both +-20deg tests equal the main test (stable_val == 0), cpp:468-475
	for (int i = 0; i < ds; ++i) dmask[i] = (uint8_t) ~((desc[i] ^ d1[i]) | (desc[i] ^ d2[i]));
}

int orc_extract(orc_extractor* e, const uint8_t* img, int w, int h, int stride, const uint8_t* mask, int mstride,
                const orc_ocam* cam, orc_keypoint* kps, int cap, uint8_t* desc, uint8_t* dmask) {  // operator(), cpp:1244-1337
	if (!img || w <= 0 || h <= 0) return 0;
	ComputePyramid(e, img, w, h, stride, mask, mstride);
	e->pyr_unblurred = e->pyr;
	std::vector<std::vector<orc_keypoint> > allKeypoints;
	ComputeKeyPointsOctTree(e, allKeypoints);
	e->selected = allKeypoints;
	int nl = e->p.nlevels, ds = e->p.descSize;
	int nkeypoints = 0;
	for (int level = 0; level < nl; ++level) nkeypoints += (int)allKeypoints[level].size();
	if (nkeypoints > cap) return -2;
	int offset = 0;
	const double scaleF = cam ? cam->p[0] : 0.0;  // camModel.Get_P().at<double>(0), :1288
	e->blurred.assign(nl, 0);
	for (int level = 0; level < nl; ++level) {
		std::vector<orc_keypoint>& keypoints = allKeypoints[level];
		int n = (int)keypoints.size();
		if (n == 0) continue;
		Img& L = e->pyr[level];
		orc_box5_inplace(L.roi(), L.w, L.h, L.stride);  // :1301
		e->blurred[level] = 1;
		std::vector<double> und(2 * n, 0.0);
		float scale = (float)e->mvScaleFactor[level];
		if (e->p.do_dBrief) {
			for (int i = 0; i < n; ++i)
				undistortPointsOcam(cam, static_cast<double>(keypoints[i].x * scale), static_cast<double>(keypoints[i].y * scale), scaleF,
				                    &und[2 * i], &und[2 * i + 1]);
		}
		uint8_t* D = desc + (size_t)offset * ds;
		uint8_t* DM = dmask + (size_t)offset * ds;
		memset(D, 0, (size_t)n * ds);   // computeDescriptors zero-fills both (:1215-1216)
		memset(DM, 0, (size_t)n * ds);
		for (int i = 0; i < n; ++i) {
			if (e->p.learnMasks) compute_mdBRIEF(L, keypoints[i], und[2 * i], und[2 * i + 1], e, cam, D + (size_t)i * ds, DM + (size_t)i * ds);
			else if (e->p.do_dBrief) compute_dBRIEF(L, keypoints[i], und[2 * i], und[2 * i + 1], e, cam, D + (size_t)i * ds);
			else compute_ORB(L, keypoints[i], e, D + (size_t)i * ds);
		}
		if (level != 0)
			for (auto& kp : keypoints) { kp.x *= scale; kp.y *= scale; }
		for (int i = 0; i < n; ++i) kps[offset + i] = keypoints[i];
		offset += n;
	}
	return nkeypoints;
}

int orc_tap_level_size(orc_extractor* e, int level, int* w, int* h) {
	if (level < 0 || level >= (int)e->pyr.size()) return -1;
	*w = e->pyr[level].w; *h = e->pyr[level].h;
	return 0;
}
int orc_tap_level_image(orc_extractor* e, int level, int blurred, uint8_t* out) {
	if (level < 0 || level >= (int)e->pyr.size()) return -1;
	const Img& L = blurred ? e->pyr[level] : e->pyr_unblurred[level];
	for (int y = 0; y < L.h; ++y) memcpy(out + (size_t)y * L.w, L.roi() + (size_t)y * L.stride, L.w);
	return blurred ? (int)e->blurred[level] : 1;
}
int orc_tap_level_mask(orc_extractor* e, int level, uint8_t* out) {
	if (level < 0 || level >= (int)e->maskpyr.size() || !e->has_mask) return -1;
	const Img& L = e->maskpyr[level];
	for (int y = 0; y < L.h; ++y) memcpy(out + (size_t)y * L.w, L.roi() + (size_t)y * L.stride, L.w);
	return 0;
}
int orc_tap_candidates(orc_extractor* e, int level, orc_keypoint* out, int cap) {
	if (level < 0 || level >= (int)e->candidates.size()) return -1;
	int n = (int)e->candidates[level].size();
	for (int i = 0; i < n && i < cap; ++i) out[i] = e->candidates[level][i];
	return n;
}
int orc_tap_selected(orc_extractor* e, int level, orc_keypoint* out, int cap) {
	if (level < 0 || level >= (int)e->selected.size()) return -1;
	int n = (int)e->selected[level].size();
	for (int i = 0; i < n && i < cap; ++i) out[i] = e->selected[level][i];
	return n;
}

void orc_rays(const orc_ocam* cam, const orc_keypoint* kps, int n, double* rays) {  // cMultiFrame.cpp:146-152
	for (int i = 0; i < n; ++i)
		orc_img2world(cam, static_cast<double>(kps[i].x), static_cast<double>(kps[i].y), &rays[3 * i], &rays[3 * i + 1], &rays[3 * i + 2]);
}

int orc_pos_in_grid(const orc_ocam* cam, float x, float y, int* gx, int* gy) {  // cMultiFrame.cpp:342-353 (64x48 grid)
	const int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64;
	double wInv = static_cast<double>(FRAME_GRID_COLS) / static_cast<double>(cam->width - 0);
	double hInv = static_cast<double>(FRAME_GRID_ROWS) / static_cast<double>(cam->height - 0);
	*gx = cvRound_((x - 0) * wInv);
	*gy = cvRound_((y - 0) * hInv);
	if (*gx < 0 || *gx >= FRAME_GRID_COLS || *gy < 0 || *gy >= FRAME_GRID_ROWS) return 0;
	return 1;
}

// ---------------------------------------------------------------- matcher
int orc_dist64(const uint64_t* a, const uint64_t* b, int dim) {  // cORBmatcher.cpp:2438-2450
	uint64_t dist = 0;
	for (int d = 0; d < dim / 8; ++d) dist += __builtin_popcountll(a[d] ^ b[d]);
	return static_cast<int>(dist);
}

int orc_dist64_masked(const uint64_t* a, const uint64_t* b, const uint64_t* ma, const uint64_t* mb, int dim) {  // :2452-2474
	uint64_t dist = 0;
	for (int i = 0; i < dim / 8; ++i) {
		uint64_t axorb = a[i] ^ b[i];
		dist += __builtin_popcountll(axorb & ma[i]);
		dist += __builtin_popcountll(axorb & mb[i]);
	}
	return static_cast<int>(dist / 2);
}

void orc_thresholds(int featDim, int havingMasks, int* th_high, int* th_low) {  // :46-65
	if (havingMasks) { *th_high = (int)floor(1.5 * featDim); *th_low = (int)floor((double)featDim); }
	else { *th_high = 3 * featDim; *th_low = 2 * featDim; }
}

static inline int dist_any(const uint8_t* d1, const uint8_t* m1, int i, const uint8_t* d2, const uint8_t* m2, int j, int dim, int masks) {
	const uint64_t* a = (const uint64_t*)(d1 + (size_t)i * dim);
	const uint64_t* b = (const uint64_t*)(d2 + (size_t)j * dim);
	if (masks) return orc_dist64_masked(a, b, (const uint64_t*)(m1 + (size_t)i * dim), (const uint64_t*)(m2 + (size_t)j * dim), dim);
	return orc_dist64(a, b, dim);
}

int orc_search_kf_kf(const uint8_t* d1, const uint8_t* m1, const uint8_t* valid1, int n1, const uint8_t* d2, const uint8_t* m2,
                     const uint8_t* valid2, int n2, int dim, int havingMasks, double nnratio, int* match12) {  // :885-966
	int TH_HIGH, TH_LOW;
	orc_thresholds(dim, havingMasks, &TH_HIGH, &TH_LOW);
	std::vector<char> vbMatched2(n2, 0);
	int nmatches = 0;
	for (int idx1 = 0; idx1 < n1; ++idx1) {
		match12[idx1] = -1;
		if (!valid1[idx1]) continue;
		int bestDist1 = INT_MAX, bestIdx2 = -1, bestDist2 = INT_MAX;
		for (int idx2 = 0; idx2 < n2; ++idx2) {
			if (vbMatched2[idx2] || !valid2[idx2]) continue;
			int dist = dist_any(d1, m1, idx1, d2, m2, idx2, dim, havingMasks);
			if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
			else if (dist < bestDist2) bestDist2 = dist;
		}
		if (bestDist1 < TH_LOW) {
			if (static_cast<double>(bestDist1) < nnratio * static_cast<double>(bestDist2)) {
				match12[idx1] = bestIdx2;
				vbMatched2[bestIdx2] = 1;
				++nmatches;
			}
		}
	}
	return nmatches;
}

int orc_search_kf_f(const uint8_t* dKF, const uint8_t* mKF, const uint8_t* validKF, int nKF, const uint8_t* dF, const uint8_t* mF,
                    int nF, int dim, int havingMasks, double nnratio, int* matchF) {  // :179-323 without the BoW-node restriction
	int TH_HIGH, TH_LOW;
	orc_thresholds(dim, havingMasks, &TH_HIGH, &TH_LOW);
	for (int j = 0; j < nF; ++j) matchF[j] = -1;
	int nmatches = 0;
	for (int iKF = 0; iKF < nKF; ++iKF) {
		if (!validKF[iKF]) continue;
		int bestDist1 = INT_MAX, bestIdxF = -1, bestDist2 = INT_MAX;
		for (int iF = 0; iF < nF; ++iF) {
			if (matchF[iF] >= 0) continue;
			int dist = dist_any(dKF, mKF, iKF, dF, mF, iF, dim, havingMasks);
			if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = iF; }
			else if (dist < bestDist2) bestDist2 = dist;
		}
		if (bestDist1 <= TH_LOW) {
			if (static_cast<double>(bestDist1) < nnratio * static_cast<double>(bestDist2)) {
				matchF[bestIdxF] = iKF;
				++nmatches;
			}
		}
	}
	return nmatches;
}

int orc_check_epipolar(const double* ray1, const double* ray2, const double* E, double thresh) {  // misc.cpp:53-69
	double t[3];
	for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += ray2[k] * E[3 * k + j]; t[j] = s; }
	double nom = 0;
	for (int k = 0; k < 3; ++k) nom += t[k] * ray1[k];
	double Ex1[3], Etx2[3];
	for (int i = 0; i < 3; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += E[3 * i + k] * ray1[k]; Ex1[i] = s; }
	for (int i = 0; i < 3; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += E[3 * k + i] * ray2[k]; Etx2[i] = s; }
	const double den = Ex1[0] * Ex1[0] + Ex1[1] * Ex1[1] + Ex1[2] * Ex1[2] + Etx2[0] * Etx2[0] + Etx2[1] * Etx2[1] + Etx2[2] * Etx2[2];
	if (den == 0.0) return 0;
	const double dsqr = (nom * nom) / den;
	return dsqr < thresh;
}

int orc_search_triangulation(const uint8_t* d1, const uint8_t* m1, const uint8_t* hasMP1, const int* cam1, const double* rays1, int n1,
                             const uint8_t* d2, const uint8_t* m2, const uint8_t* hasMP2, const int* cam2, const double* rays2, int n2,
                             const double* E, int nrCams, int dim, int havingMasks, int* match12) {  // :968-1155 (mbCheckOrientation=false)
	int TH_HIGH, TH_LOW;
	orc_thresholds(dim, havingMasks, &TH_HIGH, &TH_LOW);
	std::vector<char> vbMatched2(n2, 0);
	int nmatches = 0;
	for (int idx1 = 0; idx1 < n1; ++idx1) {
		match12[idx1] = -1;
		if (hasMP1[idx1]) continue;
		int camIdx1 = cam1[idx1];
		std::vector<std::pair<int, size_t> > vDistIndex;
		for (int idx2 = 0; idx2 < n2; ++idx2) {
			if (vbMatched2[idx2] || hasMP2[idx2]) continue;
			if (camIdx1 != cam2[idx2]) continue;
			int dist = dist_any(d1, m1, idx1, d2, m2, idx2, dim, havingMasks);
			if (dist > TH_LOW) continue;
			vDistIndex.push_back(std::make_pair(dist, (size_t)idx2));
		}
		if (vDistIndex.empty()) continue;
		std::sort(vDistIndex.begin(), vDistIndex.end());
		int BestDist = vDistIndex.front().first;
		int DistTh = cvRound_(2 * BestDist);
		for (size_t id = 0; id < vDistIndex.size(); ++id) {
			if (vDistIndex[id].first > DistTh) break;
			int currentIdx2 = (int)vDistIndex[id].second;
			int camIdx2 = cam2[currentIdx2];
			if (orc_check_epipolar(rays1 + 3 * idx1, rays2 + 3 * currentIdx2, E + 9 * ((size_t)camIdx1 * nrCams + camIdx2), 1e-2)) {
				vbMatched2[currentIdx2] = 1;
				match12[idx1] = currentIdx2;
				nmatches++;
				break;
			}
		}
	}
	return nmatches;
}

// ---------------------------------------------------------------- "next" row: SearchByProjection(F, mapPoints, th)
int orc_search_by_projection(const double* projx, const double* projy, const double* viewcos, const int* level, const int* pcam,
                             const uint8_t* pdesc, const uint8_t* pmask, int nproj, const orc_keypoint* keys, const uint8_t* fdesc,
                             const uint8_t* fmask, const int* fcam, uint8_t* assigned, int nfeat, const int* width, const int* height, int nrCams,
                             const double* scaleFactors, int nlevels, double th, double nnratio, int dim, int havingMasks, int* match) {
	const int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64;
	int TH_HIGH, TH_LOW;
	orc_thresholds(dim, havingMasks, &TH_HIGH, &TH_LOW);
	// mGrids[cam][x][y]: filled in mvKeys order (src/cMultiFrame.cpp:167-184) with PosInGrid (:342-353)
	std::vector<std::vector<std::vector<std::vector<size_t> > > > grids(nrCams);
	std::vector<double> wInv(nrCams), hInv(nrCams);
	for (int c = 0; c < nrCams; ++c) {
		grids[c].assign(FRAME_GRID_COLS, std::vector<std::vector<size_t> >(FRAME_GRID_ROWS));
		wInv[c] = static_cast<double>(FRAME_GRID_COLS) / static_cast<double>(width[c] - 0);
		hInv[c] = static_cast<double>(FRAME_GRID_ROWS) / static_cast<double>(height[c] - 0);
	}
	for (int i = 0; i < nfeat; ++i) {
		const int c = fcam[i];
		const int posX = cvRound_((keys[i].x - 0) * wInv[c]);
		const int posY = cvRound_((keys[i].y - 0) * hInv[c]);
		if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) continue;
		grids[c][posX][posY].push_back(i);
	}
	int nmatches = 0;
	const bool bFactor = th != 1.0;
	for (int p = 0; p < nproj; ++p) {
		match[p] = -1;
		const int cam = pcam[p];
		const int nPredictedLevel = level[p];
		double r = viewcos[p] > 0.998 ? 2.5 : 4.0;   // RadiusByViewingCos (:169-175)
		if (bFactor) r *= th;
		// GetFeaturesInArea(cam, x, y, r*mvScaleFactors[level], level-1, level)
		const double x = projx[p], y = projy[p], rr = r * scaleFactors[nPredictedLevel];
		const int minLevel = nPredictedLevel - 1, maxLevel = nPredictedLevel;
		std::vector<size_t> vIndices;
		do {
			int nMinCellX = (int)floor((x - 0 - rr) * wInv[cam]);
			nMinCellX = std::max(0, nMinCellX);
			if (nMinCellX >= FRAME_GRID_COLS) break;
			int nMaxCellX = (int)ceil((x - 0 + rr) * wInv[cam]);
			nMaxCellX = std::min(FRAME_GRID_COLS - 1, nMaxCellX);
			if (nMaxCellX < 0) break;
			int nMinCellY = (int)floor((y - 0 - rr) * hInv[cam]);
			nMinCellY = std::max(0, nMinCellY);
			if (nMinCellY >= FRAME_GRID_ROWS) break;
			int nMaxCellY = (int)ceil((y - 0 + rr) * hInv[cam]);
			nMaxCellY = std::min(FRAME_GRID_ROWS - 1, nMaxCellY);
			if (nMaxCellY < 0) break;
			bool bCheckLevels = true, bSameLevel = false;
			if (minLevel == -1 && maxLevel == -1) bCheckLevels = false;
			else if (minLevel == maxLevel) bSameLevel = true;
			for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
				for (int iy = nMinCellY; iy <= nMaxCellY; ++iy) {
					const std::vector<size_t>& vCell = grids[cam][ix][iy];
					for (size_t j = 0; j < vCell.size(); ++j) {
						const orc_keypoint& kpUn = keys[vCell[j]];
						if (bCheckLevels && !bSameLevel) { if (kpUn.octave < minLevel || kpUn.octave > maxLevel) continue; }
						else if (bSameLevel) { if (kpUn.octave != minLevel) continue; }
						if (std::abs(kpUn.x - x) > rr || std::abs(kpUn.y - y) > rr) continue;
						vIndices.push_back(vCell[j]);
					}
				}
		} while (0);
		(void)nlevels;
		if (vIndices.empty()) continue;
		int bestDist = INT_MAX, bestLevel = -1, bestDist2 = INT_MAX, bestLevel2 = -1, bestIdx = -1;
		for (size_t k = 0; k < vIndices.size(); ++k) {
			const size_t idx = vIndices[k];
			if (assigned[idx]) continue;
			const int dist = dist_any(pdesc, pmask, p, fdesc, fmask, (int)idx, dim, havingMasks);
			if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = keys[idx].octave; bestIdx = (int)idx; }
			else if (dist < bestDist2) { bestLevel2 = keys[idx].octave; bestDist2 = dist; }
		}
		if (bestDist <= TH_HIGH) {
			if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
			assigned[bestIdx] = 1;
			match[p] = bestIdx;
			++nmatches;
		}
	}
	return nmatches;
}

// ---------------------------------------------------------------- "next" row: the other grid-window matchers of cTracking
namespace {
const int FRAME_GRID_ROWS_ = 48, FRAME_GRID_COLS_ = 64, HISTO_LENGTH_ = 30;

struct FrameGrid {   // mGrids + GetFeaturesInArea of one cMultiFrame (src/cMultiFrame.cpp:167-184, 272-340, 342-353)
	const orc_frame_view* f;
	std::vector<std::vector<std::vector<std::vector<size_t> > > > grids;
	std::vector<double> wInv, hInv;
	explicit FrameGrid(const orc_frame_view* fv) : f(fv), grids(fv->nrCams), wInv(fv->nrCams), hInv(fv->nrCams) {
		for (int c = 0; c < f->nrCams; ++c) {
			grids[c].assign(FRAME_GRID_COLS_, std::vector<std::vector<size_t> >(FRAME_GRID_ROWS_));
			wInv[c] = static_cast<double>(FRAME_GRID_COLS_) / static_cast<double>(f->width[c] - 0);
			hInv[c] = static_cast<double>(FRAME_GRID_ROWS_) / static_cast<double>(f->height[c] - 0);
		}
		for (int i = 0; i < f->n; ++i) {
			const int c = f->cam[i];
			const int posX = cvRound_((f->keys[i].x - 0) * wInv[c]);
			const int posY = cvRound_((f->keys[i].y - 0) * hInv[c]);
			if (posX < 0 || posX >= FRAME_GRID_COLS_ || posY < 0 || posY >= FRAME_GRID_ROWS_) continue;
			grids[c][posX][posY].push_back(i);
		}
	}
	std::vector<size_t> GetFeaturesInArea(int cam, double x, double y, double r, int minLevel = -1, int maxLevel = -1) const {
		std::vector<size_t> vIndices;
		int nMinCellX = (int)floor((x - 0 - r) * wInv[cam]);
		nMinCellX = std::max(0, nMinCellX);
		if (nMinCellX >= FRAME_GRID_COLS_) return vIndices;
		int nMaxCellX = (int)ceil((x - 0 + r) * wInv[cam]);
		nMaxCellX = std::min(FRAME_GRID_COLS_ - 1, nMaxCellX);
		if (nMaxCellX < 0) return vIndices;
		int nMinCellY = (int)floor((y - 0 - r) * hInv[cam]);
		nMinCellY = std::max(0, nMinCellY);
		if (nMinCellY >= FRAME_GRID_ROWS_) return vIndices;
		int nMaxCellY = (int)ceil((y - 0 + r) * hInv[cam]);
		nMaxCellY = std::min(FRAME_GRID_ROWS_ - 1, nMaxCellY);
		if (nMaxCellY < 0) return vIndices;
		bool bCheckLevels = true, bSameLevel = false;
		if (minLevel == -1 && maxLevel == -1) bCheckLevels = false;
		else if (minLevel == maxLevel) bSameLevel = true;
		for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
			for (int iy = nMinCellY; iy <= nMaxCellY; ++iy) {
				const std::vector<size_t>& vCell = grids[cam][ix][iy];
				for (size_t j = 0; j < vCell.size(); ++j) {
					const orc_keypoint& kpUn = f->keys[vCell[j]];
					if (bCheckLevels && !bSameLevel) { if (kpUn.octave < minLevel || kpUn.octave > maxLevel) continue; }
					else if (bSameLevel) { if (kpUn.octave != minLevel) continue; }
					if (std::abs(kpUn.x - x) > r || std::abs(kpUn.y - y) > r) continue;
					vIndices.push_back(vCell[j]);
				}
			}
		return vIndices;
	}
};

void ComputeThreeMaxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {  // src/cORBmatcher.cpp:2394-2436
	int max1 = 0, max2 = 0, max3 = 0;
	for (int i = 0; i < L; i++) {
		const int s = (int)histo[i].size();
		if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
		else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
		else if (s > max3) { max3 = s; ind3 = i; }
	}
	if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
	else if (max3 < 0.1f * (float)max1) ind3 = -1;
}
}  // namespace

// cMultiCamSys_::WorldToCamHom_fast (src/cam_system_omni.cpp:92-133, flagMcMt branch: ptRot = MtMc_inv[c] * pt4) followed by
// cCamModelGeneral_::isPointInMirrorMask(u, v, 0) (src/cam_model_omni.cpp:163-178).  cv::Matx product: s = 0; s += a(i,k)*b(k).
// flags bit0 = inside the mirror mask (1 if no mask image is given and the rounded pixel is inside the bounds test), bit1 = z <= 0.
void orc_world_to_cam(const double* MtMc_inv /* nrCams x 16 row-major */, const orc_ocam* cams, const uint8_t* const* mirrorMasks /* level 0, tight, or NULL */,
                      const double* pts3 /* n x 3 */, const int* pcam, int n, double* uv /* n x 2 */, uint8_t* flags) {
	for (int i = 0; i < n; ++i) {
		const int c = pcam[i];
		const double* M = MtMc_inv + 16 * (size_t)c;
		const double pt4[4] = {pts3[3 * i], pts3[3 * i + 1], pts3[3 * i + 2], 1.0};
		double ptRot[4];
		for (int r = 0; r < 4; ++r) { double s = 0; for (int k = 0; k < 4; ++k) s += M[4 * r + k] * pt4[k]; ptRot[r] = s; }
		double u = 0.0, v = 0.0;
		orc_world2img(&cams[c], ptRot[0], ptRot[1], ptRot[2], &u, &v);
		uv[2 * i] = u; uv[2 * i + 1] = v;
		const int ur = cvRound_(u), vr = cvRound_(v);
		uint8_t fl = 0;
		if (!(ur >= cams[c].width || ur <= 0 || vr >= cams[c].height || vr <= 0)) {
			if (!mirrorMasks || !mirrorMasks[c] || mirrorMasks[c][(size_t)vr * cams[c].width + ur] > 0) fl |= 1;
		}
		if (ptRot[2] <= 0.0) fl |= 2;
		flags[i] = fl;
	}
}

// cORBmatcher::WindowSearch (src/cORBmatcher.cpp:326-473).  hasMP1[i1] = F1.mvpMapPoints[i1] && !isBad().  maxScaleLevel < 0 means INT_MAX.
// match21[i2] = i1 (vnMatches21 / the owner of vpMapPointMatches2[i2]) or -1.
int orc_window_search(const orc_frame_view* F1, const uint8_t* hasMP1, const orc_frame_view* F2, int windowSize, int minScaleLevel, int maxScaleLevel,
                      double nnratio, int dim, int havingMasks, int checkOri, int* match21) {
	int TH_HIGH, TH_LOW;
	orc_thresholds(dim, havingMasks, &TH_HIGH, &TH_LOW);
	FrameGrid G2(F2);
	int nmatches = 0;
	for (int i = 0; i < F2->n; ++i) match21[i] = -1;
	std::vector<int> rotHist[HISTO_LENGTH_];
	const double factor = 1.0f / HISTO_LENGTH_;
	const bool bMinLevel = minScaleLevel > 0;
	const bool bMaxLevel = maxScaleLevel >= 0 && maxScaleLevel < INT_MAX;
	for (int i1 = 0; i1 < F1->n; ++i1) {
		if (!hasMP1[i1]) continue;
		const orc_keypoint& kp1 = F1->keys[i1];
		const int level1 = kp1.octave;
		if (bMinLevel && level1 < minScaleLevel) continue;
		if (bMaxLevel && level1 > maxScaleLevel) continue;
		const int camIdx1 = F1->cam[i1];
		std::vector<size_t> vIndices2 = G2.GetFeaturesInArea(camIdx1, kp1.x, kp1.y, windowSize);
		if (vIndices2.empty()) continue;
		int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
		for (size_t k = 0; k < vIndices2.size(); ++k) {
			const size_t i2 = vIndices2[k];
			if (match21[i2] >= 0) continue;
			const int dist = dist_any(F1->desc, F1->mask, i1, F2->desc, F2->mask, (int)i2, dim, havingMasks);
			if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
			else if (dist < bestDist2) bestDist2 = dist;
		}
		if (bestDist <= bestDist2 * nnratio && bestDist <= TH_HIGH) {
			match21[bestIdx2] = i1;
			nmatches++;
			float rot = F1->keys[i1].angle - F2->keys[bestIdx2].angle;
			if (rot < 0.0) rot += 360.0f;
			int bin = cvRound_(rot * factor);
			if (bin == HISTO_LENGTH_) bin = 0;
			rotHist[bin].push_back(bestIdx2);
		}
	}
	if (checkOri) {
		int ind1 = -1, ind2 = -1, ind3 = -1;
		ComputeThreeMaxima(rotHist, HISTO_LENGTH_, ind1, ind2, ind3);
		for (int i = 0; i < HISTO_LENGTH_; ++i)
			if (i != ind1 && i != ind2 && i != ind3)
				for (size_t j = 0; j < rotHist[i].size(); ++j) { match21[rotHist[i][j]] = -1; --nmatches; }
	}
	return nmatches;
}

// cORBmatcher::SearchByProjection(F1, F2, windowSize, vpMapPointMatches2) (src/cORBmatcher.cpp:476-577).
// mp1[i1] / mp2[i2]: map point id (>= 0) or -1 for NULL; bad1[i1] = isBad().  uv / inMask: [n1][nrCams] projections of F1's map points
// into F2's cameras (WorldToCamHom_fast + isPointInMirrorMask, see orc_world_to_cam).  match21[i2] = i1 for the NEW matches, else -1.
int orc_search_by_projection_frames(const orc_frame_view* F1, const int* mp1, const uint8_t* bad1, const orc_frame_view* F2, const int* mp2,
                                    const double* uv, const uint8_t* inMask, int windowSize, double nnratio, int dim, int havingMasks, int* match21) {
	int TH_HIGH, TH_LOW;
	orc_thresholds(dim, havingMasks, &TH_HIGH, &TH_LOW);
	FrameGrid G2(F2);
	std::vector<char> taken(F2->n, 0);   // vpMapPointMatches2[i2] != NULL
	std::set<int> spMapPointsAlreadyFound;
	for (int i = 0; i < F2->n; ++i) { match21[i] = -1; taken[i] = mp2[i] >= 0; if (mp2[i] >= 0) spMapPointsAlreadyFound.insert(mp2[i]); }
	int nmatches = 0;
	std::set<int> mapPt_2_obs_idx;
	const int nrCams = F1->nrCams;
	for (int i1 = 0; i1 < F1->n; ++i1) {
		const int pMP1 = mp1[i1];
		if (pMP1 < 0) continue;
		if (bad1[i1] || spMapPointsAlreadyFound.count(pMP1)) continue;
		if (mapPt_2_obs_idx.count(pMP1) > 0) continue;
		mapPt_2_obs_idx.insert(pMP1);
		const int level1 = F1->keys[i1].octave;
		for (int c = 0; c < nrCams; ++c) {
			if (!inMask[(size_t)i1 * nrCams + c]) continue;
			const double u = uv[2 * ((size_t)i1 * nrCams + c)], v = uv[2 * ((size_t)i1 * nrCams + c) + 1];
			std::vector<size_t> vIndices2 = G2.GetFeaturesInArea(c, u, v, windowSize, level1, level1);
			if (vIndices2.empty()) continue;
			int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
			for (size_t k = 0; k < vIndices2.size(); ++k) {
				const size_t i2 = vIndices2[k];
				if (taken[i2]) continue;
				const int dist = dist_any(F1->desc, F1->mask, i1, F2->desc, F2->mask, (int)i2, dim, havingMasks);
				if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
				else if (dist < bestDist2) bestDist2 = dist;
			}
			if (static_cast<double>(bestDist) <= static_cast<double>(bestDist2) * nnratio && bestDist <= TH_HIGH) {
				taken[bestIdx2] = 1;
				match21[bestIdx2] = i1;
				++nmatches;
			}
		}
	}
	return nmatches;
}

// cORBmatcher::SearchForInitialization (src/cORBmatcher.cpp:579-726).  prevMatched: [n1][2] in/out (vbPrevMatched).  match12[i1] = i2 or -1.
int orc_search_for_initialization(const orc_frame_view* F1, const orc_frame_view* F2, double* prevMatched, int windowSize, double nnratio, int dim,
                                  int havingMasks, int checkOri, int* match12) {
	int TH_HIGH, TH_LOW;
	orc_thresholds(dim, havingMasks, &TH_HIGH, &TH_LOW);
	FrameGrid G2(F2);
	int nmatches = 0;
	for (int i = 0; i < F1->n; ++i) match12[i] = -1;
	std::vector<int> rotHist[HISTO_LENGTH_];
	const double factor = 1.0 / HISTO_LENGTH_;
	std::vector<int> vMatchedDistance(F2->n, INT_MAX), vnMatches21(F2->n, -1);
	for (int i1 = 0; i1 < F1->n; ++i1) {
		const orc_keypoint kp1 = F1->keys[i1];
		const int level1 = kp1.octave;
		const int camIdx1 = F1->cam[i1];
		std::vector<size_t> vIndices2 = G2.GetFeaturesInArea(camIdx1, prevMatched[2 * i1], prevMatched[2 * i1 + 1], windowSize, level1, level1);
		if (vIndices2.empty()) continue;
		int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
		for (size_t k = 0; k < vIndices2.size(); ++k) {
			const size_t i2 = vIndices2[k];
			const int dist = dist_any(F1->desc, F1->mask, i1, F2->desc, F2->mask, (int)i2, dim, havingMasks);
			if (vMatchedDistance[i2] <= dist) continue;
			if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
			else if (dist < bestDist2) bestDist2 = dist;
		}
		if (bestDist <= TH_LOW) {
			if (bestDist < (double)bestDist2 * nnratio) {
				if (vnMatches21[bestIdx2] >= 0) { match12[vnMatches21[bestIdx2]] = -1; nmatches--; }
				match12[i1] = bestIdx2;
				vnMatches21[bestIdx2] = i1;
				vMatchedDistance[bestIdx2] = bestDist;
				nmatches++;
				if (checkOri) {
					float rot = F1->keys[i1].angle - F2->keys[bestIdx2].angle;
					if (rot < 0.0) rot += 360.0f;
					int bin = (int)round(rot * factor);
					if (bin == HISTO_LENGTH_) bin = 0;
					rotHist[bin].push_back(i1);
				}
			}
		}
	}
	if (checkOri) {
		int ind1 = -1, ind2 = -1, ind3 = -1;
		ComputeThreeMaxima(rotHist, HISTO_LENGTH_, ind1, ind2, ind3);
		for (int i = 0; i < HISTO_LENGTH_; i++) {
			if (i == ind1 || i == ind2 || i == ind3) continue;
			for (size_t j = 0; j < rotHist[i].size(); j++) {
				const int idx1 = rotHist[i][j];
				if (match12[idx1] >= 0) { match12[idx1] = -1; --nmatches; }
			}
		}
	}
	for (int i1 = 0; i1 < F1->n; ++i1)
		if (match12[i1] >= 0) { prevMatched[2 * i1] = F2->keys[match12[i1]].x; prevMatched[2 * i1 + 1] = F2->keys[match12[i1]].y; }
	return nmatches;
}

// cORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th) (src/cORBmatcher.cpp:1990-2118).  lastMP[i] = map point non-NULL && !isBad();
// lastOutlier[i] = LastFrame.mvbOutlier[i]; uv / inMask: [nLast] projection of the map point into camera cam(i) of the CURRENT frame;
// curAssigned[i2] = CurrentFrame.mvpMapPoints[i2] != NULL (updated in place); matchCur[i2] = index in Last of the NEW match, else -1.
int orc_search_by_projection_last(const orc_frame_view* Cur, uint8_t* curAssigned, const orc_frame_view* Last, const uint8_t* lastMP, const uint8_t* lastOutlier,
                                  const double* uv, const uint8_t* inMask, const double* scaleFactors, double th, int dim, int havingMasks, int checkOri,
                                  int* matchCur) {
	int TH_HIGH, TH_LOW;
	orc_thresholds(dim, havingMasks, &TH_HIGH, &TH_LOW);
	FrameGrid G(Cur);
	int nmatches = 0;
	for (int i = 0; i < Cur->n; ++i) matchCur[i] = -1;
	std::vector<int> rotHist[HISTO_LENGTH_];
	const float factor = 1.0f / HISTO_LENGTH_;
	for (int i = 0; i < Last->n; ++i) {
		if (!lastMP[i]) continue;
		const int cam = Last->cam[i];
		if (lastOutlier[i]) continue;
		if (!inMask[i]) continue;
		const int nPredictedOctave = Last->keys[i].octave;
		const double radius = th * scaleFactors[nPredictedOctave];
		std::vector<size_t> vIndices2 = G.GetFeaturesInArea(cam, uv[2 * i], uv[2 * i + 1], radius, nPredictedOctave - 1, nPredictedOctave + 1);
		if (vIndices2.empty()) continue;
		int bestDist = INT_MAX, bestIdx2 = -1;
		for (size_t k = 0; k < vIndices2.size(); ++k) {
			const size_t i2 = vIndices2[k];
			if (curAssigned[i2]) continue;
			const int dist = dist_any(Last->desc, Last->mask, i, Cur->desc, Cur->mask, (int)i2, dim, havingMasks);
			if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
		}
		if (bestDist <= TH_HIGH) {
			curAssigned[bestIdx2] = 1;
			matchCur[bestIdx2] = i;
			++nmatches;
			if (checkOri) {
				float rot = Last->keys[i].angle - Cur->keys[bestIdx2].angle;
				if (rot < 0.0) rot += 360.0f;
				int bin = cvRoundf_(rot * factor);
				if (bin == HISTO_LENGTH_) bin = 0;
				rotHist[bin].push_back(bestIdx2);
			}
		}
	}
	if (checkOri) {
		int ind1 = -1, ind2 = -1, ind3 = -1;
		ComputeThreeMaxima(rotHist, HISTO_LENGTH_, ind1, ind2, ind3);
		for (int i = 0; i < HISTO_LENGTH_; i++)
			if (i != ind1 && i != ind2 && i != ind3)
				for (size_t j = 0; j < rotHist[i].size(); j++) { curAssigned[rotHist[i][j]] = 0; matchCur[rotHist[i][j]] = -1; --nmatches; }
	}
	return nmatches;
}

// The search loop shared by cORBmatcher::Fuse (src/cORBmatcher.cpp:1265-1719), SearchBySim3 (:1721-1988), SearchForTriangulationBetweenCameras
// (:1158-1263), SearchByProjection(pKF, Scw, ...) (:2265-2392) [skipTaken = 0] and SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th,
// ORBdist) (:2120-2263) [skipTaken = 1]: GetFeaturesInArea, level range, strict-'<' best, accept if bestDist <= maxDist.
// pdesc / pmask: one descriptor row per probe.  match[p] = feature or -1, dist[p] = best distance (INT_MAX for an empty window).
int orc_window_best(const double* x, const double* y, const double* radius, const int* minLevel, const int* maxLevel, const int* pcam, const uint8_t* pdesc,
                    const uint8_t* pmask, int nprobes, const orc_frame_view* F, uint8_t* assigned, int maxDist, int skipTaken, int dim, int havingMasks,
                    int* match, int* dist) {
	FrameGrid G(F);
	int nmatches = 0;
	for (int p = 0; p < nprobes; ++p) {
		match[p] = -1; dist[p] = INT_MAX;
		std::vector<size_t> vIndices = G.GetFeaturesInArea(pcam[p], x[p], y[p], radius[p], minLevel[p], maxLevel[p]);
		if (vIndices.empty()) continue;
		int bestDist = INT_MAX, bestIdx = -1;
		for (size_t k = 0; k < vIndices.size(); ++k) {
			const size_t idx = vIndices[k];
			if (skipTaken && assigned[idx]) continue;
			const int d = dist_any(pdesc, pmask, p, F->desc, F->mask, (int)idx, dim, havingMasks);
			if (d < bestDist) { bestDist = d; bestIdx = (int)idx; }
		}
		if (bestIdx < 0) continue;
		if (!skipTaken) dist[p] = bestDist;
		if (bestDist <= maxDist) {
			match[p] = bestIdx;
			if (skipTaken) { assigned[bestIdx] = 1; dist[p] = bestDist; }
			++nmatches;
		}
	}
	return nmatches;
}

// ---------------------------------------------------------------- "next" row 3: cMapPoint::ComputeDistinctiveDescriptors
// src/cMapPoint.cpp:294-382 with median() of include/misc.h:95-104 (nth_element at size/2).  desc/mask: the N observed descriptors
// in the order the reference's loop pushed them (its std::map is keyed by keyframe POINTER, so that order is the caller's to define).
// Quirks kept: the median of row i only covers j > i, the last row is never a candidate, N <= 2 -> index 0.  Returns BestIdx.
int orc_distinctive_descriptor(const uint8_t* desc, const uint8_t* mask, int N, int dim, int havingMasks) {
	if (N <= 0) return -1;
	std::vector<int> Distances((size_t)N * N, 0);
	for (int i = 0; i < N; ++i)
		for (int j = i + 1; j < N; ++j) {
			const int distij = dist_any(desc, mask, i, desc, mask, j, dim, havingMasks);
			Distances[(size_t)i * N + j] = distij;
			Distances[(size_t)j * N + i] = distij;
		}
	int BestMedian = INT_MAX, BestIdx = 0;
	if (N > 2) {
		for (int i = 0; i < N - 1; ++i) {
			std::vector<int> vDists;
			for (int j = i + 1; j < N; ++j) vDists.push_back(Distances[(size_t)i * N + j]);
			std::nth_element(vDists.begin(), vDists.begin() + vDists.size() / 2, vDists.end());
			const int medianV = vDists[vDists.size() / 2];
			if (medianV < BestMedian) { BestMedian = medianV; BestIdx = i; }
		}
	} else BestIdx = 0;
	return BestIdx;
}

// ---------------------------------------------------------------- mbCheckOrientation: the rotation-consistency filter of the searches
// Every search keeps a 30-bin histogram of rot = angle_first - angle_second (+360 if negative) of its accepted matches and, when
// mbCheckOrientation is set, drops the matches outside the three fullest bins (ComputeThreeMaxima, src/cORBmatcher.cpp:2394-2436).  The bin
// arithmetic differs from function to function (and "factor" is 1/HISTO_LENGTH, not HISTO_LENGTH/360 — bins 0..12 only; kept):
//   variant 0  SearchByBoW(KF,F) :191-194,272-282, SearchByProjection(Cur,Last) :1999,2084-2094, SearchByProjection(Cur,pKF,..) :2137,2222-2232
//              float factor = 1.0f/30;  bin = cvRound(rot*factor)               (float product)
//   variant 1  WindowSearch :339,428-438        double factor = 1.0f/30 (float division);  bin = cvRound(rot*factor)   (double product)
//   variant 2  SearchForInitialization :596,686-694   double factor = 1.0/30;  bin = round(rot*factor)      (half away from zero)
//   variant 3  SearchForTriangulationRaw :1011,1092-1101  double factor = 1.0/30; rot += 360.0 in DOUBLE;  bin = cvRound(rot*factor)
// slot s is matched to partner match[s] (or -1).  swapped = 0: rot = angle_slot[s] - angle_partner[match[s]]; 1: partner minus slot.
// accepted (optional): the partner at ACCEPTANCE time per slot, -1 if the slot never accepted — SearchForInitialization's histogram also counts
// acceptances that were stolen later (:686-694 pushes before the steal, :706-719 only clears live matches).  Returns the number removed.
static int rot_bin(int variant, float a_first, float a_second) {
	float rot = a_first - a_second;
	if (variant == 3) { if (rot < 0.0) rot += 360.0; }       // float = (double)rot + 360.0
	else if (rot < 0.0) rot += 360.0f;
	int bin;
	if (variant == 0) { const float factor = 1.0f / 30; bin = cvRoundf_(rot * factor); }
	else if (variant == 1) { const double factor = 1.0f / 30; bin = cvRound_(rot * factor); }
	else if (variant == 2) { const double factor = 1.0 / 30; bin = (int)round(rot * factor); }
	else { const double factor = 1.0 / 30; bin = cvRound_(rot * factor); }
	if (bin == 30) bin = 0;
	return bin;
}
int orc_rotation_consistency(int variant, const float* angle_slot, const float* angle_partner, const int* accepted, int* match, int n, int swapped) {
	std::vector<int> rotHist[30];
	for (int s = 0; s < n; ++s) {
		const int p = accepted ? accepted[s] : match[s];
		if (p < 0) continue;
		const float a = angle_slot[s], b = angle_partner[p];
		rotHist[rot_bin(variant, swapped ? b : a, swapped ? a : b)].push_back(s);
	}
	int ind1 = -1, ind2 = -1, ind3 = -1;
	ComputeThreeMaxima(rotHist, 30, ind1, ind2, ind3);
	int removed = 0;
	for (int i = 0; i < 30; ++i) {
		if (i == ind1 || i == ind2 || i == ind3) continue;
		for (size_t j = 0; j < rotHist[i].size(); ++j) {
			const int s = rotHist[i][j];
			if (match[s] >= 0) { match[s] = -1; ++removed; }
		}
	}
	return removed;
}

// ---------------------------------------------------------------- "next" row 4: cMultiFrame::ComputeBoW -> DBoW2 transform
// ThirdParty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259 (one feature down the tree; first strict minimum among a node's children in
// file order), FORB::distance ThirdParty/DBoW2/DBoW2/FORB.cpp:85-105 (32 bytes, the bit-trick popcount).  The vocabulary is given flat:
// node 0 = root, node_desc = 32 bytes per node, children of node i = child_idx[child_off[i] .. child_off[i+1]) in the order
// TemplatedVocabulary::load (:1573-1622) pushed them.  leaf[f] = the leaf node reached (word / weight are table look-ups on it),
// nid[f] = the node on the path at level L - levelsup (0 = root if that level is <= 0).
static inline int forb_distance(const uint8_t* a, const uint8_t* b) {
	const int32_t* pa = reinterpret_cast<const int32_t*>(a);
	const int32_t* pb = reinterpret_cast<const int32_t*>(b);
	int dist = 0;
	for (int i = 0; i < 8; i++, pa++, pb++) {
		unsigned int v = *pa ^ *pb;
		v = v - ((v >> 1) & 0x55555555);
		v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
		dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
	}
	return dist;
}

void orc_bow_transform(const uint8_t* node_desc, const int32_t* child_off, const int32_t* child_idx, int L, const uint8_t* desc, int n, int stride,
                       int levelsup, int32_t* leaf, int32_t* nid) {
	for (int f = 0; f < n; ++f) {
		const uint8_t* feature = desc + (size_t)f * stride;
		const int nid_level = L - levelsup;
		int nidv = 0;   // "if (nid_level <= 0 && nid != NULL) *nid = 0; // root"
		int final_id = 0, current_level = 0;
		do {
			++current_level;
			const int32_t* nodes = child_idx + child_off[final_id];
			const int cnt = child_off[final_id + 1] - child_off[final_id];
			final_id = nodes[0];
			double best_d = forb_distance(feature, node_desc + 32 * (size_t)final_id);
			for (int c = 1; c < cnt; ++c) {
				const int id = nodes[c];
				const double d = forb_distance(feature, node_desc + 32 * (size_t)id);
				if (d < best_d) { best_d = d; final_id = id; }
			}
			if (current_level == nid_level) nidv = final_id;
		} while (child_off[final_id + 1] - child_off[final_id] > 0);   // !isLeaf()
		leaf[f] = final_id;
		nid[f] = nidv;
	}
}

// cORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) WITH the vocabulary restriction (src/cORBmatcher.cpp:179-323): only features that
// share a FeatureVector node are compared; nodes ascending, keyframe features of a node in index order.  nodeKF / nodeF: node id per
// feature or -1 (stopped word: not in the FeatureVector).  matchF[j] = keyframe feature or -1.
int orc_search_kf_f_bow(const uint8_t* dKF, const uint8_t* mKF, const uint8_t* validKF, const int* nodeKF, int nKF, const uint8_t* dF, const uint8_t* mF,
                        const int* nodeF, int nF, int dim, int havingMasks, double nnratio, int* matchF) {
	int TH_HIGH, TH_LOW;
	orc_thresholds(dim, havingMasks, &TH_HIGH, &TH_LOW);
	std::map<int, std::vector<unsigned int> > vFeatVecKF, FeatVecF;   // DBoW2::FeatureVector (addFeature keeps index order)
	for (int i = 0; i < nKF; ++i) if (nodeKF[i] >= 0) vFeatVecKF[nodeKF[i]].push_back(i);
	for (int j = 0; j < nF; ++j) if (nodeF[j] >= 0) FeatVecF[nodeF[j]].push_back(j);
	for (int j = 0; j < nF; ++j) matchF[j] = -1;
	int nmatches = 0;
	std::map<int, std::vector<unsigned int> >::iterator KFit = vFeatVecKF.begin(), Fit = FeatVecF.begin();
	while (KFit != vFeatVecKF.end() && Fit != FeatVecF.end()) {
		if (KFit->first == Fit->first) {
			const std::vector<unsigned int>& vIndicesKF = KFit->second;
			const std::vector<unsigned int>& vIndicesF = Fit->second;
			for (size_t iKF = 0; iKF < vIndicesKF.size(); ++iKF) {
				const unsigned int realIdxKF = vIndicesKF[iKF];
				if (!validKF[realIdxKF]) continue;
				int bestDist1 = INT_MAX, bestIdxF = -1, bestDist2 = INT_MAX;
				for (size_t iF = 0; iF < vIndicesF.size(); ++iF) {
					const unsigned int realIdxF = vIndicesF[iF];
					if (matchF[realIdxF] >= 0) continue;
					const int dist = dist_any(dKF, mKF, (int)realIdxKF, dF, mF, (int)realIdxF, dim, havingMasks);
					if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
					else if (dist < bestDist2) bestDist2 = dist;
				}
				if (bestDist1 <= TH_LOW) {
					if (static_cast<double>(bestDist1) < nnratio * static_cast<double>(bestDist2)) { matchF[bestIdxF] = (int)realIdxKF; ++nmatches; }
				}
			}
			++KFit; ++Fit;
		} else if (KFit->first < Fit->first) KFit = vFeatVecKF.lower_bound(Fit->first);
		else Fit = FeatVecF.lower_bound(KFit->first);
	}
	return nmatches;
}

// ---------------------------------------------------------------- CPU baseline helper
int orc_num_threads(void) {
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

long orc_extract_many(const orc_params* p, int nimg, const uint8_t* const* imgs, int w, int h, int stride,
                      const uint8_t* const* masks, const orc_ocam* cams, int threads, orc_keypoint* kps, int cap, int* nkp,
                      uint8_t* desc, uint8_t* dmask) {
	long total = 0;
	if (threads < 1) threads = 1;
	// CPU-baseline fairness: keep the per-image level buffers in the malloc arenas instead of mmap/munmap per image — with
	// hundreds of threads the kernel's address-space lock otherwise serialises the whole run (measured: 44 -> 1360 ms/image).
	mallopt(M_MMAP_THRESHOLD, 1 << 30);
	mallopt(M_TRIM_THRESHOLD, 1 << 30);
	mallopt(M_ARENA_MAX, 1024);
#pragma omp parallel num_threads(threads) reduction(+ : total)
	{
		orc_extractor* e = orc_extractor_create(p);  // one stateful extractor per thread, like one per camera in cMultiFrame.cpp:128-139
#pragma omp for schedule(dynamic, 1)
		for (int i = 0; i < nimg; ++i) {
			int n = orc_extract(e, imgs[i], w, h, stride, masks ? masks[i] : nullptr, w, &cams[i], kps + (size_t)i * cap, cap,
			                    desc + (size_t)i * cap * p->descSize, dmask + (size_t)i * cap * p->descSize);
			nkp[i] = n;
			if (n > 0) total += n;
		}
		orc_extractor_destroy(e);
	}
	return total;
}

// The same pass with its results handed out (bench.py's in-run check compares EVERY multi-frame and pair of the CPU-baseline sample with the device's
// outputs): out_nkp[nimg], out_kps[nimg][cap], out_desc / out_mask[nimg][cap][descSize] (cap = nfeatures + 4 * nlevels), out_match[nframes][ncam * cap] = the
// m12 array of SearchByBoW(frame f, frame f - 1) over frame f's flattened keypoints (camera after camera, valid rows only; index into frame f - 1's flattened
// rows, -1 none; the tail of each row and row 0 stay -1).  Any of them may be null.
long orc_extract_match_many_out(const orc_params* p, int nframes, int ncam, const uint8_t* const* imgs, int w, int h, int stride,
                                const uint8_t* const* masks, const orc_ocam* cams, int threads, double nnratio, int* nmatch, double* seconds,
                                int* out_nkp, orc_keypoint* out_kps, uint8_t* out_desc, uint8_t* out_mask, int* out_match) {
	const int nimg = nframes * ncam, cap = p->nfeatures + 4 * p->nlevels, ds = p->descSize;
	std::vector<orc_keypoint> kps((size_t)nimg * cap);
	std::vector<int> nkp(nimg, 0);
	std::vector<uint8_t> desc((size_t)nimg * cap * ds), dmask((size_t)nimg * cap * ds);
#ifdef _OPENMP
	double t0 = omp_get_wtime();
#else
	double t0 = 0;
#endif
	long total = orc_extract_many(p, nimg, imgs, w, h, stride, masks, cams, threads, kps.data(), cap, nkp.data(), desc.data(), dmask.data());
#ifdef _OPENMP
	double t1 = omp_get_wtime();
#else
	double t1 = 0;
#endif
	// flatten every multi-frame (cameras concatenated, like cMultiFrame's mvKeys order)
	std::vector<std::vector<uint8_t> > fd(nframes), fm(nframes);
	for (int f = 0; f < nframes; ++f)
		for (int c = 0; c < ncam; ++c) {
			const int i = f * ncam + c;
			fd[f].insert(fd[f].end(), desc.begin() + (size_t)i * cap * ds, desc.begin() + ((size_t)i * cap + nkp[i]) * ds);
			fm[f].insert(fm[f].end(), dmask.begin() + (size_t)i * cap * ds, dmask.begin() + ((size_t)i * cap + nkp[i]) * ds);
		}
	nmatch[0] = 0;
	if (out_match) for (size_t i = 0; i < (size_t)nframes * ncam * cap; ++i) out_match[i] = -1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
	for (int f = 1; f < nframes; ++f) {
		const int n1 = (int)(fd[f].size() / ds), n2 = (int)(fd[f - 1].size() / ds);
		std::vector<uint8_t> v1(n1, 1), v2(n2, 1);
		std::vector<int> m12(n1 > 0 ? n1 : 1);
		nmatch[f] = orc_search_kf_kf(fd[f].data(), fm[f].data(), v1.data(), n1, fd[f - 1].data(), fm[f - 1].data(), v2.data(), n2, ds, p->learnMasks,
		                             nnratio, m12.data());
		if (out_match) memcpy(out_match + (size_t)f * ncam * cap, m12.data(), sizeof(int) * (size_t)n1);
	}
#ifdef _OPENMP
	double t2 = omp_get_wtime();
#else
	double t2 = 0;
#endif
	if (seconds) { seconds[0] = t1 - t0; seconds[1] = t2 - t1; }
	if (out_nkp) memcpy(out_nkp, nkp.data(), sizeof(int) * nimg);
	if (out_kps) memcpy(out_kps, kps.data(), sizeof(orc_keypoint) * kps.size());
	if (out_desc) memcpy(out_desc, desc.data(), desc.size());
	if (out_mask) memcpy(out_mask, dmask.data(), dmask.size());
	return total;
}

long orc_extract_match_many(const orc_params* p, int nframes, int ncam, const uint8_t* const* imgs, int w, int h, int stride,
                            const uint8_t* const* masks, const orc_ocam* cams, int threads, double nnratio, int* nmatch, double* seconds) {
	return orc_extract_match_many_out(p, nframes, ncam, imgs, w, h, stride, masks, cams, threads, nnratio, nmatch, seconds, nullptr, nullptr, nullptr, nullptr, nullptr);
}

}  // extern "C"

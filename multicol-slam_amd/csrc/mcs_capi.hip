// mcs_capi.hip — C ABI (include/mcs_c.h) of libmcs_hip.so: contexts, extractor construction (host-side tables in the
// reference's exact float/double steps), batch orchestration on one HIP stream, stage taps, matcher entry points.
// There is no CPU fallback in this library: every entry point needs a HIP device.
#include "mcs_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace mcs {
bool upload_describe_tables(const signed char* pattern);
void launch_single_distance(const uint8_t* a, const uint8_t* b, const uint8_t* ma, const uint8_t* mb, int dim, int* out, hipStream_t s);
void describe_host(int mode, int descSize, const signed char* pattern, const OcamDev* cam, int undistort, int level, float levelScale, int row, int col, float angle,
                   const HostLevel& L, uint8_t* desc, uint8_t* mask);   // mcs_tiefix.hip
void launch_tie_capture(const ExtractBuffers& b, const PyrDesc& hd, int nimg, int maxTies, uint8_t* devOut, hipStream_t s);   // mcs_tiefix.hip
void launch_selftest_fast_model(const OcamDev* cam, const double* tab, unsigned long long seed, int n, int width, int height, unsigned long long* maxDiff, hipStream_t s);
static const signed char kPattern[2048] = {
#include "learned_pattern_64_orb.inc"
};
}  // namespace mcs

using namespace mcs;

// Guard band of the fast descriptor pass: a coordinate closer than this to a rounding tie sends its keypoint to the exact pass.  2^-24 px: ~4 of
// 10 000 keypoints, and 30-40x above describe_fast_bound() for the Lafida cameras.
static constexpr double kDefaultGuardEps = 5.9604644775390625e-08;
// Default bands around the cvRound ties inside which an exact-arithmetic keypoint is recomputed on the host with the host's libm (mcs_tiefix.hip says why these)
static constexpr double kTieBandOrb = 1e-12, kTieBandDistorted = 1e-9;
static constexpr int kHostTieSlots = 64;   // entries of an extractor's own capture slot (host-kind batches); more listed keypoints take the whole-level download

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The fast pass's table of G(s) = rho(theta) / sqrt(s), theta = atan(p0 / sqrt(s)), for one camera (layout: mcs_common.h kG*), built in long double, with
// the bound on what its truncated Taylor rows leave out and the few magnitudes describe_fast_bound() needs.
//
// Row (e, k): s = 2^e (kappa + tau), kappa = 1 + (k + 1/2) / 2^kGM, |tau| <= 2^-(kGM+1); centre c = 2^e kappa, eps = tau / kappa = (s - c) / c.
//   N(eps)      = (1 + eps)^(-1/2) = sum b_j eps^j                        (1 / sqrt(s) = N / sqrt(c))
//   z(eps)      = z_c N(eps),  z_c = p0 / sqrt(c);   y = z - z_c
//   atan(z_c+y) = theta_c + sum_k A_k y^k,  A_k from 1 / ((1 + z_c^2) + 2 z_c y + y^2)
//   rho         = sum_k p_k (theta - theta_c)^k,  p_k = invP re-expanded at theta_c
//   G           = N * rho / sqrt(c), truncated at eps^kGDeg, stored in the kernel's variable tau (coefficient j divided by kappa^j).
// Tail (coefficient-wise majorants, "<<"):  |b_j| <= 1/2 (j >= 1), so y << |z_c| (eps/2) / (1 - eps);  the Taylor coefficients of atan at the real point z_c
// are (1/k) Im-parts of (z_c -+ i)^-k, at most R^-k / k with R = sqrt(1 + z_c^2), so theta - theta_c << (Y/R) / (1 - Y/R) = a eps / (1 - b eps) with
// q = |z_c| / R, a = q / 2, b = 1 + q / 2;  rho - p_0 << sum_k |p_k| (a eps / (1 - b eps))^k, whose eps^i coefficient is P_i = sum_k |p_k| a^k b^(i-k) C(i-1, k-1);
// G << c^(-1/2) N~ (|p_0| + P) with N~ = sum |b_j| eps^j:  W_j = c^(-1/2) (|p_0| |b_j| + sum_{i=1..j} P_i |b_(j-i)|).  The truncated tail of a row is at most
// sum_{j > kGDeg} W_j epsmax^j (summed to j = 80; the rest is below c^(-1/2) (|p_0| + sum |p_k|) 3 (3 epsmax)^81 / (1 - 3 epsmax) since P_i <= sum|p_k| (2b)^i, 2b <= 3).
// What the pixel coordinates see is sqrt(s) times that: tailU.
struct GTabInfo {
	double tailU = INFINITY;   // max over the rows of sqrt(s) * (truncated tail of the row)                       [pixels]
	double rhoB = 0;           // max sqrt(s) * sum_j |g_j| |eps|^j  (>= |rho| on the table's range)
	double dB = 0;             // max sqrt(s) * |s G'(s)|
	double lip = 0;            // max (|G| + 2 |s G'(s)|) >= both eigenvalues of d(x G, y G) / d(x, y)
	double inB = 0;            // max (|G| + 2 |s G'(s)|) * (sqrt(s) + 44): what one relative rounding of the rotated pattern point moves (x G, y G) by — a point
	                           // at distance n from the axis belongs to a keypoint within n + 22 of it, and the rotated offset is within 22: |terms| <= n + 44
	double seen = 0;           // largest |row polynomial - G| (times sqrt(s)) at the sample points checked against direct long double evaluation
	double f32U = INFINITY;    // max sqrt(s) * (what the FLOAT tail of the packed device rows adds: g4, g5, g6 and tau rounded to float, two float FMAs)  [pixels]
};

// The packed device form of a table (mcs_common.h kGDev*): per row [g0 g1 g2 g3] as doubles, [g4 g5 g6 0] as floats (round to nearest).
static void pack_g_table(const double* tab, uint8_t* out) {
	if (!MCS_G_PACKED) { memcpy(out, tab, (size_t)kGTabDoubles * sizeof(double)); return; }   // (the device reads the host's rows as they are)
	for (int r = 0; r < kGRows; ++r) {
		double* d = reinterpret_cast<double*>(out + (size_t)r * kGDevRowBytes);
		const double* g = tab + (size_t)r * kGRow;
		d[0] = g[0]; d[1] = g[1]; d[2] = g[2]; d[3] = g[3];
		float* f = reinterpret_cast<float*>(d + 4);
		f[0] = (float)g[4]; f[1] = (float)g[5]; f[2] = (float)g[6]; f[3] = 0.f;
	}
}

static GTabInfo build_g_table(const mcs_ocam& m, double* tab) {
	typedef long double LD;
	GTabInfo info;
	const int n = m.invP_deg, D = kGDeg, J = 80;
	if (n < 1 || n > MCS_MAX_POLY || !(std::fabs(m.p[0]) > 1e-300) || !std::isfinite(m.p[0])) return info;
	for (int i = 0; i < n; ++i) if (!std::isfinite(m.invP[i])) return info;
	LD binom[MCS_MAX_POLY][MCS_MAX_POLY];
	for (int i = 0; i < MCS_MAX_POLY; ++i)
		for (int k = 0; k <= i; ++k) binom[i][k] = (k == 0 || k == i) ? 1.0L : binom[i - 1][k - 1] + binom[i - 1][k];
	LD bj[J + 1];   // (1 + eps)^(-1/2)
	bj[0] = 1.0L;
	for (int j = 1; j <= J; ++j) bj[j] = bj[j - 1] * (-(LD)(2 * j - 1) / (LD)(2 * j));
	auto mul = [&](const LD* A, const LD* B, LD* C) {   // series product truncated at degree D
		LD T[D + 1];
		for (int i = 0; i <= D; ++i) { T[i] = 0; for (int k = 0; k <= i; ++k) T[i] += A[k] * B[i - k]; }
		for (int i = 0; i <= D; ++i) C[i] = T[i];
	};
	const LD p0 = (LD)m.p[0];
	auto direct = [&](LD sv) {   // G(s) itself
		const LD nn = sqrtl(sv), th = atanl(p0 / nn);
		LD r = 0;
		for (int i = n - 1; i >= 0; --i) r = r * th + (LD)m.invP[i];
		return r / nn;
	};
	double tailU = 0, rhoB = 0, dB = 0, lip = 0, inB = 0, seen = 0, f32U = 0;
	for (int e = kGE0; e < kGE1; ++e)
		for (int k = 0; k < (1 << kGM); ++k) {
			const LD kappa = 1.0L + ((LD)k + 0.5L) / (LD)(1 << kGM), c = ldexpl(kappa, e), sqc = sqrtl(c), zc = p0 / sqc, thc = atanl(zc);
			const double epsmax = (double)(ldexpl(1.0L, -(kGM + 1)) / kappa) * (1.0 + 1e-15);
			// y(eps), atan coefficients, composition
			LD y[D + 1], A[D + 1], f[D + 1];
			y[0] = 0;
			for (int j = 1; j <= D; ++j) y[j] = zc * bj[j];
			const LD d0 = 1.0L + zc * zc;
			f[0] = 1.0L / d0;
			for (int j = 1; j <= D; ++j) f[j] = -(2.0L * zc * f[j - 1] + (j >= 2 ? f[j - 2] : 0.0L)) / d0;
			for (int j = 1; j <= D; ++j) A[j] = f[j - 1] / (LD)j;
			LD Th[D + 1] = {0}, Yp[D + 1];
			for (int j = 0; j <= D; ++j) Yp[j] = y[j];
			for (int kk = 1; kk <= D; ++kk) {
				for (int j = 0; j <= D; ++j) Th[j] += A[kk] * Yp[j];
				mul(Yp, y, Yp);
			}
			LD pk[MCS_MAX_POLY];
			for (int kk = 0; kk < n; ++kk) {
				LD acc = 0.0L, pw = 1.0L;
				for (int j = kk; j < n; ++j) { acc += binom[j][kk] * (LD)m.invP[j] * pw; pw *= thc; }
				pk[kk] = acc;
			}
			LD H[D + 1] = {0};
			H[0] = pk[n - 1];
			for (int kk = n - 2; kk >= 0; --kk) { mul(H, Th, H); H[0] += pk[kk]; }
			LD Gs[D + 1], Nn[D + 1];
			for (int j = 0; j <= D; ++j) Nn[j] = bj[j];
			mul(Nn, H, Gs);
			double* row = tab + (size_t)((e - kGE0) * (1 << kGM) + k) * kGRow;
			LD kp = 1.0L;
			for (int j = 0; j <= D; ++j) {
				Gs[j] /= sqc;
				row[j] = (double)(Gs[j] / kp);
				kp *= kappa;
				if (!std::isfinite(row[j])) return info;
			}
			// the majorant series W_j and the row's tail
			const double q = (double)(fabsl(zc) / sqrtl(d0)), a = 0.5 * q, b = 1.0 + 0.5 * q;
			double apk[MCS_MAX_POLY], Spk = 0;
			for (int kk = 0; kk < n; ++kk) { apk[kk] = (double)fabsl(pk[kk]) * (1.0 + 1e-15); if (kk) Spk += apk[kk]; }
			double P[J + 1];
			P[0] = 0;
			for (int i = 1; i <= J; ++i) {
				double acc = 0, cb = 1.0;   // cb = C(i-1, kk-1)
				for (int kk = 1; kk < n && kk <= i; ++kk) { acc += apk[kk] * std::pow(a, kk) * std::pow(b, i - kk) * cb; cb = cb * (double)(i - kk) / (double)kk; }
				P[i] = acc;
			}
			const double isc = (double)(1.0L / sqc) * (1.0 + 1e-15);
			double tail = 0, tailD = 0, ep = 1.0;   // sum W_j eps^j and sum j W_j eps^(j-1) over j > D
			for (int j = 1; j <= J; ++j) {
				const double epm1 = ep;
				ep *= epsmax;
				if (j <= D) continue;
				double Wj = apk[0] * (double)fabsl(bj[j]);
				for (int i = 1; i <= j; ++i) Wj += P[i] * (double)fabsl(bj[j - i]);
				Wj *= isc;
				tail += Wj * ep;
				tailD += (double)j * Wj * epm1;
			}
			const double rest = isc * (apk[0] + Spk) * 3.0 * std::pow(3.0 * epsmax, J + 1) / (1.0 - 3.0 * epsmax);
			tail = (tail + rest) * 1.01;                        // the sums above are in rounded arithmetic
			tailD = (tailD + (J + 2) * rest / epsmax) * 1.01;
			if (!std::isfinite(tail) || !std::isfinite(tailD)) return info;
			double gabs = 0, gder = 0;
			ep = 1.0;
			for (int j = 0; j <= D; ++j) {
				gabs += (double)fabsl(Gs[j]) * ep;
				if (j + 1 <= D) gder += (double)(j + 1) * (double)fabsl(Gs[j + 1]) * ep;
				ep *= epsmax;
			}
			gabs += tail;
			gder = (gder + tailD) * (1.0 + epsmax);             // |s G'(s)| = |(1 + eps) dG/d eps|
			const double sq = (double)sqc * std::sqrt(1.0 + epsmax) * (1.0 + 1e-15);
			tailU = std::max(tailU, sq * tail);
			rhoB = std::max(rhoB, sq * gabs);
			dB = std::max(dB, sq * gder);
			lip = std::max(lip, gabs + 2.0 * gder);
			inB = std::max(inB, (gabs + 2.0 * gder) * (sq + 44.0));
			// The float tail of the packed rows: T = g4 + g5 tau + g6 tau^2 with g_j and tau rounded to float and two float FMAs — the g6 term meets five
			// relative perturbations of 2^-24 (its coefficient, tau twice, both FMAs), g5 four, g4 two: |T_float - T| <= 6 * 2^-24 * sum |g_j| |tau|^(j-4) (the 6
			// and the 1.01 cover the second-order terms); T enters G as T tau^4, the coordinates as sqrt(s) G.  |tau| <= 2^-(kGM+1) in the kernel's variable.
			// Coefficients that do not fit a float make the table unusable; below the smallest normal float they are worth less than 1e-37 anyway (added).
			{
				const double tm = std::ldexp(1.0, -(kGM + 1));
				double t4 = 0, tp = tm * tm * tm * tm;
				for (int j = 4; j <= D; ++j) {
					if (!(std::fabs(row[j]) < 3.0e38) || !std::isfinite((double)(float)row[j])) return info;
					t4 += std::fabs(row[j]) * tp;
					tp *= tm;
				}
				f32U = std::max(f32U, sq * (6.0 * 1.01 * std::ldexp(1.0, -24) * t4 + 1e-37));
			}
			// sanity: the row against G itself at both ends of the bin and in the middle
			for (int t = -2; t <= 2; ++t) {
				const LD tau = (LD)t * ldexpl(1.0L, -(kGM + 2)) * (1.0L - 1e-9L), sv = ldexpl(kappa + tau, e);
				LD pv = 0;
				for (int j = D; j >= 0; --j) pv = pv * tau + (LD)row[j];
				seen = std::max(seen, (double)(fabsl(pv - direct(sv)) * sqrtl(sv)));
			}
		}
	info.f32U = f32U * 1.01;
	info.tailU = tailU; info.rhoB = rhoB * 1.01; info.dB = dB * 1.01; info.lip = lip * 1.01; info.inB = inB * 1.01; info.seen = seen;
	if (!(seen <= tailU + 64 * 1.1102230246251565e-16 * rhoB)) info.tailU = INFINITY;   // the rows must reproduce G within the bound they claim (plus their own rounding to double)
	return info;
}

// Worst-case |(fast coordinate - fast mean) - (reference coordinate - reference mean)| for one camera and npoints pattern points.
// u = 2^-53.  Both arithmetics evaluate the same real function F(X, Y) = affine(X G(s), Y G(s)), s = X^2 + Y^2, of the same real rotated pattern point
// (X, Y) = R(angle) (ptx, pty) + undistorted keypoint; each differs from it by its own rounding:
//   inputs  X = ptx ax - pty ay + ukx: reference 3 roundings, fast 2 (two FMAs), each relative to |ptx ax| + |pty ay| + |ukx| <= n + 44 for a point at distance
//           n from the axis (the keypoint lies within n + 22 of it, the rotated offset within 22).  F moves by at most aff * (|G| + 2 |s G'|) per unit of
//           either input; the product with n + 44 is maximised over the table's rows (inB): near the axis G ~ rho(axis) / n is large where n + 44 is small.
//   fast    s: 2 roundings (2.01 u relative, G moves by |s G'(s)| per relative unit: dB);  the row: truncated tail (tailU) + coefficients rounded to double +
//           6 FMAs (13 u of sum |g_j| |eps|^j: rhoB; the packed rows run 4 of them in double) + the float tail g4 .. g6 of the packed rows (f32U);  x G, y G: u
//           each;  the affine map without the principal point (it cancels against the mean): 2 more
//   ref     atan's argument and atan itself (12 u S' generously: ocml / glibc stay within 2 ulp) + 24 roundings of the Horner chain, 2 divisions, 2 products
//           (96 u S) + the affine map with the principal point (8 u (|u0| + |v0|));   S = sum |invP_i| (pi/2)^i,  S' = sum i |invP_i| (pi/2)^(i-1)
//   mean    the same per-point bound, plus the order of the sum: reference npoints - 1 sequential additions and a division, fast 2 NB - 1 per lane, 6 shuffle
//           levels and a product -> (npoints + 2 NB + 7) u Umax, Umax = 16384 + 4096 (enforced by the kernel)
//   minus   reference: one rounding of a value below 8192;  fast: the subtraction happens in fixed point (coordinate + (1.5 * 2^20 + 0.5 - mean), two
//           roundings of 2^-33 each)
// Returns +inf for a camera the fast arithmetic cannot serve (p0 = 0, non-finite coefficients, a table that fails its own check).
static double describe_fast_bound(const mcs_ocam& m, int npoints, const GTabInfo& g) {
	const double u = 1.1102230246251565e-16, hp = 1.5707963267948966;
	double S = 0, Sp = 0, pw = 1.0;
	for (int i = 0; i < m.invP_deg; ++i) {
		if (!std::isfinite(m.invP[i])) return INFINITY;
		S += std::fabs(m.invP[i]) * pw;
		if (i + 1 < m.invP_deg) Sp += (i + 1) * std::fabs(m.invP[i + 1]) * pw;
		pw *= hp;
	}
	if (!(std::fabs(m.p[0]) > 1e-300) || !std::isfinite(m.p[0]) || !std::isfinite(g.tailU) || !std::isfinite(g.rhoB) || !std::isfinite(g.dB) || !std::isfinite(g.lip) || !std::isfinite(g.inB) || !std::isfinite(g.f32U)) return INFINITY;
	const double aff = 1.0 + std::fabs(m.c) + std::fabs(m.d) + std::fabs(m.e), pp = 8 * u * (std::fabs(m.u0) + std::fabs(m.v0));
	const int nb = npoints / 128;
	const double inputs = aff * g.inB * (2 * (2 + 3) * u * 1.01);
	const double fast = aff * (g.tailU + 16 * u * g.rhoB + 2.01 * u * g.dB + (MCS_G_PACKED ? g.f32U : 0.0));
	const double ref = aff * (12 * u * Sp + 96 * u * S) + pp;
	const double point = inputs + fast + ref;
	const double total = 2 * point + (npoints + 2 * nb + 7) * u * 20480.0 + 2 * u * 8192.0 + 2.3283064365386963e-10 * 1.001;
	return std::isfinite(total) ? total : INFINITY;
}

std::string& mcs_err() { static thread_local std::string e; return e; }

#include "mcs_host.h"
#include "mcs_tiecap.h"

struct mcs_extractor {
	mcs_ctx* ctx = nullptr;
	mcs_extractor_params params{};
	int maxBatch = 0;
	PyrDesc hd{};
	std::vector<CellInfo> cells;
	PyrDesc* d_desc = nullptr;
	CellInfo* d_cells = nullptr;
	ResizeTap* d_taps = nullptr;
	short* d_maskMap = nullptr;
	uint8_t *d_pyr = nullptr, *d_blur = nullptr;
	uint32_t *d_slots = nullptr, *d_dense = nullptr, *d_sel = nullptr;
	float* d_selAngle = nullptr;
	unsigned short* d_knode = nullptr;
	int *d_cellCount = nullptr, *d_denseCount = nullptr, *d_selCount = nullptr, *d_status = nullptr;
	OcamDev* d_cams = nullptr;
	std::vector<OcamDev> h_cams;
	// descriptor passes (mcs_describe.hip): fallback list of the fast pass, its running total, the guard band
	int* d_fbCount = nullptr; uint32_t *d_fbList = nullptr, *d_preList = nullptr; unsigned long long* d_fbStats = nullptr; unsigned long long* d_tieMin = nullptr; void* d_aux = nullptr;   // d_fbCount: [0] fallback list, [1] pre-list
	int describeMode = 0; double guardEps = kDefaultGuardEps;
	// rounding ties of the exact arithmetic (mcs_tiefix.hip): the batch's listed keypoints, the band (0 = the mode's default, < 0 = list nothing), totals
	uint32_t* d_tieList = nullptr; double tieBand = 0.0; unsigned long long tieFixed = 0;
	// pipelined enforcement (mcs_extractor_set_tie_capture): a ring of page-locked capture slots, one per device-kind batch in flight
	struct TieSlot { uint8_t* host = nullptr; uint8_t* dev = nullptr; hipEvent_t ev = nullptr; ExtractBuffers b{}; int nimg = 0; std::vector<OcamDev> cams; long long seq = -1; bool patched = false; };
	std::vector<TieSlot> tieRing; int tieMax = 0; long long batchSeq = 0; hipStream_t patchStream = nullptr; uint8_t* h_patchRows = nullptr;
	unsigned long long tieWindowMisses = 0;
	TieSlot hostTie;   // host-kind batches: the extractor's own capture slot (kHostTieSlots entries)
	// G(s) tables of the cameras seen so far (a rig has a handful), and the batch's distinct tables as the fast pass reads them
	struct CamFast { OcamDev key; GTabInfo info; std::vector<double> tab; };
	std::vector<CamFast> camCache;
	double* d_gTab = nullptr;
	// host-kind input staging: the caller's image / mask block as it lies in host memory (same pitch and stride), grown on demand
	uint8_t *d_inImg = nullptr, *d_inMask = nullptr; size_t inImgCap = 0, inMaskCap = 0;
	// host-kind output staging
	int* d_nkp = nullptr; mcs_keypoint* d_kps = nullptr; uint8_t *d_odesc = nullptr, *d_omask = nullptr; double* d_rays = nullptr;
	ExtractBuffers last{};
	int lastN = 0;
	// mirror masks kept on the device (mcs_extractor_set_masks): tight rows, one after the other
	uint8_t* d_resMask = nullptr; int resMaskN = 0;
	int* h_status = nullptr;   // page-locked: device status + tie count of a host-kind batch, written by k_extract_out
	// hipGraphs of the in-order launch sequence (small batches, extract_impl): keyed by the kernels' arguments
	struct Graph { ExtractBuffers key; int nimg; hipGraphExec_t exec; };
	std::vector<Graph> graphs;
	long graphHits = 0, graphMisses = 0;   // a caller that rotates through more argument sets than the cache holds would re-capture on every call: replay is given up then
};

extern "C" {

const char* mcs_last_error(void) { return mcs_err().c_str(); }
int mcs_abi_version(void) { return MCS_ABI_VERSION; }

int mcs_device_count(int* n) {
	if (!n) return fail(MCS_ERR_INVALID, "null");
	HIPCHK(hipGetDeviceCount(n));
	return MCS_OK;
}

int mcs_ctx_create(int device, void* hip_stream, mcs_ctx** out) {
	if (!out) return fail(MCS_ERR_INVALID, "out is null");
	int n = 0;
	HIPCHK(hipGetDeviceCount(&n));
	if (n <= 0 || device < 0 || device >= n) return fail(MCS_ERR_HIP, "no usable HIP device (libmcs_hip has no CPU fallback)");
	HIPCHK(hipSetDevice(device));
	mcs_ctx* c = new mcs_ctx;
	c->device = device;
	if (hip_stream) c->stream = (hipStream_t)hip_stream;
	else { HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->ownStream = true; }
	HIPCHK(hipMalloc(&c->dscalar, 64));
	if (getenv("MCS_NO_OVERLAP") == nullptr) {
		// (stream priorities — resize chain urgent, deferred matcher least urgent or most urgent, and every other combination — change nothing measurable)
		HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
		// A/B: MCS_MATCH_CU_MASK=<hex word> confines the deferred matcher's stream (MCS_GREEDY_CU_MASK: the greedy pass's) to the CUs whose bit is set in the word,
		// repeated over the chip's CUs — its workgroups (3 waves per SIMD at 168 registers) otherwise leave no room for anybody else on the CUs they hold
		auto masked = [&](hipStream_t* st, const char* var) -> hipError_t {
			const char* m = getenv(var);
			if (!m || !*m) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
			char* end = nullptr;
			const unsigned long v = strtoul(m, &end, 16);
			if (end == m || *end != '\0' || v == 0 || v > 0xFFFFFFFFul) { fprintf(stderr, "mcs: %s=\"%s\" is not a non-zero 32-bit hex word\n", var, m); return hipErrorInvalidValue; }
			// NOTE: hipExtStreamCreateWithCUMask makes a BLOCKING stream (it synchronises implicitly with the null stream), unlike the hipStreamNonBlocking ones of
			// the default path: an A/B through these switches changes more than CU placement
			uint32_t words[16];
			for (auto& w : words) w = (uint32_t)v;
			hipDeviceProp_t prop;
			if (hipGetDeviceProperties(&prop, device) != hipSuccess) return hipErrorInvalidValue;
			return hipExtStreamCreateWithCUMask(st, (uint32_t)((prop.multiProcessorCount + 31) / 32), words);
		};
		HIPCHK(masked(&c->side2, "MCS_MATCH_CU_MASK"));
		HIPCHK(masked(&c->side3, "MCS_GREEDY_CU_MASK"));
		HIPCHK(hipEventCreateWithFlags(&c->evLists, hipEventDisableTiming));
		for (int i = 0; i < 2; ++i) HIPCHK(hipEventCreateWithFlags(&c->evGreedyBuf[i], hipEventDisableTiming));

		HIPCHK(hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&c->evPyr1, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&c->evPyr, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&c->evBlur, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&c->evMatch, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&c->evGreedy, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&c->evDescFork, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&c->evDescJoin, hipEventDisableTiming));

		for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreateWithFlags(&c->evSearch[i], hipEventDisableTiming));
	}
	*out = c;
	return MCS_OK;
}

// make the context's main stream wait for everything queued on the side stream (outputs of the searches are written there)
int mcs_ctx_join(mcs_ctx* c) {
	if (!c) return fail(MCS_ERR_INVALID, "null ctx");
	if (c->side && c->greedyPending) { HIPCHK(hipStreamWaitEvent(c->stream, c->evGreedy, 0)); c->greedyPending = false; }
	return MCS_OK;
}

int mcs_ctx_set_async_search(mcs_ctx* c, int on) {
	if (!c) return fail(MCS_ERR_INVALID, "null ctx");
	HIPCHK(hipStreamSynchronize(c->stream));
	if (c->side2) { HIPCHK(hipStreamSynchronize(c->side2)); HIPCHK(hipStreamSynchronize(c->side3)); }
	const bool was = c->asyncSearch;
	c->asyncSearch = on != 0 && c->side2 != nullptr;
	if (was != c->asyncSearch) { c->upload = nullptr; c->lastResultStream = nullptr; }   // the transfer stream is re-scored for the new mode (mcs_copy.hip); the result stream follows the next search
	return MCS_OK;
}

int mcs_ctx_search_fence(mcs_ctx* c, int lag) {
	if (!c || lag < 0 || lag > 2) return fail(MCS_ERR_INVALID, "lag must be 0, 1 or 2");
	if (!c->asyncSearch) return lag == 0 ? mcs_ctx_join(c) : MCS_OK;   // in-order searches: only the latest greedy pass can still be running
	const long long want = c->searchSeq - 1 - lag;
	if (want >= 0) HIPCHK(hipStreamWaitEvent(c->stream, c->evSearch[want & 3], 0));
	if (lag == 0) c->greedyPending = false;
	return MCS_OK;
}

int mcs_ctx_destroy(mcs_ctx* c) {
	if (!c) return MCS_OK;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	while (!c->extractors.empty()) (void)mcs_extractor_destroy(c->extractors.back());   // an extractor must not outlive the stream it runs on
	for (auto& kv : c->timers) { if (kv.second.a) { (void)hipEventDestroy(kv.second.a); (void)hipEventDestroy(kv.second.b); } }
	(void)hipFree(c->partial); (void)hipFree(c->partialCount); (void)hipFree(c->stage); (void)hipFree(c->dscalar);
	(void)hipFree(c->topKeys); (void)hipFree(c->topKeys2); (void)hipFree(c->topCnt); (void)hipFree(c->exA); (void)hipFree(c->exW); (void)hipFree(c->exRows); (void)hipFree(c->stageOut); (void)hipFree(c->arena); if (c->pinned) (void)hipHostFree(c->pinned);
	if (c->side) {
		(void)hipStreamSynchronize(c->side);
		(void)hipStreamSynchronize(c->side2);
		(void)hipStreamDestroy(c->side2);
		(void)hipStreamSynchronize(c->side3); (void)hipStreamDestroy(c->side3);
		(void)hipEventDestroy(c->evLists); (void)hipEventDestroy(c->evGreedyBuf[0]); (void)hipEventDestroy(c->evGreedyBuf[1]);
		for (int i = 0; i < 4; ++i) if (c->evSearch[i]) (void)hipEventDestroy(c->evSearch[i]);
		(void)hipEventDestroy(c->evFork); (void)hipEventDestroy(c->evPyr1); (void)hipEventDestroy(c->evPyr); (void)hipEventDestroy(c->evBlur); (void)hipEventDestroy(c->evMatch); (void)hipEventDestroy(c->evGreedy);
		(void)hipEventDestroy(c->evDescFork); (void)hipEventDestroy(c->evDescJoin);
		(void)hipStreamDestroy(c->side);
	}
	for (hipStream_t ps : c->probed) { (void)hipStreamSynchronize(ps); (void)hipStreamDestroy(ps); }
	if (c->ownStream) (void)hipStreamDestroy(c->stream);
	delete c;
	return MCS_OK;
}

int mcs_ctx_synchronize(mcs_ctx* c) {
	if (!c) return fail(MCS_ERR_INVALID, "null ctx");
	HIPCHK(hipStreamSynchronize(c->stream));
	if (c->side) { HIPCHK(hipStreamSynchronize(c->side)); HIPCHK(hipStreamSynchronize(c->side2)); HIPCHK(hipStreamSynchronize(c->side3)); c->greedyPending = false; }
	return MCS_OK;
}

int mcs_ctx_enable_timing(mcs_ctx* c, int on) {
	if (!c) return fail(MCS_ERR_INVALID, "null ctx");
	if (c->timing != (on != 0)) c->lastResultStream = nullptr;   // per-kernel timing runs the searches in order on the main stream: what the last search recorded belongs to the other mode
	c->timing = on != 0;
	return MCS_OK;
}

int mcs_ctx_kernel_ms(mcs_ctx* c, const char* name, float* ms) {
	if (!c || !name || !ms) return fail(MCS_ERR_INVALID, "null");
	auto it = c->timers.find(name);
	if (it == c->timers.end() || !it->second.used) return fail(MCS_ERR_INVALID, std::string("no timing for ") + name);
	HIPCHK(hipEventSynchronize(it->second.b));
	HIPCHK(hipEventElapsedTime(ms, it->second.a, it->second.b));
	return MCS_OK;
}

// ------------------------------------------------------------------------------------------------ extractor
static void build_umax(int* umax) {   // reference src/mdBRIEFextractorOct.cpp:187-202
	int v, v0, vmax = cvFloor_(kHalfPatch * sqrt(2.f) / 2 + 1);
	int vmin = (int)ceil(kHalfPatch * sqrt(2.f) / 2);
	const double hp2 = kHalfPatch * kHalfPatch;
	for (v = 0; v <= kHalfPatch; ++v) umax[v] = 0;
	for (v = 0; v <= vmax; ++v) umax[v] = cvRound_(sqrt(hp2 - v * v));
	for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
		while (umax[v0] == umax[v0 + 1]) ++v0;
		umax[v] = v0;
		++v0;
	}
}

int mcs_extractor_create(mcs_ctx* ctx, const mcs_extractor_params* p, int width, int height, int max_batch, mcs_extractor** out) {
	if (!ctx || !p || !out) return fail(MCS_ERR_INVALID, "null argument");
	if (p->nlevels < 1 || p->nlevels > MCS_MAX_LEVELS) return fail(MCS_ERR_INVALID, "nlevels out of range");
	if (p->descSize != 16 && p->descSize != 32 && p->descSize != 64) return fail(MCS_ERR_INVALID, "descSize must be 16, 32 or 64");
	if (p->useAgast) {
		if (p->fastAgastType < 0 || p->fastAgastType > 3) return fail(MCS_ERR_INVALID, "fastAgastType must be 0 (AGAST_5_8), 1 (AGAST_7_12d), 2 (AGAST_7_12s) or 3 (OAST_9_16) with useAgast");
		// the device stores "no corner" as score 0 and a corner's score is >= the threshold; cv::AGAST's bisection never returns more than 254
		if (p->fastThreshold < 1 || p->fastThreshold > 254) return fail(MCS_ERR_UNSUPPORTED, "useAgast needs 1 <= fastThreshold <= 254");
	} else if (p->fastAgastType < 0 || p->fastAgastType > 2) return fail(MCS_ERR_INVALID, "fastAgastType must be 0 (TYPE_5_8), 1 (TYPE_7_12) or 2 (TYPE_9_16)");
	if (!(p->scaleFactor > 1.0f)) return fail(MCS_ERR_INVALID, "scaleFactor must be > 1");
	if (max_batch < 1 || width < 1 || height < 1 || p->nfeatures < 1) return fail(MCS_ERR_INVALID, "bad size");
	HIPCHK(hipSetDevice(ctx->device));

	mcs_extractor* e = new mcs_extractor;
	e->ctx = ctx; e->params = *p; e->maxBatch = max_batch;
	PyrDesc& hd = e->hd;
	const int nl = p->nlevels;
	hd.nlevels = nl; hd.width = width; hd.height = height;
	hd.fastThreshold = std::min(std::max(p->fastThreshold, 0), 255);
	hd.fastRing = p->fastAgastType == 2 ? 16 : (p->fastAgastType == 1 ? 12 : 8);
	hd.agast = p->useAgast ? p->fastAgastType : -1;
	// border of the detector inside a cell view: cv::FAST keeps 3 pixels for every ring, cv::AGAST the ring's radius (1 / 3 / 2 / 3)
	const int detB = !p->useAgast ? 3 : (p->fastAgastType == 0 ? 1 : (p->fastAgastType == 2 ? 2 : 3));
	hd.descSize = p->descSize; hd.npoints = 2 * 8 * p->descSize;
	hd.mode = p->learnMasks ? 2 : (p->do_dBrief ? 1 : 0);
	hd.undistort = p->do_dBrief ? 1 : 0;

	// scale tables in double from the FLOAT scale factor (ctor :153-179)
	std::vector<double> sc(nl, 1.0), inv(nl, 1.0);
	const double scaleFactor = p->scaleFactor;
	for (int i = 1; i < nl; i++) sc[i] = sc[i - 1] * scaleFactor;
	const double invScaleFactor = 1.0 / scaleFactor;
	for (int i = 1; i < nl; i++) inv[i] = inv[i - 1] * invScaleFactor;
	std::vector<int> nfeat(nl);
	{
		double factor = (1.0 / scaleFactor);
		double nDesired = p->nfeatures * (1 - factor) / (1 - pow(factor, nl));
		int sum = 0;
		for (int level = 0; level < nl - 1; level++) { nfeat[level] = cvRound_(nDesired); sum += nfeat[level]; nDesired *= factor; }
		nfeat[nl - 1] = std::max(p->nfeatures - sum, 0);
	}

	std::vector<ResizeTap> taps;
	std::vector<short> maps;
	int off = 0, cellBase = 0, slotBase = 0, denseBase = 0, selBase = 0;
	for (int l = 0; l < nl; ++l) {
		LevelInfo& L = hd.lv[l];
		L.w = cvRound_((double)width * inv[l]);
		L.h = cvRound_((double)height * inv[l]);
		L.stride = (L.w + 63) / 64 * 64;
		L.off = off;
		off += L.stride * L.h;
		const int minB = kMinBorder, maxBX = L.w - kEdge + 3, maxBY = L.h - kEdge + 3;
		const double wd = (maxBX - minB), ht = (maxBY - minB);
		L.nCols = (int)(wd / (double)kCellW);
		L.nRows = (int)(ht / (double)kCellW);
		if (wd < 1 || ht < 1 || (l == 0 && (L.nCols < 1 || L.nRows < 1)) || L.w - 2 * kMinBorder >= 4096 || L.h - 2 * kMinBorder >= 4096) {
			delete e;
			return fail(MCS_ERR_UNSUPPORTED, "image too small for the requested number of levels (a level is narrower than its 44-px border, or level 0 has no 30-px FAST cell) or too large");
		}
		// A level whose inner region is less than one 30-px cell high or wide: the reference's cell loops do not run (nRows or nCols = 0, :886-887), the oct-tree gets no
		// keys and the level contributes nothing — an EMPTY level here (no cells, no slots); everything else about it (resize, blur, the feature budget) stays
		const bool emptyLevel = L.nCols < 1 || L.nRows < 1;
		if (emptyLevel) { L.nCols = L.nRows = 0; L.wCell = L.hCell = 0; L.capc = 0; }
		else {
			L.wCell = (int)ceil(wd / L.nCols);
			L.hCell = (int)ceil(ht / L.nRows);
			L.capc = p->useAgast ? ((L.wCell + 6 - 2 * detB) * (L.hCell + 6 - 2 * detB) + 1) / 2 : ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);
		}
		L.cellBase = cellBase; L.slotBase = slotBase;
		for (int i = 0; i < L.nRows; i++)
			for (int j = 0; j < L.nCols; j++) {
				CellInfo c{};
				c.level = (short)l;
				const double iniY = minB + i * L.hCell, iniX = minB + j * L.wCell;
				double maxY = iniY + L.hCell + 6, maxX = iniX + L.wCell + 6;
				bool skip = (iniY >= maxBY - 3) || (iniX >= maxBX - 6);   // :897,906
				if (maxY > maxBY) maxY = maxBY;
				if (maxX > maxBX) maxX = maxBX;
				c.x0 = (short)(iniX + detB); c.y0 = (short)(iniY + detB);
				c.cw = skip ? 0 : (short)std::max(0, (int)maxX - (int)iniX - 2 * detB);
				c.ch = skip ? 0 : (short)std::max(0, (int)maxY - (int)iniY - 2 * detB);
				c.slot = slotBase + (i * L.nCols + j) * L.capc;
				{   // k_fast_cells: tile row = cw + 4 + 3 bytes in dwords, 4-pixel groups per row
					const int ndw = (c.cw + 4 + 3 + 3) >> 2, gpr = std::max((c.cw + 3) >> 2, 1);
					c.rowM = (65536 + ndw - 1) / ndw; c.grpM = (65536 + gpr - 1) / gpr;
				}
				e->cells.push_back(c);
			}
		cellBase += L.nCols * L.nRows;
		slotBase += L.nCols * L.nRows * L.capc;
		L.denseBase = denseBase; L.denseCap = L.nCols * L.nRows * L.capc;
		denseBase += L.denseCap;
		L.nfeat = nfeat[l];
		L.selBase = selBase; L.selCap = std::max(L.nfeat + 3, 4 * kMaxRoots);
		selBase += L.selCap;
		// oct-tree roots (:641-661)
		L.nIni = cvRound_(wd / ht);
		if (L.nIni > kMaxRoots || L.nfeat + 3 > kMaxNodes) {
			delete e;
			return fail(MCS_ERR_UNSUPPORTED, "aspect ratio / features per level outside the oct-tree kernel's capacity (nIni <= 32, nfeatures_level+3 <= 2048)");
		}
		// nIni = 0 (a level more than twice as tall as wide, e.g. the small top levels of a portrait image): the reference divides by it and, as soon as the level has a
		// candidate, indexes an empty root vector (:641-661, undefined).  Its only defined outcome — a level without candidates — is "no keypoint on this level", and that
		// is what such a level yields here whatever it holds (tests/test_gpu_extract.py::test_portrait_levels_without_an_octree_root).
		if (L.nIni < 1) L.nIni = 0;
		L.hX = L.nIni > 0 ? wd / L.nIni : wd;
		for (int i = 0; i <= L.nIni; ++i) L.rootX[i] = (int)(L.hX * static_cast<double>(i));
		L.scale = (float)sc[l];
		L.kpSize = (float)(int)(kPatchSize * sc[l]);
		// resize tables from level l-1 (cv::resize INTER_LINEAR, Appendix A.1) and composed nearest-neighbour maps (A.2)
		L.tabX = (int)taps.size();
		if (l > 0) {
			const LevelInfo& P = hd.lv[l - 1];
			const double scale_x = 1. / ((double)L.w / P.w), scale_y = 1. / ((double)L.h / P.h);
			for (int dx = 0; dx < L.w; dx++) {
				float fx = (float)((dx + 0.5) * scale_x - 0.5);
				int sx = cvFloor_(fx);
				fx -= sx;
				if (sx < 0) { fx = 0; sx = 0; }
				if (sx >= P.w - 1) { fx = 0; sx = P.w - 1; }
				ResizeTap t{(short)sx, sat_short((1.f - fx) * 2048), sat_short(fx * 2048), 0};
				taps.push_back(t);
			}
			// k_resize_cols: every group of four adjacent columns must read within 8 adjacent source bytes (any scale factor up to 2), rows per thread <= 64
			L.colsOk = P.w >= 8 && L.h >= 1;
			for (int dx = 0; dx < L.w; dx += 4)
				if (taps[L.tabX + std::min(dx + 3, L.w - 1)].ofs - taps[L.tabX + dx].ofs > 6) L.colsOk = 0;
			L.tabY = (int)taps.size();
			for (int dy = 0; dy < L.h; dy++) {
				float fy = (float)((dy + 0.5) * scale_y - 0.5);
				int sy = cvFloor_(fy);
				fy -= sy;
				ResizeTap t{(short)sy, sat_short((1.f - fy) * 2048), sat_short(fy * 2048), 0};
				taps.push_back(t);
			}
		} else { L.tabY = L.tabX; L.colsOk = 0; }
		L.mapX = (int)maps.size();
		for (int x = 0; x < L.w; ++x) {
			if (l == 0) maps.push_back((short)x);
			else {
				const LevelInfo& P = hd.lv[l - 1];
				const double ifx = 1. / ((double)L.w / P.w);
				maps.push_back(maps[P.mapX + std::min(cvFloor_(x * ifx), P.w - 1)]);
			}
		}
		L.mapY = (int)maps.size();
		for (int y = 0; y < L.h; ++y) {
			if (l == 0) maps.push_back((short)y);
			else {
				const LevelInfo& P = hd.lv[l - 1];
				const double ify = 1. / ((double)L.h / P.h);
				maps.push_back(maps[P.mapY + std::min(cvFloor_(y * ify), P.h - 1)]);
			}
		}
	}
	hd.pyrBytes = off;
	hd.cellsPerImage = cellBase; hd.slotsPerImage = slotBase; hd.densePerImage = denseBase; hd.selPerImage = selBase;
	// rows per image: DistributeOctTree returns at most nfeat_l + 3 keys for a level (a split adds <= 3 nodes before the `>= N` test) — but its first
	// expansion pass runs before that test, so a level can also return up to 4 * nIni keys whatever its quota (:663-700)
	hd.kpCap = 0;
	for (int l = 0; l < nl; ++l) hd.kpCap += std::max(hd.lv[l].nfeat + 3, 4 * hd.lv[l].nIni);

	// orientation disc (IC_Angle rows v = 0, +-1..+-16 with |u| <= umax[|v|], 845 pixels)
	int umax[kHalfPatch + 1];
	build_umax(umax);
	std::vector<signed char> disc;
	for (int v = -kHalfPatch; v <= kHalfPatch; ++v)
		for (int u = -umax[std::abs(v)]; u <= umax[std::abs(v)]; ++u) { disc.push_back((signed char)u); disc.push_back((signed char)v); }
	if (disc.size() != 845 * 2) { delete e; return fail(MCS_ERR_INVALID, "internal: disc size"); }
	for (int v = 0; v <= kHalfPatch; ++v) hd.umax[v] = umax[v];
	if (!upload_describe_tables(kPattern)) { delete e; return fail(MCS_ERR_HIP, "pattern tables: upload failed (or the distinct-point rounds of mcs_describe.hip do not fit this pattern)"); }

	hd.chainFits = 0; hd.chainRegOff = 0;
	// One launch for the whole resize chain (k_resize_chain, bit-exact, tests/test_gpu_env_paths.py) is opt-in: measured 0.37 ms alone against 0.22 ms for the
	// seven launches (its 6912 workgroups pay eight barriers and the small levels run at a quarter of the lanes), default step 2.20 against 2.09 ms.
	// (Also measured without gain on the seven-launch chain: FAST launched per level as soon as the level exists, 2.11 ms.)
	if (getenv("MCS_PYR_CHAIN") && !taps.empty()) {
		std::vector<int> table;
		if (pyramid_chain_table(hd, taps.data(), &table)) {   // regions of the one-launch resize chain, stored behind the taps (8 bytes per entry)
			hd.chainFits = 1; hd.chainRegOff = (int)taps.size();
			taps.resize(taps.size() + (table.size() * sizeof(int) + sizeof(ResizeTap) - 1) / sizeof(ResizeTap));
			memcpy(taps.data() + hd.chainRegOff, table.data(), table.size() * sizeof(int));
		}
	}
	const size_t B = max_batch;
#define ALLOC(ptr, bytes) do { hipError_t _e = hipMalloc((void**)&(ptr), (bytes)); if (_e != hipSuccess) { mcs_extractor_destroy(e); return fail(MCS_ERR_HIP, std::string("hipMalloc ") + #ptr + ": " + hipGetErrorString(_e)); } } while (0)
	ALLOC(e->d_desc, sizeof(PyrDesc));
	ALLOC(e->d_cells, sizeof(CellInfo) * e->cells.size());
	ALLOC(e->d_taps, sizeof(ResizeTap) * std::max<size_t>(taps.size(), 1));
	ALLOC(e->d_maskMap, sizeof(short) * maps.size());
	ALLOC(e->d_pyr, B * hd.pyrBytes);
	ALLOC(e->d_blur, B * hd.pyrBytes);
	ALLOC(e->d_slots, B * hd.slotsPerImage * sizeof(uint32_t));
	ALLOC(e->d_dense, B * hd.densePerImage * sizeof(uint32_t));
	ALLOC(e->d_knode, B * hd.densePerImage * sizeof(unsigned short));
	ALLOC(e->d_sel, B * hd.selPerImage * sizeof(uint32_t));
	ALLOC(e->d_selAngle, B * hd.selPerImage * sizeof(float));
	ALLOC(e->d_cellCount, B * hd.cellsPerImage * sizeof(int));
	ALLOC(e->d_denseCount, B * nl * sizeof(int));
	ALLOC(e->d_selCount, B * nl * sizeof(int));
	ALLOC(e->d_status, sizeof(int));
	ALLOC(e->d_cams, B * sizeof(OcamDev));
	const size_t slotsPerImage = (size_t)(hd.kpCap + kSlotAlign - 1) / kSlotAlign * kSlotAlign;
	ALLOC(e->d_fbCount, 3 * sizeof(int));   // [0] fallback list, [1] pre-list, [2] tie list
	if (hipHostMalloc((void**)&e->h_status, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); e->h_status = nullptr; }   // (without it host-kind outputs take the runtime's copies)
	ALLOC(e->d_tieList, B * slotsPerImage * sizeof(uint32_t));
	ALLOC(e->d_fbList, B * slotsPerImage * sizeof(uint32_t));
	ALLOC(e->d_preList, B * slotsPerImage * sizeof(uint32_t));
	ALLOC(e->d_fbStats, 2 * sizeof(unsigned long long));   // [0] exact-pass keypoints, [1] tie-listed keypoints
	ALLOC(e->d_tieMin, sizeof(unsigned long long));
	ALLOC(e->d_aux, B * slotsPerImage * describe_aux_bytes());
	ALLOC(e->d_gTab, B * (size_t)kGDevDoubles * sizeof(double));
	ALLOC(e->d_nkp, B * sizeof(int));
	ALLOC(e->d_kps, B * hd.kpCap * sizeof(mcs_keypoint));
	ALLOC(e->d_odesc, B * hd.kpCap * (size_t)hd.descSize);
	ALLOC(e->d_omask, B * hd.kpCap * (size_t)hd.descSize);
	ALLOC(e->d_rays, B * hd.kpCap * 3 * sizeof(double));
#undef ALLOC
	HIPCHK(hipMemcpy(e->d_desc, &hd, sizeof(PyrDesc), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(e->d_cells, e->cells.data(), sizeof(CellInfo) * e->cells.size(), hipMemcpyHostToDevice));
	if (!taps.empty()) HIPCHK(hipMemcpy(e->d_taps, taps.data(), sizeof(ResizeTap) * taps.size(), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(e->d_maskMap, maps.data(), sizeof(short) * maps.size(), hipMemcpyHostToDevice));
	HIPCHK(hipMemset(e->d_status, 0, sizeof(int)));
	HIPCHK(hipMemset(e->d_fbCount, 0, 3 * sizeof(int)));
	HIPCHK(hipMemset(e->d_fbStats, 0, 2 * sizeof(unsigned long long)));
	{ const unsigned long long inf = 0x7FF0000000000000ull; HIPCHK(hipMemcpy(e->d_tieMin, &inf, sizeof(inf), hipMemcpyHostToDevice)); }   // +inf: no coordinate seen yet
	if (hipHostMalloc((void**)&e->hostTie.host, sizeof(TieCaptureHeader) + (size_t)kHostTieSlots * sizeof(TieCaptureEntry), hipHostMallocDefault) == hipSuccess) {
		memset(e->hostTie.host, 0, sizeof(TieCaptureHeader));
		e->hostTie.dev = (uint8_t*)device_view(e->hostTie.host);
	} else { (void)hipGetLastError(); e->hostTie.host = nullptr; }   // (without it host-kind batches take the whole-level download)
	if (getenv("MCS_DESCRIBE_EXACT")) e->describeMode = 1;   // A/B and debugging: the exact pass for every keypoint
	HIPCHK(hipMemset(e->d_pyr, 0, B * hd.pyrBytes));
	HIPCHK(hipMemset(e->d_blur, 0, B * hd.pyrBytes));
	ctx->extractors.push_back(e);
	*out = e;
	return MCS_OK;
}

static void free_tie_capture(mcs_extractor* e) {
	for (mcs_extractor::TieSlot& t : e->tieRing) { if (t.ev) (void)hipEventDestroy(t.ev); if (t.host) (void)hipHostFree(t.host); }
	e->tieRing.clear(); e->tieMax = 0;
	if (e->h_patchRows) { (void)hipHostFree(e->h_patchRows); e->h_patchRows = nullptr; }
	if (e->patchStream) { (void)hipStreamSynchronize(e->patchStream); (void)hipStreamDestroy(e->patchStream); e->patchStream = nullptr; }
}

int mcs_extractor_destroy(mcs_extractor* e) {
	if (!e) return MCS_OK;
	(void)hipSetDevice(e->ctx->device);
	(void)hipStreamSynchronize(e->ctx->stream);
	{
		std::vector<mcs_extractor*>& v = e->ctx->extractors;
		v.erase(std::remove(v.begin(), v.end(), e), v.end());
	}
	void* ptrs[] = {e->d_desc, e->d_cells, e->d_taps, e->d_maskMap, e->d_pyr, e->d_blur, e->d_slots, e->d_dense, e->d_knode,
	                e->d_sel, e->d_cellCount, e->d_denseCount, e->d_selCount, e->d_status, e->d_cams, e->d_nkp, e->d_kps, e->d_odesc,
	                e->d_omask, e->d_rays, e->d_inImg, e->d_inMask, e->d_fbCount, e->d_fbList, e->d_preList, e->d_fbStats, e->d_tieMin, e->d_aux, e->d_gTab, e->d_selAngle, e->d_tieList};
	for (void* p : ptrs) (void)hipFree(p);
	for (mcs_extractor::Graph& g : e->graphs) (void)hipGraphExecDestroy(g.exec);
	(void)hipFree(e->d_resMask);
	if (e->h_status) (void)hipHostFree(e->h_status);
	free_tie_capture(e);
	if (e->hostTie.host) (void)hipHostFree(e->hostTie.host);
	delete e;
	return MCS_OK;
}

int mcs_extractor_kp_capacity(const mcs_extractor* e, int* cap) {
	if (!e || !cap) return fail(MCS_ERR_INVALID, "null");
	*cap = e->hd.kpCap;
	return MCS_OK;
}

int mcs_extractor_levels(const mcs_extractor* e, int* nlevels, int* widths, int* heights, int* features_per_level) {
	if (!e) return fail(MCS_ERR_INVALID, "null");
	if (nlevels) *nlevels = e->hd.nlevels;
	for (int l = 0; l < e->hd.nlevels; ++l) {
		if (widths) widths[l] = e->hd.lv[l].w;
		if (heights) heights[l] = e->hd.lv[l].h;
		if (features_per_level) features_per_level[l] = e->hd.lv[l].nfeat;
	}
	return MCS_OK;
}

static int grow(uint8_t** p, size_t* cap, size_t need) {   // device buffer of at least `need` bytes (host-kind calls end with a stream sync, so it is idle here)
	if (*cap >= need) return MCS_OK;
	if (*p) (void)hipFree(*p);
	*p = nullptr; *cap = 0;
	HIPCHK(hipMalloc((void**)p, need + need / 4));
	*cap = need + need / 4;
	return MCS_OK;
}

// The listed keypoints of the extractor's LAST batch (ExtractBuffers.tieList) through describe_host; the stream must be idle.  The recomputed rows replace the
// device outputs in place and, when given, the host copies (tight [image][kpCap][descSize] arrays).
static int fix_ties(mcs_extractor* e, int nties, uint8_t* h_desc, uint8_t* h_mask) {
	const PyrDesc& hd = e->hd;
	const ExtractBuffers& b = e->last;
	const int wavesPerImage = (hd.kpCap + kSlotAlign - 1) / kSlotAlign * kSlotAlign;
	const int cap = e->maxBatch * wavesPerImage;
	if (nties > cap) nties = cap;
	std::vector<uint32_t> list(nties);
	HIPCHK(hipMemcpy(list.data(), e->d_tieList, sizeof(uint32_t) * nties, hipMemcpyDeviceToHost));
	std::sort(list.begin(), list.end());   // by image, then slot: one download per (image, level)
	struct Lv { std::vector<uint8_t> blur, raw; };
	std::map<std::pair<int, int>, Lv> levels;
	std::vector<int> selCount(hd.nlevels);
	int curImg = -1;
	std::vector<uint8_t> dsc(hd.descSize), msk(hd.descSize);
	for (uint32_t gw : list) {
		const int img = (int)(gw / (uint32_t)wavesPerImage), sl = (int)(gw - (uint32_t)img * wavesPerImage);
		if (img >= e->lastN || sl >= hd.kpCap) continue;
		if (img != curImg) { HIPCHK(hipMemcpy(selCount.data(), e->d_selCount + (size_t)img * hd.nlevels, sizeof(int) * hd.nlevels, hipMemcpyDeviceToHost)); curImg = img; levels.clear(); }
		int level = -1, pos = 0, total = 0;
		for (int l = 0; l < hd.nlevels; ++l) { if (sl >= total && sl < total + selCount[l]) { level = l; pos = sl - total; } total += selCount[l]; }
		if (level < 0) continue;
		const LevelInfo& L = hd.lv[level];
		uint32_t rec = 0; float angle = 0.f;
		HIPCHK(hipMemcpy(&rec, e->d_sel + (size_t)img * hd.selPerImage + L.selBase + pos, sizeof(rec), hipMemcpyDeviceToHost));
		HIPCHK(hipMemcpy(&angle, e->d_selAngle + (size_t)img * hd.selPerImage + L.selBase + pos, sizeof(angle), hipMemcpyDeviceToHost));
		Lv& lv = levels[std::make_pair(img, level)];
		if (lv.blur.empty()) {
			lv.blur.resize((size_t)L.w * L.h); lv.raw.resize((size_t)L.w * L.h);
			int rstride = 0;
			const uint8_t* raw = level_ptr(b, hd, img, level, &rstride);
			HIPCHK(hipMemcpy2D(lv.blur.data(), L.w, e->d_blur + (size_t)img * hd.pyrBytes + L.off, L.stride, L.w, L.h, hipMemcpyDeviceToHost));
			HIPCHK(hipMemcpy2D(lv.raw.data(), L.w, raw, rstride, L.w, L.h, hipMemcpyDeviceToHost));
		}
		HostLevel hl{lv.blur.data(), lv.raw.data(), L.w, L.h};
		const int col = (int)(rec & 0xFFF) + kMinBorder, row = (int)((rec >> 12) & 0xFFF) + kMinBorder;
		const OcamDev* cam = hd.mode != 0 && (size_t)img < e->h_cams.size() ? &e->h_cams[img] : nullptr;
		if (hd.mode != 0 && !cam) return fail(MCS_ERR_INVALID, "internal: tie list without camera models");
		describe_host(hd.mode, hd.descSize, kPattern, cam, hd.undistort, level, L.scale, row, col, angle, hl, dsc.data(), msk.data());
		const size_t drow = ((size_t)img * b.outImgPitch + sl) * b.outRowStride;
		HIPCHK(hipMemcpy(b.out_desc + drow, dsc.data(), hd.descSize, hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(b.out_mask + drow, msk.data(), hd.descSize, hipMemcpyHostToDevice));
		if (h_desc) memcpy(h_desc + ((size_t)img * hd.kpCap + sl) * hd.descSize, dsc.data(), hd.descSize);
		if (h_mask) memcpy(h_mask + ((size_t)img * hd.kpCap + sl) * hd.descSize, msk.data(), hd.descSize);
		e->tieFixed++;
	}
	const int zero = 0;
	HIPCHK(hipMemcpy(e->d_fbCount + 2, &zero, sizeof(int), hipMemcpyHostToDevice));   // done: a second call finds nothing
	return MCS_OK;
}

// The listed keypoints of a batch from what k_tie_capture left in page-locked memory (mcs_tiefix.hip): entry i -> descriptor | mask rows at rowsOut + i * 2 *
// descSize, its image and slot in where[i]; returns the number of rows recomputed, < 0 = error code.  No device access at all.
static int recompute_captured(mcs_extractor* e, const uint8_t* slot, int n, const std::vector<OcamDev>& cams, uint8_t* rowsOut, std::vector<std::pair<int, int> >& where) {
	const PyrDesc& hd = e->hd;
	const int wavesPerImage = (hd.kpCap + kSlotAlign - 1) / kSlotAlign * kSlotAlign;
	where.clear();
	for (int i = 0; i < n; ++i) {
		const TieCaptureEntry* en = reinterpret_cast<const TieCaptureEntry*>(slot + sizeof(TieCaptureHeader) + (size_t)i * sizeof(TieCaptureEntry));
		if (en->level < 0 || en->level >= hd.nlevels) continue;
		const int img = (int)(en->gw / (uint32_t)wavesPerImage), sl = (int)(en->gw - (uint32_t)img * wavesPerImage);
		const LevelInfo& L = hd.lv[en->level];
		const int col = (int)(en->rec & 0xFFF) + kMinBorder, row = (int)((en->rec >> 12) & 0xFFF) + kMinBorder;
		const OcamDev* cam = hd.mode != 0 && (size_t)img < cams.size() ? &cams[img] : nullptr;
		if (hd.mode != 0 && !cam) return fail(MCS_ERR_INVALID, "internal: tie capture without camera models");
		bool miss = false;
		HostLevel hl{nullptr, nullptr, L.w, L.h, en->patch, row - kTiePatchR, col - kTiePatchR, kTiePatchDim, &miss};
		uint8_t* dsc = rowsOut + where.size() * 2 * hd.descSize;
		describe_host(hd.mode, hd.descSize, kPattern, cam, hd.undistort, en->level, L.scale, row, col, en->angle, hl, dsc, dsc + hd.descSize);
		if (miss) { ++e->tieWindowMisses; return fail(MCS_ERR_UNSUPPORTED, "a pattern sample of a listed keypoint lies outside the captured window (camera model with > 1.9x local magnification?)"); }
		where.push_back(std::make_pair(img, sl));
	}
	return (int)where.size();
}

// Host-kind batches: the listed keypoints from the extractor's own capture slot (written by k_tie_capture in front of the batch's final synchronisation) — a few
// microseconds per keypoint, no device access; the whole-level download of fix_ties only when more keypoints are listed than the slot holds (widened bands)
static int fix_ties_host(mcs_extractor* e, int nties, uint8_t* h_desc, uint8_t* h_mask) {
	const PyrDesc& hd = e->hd;
	if (!e->hostTie.host || nties > kHostTieSlots) return fix_ties(e, nties, h_desc, h_mask);
	std::vector<std::pair<int, int> > where;
	std::vector<uint8_t> rows((size_t)nties * 2 * hd.descSize);
	const int done = recompute_captured(e, e->hostTie.host, nties, e->h_cams, rows.data(), where);
	if (done < 0) return done;
	for (int i = 0; i < done; ++i) {
		const size_t r = ((size_t)where[i].first * hd.kpCap + where[i].second) * hd.descSize;
		if (h_desc) memcpy(h_desc + r, rows.data() + (size_t)i * 2 * hd.descSize, hd.descSize);
		if (h_mask) memcpy(h_mask + r, rows.data() + (size_t)i * 2 * hd.descSize + hd.descSize, hd.descSize);
	}
	e->tieFixed += (unsigned long long)done;
	return MCS_OK;
}

namespace mcs {
// Host-kind outputs in ONE launch: when the caller's output arrays are page-locked (mcs_host_alloc / hipHostMalloc: visible to the device), the valid rows of the
// five staging arrays, the counts and the batch's status words are written straight into them.  The five hipMemcpyAsync calls this replaces cost ~15 us each for
// ONE multi-frame (7 us of transfer, 8 of runtime per call), plus two more for the status words: a third of the extraction's latency.
struct ExtractOut { int32_t* nkp; uint32_t* kps; uint32_t* desc; uint32_t* mask; uint32_t* rays; int* status; };
__global__ __launch_bounds__(256) void k_extract_out(const int* __restrict__ d_nkp, const uint32_t* __restrict__ kps, const uint32_t* __restrict__ desc, const uint32_t* __restrict__ mask,
                                                     const uint32_t* __restrict__ rays, const int* __restrict__ d_status, const int* __restrict__ d_ties, ExtractOut o, int kpCap, int descDw,
                                                     ExtractBuffers b, int nimg, int wavesPerImage, int maxTies, uint8_t* __restrict__ tieOut) {
	if ((int)blockIdx.y == nimg) {   // the extra grid row: the rounding-tie capture of this batch (mcs_tiecap.h), in the same launch
		tie_capture_body(b, nimg, wavesPerImage, maxTies, tieOut, blockIdx.x, gridDim.x);
		return;
	}
	const int img = blockIdx.y, n = d_nkp[img], part = blockIdx.x, parts = gridDim.x;
	auto copy = [&](uint32_t* dst, const uint32_t* src, int rowDw) {   // rows [0, n) of image img: one contiguous run of dwords
		const size_t base = (size_t)img * kpCap * rowDw;
		const int total = n * rowDw;
		for (int i = part * 256 + threadIdx.x; i < total; i += parts * 256) dst[base + i] = src[base + i];
	};
	copy(o.kps, kps, (int)(sizeof(mcs_keypoint) / 4));
	copy(o.desc, desc, descDw);
	copy(o.mask, mask, descDw);
	if (o.rays) copy(o.rays, rays, 6);
	if (part == 0 && threadIdx.x == 0) {
		o.nkp[img] = n;
		if (img == 0) { o.status[0] = *d_status; o.status[1] = *d_ties; }
	}
}
}  // namespace mcs

static bool use_graphs() {   // MCS_GRAPHS=0: every launch of a small batch enqueued one by one (A/B, tests)
	static const bool on = !(getenv("MCS_GRAPHS") && atoi(getenv("MCS_GRAPHS")) == 0);
	return on;
}

static int extract_impl(mcs_extractor* e, int nimg, const uint8_t* images, size_t image_pitch, int image_stride, const uint8_t* masks,
                        size_t mask_pitch, int mask_stride, const mcs_ocam* cams, mcs_mem_kind kind, int32_t* nkp, mcs_keypoint* keypoints,
                        uint8_t* desc, uint8_t* descmask, double* rays, size_t out_image_pitch_rows, int out_row_stride) {
	if (!e || !images || !nkp || !keypoints || !desc || !descmask) return fail(MCS_ERR_INVALID, "null argument");
	if (nimg < 1 || nimg > e->maxBatch) return fail(MCS_ERR_INVALID, "nimg exceeds the extractor's max_batch");
	const PyrDesc& hd = e->hd;
	const bool resMask = masks == MCS_MASKS_RESIDENT;
	if (resMask) {
		if (!e->d_resMask || nimg > e->resMaskN) return fail(MCS_ERR_INVALID, "MCS_MASKS_RESIDENT: mcs_extractor_set_masks has not been called for this many images");
		masks = e->d_resMask; mask_pitch = (size_t)hd.width * hd.height; mask_stride = hd.width;
	}
	if (image_stride < hd.width || (masks && mask_stride < hd.width)) return fail(MCS_ERR_INVALID, "stride smaller than the image width");
	if (hd.mode != 0 && !cams) return fail(MCS_ERR_INVALID, "dBRIEF/mdBRIEF need camera models");
	if (rays && !cams) return fail(MCS_ERR_INVALID, "rays need camera models");
	mcs_ctx* c = e->ctx;
	hipStream_t s = c->stream;
	HIPCHK(hipSetDevice(c->device));

	ExtractBuffers b{};
	b.desc = e->d_desc; b.cells = e->d_cells; b.taps = e->d_taps; b.maskMap = e->d_maskMap;
	b.pyr = e->d_pyr; b.blur = e->d_blur; b.slots = e->d_slots; b.cellCount = e->d_cellCount; b.dense = e->d_dense; b.knode = e->d_knode;
	b.denseCount = e->d_denseCount; b.sel = e->d_sel; b.selCount = e->d_selCount; b.selAngle = e->d_selAngle; b.status = e->d_status;
	b.gTab = e->d_gTab; b.aux = e->d_aux; b.fbCount = e->d_fbCount; b.fbList = e->d_fbList; b.preCount = e->d_fbCount + 1; b.preList = e->d_preList;
	b.fbStats = e->d_fbStats; b.tieMin = e->d_tieMin; b.guardEps = e->guardEps; b.describeMode = e->describeMode;
	b.tieCount = e->d_fbCount + 2; b.tieList = e->d_tieList; b.tieTotal = e->d_fbStats + 1;
	b.tieBand = e->tieBand > 0.0 ? e->tieBand : (e->tieBand < 0.0 ? -1.0 : (hd.mode == 0 ? kTieBandOrb : kTieBandDistorted));
	b.sideStream = nullptr; b.evDescFork = nullptr; b.evDescJoin = nullptr;
	b.outImgPitch = out_image_pitch_rows ? out_image_pitch_rows : (size_t)hd.kpCap;
	b.outRowStride = out_row_stride ? out_row_stride : hd.descSize;
	if (b.outImgPitch < (size_t)hd.kpCap || b.outRowStride < hd.descSize) return fail(MCS_ERR_INVALID, "output image pitch / row stride smaller than the rows they hold");
	// the descriptor kernels store rows as 8-byte words
	if (b.outRowStride % 8 != 0 || (kind != MCS_MEM_HOST && (((uintptr_t)desc | (uintptr_t)descmask) & 7) != 0))
		return fail(MCS_ERR_INVALID, "descriptor / mask rows must be 8-byte aligned (row stride a multiple of 8)");
	if (kind != MCS_MEM_HOST && (out_image_pitch_rows || out_row_stride)) {
		// strided outputs: mask rows either interleaved with the descriptor rows (inside the same stride, behind the descriptor) or a separate array
		const uintptr_t lo = std::min((uintptr_t)desc, (uintptr_t)descmask), hi = std::max((uintptr_t)desc, (uintptr_t)descmask);
		const size_t span = ((size_t)(nimg - 1) * b.outImgPitch + (size_t)hd.kpCap - 1) * (size_t)b.outRowStride + (size_t)hd.descSize;
		const bool interleaved = hi - lo >= (uintptr_t)hd.descSize && hi - lo + (uintptr_t)hd.descSize <= (uintptr_t)b.outRowStride;
		if (!interleaved && hi - lo < span) return fail(MCS_ERR_INVALID, "descriptor and mask rows overlap for this pitch / stride");
	}
	if (kind == MCS_MEM_HOST && (b.outImgPitch != (size_t)hd.kpCap || b.outRowStride != hd.descSize)) return fail(MCS_ERR_UNSUPPORTED, "strided descriptor outputs need device memory");
	{
		// k_blur / k_resize_cols form a lane's level-0 source offset as (image - first image of the wave) * image_pitch in 32 bits; a wave spans 64 / (column groups
		// per image) + 2 images at most
		const unsigned long long ncg = (unsigned long long)((hd.width + 3) / 4), span = 64ull / (ncg ? ncg : 1ull) + 2ull;
		if ((unsigned long long)image_pitch * span >= (1ull << 32)) return fail(MCS_ERR_UNSUPPORTED, "image_pitch too large for the level-0 readers (pitch * images per wave must stay below 4 GiB)");
	}
	if (kind == MCS_MEM_HOST) {
		// ONE linear copy per block, in the caller's own layout; the kernels take any pitch / stride for level 0.  (A pitched hipMemcpy2D from pageable
		// host memory is carried out row by row by the runtime: 2 x 480 small transfers per image, ~9 ms per image.)
		const size_t imgSpan = (size_t)(nimg - 1) * image_pitch + (size_t)(hd.height - 1) * image_stride + hd.width;
		if (int r = grow(&e->d_inImg, &e->inImgCap, imgSpan)) return r;
		HIPCHK(hipMemcpyAsync(e->d_inImg, images, imgSpan, hipMemcpyHostToDevice, s));
		b.img0 = e->d_inImg; b.img0Pitch = image_pitch; b.img0Stride = image_stride;
		if (resMask) { b.mask0 = masks; b.mask0Pitch = mask_pitch; b.mask0Stride = mask_stride; }
		else if (masks) {
			const size_t maskSpan = (size_t)(nimg - 1) * mask_pitch + (size_t)(hd.height - 1) * mask_stride + hd.width;
			if (int r = grow(&e->d_inMask, &e->inMaskCap, maskSpan)) return r;
			HIPCHK(hipMemcpyAsync(e->d_inMask, masks, maskSpan, hipMemcpyHostToDevice, s));
			b.mask0 = e->d_inMask; b.mask0Pitch = mask_pitch; b.mask0Stride = mask_stride;
		}
		b.nkp = e->d_nkp; b.kps = e->d_kps; b.out_desc = e->d_odesc; b.out_mask = e->d_omask; b.rays = rays ? e->d_rays : nullptr;
	} else {
		b.img0 = images; b.img0Pitch = image_pitch; b.img0Stride = image_stride;
		b.mask0 = masks; b.mask0Pitch = mask_pitch; b.mask0Stride = mask_stride;
		b.nkp = nkp; b.kps = keypoints; b.out_desc = desc; b.out_mask = descmask; b.rays = rays;
	}
	if (cams) {
		std::vector<OcamDev> hc(nimg);
		std::vector<int> which(nimg), uniq;   // uniq: the cache entries this batch uses, in order of first use (their tables are uploaded once each)
		for (int i = 0; i < nimg; ++i) {
			const mcs_ocam& m = cams[i];
			if (m.p_deg < 1 || m.p_deg > MCS_MAX_POLY || m.invP_deg < 1 || m.invP_deg > MCS_MAX_POLY) return fail(MCS_ERR_INVALID, "bad polynomial degree");
			OcamDev& o = hc[i];
			memset(&o, 0, sizeof(o));
			o.c = m.c; o.d = m.d; o.e = m.e; o.u0 = m.u0; o.v0 = m.v0; o.invAffine = m.c - m.d * m.e;
			for (int k = 0; k < m.p_deg; ++k) o.p[k] = m.p[k];
			for (int k = 0; k < m.invP_deg; ++k) o.invP[k] = m.invP[k];
			o.p_deg = m.p_deg; o.invP_deg = m.invP_deg;
			// the camera's G(s) table and its bounds: built once per distinct camera
			int w = -1;
			for (size_t k = 0; k < e->camCache.size() && w < 0; ++k) if (memcmp(&e->camCache[k].key, &o, sizeof(o)) == 0) w = (int)k;
			if (w < 0) {
				// (no eviction here: which[] of the earlier images of this batch indexes the cache — it is trimmed after the batch's tables are copied)
				mcs_extractor::CamFast cf;
				cf.key = o; cf.tab.assign(kGTabDoubles, 0.0);
				cf.info = build_g_table(m, cf.tab.data());
				if (!std::isfinite(cf.info.tailU)) cf.tab.assign(kGTabDoubles, 0.0);
				e->camCache.push_back(std::move(cf));
				w = (int)e->camCache.size() - 1;
			}
			which[i] = w;
			const double bound = describe_fast_bound(m, hd.npoints, e->camCache[w].info);
			o.fastOk = bound <= 0.5 * e->guardEps ? 1 : 0;   // a factor 2 between the worst case and the band
			size_t up = std::find(uniq.begin(), uniq.end(), w) - uniq.begin();
			if (up == uniq.size()) uniq.push_back(w);
			o.tabIdx = (int)up;
		}
		if (e->h_cams.size() != hc.size() || memcmp(e->h_cams.data(), hc.data(), sizeof(OcamDev) * hc.size()) != 0) {
			HIPCHK(hipStreamSynchronize(s));   // the previous batch may still read d_cams / d_gTab
			HIPCHK(hipMemcpy(e->d_cams, hc.data(), sizeof(OcamDev) * hc.size(), hipMemcpyHostToDevice));
			std::vector<double> tabs(uniq.size() * (size_t)kGDevDoubles);   // the packed device rows
			for (size_t i = 0; i < uniq.size(); ++i) pack_g_table(e->camCache[uniq[i]].tab.data(), reinterpret_cast<uint8_t*>(&tabs[i * kGDevDoubles]));
			HIPCHK(hipMemcpy(e->d_gTab, tabs.data(), tabs.size() * sizeof(double), hipMemcpyHostToDevice));
			e->h_cams = hc;
		}
		if (e->camCache.size() > 64) e->camCache.clear();   // a rig has a handful of cameras; a caller that streams distinct models rebuilds (which[] is dead here)
		b.cams = e->d_cams;
	}
	// (Other launch orders measured in round 3: the whole resize chain first on the main stream, then FAST on all levels in one launch with the blur beside it:
	// 2.55 instead of 2.21 ms per step; level 1 first, then FAST on levels 0 and 1 beside the rest of the chain: 2.51.)
	// Small batches run IN ORDER on the context's stream (round 5): the forks pay for themselves only when the kernels are long — for ONE 3-camera multi-frame the
	// kernels take 5-50 us each, and every cross-stream event costs about as much as one of them (extraction of 3 images: 0.33 -> 0.29 ms).
	const bool forked = c->overlap() && (long long)nimg * hd.width * hd.height >= 4000000ll;
	bool replayed = false;   // the whole launch sequence, descriptor stage included, came from a graph
	if (forked) {
		// Two chains until the descriptors: the side stream runs the resize chain (7 dependent, latency-bound launches) and then the blur, which needs
		// nothing else; the main stream starts FAST on level 0 — the input image itself, a third of all FAST work — at once, picks up the other levels
		// when the pyramid is there, and runs the oct-tree.  Both chains leave most of the chip idle on their own.
		HIPCHK(hipEventRecord(c->evFork, s));
		HIPCHK(hipStreamWaitEvent(c->side, c->evFork, 0));
		if (hd.chainFits) {
			// the resize chain is one launch (k_resize_chain): FAST on level 0 beside it, then FAST on all other levels in one launch beside the blur
			launch_pyramid(b, hd, nimg, c->side);
			HIPCHK(hipEventRecord(c->evPyr, c->side));
			launch_blur(b, hd, nimg, c->side);
			HIPCHK(hipEventRecord(c->evBlur, c->side));
			launch_fast(b, hd, nimg, s, 0, 1);
			HIPCHK(hipStreamWaitEvent(s, c->evPyr, 0));
			launch_fast(b, hd, nimg, s, 1, hd.nlevels);
		} else {
		static const int sched = getenv("MCS_SCHED") ? atoi(getenv("MCS_SCHED")) : 3;   // launch order; 3 (default) since round 4, the others for A/B
		if (sched == 5) {   // as 3, with the chain on the MAIN stream: no cross-stream event in front of the first resize and none between the chain and FAST; the blur forks
			launch_pyramid(b, hd, nimg, s, 1, hd.nlevels);
			HIPCHK(hipEventRecord(c->evPyr, s));
			HIPCHK(hipStreamWaitEvent(c->side, c->evPyr, 0));
			launch_blur(b, hd, nimg, c->side);
			HIPCHK(hipEventRecord(c->evBlur, c->side));
			launch_fast(b, hd, nimg, s, 0, hd.nlevels);
		} else if (sched == 6) {   // FAST alone, the blur beside the oct-trees (latency-bound: one workgroup per image and level, the level-0 ones set its time)
			launch_pyramid(b, hd, nimg, c->side, 1, hd.nlevels);
			HIPCHK(hipEventRecord(c->evPyr, c->side));
			HIPCHK(hipStreamWaitEvent(s, c->evPyr, 0));
			launch_fast(b, hd, nimg, s, 0, hd.nlevels);
			HIPCHK(hipEventRecord(c->evPyr1, s));
			HIPCHK(hipStreamWaitEvent(c->side, c->evPyr1, 0));
			launch_blur(b, hd, nimg, c->side);
			HIPCHK(hipEventRecord(c->evBlur, c->side));
		} else {
		launch_pyramid(b, hd, nimg, c->side, 1, 2);
		HIPCHK(hipEventRecord(c->evPyr1, c->side));
		launch_pyramid(b, hd, nimg, c->side, 2, hd.nlevels);
		HIPCHK(hipEventRecord(c->evPyr, c->side));
		launch_blur(b, hd, nimg, c->side);
		HIPCHK(hipEventRecord(c->evBlur, c->side));
		if (sched == 1) {          // FAST level 0, then every other level in one launch behind the whole chain
			launch_fast(b, hd, nimg, s, 0, 1);
			HIPCHK(hipStreamWaitEvent(s, c->evPyr, 0));
			launch_fast(b, hd, nimg, s, 1, hd.nlevels);
		} else if (sched == 2) {   // head start for the chain: levels 0 + 1 in one launch once level 1 exists, the rest behind the chain
			HIPCHK(hipStreamWaitEvent(s, c->evPyr1, 0));
			launch_fast(b, hd, nimg, s, 0, 2);
			HIPCHK(hipStreamWaitEvent(s, c->evPyr, 0));
			launch_fast(b, hd, nimg, s, 2, hd.nlevels);
		} else if (sched == 3) {   // the whole chain first, then FAST on all levels in one launch (the blur beside it)
			HIPCHK(hipStreamWaitEvent(s, c->evPyr, 0));
			launch_fast(b, hd, nimg, s, 0, hd.nlevels);
		} else {
		launch_fast(b, hd, nimg, s, 0, 1);           // levels 0 and 1 carry more than half of the FAST work: the rest of the resize chain finishes behind them
		HIPCHK(hipStreamWaitEvent(s, c->evPyr1, 0));
		launch_fast(b, hd, nimg, s, 1, 2);
		HIPCHK(hipStreamWaitEvent(s, c->evPyr, 0));
		launch_fast(b, hd, nimg, s, 2, hd.nlevels);
		}
		}
		}
		// (holding the previous step's deferred matcher back until here, so that it runs beside the oct-tree / orientation / descriptor kernels instead of
		// beside FAST and the resize chain: measured, 2.22 -> 2.55 ms per step — the descriptor kernel on the critical path suffers more from the company)
		launch_octree(b, hd, nimg, s);   // (oct-trees of levels 0 / 1 on a further stream beside FAST of the rest: measured, no gain)
		HIPCHK(hipStreamWaitEvent(s, c->evBlur, 0));
	} else if (!c->timing && use_graphs() && !(e->graphMisses > 8 && e->graphMisses > e->graphHits)) {
		// ONE multi-frame per call is launch-bound: a dozen dependent kernels of 4 - 50 us each, most of them shorter than the host takes to enqueue the next.  The
		// sequence is captured once per argument set into a hipGraph and replayed (the copies in and out stay ordinary stream operations around it).
		hipGraphExec_t exec = nullptr;
		for (mcs_extractor::Graph& g : e->graphs)
			if (g.nimg == nimg && memcmp(&g.key, &b, sizeof(b)) == 0) exec = g.exec;
		if (exec) ++e->graphHits; else ++e->graphMisses;
		if (!exec) {
			// Capture with an error path: whatever fails between Begin and End, the capture is ENDED (a stream left capturing fails every later call on the context) and
			// the graph destroyed; this call and all later ones of the extractor then enqueue their launches one by one (graphs given up: graphMisses forced high).
			hipGraph_t graph = nullptr;
			hipError_t ge = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
			if (ge == hipSuccess) {
				launch_pyramid(b, hd, nimg, s);
				launch_fast(b, hd, nimg, s);
				launch_octree(b, hd, nimg, s);
				launch_blur(b, hd, nimg, s);
				launch_describe(b, hd, nimg, s);
				const hipError_t le = hipGetLastError();
				ge = hipStreamEndCapture(s, &graph);
				if (ge == hipSuccess && le != hipSuccess) ge = le;
				if (ge == hipSuccess) ge = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
				if (graph) (void)hipGraphDestroy(graph);
			}
			if (ge != hipSuccess) {
				(void)hipGetLastError();
				exec = nullptr;
				e->graphMisses = 1000000;   // no further attempts on this extractor
			} else {
				if (e->graphs.size() >= 4) { (void)hipStreamSynchronize(s); (void)hipGraphExecDestroy(e->graphs.front().exec); e->graphs.erase(e->graphs.begin()); }   // a caller that rotates output buffers: a few sets
				e->graphs.push_back(mcs_extractor::Graph{b, nimg, exec});
			}
		}
		if (!exec) {   // the capture failed: the plain sequence (nothing was enqueued by the failed capture)
			launch_pyramid(b, hd, nimg, s);
			launch_fast(b, hd, nimg, s);
			launch_octree(b, hd, nimg, s);
			launch_blur(b, hd, nimg, s);
			launch_describe(b, hd, nimg, s);
			replayed = true;   // (the descriptor stage is in: nothing left to launch below)
		} else {
		HIPCHK(hipGraphLaunch(exec, s));
		replayed = true;
		}
	} else {
		c->tic("pyramid"); launch_pyramid(b, hd, nimg, s); c->toc("pyramid");
		c->tic("fast"); launch_fast(b, hd, nimg, s); c->toc("fast");
		c->tic("octree"); launch_octree(b, hd, nimg, s); c->toc("octree");
		c->tic("blur"); launch_blur(b, hd, nimg, s); c->toc("blur");
	}
	if (forked) { b.sideStream = c->side; b.evDescFork = c->evDescFork; b.evDescJoin = c->evDescJoin; }   // the side stream is idle again: the main stream has waited for the blur
	if (!replayed) {
		ExtractBuffers bd = b;
		if (c->timing) {   // the dominant kernel alone (k_describe_fast; ORB: k_describe), inside the "describe" bracket
			Timer& t = c->timers["describe_fast"];
			if (!t.a) { (void)hipEventCreate(&t.a); (void)hipEventCreate(&t.b); }
			bd.evFastA = t.a; bd.evFastB = t.b; t.used = true;
		}
		c->tic("describe"); launch_describe(bd, hd, nimg, s); c->toc("describe");
	}
	HIPCHK(hipGetLastError());
	e->last = b; e->lastN = nimg;
	if (kind != MCS_MEM_HOST && !e->tieRing.empty()) {
		// pipelined enforcement: what the host needs of this batch's listed keypoints leaves for page-locked memory right behind the descriptor kernels; the
		// caller patches the rows one step later (mcs_extractor_patch_ties), while the pyramid buffers already hold the next batch
		mcs_extractor::TieSlot& t = e->tieRing[(size_t)(e->batchSeq % (long long)e->tieRing.size())];
		launch_tie_capture(b, hd, nimg, e->tieMax, t.dev, s);
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord(t.ev, s));
		t.b = b; t.nimg = nimg; t.seq = e->batchSeq; t.patched = false;
		if (hd.mode != 0) t.cams = e->h_cams; else t.cams.clear();
		++e->batchSeq;
	}
	if (kind == MCS_MEM_HOST) {
		const size_t rows = (size_t)nimg * hd.kpCap;
		int st = 0, nties = 0;
		mcs::ExtractOut o{(int32_t*)device_view(nkp), (uint32_t*)device_view(keypoints), (uint32_t*)device_view(desc), (uint32_t*)device_view(descmask),
		                  (uint32_t*)device_view(rays), e->h_status ? (int*)device_view(e->h_status) : nullptr};
		static const bool outKernel = !(getenv("MCS_OUT_KERNEL") && atoi(getenv("MCS_OUT_KERNEL")) == 0);   // A/B, tests: 0 = always the runtime's copies
		if (outKernel && o.nkp && o.kps && o.desc && o.mask && o.status && (o.rays || !rays)) {
			// page-locked outputs: one launch writes the valid rows (rows past an image's count are left as they are), the counts and the status words
			// (+ one grid row when the extractor has its capture slot: the listed keypoints' windows, usually none)
			hipLaunchKernelGGL(mcs::k_extract_out, dim3(4, nimg + (e->hostTie.dev ? 1 : 0)), dim3(256), 0, s, e->d_nkp, (const uint32_t*)e->d_kps, (const uint32_t*)e->d_odesc, (const uint32_t*)e->d_omask,
			                   (const uint32_t*)e->d_rays, e->d_status, e->d_fbCount + 2, o, hd.kpCap, hd.descSize / 4, b, nimg, (hd.kpCap + kSlotAlign - 1) / kSlotAlign * kSlotAlign,
			                   kHostTieSlots, e->hostTie.dev);
			HIPCHK(hipGetLastError());
			HIPCHK(hipStreamSynchronize(s));
			st = e->h_status[0]; nties = e->h_status[1];
			if (st != 0) { (void)hipMemset(e->d_status, 0, sizeof(int)); return fail(st, "device capacity exceeded during extraction"); }
			if (nties > 0) { if (int r = fix_ties_host(e, nties, desc, descmask)) return r; }
			return MCS_OK;
		}
		if (e->hostTie.dev) launch_tie_capture(b, hd, nimg, kHostTieSlots, e->hostTie.dev, s);
		HIPCHK(hipMemcpyAsync(nkp, e->d_nkp, nimg * sizeof(int), hipMemcpyDeviceToHost, s));
		HIPCHK(hipMemcpyAsync(keypoints, e->d_kps, rows * sizeof(mcs_keypoint), hipMemcpyDeviceToHost, s));
		HIPCHK(hipMemcpyAsync(desc, e->d_odesc, rows * hd.descSize, hipMemcpyDeviceToHost, s));
		HIPCHK(hipMemcpyAsync(descmask, e->d_omask, rows * hd.descSize, hipMemcpyDeviceToHost, s));
		if (rays) HIPCHK(hipMemcpyAsync(rays, e->d_rays, rows * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
		HIPCHK(hipMemcpyAsync(&st, e->d_status, sizeof(int), hipMemcpyDeviceToHost, s));
		HIPCHK(hipMemcpyAsync(&nties, e->d_fbCount + 2, sizeof(int), hipMemcpyDeviceToHost, s));
		HIPCHK(hipStreamSynchronize(s));
		if (st != 0) { (void)hipMemset(e->d_status, 0, sizeof(int)); return fail(st, "device capacity exceeded during extraction"); }
		// keypoints whose exact arithmetic came within the band of a rounding tie: recomputed here with the host's libm before the results are final
		if (nties > 0) { if (int r = fix_ties_host(e, nties, desc, descmask)) return r; }
	}
	return MCS_OK;
}

int mcs_extractor_set_masks(mcs_extractor* e, int nimg, const uint8_t* masks, size_t mask_pitch, int mask_stride, mcs_mem_kind kind) {
	if (!e || !masks || nimg < 1 || nimg > e->maxBatch) return fail(MCS_ERR_INVALID, "bad argument");
	const PyrDesc& hd = e->hd;
	if (mask_stride < hd.width) return fail(MCS_ERR_INVALID, "stride smaller than the image width");
	HIPCHK(hipSetDevice(e->ctx->device));
	HIPCHK(hipStreamSynchronize(e->ctx->stream));   // an earlier batch may still read the previous masks
	const size_t plane = (size_t)hd.width * hd.height;
	if (nimg > e->resMaskN) { (void)hipFree(e->d_resMask); e->d_resMask = nullptr; e->resMaskN = 0; HIPCHK(hipMalloc((void**)&e->d_resMask, plane * nimg)); }
	for (int i = 0; i < nimg; ++i)
		HIPCHK(hipMemcpy2D(e->d_resMask + (size_t)i * plane, hd.width, masks + (size_t)i * mask_pitch, mask_stride, hd.width, hd.height,
		                   kind == MCS_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice));
	e->resMaskN = nimg;
	return MCS_OK;
}

int mcs_extract_batch(mcs_extractor* e, int nimg, const uint8_t* images, size_t image_pitch, int image_stride, const uint8_t* masks,
                      size_t mask_pitch, int mask_stride, const mcs_ocam* cams, mcs_mem_kind kind, int32_t* nkp, mcs_keypoint* keypoints,
                      uint8_t* desc, uint8_t* descmask, double* rays) {
	return extract_impl(e, nimg, images, image_pitch, image_stride, masks, mask_pitch, mask_stride, cams, kind, nkp, keypoints, desc, descmask, rays, 0, 0);
}

int mcs_extract_batch_strided(mcs_extractor* e, int nimg, const uint8_t* images, size_t image_pitch, int image_stride, const uint8_t* masks,
                              size_t mask_pitch, int mask_stride, const mcs_ocam* cams, int32_t* nkp, mcs_keypoint* keypoints, uint8_t* desc,
                              uint8_t* descmask, double* rays, size_t out_image_pitch_rows, int out_row_stride) {
	return extract_impl(e, nimg, images, image_pitch, image_stride, masks, mask_pitch, mask_stride, cams, MCS_MEM_DEVICE, nkp, keypoints, desc, descmask, rays,
	                    out_image_pitch_rows, out_row_stride);
}

int mcs_extractor_status(mcs_extractor* e) {
	if (!e) return fail(MCS_ERR_INVALID, "null");
	int st = 0;
	HIPCHK(hipStreamSynchronize(e->ctx->stream));
	HIPCHK(hipMemcpy(&st, e->d_status, sizeof(int), hipMemcpyDeviceToHost));
	if (st != 0) { (void)hipMemset(e->d_status, 0, sizeof(int)); return fail(st, "device capacity exceeded during extraction"); }
	return MCS_OK;
}

int mcs_extractor_set_describe(mcs_extractor* e, int exact_only, double guard_eps) {
	if (!e) return fail(MCS_ERR_INVALID, "null");
	if (guard_eps < 0.0 || !(guard_eps < 0.5)) return fail(MCS_ERR_INVALID, "guard band must be in [0, 0.5) pixels (0 = default)");
	HIPCHK(hipStreamSynchronize(e->ctx->stream));
	e->describeMode = exact_only ? 1 : 0;
	e->guardEps = guard_eps > 0.0 ? guard_eps : kDefaultGuardEps;
	e->h_cams.clear();   // fastOk depends on the band: rebuild the device camera table on the next batch
	return MCS_OK;
}

int mcs_extractor_describe_stats(mcs_extractor* e, uint64_t* exact_pass_keypoints, double* guard_eps) {
	if (!e) return fail(MCS_ERR_INVALID, "null");
	HIPCHK(hipStreamSynchronize(e->ctx->stream));
	unsigned long long v = 0;
	HIPCHK(hipMemcpy(&v, e->d_fbStats, sizeof(v), hipMemcpyDeviceToHost));
	if (exact_pass_keypoints) *exact_pass_keypoints = v;
	if (guard_eps) *guard_eps = e->guardEps;
	return MCS_OK;
}

int mcs_extractor_tie_stats(mcs_extractor* e, double* min_tie_distance, int reset) {
	if (!e) return fail(MCS_ERR_INVALID, "null");
	HIPCHK(hipStreamSynchronize(e->ctx->stream));
	unsigned long long v = 0;
	HIPCHK(hipMemcpy(&v, e->d_tieMin, sizeof(v), hipMemcpyDeviceToHost));
	if (min_tie_distance) memcpy(min_tie_distance, &v, sizeof(double));
	if (reset) { const unsigned long long inf = 0x7FF0000000000000ull; HIPCHK(hipMemcpy(e->d_tieMin, &inf, sizeof(inf), hipMemcpyHostToDevice)); }
	return MCS_OK;
}

int mcs_extractor_set_tie_band(mcs_extractor* e, double band_px) {
	if (!e) return fail(MCS_ERR_INVALID, "null");
	if (!(band_px <= 0.5)) return fail(MCS_ERR_INVALID, "tie band must be <= 0.5 pixels (0 = default, < 0 = list nothing)");
	HIPCHK(hipStreamSynchronize(e->ctx->stream));
	e->tieBand = band_px;
	return MCS_OK;
}

int mcs_extractor_fix_ties(mcs_extractor* e, int* recomputed) {
	if (!e) return fail(MCS_ERR_INVALID, "null");
	if (recomputed) *recomputed = 0;
	if (e->lastN <= 0) return MCS_OK;
	HIPCHK(hipSetDevice(e->ctx->device));
	HIPCHK(hipStreamSynchronize(e->ctx->stream));
	int nties = 0;
	HIPCHK(hipMemcpy(&nties, e->d_fbCount + 2, sizeof(int), hipMemcpyDeviceToHost));
	if (nties <= 0) return MCS_OK;
	const unsigned long long before = e->tieFixed;
	if (int r = fix_ties(e, nties, nullptr, nullptr)) return r;
	if (recomputed) *recomputed = (int)(e->tieFixed - before);
	return MCS_OK;
}

int mcs_extractor_set_tie_capture(mcs_extractor* e, int depth, int max_ties) {
	if (!e || depth < 0 || depth > 16 || max_ties < 0 || max_ties > 4096) return fail(MCS_ERR_INVALID, "depth must be 0..16, max_ties 0..4096 (0 = default 64)");
	HIPCHK(hipSetDevice(e->ctx->device));
	HIPCHK(hipStreamSynchronize(e->ctx->stream));
	free_tie_capture(e);
	if (depth == 0) return MCS_OK;
	if (max_ties == 0) max_ties = 64;
	const size_t bytes = sizeof(TieCaptureHeader) + (size_t)max_ties * sizeof(TieCaptureEntry);
	e->tieRing.resize(depth);
	for (mcs_extractor::TieSlot& t : e->tieRing) {
		if (hipHostMalloc((void**)&t.host, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); free_tie_capture(e); return fail(MCS_ERR_HIP, "page-locked capture slot"); }
		memset(t.host, 0, sizeof(TieCaptureHeader));
		t.dev = (uint8_t*)device_view(t.host);
		if (!t.dev || hipEventCreateWithFlags(&t.ev, hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess) { (void)hipGetLastError(); free_tie_capture(e); return fail(MCS_ERR_HIP, "capture slot: device view / event"); }
	}
	if (hipHostMalloc((void**)&e->h_patchRows, (size_t)max_ties * 2 * e->hd.descSize, hipHostMallocDefault) != hipSuccess ||
	    hipStreamCreateWithFlags(&e->patchStream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); free_tie_capture(e); return fail(MCS_ERR_HIP, "patch stream / staging rows"); }
	e->tieMax = max_ties;
	return MCS_OK;
}

int mcs_extractor_patch_ties(mcs_extractor* e, int back, int* listed, int* recomputed) {
	if (listed) *listed = 0;
	if (recomputed) *recomputed = 0;
	if (!e || back < 0) return fail(MCS_ERR_INVALID, "bad argument");
	if (e->tieRing.empty()) return fail(MCS_ERR_INVALID, "mcs_extractor_set_tie_capture has not been called");
	const long long seq = e->batchSeq - 1 - back;
	if (seq < 0) return MCS_OK;   // no such batch yet (the first steps of a pipeline)
	mcs_extractor::TieSlot& t = e->tieRing[(size_t)(seq % (long long)e->tieRing.size())];
	if (t.seq != seq) return fail(MCS_ERR_INVALID, "that batch's capture slot has been reused (capture depth too small for this lag)");
	HIPCHK(hipSetDevice(e->ctx->device));
	HIPCHK(hipEventSynchronize(t.ev));
	const TieCaptureHeader* hdr = reinterpret_cast<const TieCaptureHeader*>(t.host);
	const int n = hdr->count;
	if (listed) *listed = n;
	if (n <= 0 || t.patched) return MCS_OK;
	if (n > e->tieMax) return fail(MCS_ERR_CAPACITY, "more keypoints inside the tie band than the capture slots hold (mcs_extractor_set_tie_capture max_ties); use mcs_extractor_fix_ties for this band");
	const PyrDesc& hd = e->hd;
	const ExtractBuffers& b = t.b;
	std::vector<std::pair<int, int> > where;
	const int done = recompute_captured(e, t.host, n, t.cams, e->h_patchRows, where);
	if (done < 0) return done;
	for (int i = 0; i < done; ++i) {
		const uint8_t* dsc = e->h_patchRows + (size_t)i * 2 * hd.descSize;
		const size_t drow = ((size_t)where[i].first * b.outImgPitch + where[i].second) * b.outRowStride;
		HIPCHK(hipMemcpyAsync(b.out_desc + drow, dsc, hd.descSize, hipMemcpyHostToDevice, e->patchStream));
		HIPCHK(hipMemcpyAsync(b.out_mask + drow, dsc + hd.descSize, hd.descSize, hipMemcpyHostToDevice, e->patchStream));
	}
	// the rows are in place before this returns: whatever the caller enqueues next (matcher, exchange, download) reads the host's arithmetic
	HIPCHK(hipStreamSynchronize(e->patchStream));
	e->tieFixed += (unsigned long long)done;
	t.patched = true;
	if (recomputed) *recomputed = done;
	return MCS_OK;
}

int mcs_extractor_tie_counts(mcs_extractor* e, uint64_t* listed, uint64_t* recomputed, double* band_px) {
	if (!e) return fail(MCS_ERR_INVALID, "null");
	HIPCHK(hipStreamSynchronize(e->ctx->stream));
	unsigned long long v = 0;
	HIPCHK(hipMemcpy(&v, e->d_fbStats + 1, sizeof(v), hipMemcpyDeviceToHost));
	if (listed) *listed = v;
	if (recomputed) *recomputed = e->tieFixed;
	if (band_px) *band_px = e->tieBand > 0.0 ? e->tieBand : (e->tieBand < 0.0 ? -1.0 : (e->hd.mode == 0 ? kTieBandOrb : kTieBandDistorted));
	return MCS_OK;
}

int mcs_describe_fast_bound(const mcs_ocam* cam, int desc_size, double* bound) {
	if (!cam || !bound || cam->invP_deg < 1 || cam->invP_deg > MCS_MAX_POLY || cam->p_deg < 1) return fail(MCS_ERR_INVALID, "bad argument");
	std::vector<double> tab(kGTabDoubles, 0.0);
	*bound = describe_fast_bound(*cam, 2 * 8 * desc_size, build_g_table(*cam, tab.data()));
	return MCS_OK;
}

int mcs_describe_fast_table(const mcs_ocam* cam, double* table, int* rows, int* row_len, int* e0, int* bins_per_octave, double* info6) {
	if (!cam || cam->invP_deg < 1 || cam->invP_deg > MCS_MAX_POLY || cam->p_deg < 1) return fail(MCS_ERR_INVALID, "bad argument");
	std::vector<double> tab(kGTabDoubles, 0.0);
	const GTabInfo g = build_g_table(*cam, tab.data());
	if (table) memcpy(table, tab.data(), tab.size() * sizeof(double));
	if (rows) *rows = kGRows;
	if (row_len) *row_len = kGRow;
	if (e0) *e0 = kGE0;
	if (bins_per_octave) *bins_per_octave = 1 << kGM;
	if (info6) { info6[0] = g.tailU; info6[1] = g.rhoB; info6[2] = g.dB; info6[3] = g.lip; info6[4] = g.seen; info6[5] = g.inB; }
	return MCS_OK;
}

int mcs_describe_fast_table_packed(const mcs_ocam* cam, void* packed, int* row_bytes, double* f32_term) {
	if (!cam || cam->invP_deg < 1 || cam->invP_deg > MCS_MAX_POLY || cam->p_deg < 1) return fail(MCS_ERR_INVALID, "bad argument");
	std::vector<double> tab(kGTabDoubles, 0.0);
	const GTabInfo g = build_g_table(*cam, tab.data());
	if (packed) pack_g_table(tab.data(), reinterpret_cast<uint8_t*>(packed));
	if (row_bytes) *row_bytes = kGDevRowBytes;
	if (f32_term) *f32_term = MCS_G_PACKED ? g.f32U : 0.0;   // (0: this build's device rows are the doubles themselves)
	return MCS_OK;
}

int mcs_selftest_describe_fast(mcs_ctx* c, const mcs_ocam* cam, uint64_t seed, int n, double* max_abs_diff) {
	if (!c || !cam || !max_abs_diff || n < 1) return fail(MCS_ERR_INVALID, "bad argument");
	if (cam->p_deg < 1 || cam->p_deg > MCS_MAX_POLY || cam->invP_deg < 1 || cam->invP_deg > MCS_MAX_POLY) return fail(MCS_ERR_INVALID, "bad polynomial degree");
	HIPCHK(hipSetDevice(c->device));
	OcamDev o;
	memset(&o, 0, sizeof(o));
	o.c = cam->c; o.d = cam->d; o.e = cam->e; o.u0 = cam->u0; o.v0 = cam->v0; o.invAffine = cam->c - cam->d * cam->e;
	for (int k = 0; k < cam->p_deg; ++k) o.p[k] = cam->p[k];
	for (int k = 0; k < cam->invP_deg; ++k) o.invP[k] = cam->invP[k];
	o.p_deg = cam->p_deg; o.invP_deg = cam->invP_deg; o.fastOk = 1;
	std::vector<double> tab(kGTabDoubles, 0.0);
	if (!std::isfinite(build_g_table(*cam, tab.data()).tailU)) return fail(MCS_ERR_UNSUPPORTED, "the fast pass does not serve this camera");
	uint8_t* buf = nullptr;
	HIPCHK(hipStreamSynchronize(c->stream));
	const size_t tabOff = (64 + sizeof(OcamDev) + 63) / 64 * 64;
	std::vector<double> packed(kGDevDoubles);
	pack_g_table(tab.data(), reinterpret_cast<uint8_t*>(packed.data()));
	HIPCHK(ctx_arena(c, tabOff + kGDevDoubles * sizeof(double), &buf));
	unsigned long long zero = 0, got = 0;
	HIPCHK(hipMemcpy(buf, &zero, sizeof(zero), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(buf + 64, &o, sizeof(o), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(buf + tabOff, packed.data(), kGDevDoubles * sizeof(double), hipMemcpyHostToDevice));
	launch_selftest_fast_model((const OcamDev*)(buf + 64), (const double*)(buf + tabOff), seed, n, cam->width, cam->height, (unsigned long long*)buf, c->stream);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipMemcpy(&got, buf, sizeof(got), hipMemcpyDeviceToHost));
	memcpy(max_abs_diff, &got, sizeof(double));
	return MCS_OK;
}

int mcs_extractor_tap_level(mcs_extractor* e, int img, int level, int blurred, uint8_t* out) {
	if (!e || !out || img < 0 || img >= e->lastN || level < 0 || level >= e->hd.nlevels) return fail(MCS_ERR_INVALID, "bad tap");
	HIPCHK(hipStreamSynchronize(e->ctx->stream));
	const LevelInfo& L = e->hd.lv[level];
	const uint8_t* src; int stride;
	if (blurred) { src = e->d_blur + (size_t)img * e->hd.pyrBytes + L.off; stride = L.stride; }
	else src = level_ptr(e->last, e->hd, img, level, &stride);
	HIPCHK(hipMemcpy2D(out, L.w, src, stride, L.w, L.h, hipMemcpyDeviceToHost));
	return MCS_OK;
}

static int tap_list(mcs_extractor* e, int img, int level, const uint32_t* base, const int* counts, int perImage, int lvBase, uint32_t* out,
                    int cap, int* n) {
	HIPCHK(hipStreamSynchronize(e->ctx->stream));
	int cnt = 0;
	HIPCHK(hipMemcpy(&cnt, counts + (size_t)img * e->hd.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
	*n = cnt;
	const int m = std::min(cnt, cap);
	if (m > 0) HIPCHK(hipMemcpy(out, base + (size_t)img * perImage + lvBase, (size_t)m * sizeof(uint32_t), hipMemcpyDeviceToHost));
	return MCS_OK;
}

int mcs_extractor_tap_candidates(mcs_extractor* e, int img, int level, uint32_t* out, int cap, int* n) {
	if (!e || !out || !n || img < 0 || img >= e->lastN || level < 0 || level >= e->hd.nlevels) return fail(MCS_ERR_INVALID, "bad tap");
	return tap_list(e, img, level, e->d_dense, e->d_denseCount, e->hd.densePerImage, e->hd.lv[level].denseBase, out, cap, n);
}

int mcs_extractor_tap_selected(mcs_extractor* e, int img, int level, uint32_t* out, int cap, int* n) {
	if (!e || !out || !n || img < 0 || img >= e->lastN || level < 0 || level >= e->hd.nlevels) return fail(MCS_ERR_INVALID, "bad tap");
	return tap_list(e, img, level, e->d_sel, e->d_selCount, e->hd.selPerImage, e->hd.lv[level].selBase, out, cap, n);
}

}  // extern "C"

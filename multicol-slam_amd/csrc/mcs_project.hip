// mcs_project.hip — grid-window matchers, SURVEY §8f "next" row 1:
//   cORBmatcher::SearchByProjection(cMultiFrame&, const vector<cMapPoint*>&, th)   src/cORBmatcher.cpp:67-166      (rule 0)
//   cORBmatcher::WindowSearch, SearchByProjection(F1, F2, windowSize, ...)          :326-577                        (rule 1)
//   cORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th)                    :1990-2118                      (rule 2)
//   cORBmatcher::SearchForInitialization                                            :579-726                        (rule 3)
//   cMultiCamSys_::WorldToCamHom_fast + isPointInMirrorMask                         src/cam_system_omni.cpp:92-133, src/cam_model_omni.cpp:163-178
//   cMultiFrame::GetFeaturesInArea                                                src/cMultiFrame.cpp:272-340
//   cMultiFrame::PosInGrid (cvRound binning, bins 64 / 48 dropped)                :342-353
//   cORBmatcher::RadiusByViewingCos                                               src/cORBmatcher.cpp:169-175
// The reference visits the 64x48 grid cells of the window column by column (ix outer, iy inner) and the features of a cell
// in insertion (= mvKeys index) order; with strict '<' updates the winner is the candidate with the smallest
// (distance, cell, index) and the runner-up the next one.  So every (map point, camera) projection gets the sorted short list of its window
// members' smallest keys  dist<<42 | cell<<20 | index  (k_proj_candidates, one wave per projection, fully parallel), and the greedy
// part — a feature taken by an earlier projection is skipped by all later ones — runs as the same speculative wave-parallel
// commit as the brute-force searches (k_proj_greedy).  Pure integer + IEEE double arithmetic: bit-exact.
#include "mcs_common.h"

namespace mcs {

constexpr int kGridCols = 64, kGridRows = 48;   // FRAME_GRID_COLS / ROWS (include/cMultiFrame.h)

struct Window { int minCX, maxCX, minCY, maxCY, minLevel, maxLevel; double x, y, rr, wInv, hInv; bool empty; };

__device__ __forceinline__ Window make_window(const ProjArgs& a, int p) {
	Window w;
	const int cam = a.pcam[p];
	w.x = a.px[p]; w.y = a.py[p];
	if (a.rule == 0) {
		const int lvl = a.level[p];
		double r = a.vcos[p] > 0.998 ? 2.5 : 4.0;
		if (a.th != 1.0) r *= a.th;
		w.rr = r * a.scales[lvl];
		w.minLevel = lvl - 1; w.maxLevel = lvl;
	} else {
		w.rr = a.rad[p]; w.minLevel = a.minLvl[p]; w.maxLevel = a.maxLvl[p];
	}
	w.wInv = static_cast<double>(kGridCols) / static_cast<double>(a.width[cam] - 0);
	w.hInv = static_cast<double>(kGridRows) / static_cast<double>(a.height[cam] - 0);
	w.empty = false;
	int v = (int)floor((w.x - 0 - w.rr) * w.wInv); v = max(0, v); if (v >= kGridCols) w.empty = true; w.minCX = v;
	v = (int)ceil((w.x - 0 + w.rr) * w.wInv); v = min(kGridCols - 1, v); if (v < 0) w.empty = true; w.maxCX = v;
	v = (int)floor((w.y - 0 - w.rr) * w.hInv); v = max(0, v); if (v >= kGridRows) w.empty = true; w.minCY = v;
	v = (int)ceil((w.y - 0 + w.rr) * w.hInv); v = min(kGridRows - 1, v); if (v < 0) w.empty = true; w.maxCY = v;
	return w;
}

// visiting-order part of the key (cell<<20 | index) if feature i is returned by GetFeaturesInArea for this window, else ~0
__device__ __forceinline__ unsigned long long member_key(const ProjArgs& a, const Window& w, int cam, int i) {
	if (a.fcam[i] != cam) return ~0ull;
	const mcs_keypoint kp = a.keys[i];
	const int gx = __double2int_rn((kp.x - 0) * w.wInv), gy = __double2int_rn((kp.y - 0) * w.hInv);   // PosInGrid: cvRound
	if (gx < 0 || gx >= kGridCols || gy < 0 || gy >= kGridRows) return ~0ull;                            // never entered a cell
	if (gx < w.minCX || gx > w.maxCX || gy < w.minCY || gy > w.maxCY) return ~0ull;
	const bool checkLevels = !(w.minLevel == -1 && w.maxLevel == -1), same = checkLevels && w.minLevel == w.maxLevel;
	if (checkLevels && !same) { if (kp.octave < w.minLevel || kp.octave > w.maxLevel) return ~0ull; }
	else if (same) { if (kp.octave != w.minLevel) return ~0ull; }
	if (fabs(kp.x - w.x) > w.rr || fabs(kp.y - w.y) > w.rr) return ~0ull;
	return ((unsigned long long)(gx * kGridRows + gy) << 20) | (unsigned long long)i;
}

__device__ __forceinline__ int proj_distance(const ProjArgs& a, int p, int i) {
	const uint32_t* q = reinterpret_cast<const uint32_t*>(a.pdesc + (size_t)p * a.pstride);
	const uint32_t* t = reinterpret_cast<const uint32_t*>(a.fdesc + (size_t)i * a.fstride);
	int acc = 0;
	if (a.pmask) {
		const uint32_t* qm = reinterpret_cast<const uint32_t*>(a.pmask + (size_t)p * a.pstride);
		const uint32_t* tm = reinterpret_cast<const uint32_t*>(a.fmask + (size_t)i * a.fstride);
		for (int w = 0; w < a.dim / 4; ++w) { const uint32_t x = q[w] ^ t[w]; acc += __popc(x & qm[w]); acc += __popc(x & tm[w]); }
		return acc >> 1;
	}
	for (int w = 0; w < a.dim / 4; ++w) acc += __popc(q[w] ^ t[w]);
	return acc;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
		const unsigned long long y = ((unsigned long long)hi << 32) | lo;
		v = y < v ? y : v;
	}
	return v;
}

// Stage 1, one wave per probe: the kListK smallest keys (distance << 42 | cell << 20 | index, i.e. the reference's visiting order on distance ties) of
// the window's members, ascending, written [kListK][nproj] so that stage 2 reads them coalesced, plus the member count.  Only the best and the
// runner-up among the still-FREE members decide a probe, so a short sorted list is enough unless most of it has been taken meanwhile; then the
// probe is rescanned exactly.  (The first version listed ALL members, up to 128 per probe, and stage 2 walked those lists from global memory in every
// round: 2-6 ms per multi-frame, 50-70 ms once windows of 100 px overflowed the lists.)
constexpr int kListK = kProjListK;
__global__ __launch_bounds__(64) void k_proj_candidates(ProjArgs a) {
	const int p = blockIdx.x, lane = threadIdx.x;
	const unsigned long long NONE = ~0ull;
	const Window w = make_window(a, p);
	const int cam = a.pcam[p];
	unsigned long long loc[kListK];   // this lane's smallest keys, ascending
#pragma unroll
	for (int e = 0; e < kListK; ++e) loc[e] = NONE;
	int n = 0;
	if (!w.empty) {
		for (int i = lane; i < a.nfeat; i += 64) {
			const unsigned long long mk = member_key(a, w, cam, i);
			if (mk == NONE) continue;
			++n;
			const unsigned long long key = ((unsigned long long)proj_distance(a, p, i) << 42) | mk;
			if (key < loc[kListK - 1]) {
				loc[kListK - 1] = key;
#pragma unroll
				for (int e = kListK - 1; e > 0; --e) {
					const unsigned long long lo = loc[e] < loc[e - 1] ? loc[e] : loc[e - 1], hi = loc[e] < loc[e - 1] ? loc[e - 1] : loc[e];
					loc[e - 1] = lo; loc[e] = hi;
				}
			}
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
#pragma unroll
	for (int e = 0; e < kListK; ++e) {   // the wave's e-th smallest key: keys are unique (index bits), exactly one lane owns it and pops it
		const unsigned long long m = wave_min_u64(loc[0]);
		if (lane == 0) a.lists[(size_t)e * a.nproj + p] = m;
		const bool mine = loc[0] == m && m != NONE;
#pragma unroll
		for (int j = 0; j < kListK - 1; ++j) loc[j] = mine ? loc[j + 1] : loc[j];
		loc[kListK - 1] = mine ? NONE : loc[kListK - 1];
	}
	if (lane == 0) a.counts[p] = n;
}

__global__ __launch_bounds__(64) void k_proj_greedy(ProjArgs a) {
	__shared__ uint32_t taken[65536 / 32];   // frame features <= 65536
	constexpr int kClaimHash = 4096;
	__shared__ uint32_t claimH[kClaimHash];
	for (int i = threadIdx.x; i < kClaimHash; i += 64) claimH[i] = 0xFFFFFFFFu;
	const int lane = threadIdx.x;
	const bool steal = a.rule == 3;
	if (steal) {
		for (int i = lane; i < a.nfeat; i += 64) { a.owner[i] = -1; a.mdist[i] = 0x7FFFFFFF; }
		__threadfence_block();
	} else {
		for (int i = lane; i < (a.nfeat + 31) / 32; i += 64) taken[i] = 0;
		__syncthreads();
		for (int i = lane; i < a.nfeat; i += 64) if (a.assigned[i]) atomicOr(&taken[i >> 5], 1u << (i & 31));
	}
	__syncthreads();
	// is frame feature idx (at distance dist from the probe) still a candidate?
	auto free_for = [&](int idx, int dist) -> bool {
		if (steal) return a.mdist[idx] > dist;                  // "if (vMatchedDistance[i2] <= dist) continue" (:652)
		return !((taken[idx >> 5] >> (idx & 31)) & 1u);         // "if (vpMapPointMatches2[i2]) continue"
	};
	int nmatches = 0;
	const unsigned long long NONE = ~0ull;
	for (int p0 = 0; p0 < a.nproj; p0 += 64) {
		const int p = p0 + lane;
		bool resolved = p >= a.nproj;
		const int cnt = resolved ? 0 : a.counts[p];
		if (!resolved) { a.match[p] = -1; if (a.accepted) a.accepted[p] = -1; }
		if (!resolved && cnt == 0) resolved = true;
		unsigned long long key[kListK];   // my probe's sorted short list (registers for all rounds)
#pragma unroll
		for (int e = 0; e < kListK; ++e) key[e] = resolved ? NONE : a.lists[(size_t)e * a.nproj + p];
		const bool full = cnt > kListK;                                    // members beyond the list exist; their distance is >= dK
		const int dK = full ? (int)(key[kListK - 1] >> 42) : 0x7FFFFFFF;
		// verdict from best / runner-up among the free members; `bound` = the runner-up is unknown but at least that far away
		auto decide = [&](unsigned long long k1, unsigned long long k2, bool bound, int& state, int& bestIdx, int& secondIdx, int& best) {
			state = 0; bestIdx = -1; secondIdx = -1; best = 0;
			if (k1 == NONE) return;
			best = (int)(k1 >> 42);
			bestIdx = (int)(k1 & 0xFFFFFu);
			int second = 0x7FFFFFFF;
			if (bound) second = dK;
			else if (k2 != NONE) { second = (int)(k2 >> 42); secondIdx = (int)(k2 & 0xFFFFFu); }
			if (a.rule == 0) {          // :153-163
				const int lvl2 = secondIdx >= 0 ? a.keys[secondIdx].octave : -1;
				const int lvl1 = a.keys[bestIdx].octave;
				const bool ratioFails = static_cast<double>(best) > a.ratio * static_cast<double>(second);
				if (best <= a.thHigh && !((bound || lvl1 == lvl2) && ratioFails)) state = 1;
				else if (bound && best <= a.thHigh) state = 2;       // only a real runner-up (its level, its distance) can tell
			} else if (a.rule == 1) {   // :416, :558-560
				if (static_cast<double>(best) <= static_cast<double>(second) * a.ratio && best <= a.thHigh) state = 1;
				else if (bound && best <= a.thHigh) state = 2;
			} else if (a.rule == 2) {   // :2072 (no runner-up)
				secondIdx = -1;
				if (best <= a.thHigh) state = 1;
			} else {                    // :670-672
				if (best <= a.thLow && static_cast<double>(best) < static_cast<double>(second) * a.ratio) state = 1;
				else if (bound && best <= a.thLow) state = 2;
			}
		};
		for (int round = 0; round < 130; ++round) {
			const unsigned long long pend = __ballot(!resolved);
			if (pend == 0ull) break;
			const int low = __ffsll((long long)pend) - 1;
			// tentative decision from the first two FREE entries of my list
			int state = 0, bestIdx = -1, secondIdx = -1, best = 0;   // 0 no match, 1 match, 2 needs an exact rescan
			if (!resolved) {
				unsigned long long k1 = NONE, k2 = NONE;
#pragma unroll
				for (int e = 0; e < kListK; ++e) {
					const unsigned long long k = key[e];
					if (k != NONE && k2 == NONE && free_for((int)(k & 0xFFFFFu), (int)(k >> 42))) { if (k1 == NONE) k1 = k; else k2 = k; }
				}
				if (k2 != NONE || !full || (a.rule == 2 && k1 != NONE)) decide(k1, k2, false, state, bestIdx, secondIdx, best);
				else if (k1 != NONE) decide(k1, NONE, true, state, bestIdx, secondIdx, best);   // one free entry: the runner-up is hidden, >= dK away
				else state = dK <= (a.rule == 3 ? a.thLow : a.thHigh) ? 2 : 0;                   // none: a hidden member could still qualify
			}
			// the lowest pending probe is rescanned exactly by the whole wave if its list cannot decide
			if (__shfl(state, low) == 2) {
				const int pp = p0 + low;
				const Window w = make_window(a, pp);
				const int cam = a.pcam[pp];
				unsigned long long b1 = NONE, b2 = NONE;
				for (int i = lane; i < a.nfeat; i += 64) {
					const unsigned long long mk = member_key(a, w, cam, i);
					if (mk == NONE) continue;
					const int dist = proj_distance(a, pp, i);
					if (!free_for(i, dist)) continue;
					const unsigned long long k = ((unsigned long long)dist << 42) | mk;
					if (k < b1) { b2 = b1; b1 = k; } else if (k < b2) b2 = k;
				}
				const unsigned long long m1 = wave_min_u64(b1);
				const unsigned long long m2 = wave_min_u64(b1 == m1 ? b2 : b1);
				if (lane == low) decide(m1, m2, false, state, bestIdx, secondIdx, best);
			}
			// finality: no lower pending lane of this round may take my best or my second feature.  Accepting lanes publish their target in a hashed
			// LDS claim table (atomicMin of the lane id); a collision of two different features can only block a lane that was free to go, which
			// costs a round but never changes a result, and the lowest pending lane has no claim below it.  (One shuffle per accepting lane before.)
			const bool claims = !resolved && state == 1;
			if (claims) atomicMin(&claimH[bestIdx & (kClaimHash - 1)], (uint32_t)lane);
			__syncthreads();
			bool blocked = !resolved && state == 2;   // waits until it is the lowest pending probe
			if (!resolved && state != 2) {
				if (bestIdx >= 0 && claimH[bestIdx & (kClaimHash - 1)] < (uint32_t)lane) blocked = true;
				if (secondIdx >= 0 && claimH[secondIdx & (kClaimHash - 1)] < (uint32_t)lane) blocked = true;
			}
			__syncthreads();
			if (claims) claimH[bestIdx & (kClaimHash - 1)] = 0xFFFFFFFFu;
			const unsigned long long blk = __ballot(blocked);
			const int firstBlocked = blk ? __ffsll((long long)blk) - 1 : 64;
			const bool commit = !resolved && lane < firstBlocked;
			bool stole = false;
			if (commit) {
				if (state == 1) {
					if (steal) {            // :674-681
						const int prev = a.owner[bestIdx];
						if (prev >= 0) { a.match[prev] = -1; stole = true; }
						a.owner[bestIdx] = p; a.mdist[bestIdx] = best;
						if (a.accepted) a.accepted[p] = bestIdx;   // also kept if the match is stolen later (rotation histogram, :686-694)
					} else {
						atomicOr(&taken[bestIdx >> 5], 1u << (bestIdx & 31));
						a.assigned[bestIdx] = 1;
					}
					a.match[p] = bestIdx;
				}
				resolved = true;
			}
			nmatches += __popcll(__ballot(commit && state == 1)) - __popcll(__ballot(stole));
			__threadfence_block();
			__syncthreads();
		}
	}
	if (lane == 0) *a.nmatches = nmatches;
}

// ---------------------------------------------------------------------------------------------- WorldToCamHom_fast
// include/misc.h:115-122 (zero-padded fixed-length form, see mcs_describe.hip)
__device__ __forceinline__ double horner_fixed(const double* coeffs, double x) {
	double res = 0.0;
#pragma unroll
	for (int i = MCS_MAX_POLY - 1; i >= 0; i--) res = res * x + coeffs[i];
	return res;
}

__global__ __launch_bounds__(64) void k_world_to_cam(WorldToCamArgs a) {
	const int i = blockIdx.x * 64 + threadIdx.x;
	if (i >= a.n) return;
	const int c = a.pcam[i];
	const double* M = a.M + 16 * (size_t)c;
	const double pt4[4] = {a.pts[3 * (size_t)i], a.pts[3 * (size_t)i + 1], a.pts[3 * (size_t)i + 2], 1.0};
	double r[4];
#pragma unroll
	for (int row = 0; row < 4; ++row) {   // cv::Matx product: s = 0; s += a(i,k) * b(k)
		double s = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) s += M[4 * row + k] * pt4[k];
		r[row] = s;
	}
	const OcamDev& cam = a.cams[c];
	double norm = sqrt(r[0] * r[0] + r[1] * r[1]);   // cCamModelGeneral_::WorldToImg (src/cam_model_omni.cpp:146-161)
	if (norm == 0.0) norm = 1e-14;
	const double theta = atan(-r[2] / norm);
	const double rho = horner_fixed(cam.invP, theta);
	const double uu = r[0] / norm * rho;
	const double vv = r[1] / norm * rho;
	const double u = uu * cam.c + vv * cam.d + cam.u0;
	const double v = uu * cam.e + vv + cam.v0;
	a.uv[2 * (size_t)i] = u; a.uv[2 * (size_t)i + 1] = v;
	const int ur = __double2int_rn(u), vr = __double2int_rn(v);
	unsigned fl = 0;
	const int W = a.width[c], H = a.height[c];
	if (!(ur >= W || ur <= 0 || vr >= H || vr <= 0)) {
		const uint8_t* m = a.masks ? a.masks[c] : nullptr;
		if (!m || m[(size_t)vr * W + ur] > 0) fl |= 1u;
	}
	if (r[2] <= 0.0) fl |= 2u;
	a.flags[i] = (uint8_t)fl;
}

void launch_world_to_cam(const WorldToCamArgs& a, hipStream_t s) {
	if (a.n > 0) hipLaunchKernelGGL(k_world_to_cam, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
}

// Independent best-in-window (no feature is ever "taken"): the search loops of Fuse (:1265-1719), SearchBySim3 (:1721-1988),
// SearchForTriangulationBetweenCameras (:1158-1263) and SearchByProjection(pKF, Scw, ...) (:2265-2392).  One wave per probe, the
// smallest (dist, cell, index) key is the reference's strict-'<' winner; accepted if dist <= thHigh.
__global__ __launch_bounds__(64) void k_proj_best(ProjArgs a, int* outDist) {
	const int p = blockIdx.x, lane = threadIdx.x;
	const Window w = make_window(a, p);
	const int cam = a.pcam[p];
	unsigned long long best = ~0ull;
	if (!w.empty) {
		for (int i = lane; i < a.nfeat; i += 64) {
			const unsigned long long mk = member_key(a, w, cam, i);
			if (mk == ~0ull) continue;
			const unsigned long long k = ((unsigned long long)proj_distance(a, p, i) << 42) | mk;
			best = k < best ? k : best;
		}
	}
	best = wave_min_u64(best);
	if (lane == 0) {
		const bool found = best != ~0ull;
		const int d = found ? (int)(best >> 42) : 0x7FFFFFFF;
		const bool ok = found && d <= a.thHigh;
		a.match[p] = ok ? (int)(best & 0xFFFFFu) : -1;
		if (outDist) outDist[p] = d;
		if (ok) atomicAdd(a.nmatches, 1);
	}
}

// distances of the accepted matches of a greedy pass (mcs_window_best with skip_taken)
__global__ __launch_bounds__(64) void k_proj_fill_dist(ProjArgs a, int* outDist) {
	const int p = blockIdx.x * 64 + threadIdx.x;
	if (p >= a.nproj) return;
	const int j = a.match[p];
	outDist[p] = j >= 0 ? proj_distance(a, p, j) : 0x7FFFFFFF;
}

// ---------------------------------------------------------------------------------------------- mbCheckOrientation
// Rotation-consistency filter of the searches (ComputeThreeMaxima, src/cORBmatcher.cpp:2394-2436): 30-bin histogram of
// rot = angle_first - angle_second (+360 if negative) over the accepted matches, matches outside the three fullest bins are dropped.
// The four bin-arithmetic variants of the reference are listed in include/mcs_c.h (mcs_rotation_consistency).  One workgroup.
__device__ __forceinline__ int rot_bin(int variant, float aFirst, float aSecond) {
	float rot = aFirst - aSecond;
	if (variant == 3) { if (rot < 0.0f) rot = (float)((double)rot + 360.0); }
	else if (rot < 0.0f) rot += 360.0f;
	int bin;
	if (variant == 0) { const float factor = 1.0f / 30; bin = __double2int_rn((double)(rot * factor)); }
	else if (variant == 1) { const double factor = (double)(1.0f / 30); bin = __double2int_rn((double)rot * factor); }
	else if (variant == 2) { const double factor = 1.0 / 30; bin = (int)round((double)rot * factor); }
	else { const double factor = 1.0 / 30; bin = __double2int_rn((double)rot * factor); }
	return bin == 30 ? 0 : bin;
}

__global__ __launch_bounds__(256) void k_rotation_consistency(int variant, const float* angleSlot, int strideSlot, const float* anglePartner, int stridePartner,
                                                              const int* accepted, int* match, int n, int swapped, int* removedOut) {
	__shared__ int hist[30];
	__shared__ int keep[3];
	__shared__ int removed;
	const int tid = threadIdx.x;
	if (tid < 30) hist[tid] = 0;
	if (tid == 0) removed = 0;
	__syncthreads();
	auto angle = [](const float* base, int strideBytes, int i) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)i * strideBytes); };
	for (int s = tid; s < n; s += 256) {
		const int p = accepted ? accepted[s] : match[s];
		if (p < 0) continue;
		const float a = angle(angleSlot, strideSlot, s), b = angle(anglePartner, stridePartner, p);
		atomicAdd(&hist[rot_bin(variant, swapped ? b : a, swapped ? a : b)], 1);
	}
	__syncthreads();
	if (tid == 0) {   // ComputeThreeMaxima
		int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
		for (int i = 0; i < 30; i++) {
			const int sz = hist[i];
			if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
			else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
			else if (sz > max3) { max3 = sz; ind3 = i; }
		}
		if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
		else if ((float)max3 < 0.1f * (float)max1) ind3 = -1;
		keep[0] = ind1; keep[1] = ind2; keep[2] = ind3;
	}
	__syncthreads();
	for (int s = tid; s < n; s += 256) {
		const int p = accepted ? accepted[s] : match[s];
		if (p < 0) continue;
		const float a = angle(angleSlot, strideSlot, s), b = angle(anglePartner, stridePartner, p);
		const int bin = rot_bin(variant, swapped ? b : a, swapped ? a : b);
		if (bin != keep[0] && bin != keep[1] && bin != keep[2] && match[s] >= 0) { match[s] = -1; atomicAdd(&removed, 1); }
	}
	__syncthreads();
	if (tid == 0) *removedOut = removed;
}

void launch_rotation_consistency(int variant, const float* angleSlot, int strideSlot, const float* anglePartner, int stridePartner, const int* accepted, int* match,
                                 int n, int swapped, int* removedOut, hipStream_t s) {
	hipLaunchKernelGGL(k_rotation_consistency, dim3(1), dim3(256), 0, s, variant, angleSlot, strideSlot, anglePartner, stridePartner, accepted, match, n, swapped, removedOut);
}

void launch_proj_candidates(const ProjArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_proj_candidates, dim3(a.nproj), dim3(64), 0, s, a); }
void launch_proj_greedy(const ProjArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_proj_greedy, dim3(1), dim3(64), 0, s, a); }
void launch_projection(const ProjArgs& a, hipStream_t s) {
	launch_proj_candidates(a, s);
	launch_proj_greedy(a, s);
}

void launch_window_best(const ProjArgs& a, bool skipTaken, int* outDist, hipStream_t s) {
	if (skipTaken) {
		launch_projection(a, s);   // rule 2 with thHigh = the caller's threshold
		if (outDist) hipLaunchKernelGGL(k_proj_fill_dist, dim3((a.nproj + 63) / 64), dim3(64), 0, s, a, outDist);
	} else {
		(void)hipMemsetAsync(a.nmatches, 0, sizeof(int), s);
		hipLaunchKernelGGL(k_proj_best, dim3(a.nproj), dim3(64), 0, s, a, outDist);
	}
}

}  // namespace mcs

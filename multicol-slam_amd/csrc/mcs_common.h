// mcs_common.h — shared definitions of the gfx950 feature front end (device tables, launch prototypes).
// Written for CDNA4 only (wave64, 160 KiB LDS/CU, 8 XCDs); no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <stdint.h>
#include "../../include/mcs_c.h"

namespace mcs {

constexpr int kEdge = 25;          // EDGE_THRESHOLD   (reference src/mdBRIEFextractorOct.cpp:86)
constexpr int kHalfPatch = 16;     // HALF_PATCH_SIZE  (:85)
constexpr int kPatchSize = 32;     // PATCH_SIZE       (:84)
constexpr int kMinBorder = kEdge - 3;  // 22 (:876)
constexpr int kCellW = 30;         // W (:868)
constexpr int kMaxRoots = 32;
constexpr int kMaxNodes = 2048;    // oct-tree node capacity per (image, level): nfeatures_level + 3 must fit (k_octree<2048, ...>: 140 of the CU's 160 KB of LDS)
constexpr int kNumXCD = 8;

// One pyramid level of one image (identical for every image of the batch).
struct LevelInfo {
	int w, h, stride;       // size and row pitch in bytes (pitch is a multiple of 64)
	int off;                // byte offset inside the per-image pyramid block (level 0 of the UNBLURRED pyramid may alias the input)
	// FAST cell grid (reference :884-890)
	int nCols, nRows, wCell, hCell;
	int cellBase;           // first cell of this level in the per-image cell list
	int capc;               // candidate slots per cell  = ceil(wCell/2)*ceil(hCell/2)  (strict local maxima cannot be 8-adjacent); AGAST: half the cell's pixels (survivors are not 4-adjacent)
	int slotBase;           // first candidate slot of this level in the per-image slot array
	int denseBase, denseCap;// dense (compacted, ordered) candidate list of this level
	int nfeat;              // mnFeaturesPerLevel[level]
	int selBase, selCap;    // selected-keypoint region (nfeat + 3 slots)
	int nIni;               // oct-tree roots = cvRound(width/height)
	int rootX[kMaxRoots + 1];
	double hX;
	int tabX, tabY;         // offsets of this level's resize tables (from level-1), in entries
	int colsOk;             // the column-marching resize kernel serves this level (four adjacent columns read <= 8 adjacent source bytes)
	int mapX, mapY;         // offsets of this level's composed nearest-neighbour maps to level-0 mask coordinates
	float scale;            // (float)mvScaleFactor[level]
	float kpSize;           // (float)(int)(PATCH_SIZE*mvScaleFactor[level])
};

struct PyrDesc {
	int nlevels;
	int width, height;
	int pyrBytes;           // per-image pyramid block size
	int cellsPerImage, slotsPerImage, densePerImage, selPerImage;
	int kpCap;              // output rows per image
	int fastThreshold;
	int fastRing;           // 16 / 12 / 8: FastFeatureDetector TYPE_9_16 / TYPE_7_12 / TYPE_5_8
	int agast;              // -1: FAST; 0 / 1 / 2 / 3: AgastFeatureDetector AGAST_5_8 / AGAST_7_12d / AGAST_7_12s / OAST_9_16 (useAgast)
	int descSize, npoints;
	int umax[kHalfPatch + 1];   // half-width of orientation-disc row |v| (reference :187-202)
	int chainFits;          // the one-launch resize chain (k_resize_chain) serves this geometry
	int chainRegOff;        // its region table [tile][8 levels][8 ints], appended to the tap array (offset in tap entries)
	int mode;               // 0 ORB, 1 dBRIEF, 2 mdBRIEF
	int undistort;          // do_dBrief (reference gates undistortion on it only, Appendix B.1)
	LevelInfo lv[MCS_MAX_LEVELS];
};

struct CellInfo { short level; short x0, y0; short cw, ch; short pad; int slot; int rowM, grpM; };  // processed region [x0,x0+cw) x [y0,y0+ch) in level ROI coords;
// rowM = ceil(2^16 / dwords per staged tile row), grpM = ceil(2^16 / 4-pixel groups per row): the FAST kernel's divisions by multiplication

// bilinear resize tables (Appendix A.1), one entry per destination column / row of a level
struct ResizeTap { short ofs; short a0; short a1; short pad; };

// device-side camera model (double precision, Scaramuzza)
struct OcamDev {
	double c, d, e, u0, v0, invAffine;
	double p[MCS_MAX_POLY];
	double invP[MCS_MAX_POLY];
	int p_deg, invP_deg;
	// fast pass of the descriptor kernel (mcs_describe.hip): whether this camera's worst-case arithmetic difference stays below the guard band
	int fastOk;
	int tabIdx;             // this camera's G table among the batch's distinct tables (ExtractBuffers.gTab)
};

// Per-camera table of G(s) = rho(theta(s)) / sqrt(s), s = x^2 + y^2 the squared norm of an undistorted pattern point, theta = atan(p0 / sqrt(s)), for the
// fast descriptor pass: u, v = affine(x * G, y * G) needs neither the square root, the reciprocal nor atan.  Log-spaced bins: row = exponent and top kGM
// mantissa bits of s (read from the bit pattern), s in [2^kGE0, 2^kGE1); a row holds the degree-kGDeg Taylor coefficients of G about the bin centre in the
// variable tau = (low mantissa fraction of s) - 2^-(kGM+1), which is exact.  Built by build_g_table() in mcs_capi.hip, which also bounds the truncated tail.
#ifndef MCS_G_M
#define MCS_G_M 5     // 32 bins per octave, degree 6: 56-byte rows, 54 KB (64 bins, degree 5: 48-byte rows, 0.675 against 0.696 ms, but 92 KB for the same range; 16 bins, degree 8: 0.753 ms)
#define MCS_G_DEG 6
#endif
#ifndef MCS_G_E0
#define MCS_G_E0 (-10)
#define MCS_G_E1 24
#endif
constexpr int kGM = MCS_G_M, kGDeg = MCS_G_DEG, kGE0 = MCS_G_E0, kGE1 = MCS_G_E1, kGRows = (kGE1 - kGE0) << kGM, kGRow = kGDeg + 1, kGTabDoubles = kGRows * kGRow;
// ... as the HOST builds it (kGRow doubles per row).  The DEVICE reads a packed form (round 6): 48-byte rows in three 16-byte slots, [g0 g1] [g2 g3] doubles and
// [g4 g5 g6 0] floats — three ds_read_b128 per row from LDS instead of seven 8-byte loads; the float tail's roundings are bounded by GTabInfo.f32U (mcs_capi.hip).
#ifndef MCS_G_PACKED
#define MCS_G_PACKED 1   // 1: the packed rows (three ds_read_b128 per gather: 27 % fewer LDS-array cycles per keypoint).  0 (A/B): rows of kGRow doubles, which the compiler reads as
#endif                   // three ds_read2_b64 + one ds_read_b64 (half rate).  Round 6: 0.456 against 0.493 ms once the kernel no longer spills (profiles/NOTES.md).
static_assert(!MCS_G_PACKED || kGDeg == 6, "the packed device rows hold degree 6");
constexpr int kGDevRowBytes = MCS_G_PACKED ? 48 : kGRow * 8, kGDevDoubles = kGRows * kGDevRowBytes / 8;
// The table starts at s = 2^kGE0, i.e. 1/32 pixel from the optical axis: G has a sqrt-type branch point at s = 0 (rho(theta) of a fitted backward polynomial
// does not vanish exactly on the axis), so only log-spaced bins reach down there.  One keypoint in 200 has the axis inside its pattern's footprint, and of
// those one in 300 a point within 1/32 px of it: that keypoint takes the exact pass.  (Starting the table at s = 16 sent 1 % of all keypoints there, at
// s = 2^-6 still 60 per 193 000.)
#ifndef MCS_FAST_WAVES
#define MCS_FAST_WAVES 16   // waves per workgroup of the fast descriptor pass (mcs_describe.hip): they share the camera's table in LDS
#endif
constexpr int kSlotAlign = MCS_FAST_WAVES > 8 ? MCS_FAST_WAVES : 8;       // keypoint slots per image are a multiple of this (the fast pass walks groups of kFastWaves keypoints of ONE image, a wave each)

// Per-keypoint scratch of the descriptor passes (mcs_describe.hip), one array per field over all keypoint slots of the batch (thread-per-keypoint kernels
// write full cache lines).  lvl: -1 no keypoint, else level | kAuxExact if the keypoint is on the exact pass's list already.
constexpr int kAuxExact = 0x100;
struct KpAuxSoA {
	int* lvl; int* rc;                  // level (bits 0..7) | flags | row stride of the blurred level << 16;  row | col << 16
	unsigned* poff;                     // byte offset of the patch origin (row - R, col - R) inside the image's blurred pyramid
	double* d8;                         // [8][slots]: undistorted keypoint x, y; cos, sin of the (up to) three pattern angles
	int slots;
	__host__ __device__ static size_t bytes_per_slot() { return 3 * sizeof(int) + 8 * sizeof(double); }
	__host__ __device__ void carve(void* base, int nslots) {   // nslots is a multiple of kSlotAlign
		slots = nslots;
		d8 = reinterpret_cast<double*>(base);
		lvl = reinterpret_cast<int*>(d8 + (size_t)8 * nslots); rc = lvl + nslots; poff = reinterpret_cast<unsigned*>(rc + nslots);
	}
};
struct ExtractBuffers {
	const PyrDesc* desc;          // device copy
	const CellInfo* cells;        // [cellsPerImage]
	const ResizeTap* taps;        // x and y tables, all levels
	const short* maskMap;         // composed NN maps, all levels
	const uint8_t* img0; size_t img0Pitch; int img0Stride;       // level 0 (input images, device)
	const uint8_t* mask0; size_t mask0Pitch; int mask0Stride;    // level-0 masks or nullptr
	uint8_t* pyr;                 // [B][pyrBytes] unblurred levels 1.. (level 0 region unused when aliasing the input)
	uint8_t* blur;                // [B][pyrBytes] blurred levels 0..
	uint32_t* slots;              // [B][slotsPerImage] per-cell candidate slots  (x | y<<12 | score<<24, border-relative coords)
	int* cellCount;               // [B][cellsPerImage]
	uint32_t* dense;              // [B][densePerImage] ordered candidates per level
	unsigned short* knode;        // [B][densePerImage] oct-tree scratch: list position of the node holding the key
	int* denseCount;              // [B][nlevels]
	uint32_t* sel;                // [B][selPerImage] selected keys in final list order
	int* selCount;                // [B][nlevels]
	float* selAngle;              // [B][selPerImage] orientation of the selected keys (degrees), written by the oct-tree kernel
	const OcamDev* cams;          // [B] or nullptr
	int* status;                  // device error word (capacity overflows)
	// dBRIEF / mdBRIEF: fast pass + exact pass over the fast pass's fallback list (mcs_describe.hip)
	const double* gTab;                  // [distinct cameras of the batch][kGDevDoubles] packed rows, indexed by OcamDev.tabIdx
	void* aux;                           // KpAuxSoA over [B][roundup(kpCap, kSlotAlign)] slots: orientation / undistorted keypoint / pattern angles for the fast pass
	int* fbCount; uint32_t* fbList;      // keypoint slots (image * wavesPerImage + slot) the fast pass handed to the exact pass, this batch
	int* preCount; uint32_t* preList;    // ... and the ones k_orient_b sent there before the fast pass ran (camera not served, keypoint next to the optical axis)
	unsigned long long* fbStats;         // running total of both (all batches of the extractor)
	unsigned long long* tieMin;          // bits of the smallest distance to a rounding tie (| |frac| - 1/2 |, pixels) seen among the cvRound arguments of the EXACT arithmetic
	// keypoints of this batch whose EXACT arithmetic had a cvRound argument within tieBand of a tie: the host recomputes them with its own libm (mcs_tiefix.hip)
	int* tieCount; uint32_t* tieList;    // slots (image * wavesPerImage + slot); tieCount = fbCount + 2, cleared with its neighbours by k_octree
	unsigned long long* tieTotal;        // running total (all batches of the extractor)
	double tieBand;                      // pixels; < 0: nothing is listed
	hipStream_t sideStream; hipEvent_t evDescFork, evDescJoin;   // optional: the exact pass over preList runs here, beside the fast pass
	hipEvent_t evFastA, evFastB;         // optional (per-kernel timing passes only): recorded around k_describe_fast / k_describe<0> alone ("describe_fast")
	double guardEps;                     // half-width of the guard band around the rounding ties
	int describeMode;                    // 0 fast + exact fallback, 1 exact pass for every keypoint
	// outputs
	int* nkp; mcs_keypoint* kps; uint8_t* out_desc; uint8_t* out_mask; double* rays;
	size_t outImgPitch; int outRowStride;   // descriptor / mask row k of image i at (i * outImgPitch + k) * outRowStride (default kpCap rows, descSize bytes)
};

// Host recomputation of rounding ties (mcs_tiefix.hip).  HostLevel: either the whole level (tight rows, stride = w, blurred and unblurred, both the size of the
// level's ROI) or — the pipelined form, k_tie_capture — a square window of Sampler::at values around the keypoint: patch[(r - prow) * pdim + (c - pcol)],
// `miss` set when a sample falls outside it.
struct HostLevel { const uint8_t* blur; const uint8_t* raw; int w, h; const uint8_t* patch = nullptr; int prow = 0, pcol = 0, pdim = 0; bool* miss = nullptr; };
// What k_tie_capture leaves in page-locked memory per batch: a header and up to max_ties entries (slot, level, selected-key record, angle, window of samples)
constexpr int kTiePatchR = 40;
constexpr int kTiePatchDim = 2 * kTiePatchR + 1;
struct TieCaptureHeader { int count; int status; int pad[14]; };
struct TieCaptureEntry { uint32_t gw; int level; uint32_t rec; float angle; uint8_t patch[kTiePatchDim * kTiePatchDim + 15]; };   // 16 + 6576 bytes

__host__ __device__ inline const uint8_t* level_ptr(const ExtractBuffers& b, const PyrDesc& d, int img, int level, int* stride) {
	if (level == 0) { *stride = b.img0Stride; return b.img0 + (size_t)img * b.img0Pitch; }
	*stride = d.lv[level].stride;
	return b.pyr + (size_t)img * d.pyrBytes + d.lv[level].off;
}

// kernel launchers (each enqueues on `s`)
void launch_pyramid(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s, int level0 = 1, int level1 = MCS_MAX_LEVELS);
bool pyramid_chain_table(const PyrDesc& d, const ResizeTap* taps, std::vector<int>* table);
void launch_fast(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s, int level0 = 0, int level1 = MCS_MAX_LEVELS);
void launch_octree(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s, int level0 = 0, int level1 = MCS_MAX_LEVELS);
void launch_blur(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s);
void launch_describe(const ExtractBuffers& b, const PyrDesc& hd, int nimg, hipStream_t s);
size_t describe_aux_bytes();   // scratch bytes per keypoint slot

// Row i of a descriptor set -> row of the caller's array.  A set is either contiguous (blk = 0) or made of blocks of `blk` rows lying `bpitch` rows
// apart (mcs_desc_set.block_rows / block_pitch_rows: the cameras of one multi-frame inside a gathered [camera][frame][row] buffer).
struct RowMap {
	size_t base; int blk; size_t bpitch;
	__device__ __forceinline__ size_t operator()(int i) const {
		if (blk == 0) return base + (size_t)i;
		const int b = i / blk;
		return base + (size_t)b * bpitch + (size_t)(i - b * blk);
	}
};

struct MatchArgs {
	const uint8_t* qd; const uint8_t* qm; const uint8_t* qvalid; const int* qgroup; int nq; int qstride; size_t qpitch;
	const uint8_t* td; const uint8_t* tm; const uint8_t* tvalid; const int* tgroup; int nt; int tstride; size_t tpitch;
	int nsets; int dim; int K; int countThresh;
	int qblk, tblk; size_t qbpitch, tbpitch;   // block structure of a set's rows (RowMap), 0 = contiguous
	int toff, tmod;            // train set of pair s = ((s / tdiv) + toff) % tmod: a ring of multi-frames, each against its predecessor (toff = 0, tmod = INT_MAX otherwise)
	int qmod, tdiv;            // set s reads query set (s % qmod) and train set (s / tdiv): plain batches qmod = nsets, tdiv = 1; a database sweep of
	                           // nkf keyframes x nframes frames (s = f*nkf + k) qmod = tdiv = nkf
	int maxDist;               // rows farther than this never enter a list (INT_MAX for plain top-K)
	int splits;                // train-range splits per (set, query tile)
	uint32_t* partial;         // [nsets][splits][K][nq] packed (dist<<20 | idx), ascending in K
	int* partialCount;         // [nsets][splits][nq]
	uint32_t* keys;            // [nsets][K][nq] final packed lists (0xFFFFFFFF = empty)
	int* outDist; int* outIdx; // optional public [nsets][nq][K] form
	int* outCount;
	// mcs_match_mfma.hip: the train sets, compacted and expanded to matrix-core operands once per call (nullptr: the v_bcnt kernel serves)
	uint4* exA; float* exW; int* exRows; int exStages; int tsets;   // tsets distinct train sets, exStages 64-row stages reserved per set
	int exDone;                // the caller has launched launch_match_expand itself
};
void launch_match(const MatchArgs& a, hipStream_t s);
bool match_mfma_serves(const MatchArgs& a);
void launch_match_expand(const MatchArgs& a, hipStream_t s);
bool match_mfma_shape(const MatchArgs& a);
void match_mfma_scratch(const MatchArgs& a, int tsets, size_t* bytesA, size_t* bytesW, int* stages);

// greedy, order-dependent resolution of the reference's brute-force searches on top of the top-K lists
struct GreedyArgs {
	const uint8_t* qd; const uint8_t* qm; const uint8_t* qvalid; const int* qgroup; int nq; int qstride; size_t qpitch;
	const uint8_t* td; const uint8_t* tm; const uint8_t* tvalid; const int* tgroup; int nt; int tstride; size_t tpitch;
	int nsets; int dim; int K;
	int qblk, tblk; size_t qbpitch, tbpitch; // as in MatchArgs
	int toff, tmod;                          // as in MatchArgs
	int qmod, tdiv;                          // as in MatchArgs
	const uint32_t* keys;                    // [nsets][K][nq] packed (dist<<20 | idx) ascending in K, 0xFFFFFFFF = empty
	int thLow; int thInclusive; double ratio;
	int mode;                                // 0 SearchByBoW(KF,KF)  1 SearchByBoW(KF,F)  2 SearchForTriangulationRaw
	const double* rays1; const double* rays2; const double* E; int nrCams;
	size_t Epitch;                           // doubles between the essential-matrix blocks of consecutive sets (0: one block for all sets)
	int* outMatch; int* outCount; int* outFallbacks;
	int jacMaxSweeps;                        // k_greedy_jacobi: sweeps + rebuilds before the in-order exact pass takes over (0: default 256; MCS_JACOBI_MAX_SWEEPS for tests)
};
void launch_greedy(const GreedyArgs& g, hipStream_t s);

// window matcher (SearchByProjection), mcs_project.hip
constexpr int kProjListK = 16;        // sorted candidate keys kept per probe (mcs_project.hip); longer windows fall back to exact rescans
// rule 0  SearchByProjection(F, mapPoints, th)   window from vcos / level / th, level-aware ratio test        (src/cORBmatcher.cpp:67-166)
// rule 1  WindowSearch, SearchByProjection(F1,F2) explicit window, taken features skipped, best <= second*ratio && best <= TH_HIGH (:326-577)
// rule 2  SearchByProjection(Cur, Last, th)       explicit window, taken features skipped, best <= TH_HIGH     (:1990-2118)
// rule 3  SearchForInitialization                 explicit window, matched-distance stealing, best <= TH_LOW && best < second*ratio (:579-726)
struct ProjArgs {
	const double* px; const double* py; const double* vcos; const int* level; const int* pcam;
	const uint8_t* pdesc; const uint8_t* pmask; int nproj; int pstride;
	const mcs_keypoint* keys; const uint8_t* fdesc; const uint8_t* fmask; const int* fcam; uint8_t* assigned; int nfeat; int fstride;
	const int* width; const int* height; const double* scales; int nrCams;
	double th; double ratio; int dim; int thHigh;
	unsigned long long* lists; int* counts;   // [kProjListK][nproj] smallest keys ascending, [nproj] window members
	int* match; int* nmatches;
	// rules 1-3
	int rule; int cap; int thLow;
	const double* rad; const int* minLvl; const int* maxLvl;   // explicit window per probe
	int* owner; int* mdist;                                    // rule 3: vnMatches21 / vMatchedDistance, [nfeat]
	int* accepted;                                             // rule 3, optional: partner at acceptance time per probe (rotation histogram)
};

void launch_projection(const ProjArgs& a, hipStream_t s);
void launch_proj_candidates(const ProjArgs& a, hipStream_t s);   // the two stages of launch_projection, for per-kernel timing
void launch_proj_greedy(const ProjArgs& a, hipStream_t s);

struct WorldToCamArgs {   // cMultiCamSys_::WorldToCamHom_fast + isPointInMirrorMask (src/cam_system_omni.cpp:92-133, src/cam_model_omni.cpp:163-178)
	const double* M;              // [nrCams][16] MtMc_inv, row-major
	const OcamDev* cams;          // [nrCams]
	const int* width; const int* height;
	const uint8_t* const* masks;  // [nrCams] level-0 mirror masks (tight rows) or nullptr
	const double* pts; const int* pcam; int n;
	double* uv; uint8_t* flags;
};
void launch_world_to_cam(const WorldToCamArgs& a, hipStream_t s);

struct DistinctArgs {   // cMapPoint::ComputeDistinctiveDescriptors for a batch of map points (mcs_distinct.hip)
	const uint8_t* desc; const uint8_t* mask; int stride; int dim;
	const int* offsets;   // [npoints + 1] CSR row offsets of the observations of each map point
	int npoints; int* bestIdx;
};
void launch_distinct(const DistinctArgs& a, hipStream_t s);

}  // namespace mcs

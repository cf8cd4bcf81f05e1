/*
 * mcs_c.h — C ABI of libmcs_hip.so: the MI355X (gfx950) feature front end + brute-force Hamming matcher
 * of MultiCol-SLAM.  This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 *
 * The reference has no FFI for this path; its boundary is three C++ class surfaces.  Each entry point below
 * names the reference interface it replaces (file:line relative to the reference tree):
 *
 *   mcs_extractor_create / _destroy    mdBRIEFextractorOct::mdBRIEFextractorOct   include/mdBRIEFextractorOct.h:339-351, src/mdBRIEFextractorOct.cpp:134-203
 *   mcs_extract_batch                  mdBRIEFextractorOct::operator()            include/mdBRIEFextractorOct.h:355-361, src/mdBRIEFextractorOct.cpp:1244-1337
 *                                      (+ rays: the per-camera loop body of cMultiFrame::cMultiFrame, src/cMultiFrame.cpp:128-152)
 *   mcs_match_topk                     inner loops of cORBmatcher::SearchByBoW(KF,KF) src/cORBmatcher.cpp:885-966,
 *                                      SearchByBoW(KF,F) :179-323, SearchForTriangulationRaw :968-1155
 *   mcs_descriptor_distance[_masked]   DescriptorDistance64[_Masked]             src/cORBmatcher.cpp:2438-2474
 *
 * The C++ facade with the reference's class names (headers under include/mcs/, namespace MultiColSLAM) sits on top.
 * Every function returns MCS_OK (0) or a negative error code, never throws, and the caller owns every buffer.
 * A handle must be used from one host thread at a time (the reference runs one extractor instance per camera
 * thread, src/cMultiFrame.cpp:128-139).  There is NO CPU fallback: without a usable HIP device every call fails.
 */
#ifndef MCS_C_H
#define MCS_C_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MCS_OK 0
#define MCS_ERR_INVALID (-1)   /* bad argument */
#define MCS_ERR_HIP (-2)       /* HIP runtime error (mcs_last_error() has the text) */
#define MCS_ERR_CAPACITY (-3)  /* an output / internal capacity was exceeded */
#define MCS_ERR_UNSUPPORTED (-4)

/* ABI revision of this header: bumped whenever a struct layout or an entry point's signature changes (3: mcs_desc_set carries block_rows / block_pitch_rows
 * since round 2 — callers built against an older header must be recompiled; mcs_describe_fast_table, FAST types 0 / 1 in round 3; 4: mcs_extractor_tie_stats; 5: mcs_copy_narrow,
 * mcs_ctx_result_stream, mcs_ctx_stream_conflicts, mcs_ctx_transfer_stream in round 4; 8: mcs_extractor_set_tie_capture / _patch_ties in round 6).  mcs_abi_version() returns
 * the value the LIBRARY was built with: compare it with MCS_ABI_VERSION after dlopen. */
#define MCS_ABI_VERSION 8

#define MCS_MAX_POLY 16
#define MCS_MAX_LEVELS 16

typedef struct mcs_ctx mcs_ctx;
typedef struct mcs_extractor mcs_extractor;

/* cv::KeyPoint layout (28 bytes): pt.x pt.y size angle response octave class_id */
typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } mcs_keypoint;

/* cCamModelGeneral_ (include/cam_model_omni.h:60-110): affine c,d,e, principal point, forward / backward polynomials */
typedef struct {
	double c, d, e, u0, v0;
	double p[MCS_MAX_POLY];    int32_t p_deg;     /* number of forward coefficients (Lafida: 5)  */
	double invP[MCS_MAX_POLY]; int32_t invP_deg;  /* number of backward coefficients (Lafida: 12) */
	int32_t width, height;
} mcs_ocam;

/* the 13 constructor arguments of mdBRIEFextractorOct, same order and meaning (h:339-351).  useAgast 0: cv::FastFeatureDetector, fastAgastType 0 / 1 / 2 =
   TYPE_5_8 / TYPE_7_12 / TYPE_9_16; useAgast 1: cv::AgastFeatureDetector, fastAgastType 0 / 1 / 2 / 3 = AGAST_5_8 / AGAST_7_12d / AGAST_7_12s / OAST_9_16
   (src/mdBRIEFextractorOct.cpp:869-872, 912-917), 1 <= fastThreshold <= 254 */
typedef struct {
	int32_t nfeatures; float scaleFactor; int32_t nlevels; int32_t edgeThreshold; int32_t firstLevel; int32_t scoreType;
	int32_t patchSize; int32_t fastThreshold; int32_t useAgast; int32_t fastAgastType; int32_t do_dBrief; int32_t learnMasks;
	int32_t descSize;
} mcs_extractor_params;

typedef enum { MCS_MEM_HOST = 0, MCS_MEM_DEVICE = 1 } mcs_mem_kind;

const char* mcs_last_error(void);              /* thread-local text of the last failure */
int mcs_abi_version(void);                     /* MCS_ABI_VERSION of the library build */
int mcs_device_count(int* n);

/* One context per (process, GPU).  stream: a hipStream_t to run on (NULL = the context creates its own). */
int mcs_ctx_create(int device, void* hip_stream, mcs_ctx** out);
int mcs_ctx_destroy(mcs_ctx*);                 /* also destroys the extractors still alive on this context (their handles become invalid) */
int mcs_ctx_synchronize(mcs_ctx*);
/* DEVICE-kind calls only enqueue work.  Internally the library forks independent / latency-bound kernels (the blur, the
 * greedy match resolution) onto a second stream so they overlap the VALU-bound ones.  Everything an extraction produces is
 * ordered on the context's stream again before mcs_extract_batch returns; the outputs of mcs_search_* (match arrays, counts)
 * are complete for later work on the context's stream only after mcs_ctx_join (a stream-side wait, no host block), after
 * the next mcs_search_* / mcs_match_* call, or after mcs_ctx_synchronize.  MCS_NO_OVERLAP=1 in the environment disables the fork. */
int mcs_ctx_join(mcs_ctx*);
/* Deferred searches (optional, device memory only).  With mcs_ctx_set_async_search(ctx, 1) a mcs_search_* call runs ENTIRELY beside the context's stream
 * (top-K lists and greedy pass on the library's own stream, ordered behind everything enqueued on the context's stream before the call and behind the
 * previous search): the caller's stream continues at once, so the next batch's extraction overlaps the matcher.  The caller then owes the ordering a
 * plain stream would have given it: the search's INPUT buffers must not be overwritten, and its outputs not read, before
 *   mcs_ctx_search_fence(ctx, lag)   the context's stream waits for the search issued `lag` calls before the latest one (0 = the latest = mcs_ctx_join)
 * (a stream-side wait, no host block).  bench.py alternates two buffer sets and fences with lag 1 before it reuses a set.
 * The same holds, to a lesser degree, WITHOUT deferred searches: the greedy pass of a search runs on the library's side stream and rescans the search's input
 * rows and valid flags, so they must stay intact until mcs_ctx_join / the next mcs_search_* call / mcs_ctx_synchronize — a caller that refills them on its own
 * stream in between (the native rig host's exchange) joins first or rotates buffer sets. */
int mcs_ctx_set_async_search(mcs_ctx*, int on);
int mcs_ctx_search_fence(mcs_ctx*, int lag);

/* ------------------------------------------------------------------ extractor
 * An extractor is built for one image size and a maximum batch (images per launch).  It owns the device pyramid,
 * candidate and keypoint buffers (sized once; nothing is allocated per call).
 * MCS_ERR_UNSUPPORTED at creation: more than 32 oct-tree roots on a level (nIni = round(width / height) of the processed area: panoramas wider than 32:1) or more than
 * 2045 features on one level.  A level with NO root (nIni = 0: more than twice as tall as wide, e.g. the small top levels of a portrait image) yields no keypoint: the
 * reference divides by nIni there and indexes an empty root vector as soon as the level has a candidate (DistributeOctTree, src/mdBRIEFextractorOct.cpp:641-661,
 * undefined); its one defined outcome, a level without candidates, is "nothing from this level", which is also the oracle's reading.                          */
int mcs_extractor_create(mcs_ctx*, const mcs_extractor_params*, int width, int height, int max_batch, mcs_extractor** out);
int mcs_extractor_destroy(mcs_extractor*);
int mcs_extractor_kp_capacity(const mcs_extractor*, int* cap);  /* rows per image in the outputs: sum over levels of max(nfeatures_level + 3, 4 * oct-tree roots);
                                                                 * = nfeatures + 3*nlevels for every usual configuration */
int mcs_extractor_levels(const mcs_extractor*, int* nlevels, int* widths, int* heights, int* features_per_level);

/* Run mdBRIEFextractorOct::operator() on nimg images at once.
 *   images   nimg rows of `image_stride` bytes each ... image i starts at images + i*image_pitch (w x h, 8UC1)
 *   masks    same layout (mirror mask level 0; 0 = reject) or NULL for "no mask"
 *   cams     nimg camera models (needed for dBRIEF/mdBRIEF and rays; may be NULL in ORB mode when rays == NULL)
 *   kind     where images/masks AND all outputs live (host pointers, or device pointers on the context's GPU)
 * outputs (row i*cap + k is keypoint k of image i, cap = mcs_extractor_kp_capacity):
 *   nkp[nimg]  keypoints[nimg*cap]  desc[nimg*cap*descSize]  descmask[nimg*cap*descSize] (zeros unless learnMasks)
 *   rays[nimg*cap*3] (ImgToWorld of every keypoint, optional)
 * With kind == DEVICE the call only enqueues work on the context's stream (no host synchronisation).
 * DEVICE-kind desc / descmask pointers (and out_row_stride of the strided form) must be 8-byte aligned: a descriptor leaves the kernel as 64-bit words
 * (the reference's matcher reads its rows as const uint64_t* too, src/cMultiKeyFrame.cpp:356-364); MCS_ERR_INVALID otherwise.                          */
/* Mirror masks that do not change from frame to frame (the reference builds them once per camera: cCamModelGeneral_::CreateMirrorMask, src/cam_model_omni.cpp) can be
 * left on the device: mcs_extractor_set_masks copies nimg masks (same layout rules as above; synchronous), after which MCS_MASKS_RESIDENT as the `masks` argument of
 * mcs_extract_batch / _strided means "mask i of that set for image i" — for ONE multi-frame per call the mask upload is a tenth of the extraction's latency.
 * Host-kind outputs: when nkp / keypoints / desc / descmask (/ rays) are page-locked (mcs_host_alloc), the call writes the VALID rows (k < nkp[i]) straight into
 * them with one launch instead of five copies; rows past an image's count are then left as they were.                                                      */
#define MCS_MASKS_RESIDENT ((const uint8_t*)(uintptr_t)1)
int mcs_extractor_set_masks(mcs_extractor*, int nimg, const uint8_t* masks, size_t mask_pitch, int mask_stride, mcs_mem_kind kind);
int mcs_extract_batch(mcs_extractor*, int nimg, const uint8_t* images, size_t image_pitch, int image_stride,
                      const uint8_t* masks, size_t mask_pitch, int mask_stride, const mcs_ocam* cams, mcs_mem_kind kind,
                      int32_t* nkp, mcs_keypoint* keypoints, uint8_t* desc, uint8_t* descmask, double* rays);

/* dBRIEF / mdBRIEF descriptors are computed in two passes (csrc/mcs_describe.hip): a fast pass whose omni-model arithmetic and pattern
 * mean differ from the reference's roundings by less than a per-camera bound, used only for keypoints none of whose pattern coordinates lies within
 * `guard_eps` pixels of a cvRound tie, and the reference's exact arithmetic (rotateAndDistortPattern, src/mdBRIEFextractorOct.cpp:250-283) for the
 * rest — the outputs are bit-identical either way.
 *   mcs_extractor_set_describe     exact_only != 0: every keypoint through the exact pass.  guard_eps: half-width of the band (0 = default 2^-24 px);
 *                                  a camera whose bound exceeds guard_eps / 2 runs exact-only by itself.
 *   mcs_extractor_describe_stats   keypoints the exact pass has handled since the extractor was created (fast-pass mode), and the band in use
 *   mcs_extractor_tie_stats        the checked invariant behind "bit-identical": the smallest distance to a rounding tie (|frac| = 1/2), in pixels, among ALL
 *                                  cvRound arguments the extractor has computed with the reference's exact arithmetic since creation / the last reset (ORB
 *                                  rotation :295-296; rotateAndDistortPattern :280-281 in the exact pass) — device libm (ocml) and the reference's (glibc) can
 *                                  only round such a coordinate differently within ~1e-13 px of a tie.  +inf before the first one.  Fast-pass coordinates are
 *                                  not listed: they are at least guard_eps from a tie by construction.
 *   mcs_describe_fast_bound        the worst-case coordinate difference the library assumes for a camera and descriptor size
 *   mcs_selftest_describe_fast     n pseudo-random pattern points of camera `cam` through both arithmetics on the device: the largest difference seen
 *                                  (must stay below mcs_describe_fast_bound)
 *   mcs_describe_fast_table        (host only, no device needed) the camera's table of G(s) = rho(atan(p0 / sqrt(s))) / sqrt(s) the fast pass reads: `rows`
 *                                  rows of `row_len` Taylor coefficients, row = (exponent of s - e0) * bins_per_octave + top mantissa bits, variable =
 *                                  low mantissa fraction - half a bin;  info6 = {tail bound in pixels, max |rho|, max sqrt(s)|s G'|, Lipschitz bound of
 *                                  (x G, y G), largest row error the builder itself measured, max Lipschitz bound x (sqrt(s) + 44)}; any output may be NULL                                */
int mcs_extractor_set_describe(mcs_extractor*, int exact_only, double guard_eps);
int mcs_extractor_describe_stats(mcs_extractor*, uint64_t* exact_pass_keypoints, double* guard_eps);
int mcs_extractor_tie_stats(mcs_extractor*, double* min_tie_distance, int reset);
/* Rounding ties are ENFORCED, not only watched (round 5; csrc/mcs_tiefix.hip; reference src/mdBRIEFextractorOct.cpp:280-281, 295-296).  Every keypoint whose
 * exact arithmetic (ORB rotation; rotateAndDistortPattern in the exact pass) produced a cvRound argument within `band` pixels of a tie is listed by the device,
 * and its descriptor (+ mask) is recomputed ON THE HOST with the host's libm — the one the reference links — before the results are final:
 *   - host-kind mcs_extract_batch does it by itself before it returns (it synchronises anyway);
 *   - device-kind calls only enqueue work: the caller runs mcs_extractor_fix_ties once the batch may be synchronised and BEFORE the extractor's next batch (the
 *     list belongs to the last batch); it waits for the context's stream, patches the device rows in place and reports how many it recomputed.  A caller that
 *     consumes the rows on-stream without it accepts the device libm's rounding for the listed keypoints (counted: mcs_extractor_tie_counts).
 *   mcs_extractor_set_tie_band   band in pixels; 0 = default (1e-9 for dBRIEF / mdBRIEF, 1e-12 for ORB — csrc/mcs_tiefix.hip derives both from a two-ulp libm
 *                                difference), < 0 = list nothing (round 4's behaviour), at most 0.5 (= every exact-pass keypoint: the test suite's setting)
 *   mcs_extractor_tie_counts     keypoints listed / recomputed since the extractor was created, and the band in use; any output may be NULL                        */
int mcs_extractor_set_tie_band(mcs_extractor*, double band_px);
int mcs_extractor_fix_ties(mcs_extractor*, int* recomputed);
int mcs_extractor_tie_counts(mcs_extractor*, uint64_t* listed, uint64_t* recomputed, double* band_px);
/* Pipelined enforcement (round 6): a caller that keeps its outputs on the device and consumes them on-stream — the batched front end whose matcher runs one
 * step behind the extraction, the multi-GPU rig whose descriptor blocks leave for the other ranks — cannot stop for mcs_extractor_fix_ties, and by the time it
 * could, the extractor's pyramid buffers hold the next batch.  With a capture ring every DEVICE-kind batch ends with one small launch that writes, for each
 * listed keypoint, its slot, level, position, angle and the 81 x 81 window of blurred / reflected samples around it (the values the reference's
 * `img.at<uchar>(…)` reads, src/mdBRIEFextractorOct.cpp:340-352, 437-470) into page-locked memory, and records an event.
 *   mcs_extractor_set_tie_capture(e, depth, max_ties)   depth = batches in flight (ring slots; 0 = off), max_ties = entries per slot (0 = 64); synchronous
 *   mcs_extractor_patch_ties(e, back, &listed, &recomputed)   batch `back` calls before the latest one (0 = the latest): waits ON THE HOST for that batch's
 *        event only — later batches keep running —, recomputes the listed descriptors with the host's libm (the same code path as mcs_extractor_fix_ties) and
 *        writes them over the batch's device rows; the rows are in place when it returns, so whatever is enqueued afterwards (search, exchange, download) reads
 *        them.  MCS_ERR_CAPACITY: more listed keypoints than max_ties (nothing patched; widen the slots or use mcs_extractor_fix_ties).
 * The pipelined order of bench.py / host/rig_host.cpp: enqueue extract(n); patch_ties(back = 1) — batch n - 1 has finished or is about to, the device already
 * holds batch n's work —; then enqueue search / exchange of batch n - 1.  One step of result latency, no device idle time, every consumed row the host's. */
int mcs_extractor_set_tie_capture(mcs_extractor*, int depth, int max_ties);
int mcs_extractor_patch_ties(mcs_extractor*, int back, int* listed, int* recomputed);
int mcs_describe_fast_bound(const mcs_ocam* cam, int desc_size, double* bound);
int mcs_selftest_describe_fast(mcs_ctx*, const mcs_ocam* cam, uint64_t seed, int n, double* max_abs_diff);
int mcs_describe_fast_table(const mcs_ocam* cam, double* table, int* rows, int* row_len, int* e0, int* bins_per_octave, double* info6);
/* (host only) the same table as the DEVICE reads it (round 6): `rows` rows of row_bytes = 48 bytes — g0 .. g3 as doubles, g4 g5 g6 as floats (+ 4 bytes of padding): three
 * 16-byte LDS reads per row instead of seven 8-byte ones; the tail g4 + g5 tau + g6 tau^2 is evaluated in float.  f32_term = what that adds to the coordinates at
 * most, in pixels (a term of mcs_describe_fast_bound).  Any output may be NULL. */
int mcs_describe_fast_table_packed(const mcs_ocam* cam, void* packed, int* row_bytes, double* f32_term);

/* mcs_extract_batch (device memory) with the descriptor / mask rows laid out for an exchange: row k of image i is written at
 * desc + (i * out_image_pitch_rows + k) * out_row_stride (descmask alike); 0 = the defaults (capacity rows, descSize bytes).  The camera-sharded rig
 * interleaves descriptor and mask in 2*descSize-byte rows and leaves one header row per image (mcs_rig_pack_headers).  keypoints / rays / nkp stay compact. */
int mcs_extract_batch_strided(mcs_extractor*, int nimg, const uint8_t* images, size_t image_pitch, int image_stride, const uint8_t* masks,
                              size_t mask_pitch, int mask_stride, const mcs_ocam* cams, int32_t* nkp, mcs_keypoint* keypoints, uint8_t* desc,
                              uint8_t* descmask, double* rays, size_t out_image_pitch_rows, int out_row_stride);

/* synchronise and report a device-side capacity overflow of earlier DEVICE-kind batches (MCS_OK if none) */
int mcs_extractor_status(mcs_extractor*);

/* stage taps for stage-level parity tests (device -> host copies of the LAST batch; synchronises) */
int mcs_extractor_tap_level(mcs_extractor*, int img, int level, int blurred, uint8_t* out /* w*h tight */);
int mcs_extractor_tap_candidates(mcs_extractor*, int img, int level, uint32_t* out /* x | y<<12 | score<<24, border-relative */, int cap, int* n);
int mcs_extractor_tap_selected(mcs_extractor*, int img, int level, uint32_t* out, int cap, int* n);

/* ------------------------------------------------------------------ brute-force Hamming matcher
 * For every query row the K nearest train rows by (distance, train index) among the train rows that are
 *   valid (t_valid[j] != 0, NULL = all) and, if q_group/t_group are given, have t_group[j] == q_group[i]
 *   (SearchForTriangulationRaw's same-camera rule, cORBmatcher.cpp:1047).
 * Queries with q_valid[i] == 0 get count 0.  Distances are DescriptorDistance64 (masks NULL) or
 * DescriptorDistance64Masked.  out_dist/out_idx are [nq*K] (unused tail: dist = INT32_MAX, idx = -1);
 * out_count_le[i] = number of eligible train rows with distance <= count_thresh (to detect K overflow); count_thresh < 0: not wanted (out_count_le is
 * zero-filled) — without it and without camera groups, 16- and 32-byte descriptors take the matrix-core kernel (csrc/mcs_match_mfma.hip).
 * dim = descriptor bytes (16/32/64).  The greedy, order-dependent part of the reference's searches is host logic
 * in the facade (include/mcs/cORBmatcher.hpp) on top of these lists.                                                */
typedef struct {
	const uint8_t* desc; const uint8_t* mask; const uint8_t* valid; const int32_t* group; int32_t n; int32_t stride;
	/* optional block structure (0 = the n rows are contiguous): the set consists of n / block_rows blocks of block_rows rows each, block b starting
	 * block_pitch_rows * b rows after the set's first row — the cameras of one multi-frame inside a gathered [camera][frame][row] buffer (rig.py).  Row i of
	 * the set (the index the results refer to) is row (i / block_rows) * block_pitch_rows + i % block_rows of the arrays; valid / group / rays follow the
	 * same rule.  n must be a multiple of block_rows. */
	int32_t block_rows; int64_t block_pitch_rows;
} mcs_desc_set;
int mcs_match_topk(mcs_ctx*, const mcs_desc_set* q, const mcs_desc_set* t, int dim, int K, int count_thresh, mcs_mem_kind kind,
                   int32_t* out_dist, int32_t* out_idx, int32_t* out_count_le);

/* batched form: nsets independent (query set, train set) pairs of identical shape laid out back to back
 * (set s starts at base + s*set_pitch rows).  One launch for a whole keyframe database sweep.                  */
int mcs_match_topk_batched(mcs_ctx*, int nsets, const mcs_desc_set* q, size_t q_set_pitch_rows, const mcs_desc_set* t,
                           size_t t_set_pitch_rows, int dim, int K, int count_thresh, mcs_mem_kind kind, int32_t* out_dist,
                           int32_t* out_idx, int32_t* out_count_le);

/* ------------------------------------------------------------------ cORBmatcher's three brute-force searches, complete
 * (top-K lists + the greedy, order-dependent resolution, both on the device; exact for any K — K only trades list size
 * against per-query rescans, reported in fallbacks[set]).  nsets independent (set 1, set 2) pairs per call, laid out
 * like mcs_match_topk_batched.  Thresholds follow cORBmatcher::cORBmatcher (src/cORBmatcher.cpp:46-65):
 * masks given -> TH_LOW = dim, else 2*dim.  mbCheckOrientation is false at every reference call site and not implemented.
 *
 * mcs_search_kf_kf         int cORBmatcher::SearchByBoW(cMultiKeyFrame*, cMultiKeyFrame*, vector<cMapPoint*>&)  (:885-966)
 *     valid = "has a good map point" on both sides;  match12[set*n1 + i] = index in set 2 or -1
 * mcs_search_kf_f          int cORBmatcher::SearchByBoW(cMultiKeyFrame*, cMultiFrame&, vector<cMapPoint*>&)     (:179-323)
 *     with the vocabulary-node restriction removed (BASELINE config 3): kf.valid = "has a good map point";
 *     matchF[set*nF + j] = keyframe feature index matched to FRAME feature j, or -1
 * mcs_search_triangulation int cORBmatcher::SearchForTriangulationRaw(...)                                      (:968-1155)
 *     valid = "has NO map point yet", group = camera index (required), rays = 3 doubles per row (same set pitch as the
 *     descriptors), E = nrCams*nrCams essential matrices (3x3 row-major, E[c1*nrCams+c2]);  match12 as above          */
int mcs_search_kf_kf(mcs_ctx*, int nsets, const mcs_desc_set* kf1, size_t pitch1_rows, const mcs_desc_set* kf2, size_t pitch2_rows, int dim,
                     double nnratio, int K, mcs_mem_kind kind, int32_t* match12, int32_t* nmatches, int32_t* fallbacks /* optional */);
int mcs_search_kf_f(mcs_ctx*, int nsets, const mcs_desc_set* kf, size_t pitchKF_rows, const mcs_desc_set* frame, size_t pitchF_rows, int dim,
                    double nnratio, int K, mcs_mem_kind kind, int32_t* matchF, int32_t* nmatches, int32_t* fallbacks);
int mcs_search_triangulation(mcs_ctx*, int nsets, const mcs_desc_set* kf1, size_t pitch1_rows, const mcs_desc_set* kf2, size_t pitch2_rows,
                             const double* rays1, const double* rays2, const double* E, int nrCams, int dim, int K, mcs_mem_kind kind,
                             int32_t* match12, int32_t* nmatches, int32_t* fallbacks);

/* Database sweeps: the loops the reference runs AROUND those searches, one call each (a set pitch of 0 rows = the same set for every pair).
 *
 * mcs_search_kf_f_sweep           the relocalisation loop of cTracking::Relocalisation (src/cTracking.cpp:1125-1221: for every candidate keyframe
 *     SearchByBoW(pKF, mCurrentFrame, ...)) for nframes frames at once — BASELINE configs[2]/[4]: every multi-frame against every stored keyframe.
 *     Pair s = f*nkf + k reads keyframe k (rows k*pitchKF_rows ...) and frame f (rows f*pitchF_rows ...);
 *     matchF[(f*nkf + k)*nF + j] = feature of keyframe k matched to feature j of frame f, or -1;  nmatches / fallbacks[f*nkf + k].
 *     Same semantics per pair as mcs_search_kf_f.
 * mcs_search_triangulation_sweep  the neighbour loop of cLocalMapping::CreateNewMapPoints (src/cLocalMapping.cpp:223-270: the current keyframe against each
 *     covisible keyframe, ComputeE per pair, then SearchForTriangulationRaw): like mcs_search_triangulation with pitch1_rows = 0 (the current keyframe
 *     and its rays shared by all pairs) and one block of nrCams*nrCams essential matrices PER pair: pair s reads E + s*E_set_pitch (doubles; 0 = one
 *     block for all pairs, >= 9*nrCams*nrCams otherwise).                                                                                             */
/* mcs_search_kf_kf_ring  a stream of multi-frames, each matched (SearchByBoW(KF,KF) semantics) against the one before it — BASELINE configs[1] on the
 *     gathered buffer of the camera-sharded rig: `frames` describes frame 0 of a ring of nframes_total frames lying pitch_rows rows apart (device memory);
 *     pair s = (frame first + s, frame (first + s - 1) mod nframes_total), s = 0 .. count-1;  match12[s*n + i], nmatches[s] as in mcs_search_kf_kf.
 *     nframes_total >= 2, first >= 0, count >= 1, first + count <= nframes_total (only the predecessor wraps); MCS_ERR_INVALID otherwise. */
int mcs_search_kf_kf_ring(mcs_ctx*, int nframes_total, int first, int count, const mcs_desc_set* frames, size_t pitch_rows, int dim, double nnratio, int K,
                          mcs_mem_kind kind, int32_t* match12, int32_t* nmatches, int32_t* fallbacks);
int mcs_search_kf_f_sweep(mcs_ctx*, int nkf, const mcs_desc_set* kf, size_t pitchKF_rows, int nframes, const mcs_desc_set* frame, size_t pitchF_rows,
                          int dim, double nnratio, int K, mcs_mem_kind kind, int32_t* matchF, int32_t* nmatches, int32_t* fallbacks);
int mcs_search_triangulation_sweep(mcs_ctx*, int nsets, const mcs_desc_set* kf1, size_t pitch1_rows, const mcs_desc_set* kf2, size_t pitch2_rows,
                                   const double* rays1, const double* rays2, const double* E, size_t E_set_pitch, int nrCams, int dim, int K,
                                   mcs_mem_kind kind, int32_t* match12, int32_t* nmatches, int32_t* fallbacks);

/* ------------------------------------------------------------------ window matcher (SURVEY §8f "next" row 1)
 * int cORBmatcher::SearchByProjection(cMultiFrame& F, const vector<cMapPoint*>& vpMapPoints, const double th)  (src/cORBmatcher.cpp:67-166)
 * including cMultiFrame::GetFeaturesInArea (src/cMultiFrame.cpp:272-340), PosInGrid (:342-353) and RadiusByViewingCos (:169-175).
 * A "projection" is one (map point, camera) pair that isInFrustum() marked mbTrackInView, listed in the reference's visiting
 * order (map point index, then camera): proj_x/y = mTrackProjX/Y, view_cos = mTrackViewCos, level = mnTrackScaleLevel, desc/mask =
 * the map point's descriptor.  The frame is given flat in mvKeys order (cameras concatenated): keys, desc, mask, keypoint_to_cam,
 * assigned[i] = (F.mvpMapPoints[i] != NULL) on entry (updated in place like the reference), image size per camera, mvScaleFactors.
 * match[p] = frame feature index matched to projection p or -1.  Thresholds per cORBmatcher's constructor (TH_HIGH).          */
typedef struct {
	const double* proj_x; const double* proj_y; const double* view_cos; const int32_t* level; const int32_t* cam;
	const uint8_t* desc; const uint8_t* mask; int32_t n; int32_t stride;
} mcs_projection_set;
typedef struct {
	const mcs_keypoint* keys; const uint8_t* desc; const uint8_t* mask; const int32_t* cam; uint8_t* assigned; int32_t n; int32_t stride;
	int32_t nr_cams; const int32_t* width; const int32_t* height; const double* scale_factors; int32_t nlevels;
} mcs_frame_view;
int mcs_search_by_projection(mcs_ctx*, const mcs_projection_set* mp, const mcs_frame_view* frame, double th, double nnratio, int dim,
                             mcs_mem_kind kind, int32_t* match, int32_t* nmatches);

/* ------------------------------------------------------------------ the other grid-window matchers of cTracking
 * One device routine serves them; a "probe" is one window the reference opens with cMultiFrame::GetFeaturesInArea(cam, x, y, radius,
 * min_level, max_level) (src/cMultiFrame.cpp:272-340; min_level = max_level = -1: no level check) together with the descriptor (+mask)
 * compared against the window's features, listed in the reference's visiting order.  Rules:
 *   MCS_WINDOW_RATIO      int cORBmatcher::WindowSearch(F1, F2, windowSize, vpMapPointMatches2, minScaleLevel, maxScaleLevel)   (:326-473)
 *                         int cORBmatcher::SearchByProjection(F1, F2, windowSize, vpMapPointMatches2)                            (:476-577)
 *                         features already taken are skipped; accept if best <= second*nnratio && best <= TH_HIGH
 *   MCS_WINDOW_BEST       int cORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th)                                       (:1990-2118)
 *                         features already taken are skipped; accept if best <= TH_HIGH
 *   MCS_WINDOW_INITIALIZE int cORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)               (:579-726)
 *                         a feature matched at distance d is skipped by later probes whose distance is >= d and stolen by closer ones;
 *                         accept if best <= TH_LOW && best < second*nnratio
 * frame->assigned[i] = "feature i is taken" on entry (updated in place; ignored by MCS_WINDOW_INITIALIZE).  match[p] = frame feature
 * index matched to probe p, or -1 (for MCS_WINDOW_INITIALIZE after all steals).  mbCheckOrientation (false at every reference call site,
 * include/cORBmatcher.h:40) is a separate pass: mcs_rotation_consistency.  frame->scale_factors / nlevels are not read by these rules.  */
typedef struct {
	const double* x; const double* y; const double* radius; const int32_t* min_level; const int32_t* max_level; const int32_t* cam;
	const uint8_t* desc; const uint8_t* mask; int32_t n; int32_t stride;
	int32_t* accepted_out;   /* optional, MCS_WINDOW_INITIALIZE only: the feature a probe held when it was accepted, -1 if never (also kept when the match
	                          * is stolen later) — the input of the rotation histogram, see mcs_rotation_consistency.  Same memory kind as `match`. */
} mcs_window_probes;
typedef enum { MCS_WINDOW_RATIO = 1, MCS_WINDOW_BEST = 2, MCS_WINDOW_INITIALIZE = 3 } mcs_window_rule;
int mcs_window_match(mcs_ctx*, const mcs_window_probes* probes, const mcs_frame_view* frame, mcs_window_rule rule, double nnratio, int dim,
                     mcs_mem_kind kind, int32_t* match, int32_t* nmatches);

/* mbCheckOrientation: the rotation-consistency filter every search applies when cORBmatcher was built with checkOri = true
 * (ComputeThreeMaxima, src/cORBmatcher.cpp:2394-2436).  Slot s is matched to partner match[s] (or -1); a 30-bin histogram of
 * rot = angle_first - angle_second (+360 if negative) is taken over the matches, those outside the three fullest bins are cleared in place.
 * angle_slot / angle_partner point at the `angle` field of the first keypoint of each side, stride_* = bytes between keypoints
 * (sizeof(mcs_keypoint) for keypoint arrays, 4 for plain float arrays).  swapped = 0: rot = slot angle - partner angle, 1: partner - slot.
 * variant = the reference's bin arithmetic ("factor" is 1/HISTO_LENGTH there, so only bins 0..12 are ever hit; reproduced):
 *   0  float factor, bin = cvRound(rot*factor) in float   SearchByBoW(KF,F) :191-282 [swapped 1], SearchByProjection(Cur,Last) :1999-2094 [1],
 *                                                          SearchByProjection(Cur, pKF, ...) :2137-2232 [1]
 *   1  double factor = (double)(1.0f/30), cvRound          WindowSearch :339-438 [1]
 *   2  double factor = 1.0/30, round() half away from zero SearchForInitialization :596-694 [0]; pass probes->accepted_out as `accepted`: its
 *                                                          histogram also counts acceptances that were stolen later
 *   3  double factor, rot += 360.0 in double, cvRound      SearchForTriangulationRaw :1011-1101 [0]
 * removed = number of matches cleared (the searches subtract it from their return value). */
int mcs_rotation_consistency(mcs_ctx*, int variant, const float* angle_slot, int stride_slot, const float* angle_partner, int stride_partner,
                             const int32_t* accepted /* optional */, int32_t* match, int n, int n_partner, int swapped, mcs_mem_kind kind, int32_t* removed);

/* Best feature per probe with a caller-given distance threshold — the search loops of the mapping / loop-closing matchers, whose
 * surrounding map-point surgery stays on the host:
 *   skip_taken = 0  every probe independently: cORBmatcher::Fuse (three overloads, :1265-1719; accept <= TH_LOW), SearchBySim3 (:1721-1988,
 *                   both directions, <= TH_HIGH), SearchForTriangulationBetweenCameras (:1158-1263, <= 100), SearchByProjection(pKF, Scw, ...)
 *                   (:2265-2392).  Their "kpLevel < nPredictedLevel-1 || kpLevel > nPredictedLevel" filter is the probe's level range.
 *   skip_taken = 1  in probe order, features with frame->assigned set are skipped and an accepted feature becomes assigned:
 *                   SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (:2120-2263, accept <= ORBdist).
 *   Fuse(pKF, vpMapPoints, th) (:1420-1568) never assigns the distance it computes (:1513-1516; `dist` stays 0), so the first feature of the window inside
 *   the level range wins there: pass all-zero descriptors on both sides (no masks) to reproduce it — integration/cORBmatcher_mcs.cpp does.
 * match[p] = feature index with the smallest distance (first in GetFeaturesInArea order on ties) if that distance <= max_dist, else -1;
 * dist[p] (optional) = that smallest distance (INT_MAX if the window is empty; with skip_taken only for accepted probes). */
int mcs_window_best(mcs_ctx*, const mcs_window_probes* probes, const mcs_frame_view* frame, int max_dist, int skip_taken, int dim, mcs_mem_kind kind,
                    int32_t* match, int32_t* dist /* optional */, int32_t* nmatches);

/* void cMultiCamSys_::WorldToCamHom_fast(int c, cv::Vec3d& pt3, cv::Vec2d& pt2) (src/cam_system_omni.cpp:114-133, the flagMcMt branch:
 * ptRot = MtMc_inv[c] * (pt3, 1), then cCamModelGeneral_::WorldToImg) for n points, point i into camera cam[i], followed by
 * cCamModelGeneral_::isPointInMirrorMask(u, v, 0) (src/cam_model_omni.cpp:163-178).  MtMc_inv: nr_cams 4x4 row-major matrices;
 * mirror_masks: nr_cams pointers to the level-0 masks (rows of cams[c].width bytes) or NULL (bounds test only).
 * uv = 2 doubles per point; flags bit0 = inside the mirror mask, bit1 = ptRot.z <= 0 (the bool the Vec4d overload returns, :92-112). */
int mcs_world_to_cam(mcs_ctx*, const double* MtMc_inv, const mcs_ocam* cams, int nr_cams, const uint8_t* const* mirror_masks,
                     const double* pts3, const int32_t* cam, int n, mcs_mem_kind kind, double* uv, uint8_t* flags);

/* void cMapPoint::ComputeDistinctiveDescriptors(bool havingMasks) (src/cMapPoint.cpp:294-382) for a batch of map points.
 * desc / mask: the observed descriptors (+ masks, or NULL) of all map points, row-major, `stride` bytes per row; map point k owns rows
 * offsets[k] .. offsets[k+1]-1 in the order the reference's loop collects them (keyframe, then observation).  best_idx[k] = the row
 * (relative to offsets[k]) whose descriptor becomes mDescriptor: least median distance to the later rows, first one on ties,
 * 0 for N <= 2, -1 for N = 0 (the reference returns early and keeps the old descriptor).  Rows per map point <= 65535. */
int mcs_distinctive_descriptors(mcs_ctx*, const uint8_t* desc, const uint8_t* mask, int stride, int dim, const int32_t* offsets, int npoints,
                                mcs_mem_kind kind, int32_t* best_idx);

/* ------------------------------------------------------------------ cMultiFrame::ComputeBoW (src/cMultiFrame.cpp:356-363), SURVEY §8f row 4
 * mpORBvocabulary->transform(descriptors, mBowVec, mFeatVec, levelsup) = one DBoW2 tree descent per descriptor
 * (ThirdParty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259, FORB::distance over 32 bytes).  The vocabulary is passed flat: node 0 is the
 * root, node_desc holds 32 bytes per node, the children of node i are child_idx[child_off[i] .. child_off[i+1]) in the order
 * TemplatedVocabulary::load (:1573-1622) appended them (file order), L = depth levels.  Per feature the call returns the leaf node reached
 * (its word id and weight are look-ups in the caller's node table) and the node of the path at level L - levelsup (0 if that level is
 * <= 0), i.e. the FeatureVector key.  Building the BowVector / FeatureVector maps from these (weights, L1 normalisation) stays on the host.
 * With node ids as `group` on both sides, mcs_search_kf_f is the vocabulary-restricted SearchByBoW(KF, F) of the reference
 * (keyframe rows ordered by (node, index), src/cORBmatcher.cpp:179-323). */
typedef struct mcs_vocabulary mcs_vocabulary;
int mcs_vocabulary_create(mcs_ctx*, int n_nodes, const uint8_t* node_desc, const int32_t* child_off, const int32_t* child_idx, int L, mcs_vocabulary** out);
void mcs_vocabulary_destroy(mcs_vocabulary*);
int mcs_bow_transform(mcs_vocabulary*, const uint8_t* desc, int n, int stride, int levelsup, mcs_mem_kind kind, int32_t* leaf_node, int32_t* node_at_level);

/* self-test of an arithmetic shortcut of the descriptor kernel: the omni model's three divisions by the same norm (src/cam_model_omni.cpp:
 * 146-161) share one refined reciprocal; this runs n pseudo-random (numerator, denominator) pairs of the magnitudes the kernel sees through
 * both forms on the device and returns the number of results that are not bit-identical to a / d (must be 0). */
int mcs_selftest_shared_reciprocal(mcs_ctx*, uint64_t seed, int n, int32_t* mismatches);

/* device helper: valid[i*cap + k] = (k < nkp[i]) for the row layout produced by mcs_extract_batch (all pointers on the GPU) */
int mcs_rows_valid(mcs_ctx*, const int32_t* nkp_dev, int nimg, int cap, uint8_t* valid_dev);

/* Exchange blocks of the camera-sharded rig (BASELINE configs[3]/[4]; multicol-slam_amd/rig.py): per image cap descriptor rows followed by ONE header row
 * whose first 4 bytes hold the image's keypoint count, so that descriptors, masks and counts of a rank's cameras travel in ONE all-gather.
 *   mcs_rig_pack_headers  nkp[i] -> header row of image block i (after mcs_extract_batch_strided with out_image_pitch_rows = cap + 1)
 *   mcs_rig_rows_valid    gathered blocks -> valid[i*(cap+1) + k] = (k < count of image i) in the SAME row geometry (header rows 0), counts optional
 * All pointers on the context's GPU; both only enqueue on the context's stream. */
int mcs_rig_pack_headers(mcs_ctx*, const int32_t* nkp_dev, int nimg, int cap, uint8_t* blocks_dev, int row_stride);
int mcs_rig_rows_valid(mcs_ctx*, const uint8_t* blocks_dev, int nimg, int cap, int row_stride, uint8_t* valid_dev, int32_t* nkp_out_dev);

/* A copy between page-locked host memory and the device (either direction, or device to device) that occupies at most `workgroups` workgroups of 256
 * threads instead of the runtime's chip-wide blit kernel: for the keypoints / descriptors / matches that leave for the host (the cv::KeyPoint vectors and
 * cv::Mat descriptors of src/cMultiFrame.cpp:92-216) BESIDE the kernels of the neighbouring steps.  Both pointers must be addressable from the context's GPU
 * (device memory; hipHostMalloc / hipHostRegister'ed host memory).  Towards the host TWO workgroups saturate PCIe 5 x16 (48 GB/s) — more of them stall the
 * memory traffic of every kernel running beside the copy.  Images coming FROM the host are better served by hipMemcpyAsync (SDMA engine).  Enqueues on
 * `hip_stream` (NULL: the context's stream) and returns; buffers that are not 16-byte aligned relative to each other go through hipMemcpyAsync. */
int mcs_copy_narrow(mcs_ctx*, void* dst, const void* src, size_t bytes, int workgroups, void* hip_stream);
/* The stream on which the outputs of the latest mcs_search_* call on device memory become complete IN STREAM ORDER (the greedy pass's stream; the context's own
 * stream when nothing is overlapped).  Results leave for the host from here without an event in front and without another stream: enqueue mcs_copy_narrow on it
 * right after the search call, record an event behind the copies, and wait for that event before the buffers are written again.
 * The answer belongs to the LATEST search (each search records the stream it used); before the first search it is the stream the next one would use in the
 * context's current mode — so query it AFTER the search whose results are to be copied, or at least after mcs_ctx_set_async_search / mcs_ctx_enable_timing:
 * both change which stream completes a search (in-order: the greedy pass's stream; deferred: a further one behind it). */
int mcs_ctx_result_stream(mcs_ctx*, void** hip_stream);
/* Long transfers beside the step — the image upload (hipMemcpyAsync from page-locked memory), the descriptor exchange of a multi-GPU rig (RCCL) — run on a
 * stream of the caller's, and which HARDWARE QUEUE that stream gets is the runtime's choice: HIP streams are dealt onto four queues, a queue runs its packets in
 * order, and an upload holds its queue for its whole duration (1.3 ms for the 69.5 MB of a default step).  Measured per step, by the context stream the upload
 * shared a queue with: deferred matcher 1.60 ms, greedy pass 2.17 ms, main stream 2.89 ms (alternating from one freshly created stream to the next).
 *   mcs_ctx_stream_conflicts   bit i of *mask: `hip_stream` shares a queue with the context's stream i (0 main, 1 extraction side stream, 2 deferred matcher,
 *                              3 greedy pass = result stream); probes with a held kernel and a marker, ~2 ms, synchronises the streams involved
 *   mcs_ctx_transfer_stream    a stream created and probed by the library until one shares a queue with none of the context's streams or only with the
 *                              deferred matcher's, which has a step of slack (owned by the context; *conflicts = its mask, may be NULL) */
int mcs_ctx_stream_conflicts(mcs_ctx*, void* hip_stream, unsigned* mask);
int mcs_ctx_transfer_stream(mcs_ctx*, void** hip_stream, unsigned* conflicts);

/* Page-locked host memory (hipHostMalloc) for callers that only see this header: images staged in it are uploaded by DMA without the runtime's pageable-memory
 * bounce buffer, results copied into it arrive at PCIe rate.  integration/cMultiFrame_mcs.cpp stages the rig's images and receives keypoints / descriptors /
 * rays through it (the cv::Mat / std::vector side of src/cMultiFrame.cpp:92-216).  Free before mcs_ctx_destroy. */
int mcs_host_alloc(mcs_ctx*, size_t bytes, void** out);
int mcs_host_free(mcs_ctx*, void* p);

/* single-pair distances on the device (known-answer / spot checks) */
int mcs_descriptor_distance(mcs_ctx*, const uint8_t* a, const uint8_t* b, int dim, int* out);
int mcs_descriptor_distance_masked(mcs_ctx*, const uint8_t* a, const uint8_t* b, const uint8_t* ma, const uint8_t* mb, int dim, int* out);

/* per-kernel device timing of the last mcs_extract_batch / mcs_match_* call on this context, measured with HIP
 * events on the context's stream when enabled (bench.py's roofline leg).  names: "pyramid","fast","octree","blur","describe","match" */
int mcs_ctx_enable_timing(mcs_ctx*, int on);
int mcs_ctx_kernel_ms(mcs_ctx*, const char* name, float* ms);

#ifdef __cplusplus
}
#endif
#endif

/*
 * mcs_oracle.h — C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a plain CPU restatement of the reference algorithm
 * (urbste/MultiCol-SLAM feature front end + brute-force matcher).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker /
 * the timed CPU baseline.  The product (libmcs_hip.so and the host facade) never links or loads it.
 *
 * PINNING STATUS.  The reference ships no tests / golden vectors for this path and cannot be built as a whole here (OpenCV >= 3.0, Pangolin,
 * compiled g2o ... are absent).  But almost all of ITS OWN sources compile UNMODIFIED, from /root/reference, against oracle/cvshim (OpenCV's types
 * re-implemented as far as they are used, type-only stubs for g2o's SE3Quat / Sim3): `make -C oracle ref` -> oracle/_ref/libmcs_ref.so =
 * mdBRIEFextractorOct, cam_model_omni, cam_system_omni, cConverter, misc, cORBmatcher, cMultiFrame, cMultiKeyFrame, cMapPoint, cMap,
 * cMultiKeyFrameDatabase + the vendored DBoW2.
 *   PINNED, bit for bit against that library (tests/test_oracle_vs_ref.py, tests/test_oracle_vs_ref_match.py, golden vectors from it in
 *     tests/golden/ref_extract.npz for machines without the checkout): the extractor in all three modes, the camera model and camera system
 *     (poses, WorldToCamHom_fast, mirror-mask test), cMultiFrame's constructor fields and ComputeBoW (DBoW2 load + transform), SearchByBoW KF-KF and KF-F
 *     (vocabulary-restricted and, through a one-leaf vocabulary, brute force), SearchForTriangulationRaw incl. ComputeE / CheckDistEpipolarLine,
 *     WindowSearch, SearchForInitialization, the three SearchByProjection overloads, the rotation-consistency filter (mbCheckOrientation = true), the
 *     search loop of Fuse, cMapPoint::ComputeDistinctiveDescriptors — driven through REAL
 *     cMultiFrame / cMultiKeyFrame / cMapPoint / cORBmatcher objects (oracle/ref_wrap_match.cpp).
 *   UNPINNED: the OpenCV image primitives the sources call (resize, copyMakeBorder, FAST, boxFilter, fastAtan2, cvRound, Matx products) — un-vendored
 *     third-party code restated from its published generic C++ algorithm; the shim forwards them to the restatements below, so
 *     _ref cannot check them (hand-derived known-answer tests do: tests/test_oracle_kat.py).  (orc_window_best is pinned through Fuse; the other
 *     users of that loop — SearchBySim3, SearchForTriangulationBetweenCameras — need Sim3 / relative-pose state the scene driver does not build.)
 *   The reference's DistributeOctTree orders equal nodes by heap address; _ref runs it on a bump arena (increasing addresses = creation order), the
 *   order the oracle uses; with the system allocator the reference's own output varies from run to run.  _ref is built without -fopenmp (the
 *   per-camera `#pragma omp parallel for` of cMultiFrame's constructor only changes timing).
 */
#ifndef MCS_ORACLE_H
#define MCS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, size, angle, response; int octave, class_id; } orc_keypoint; /* cv::KeyPoint, 28 B */

typedef struct {
	double c, d, e, u0, v0;
	double p[16];    int p_deg;      /* forward poly  (cam_model_omni.h: p)    */
	double invP[16]; int invP_deg;   /* backward poly (cam_model_omni.h: invP) */
	int width, height;
} orc_ocam;

typedef struct {       /* the 13 ctor arguments, include/mdBRIEFextractorOct.h:339-351 */
	int nfeatures; float scaleFactor; int nlevels; int edgeThreshold; int firstLevel; int scoreType;
	int patchSize; int fastThreshold; int useAgast; int fastAgastType; int do_dBrief; int learnMasks;
	int descSize;
} orc_params;

/* ---- E0 tables ---- */
void orc_features_per_level(int nfeatures, float scaleFactor, int nlevels, int* out);
void orc_umax(int* out17);
void orc_level_sizes(int W, int H, float scaleFactor, int nlevels, int* w, int* h);
int  orc_pattern(int descSize, int* xy /* 2*16*descSize ints */);

/* ---- OpenCV primitive restatements (Appendix A) ---- */
void  orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
void  orc_resize_nearest(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
void  orc_border_reflect101(uint8_t* buf, int w, int h, int stride, int border); /* fills the frame of a (w+2b)x(h+2b) buffer */
int   orc_fast9_16(const uint8_t* img, int w, int h, int stride, const uint8_t* mask, int mstride,
                   int threshold, orc_keypoint* out, int cap);
int   orc_fast_score(const uint8_t* center, int stride, int threshold);
int   orc_fast_score_type(int type, const uint8_t* center, int stride, int threshold);   /* cornerScore<8 / 12 / 16> for type 0 / 1 / 2 */
int   orc_fast_type(int type, const uint8_t* img, int w, int h, int stride, const uint8_t* mask, int mstride,
                    int threshold, orc_keypoint* out, int cap);                          /* FastFeatureDetector TYPE_5_8 / 7_12 / 9_16 */
int   orc_agast_type(int type, const uint8_t* img, int w, int h, int stride, const uint8_t* mask, int mstride,
                     int threshold, orc_keypoint* out, int cap);                         /* AgastFeatureDetector AGAST_5_8 / 7_12d / 7_12s / OAST_9_16, nonmax = true */
int   orc_agast_corners(int type, const uint8_t* img, int w, int h, int stride, int threshold, orc_keypoint* out, int cap); /* before the suppression */
int   orc_agast_score_type(int type, const uint8_t* center, int stride, int threshold);  /* agast_cornerScore<type> (bisection) */
void  orc_box5_inplace(uint8_t* roi, int w, int h, int stride); /* roi sits inside a >=2px frame */
float orc_fastAtan2(float y, float x);
int   orc_cvRound(double v);
float orc_ic_angle(const uint8_t* img, int stride, float ptx, float pty);

/* ---- omni camera model (src/cam_model_omni.cpp) ---- */
void orc_world2img(const orc_ocam* cam, double x, double y, double z, double* u, double* v);
void orc_img2world(const orc_ocam* cam, double u, double v, double* x, double* y, double* z);
void orc_mirror_mask(const orc_ocam* cam, uint8_t* mask /* h*w */);

/* ---- oct-tree (E3/E4) ---- */
int orc_distribute_octtree(const orc_keypoint* in, int n, int minX, int maxX, int minY, int maxY, int N,
                           orc_keypoint* out, int cap);

/* ---- extractor (E1..E8) ---- */
typedef struct orc_extractor orc_extractor;
orc_extractor* orc_extractor_create(const orc_params* p);
void orc_extractor_destroy(orc_extractor*);
/* returns number of keypoints (<= cap) or <0 on error; desc/dmask are nkp x descSize */
int  orc_extract(orc_extractor*, const uint8_t* img, int w, int h, int stride, const uint8_t* mask, int mstride,
                 const orc_ocam* cam, orc_keypoint* kps, int cap, uint8_t* desc, uint8_t* dmask);
/* stage taps of the LAST orc_extract call (for stage-level parity tests) */
int  orc_tap_level_size(orc_extractor*, int level, int* w, int* h);
int  orc_tap_level_image(orc_extractor*, int level, int blurred, uint8_t* out /* w*h tight */);
int  orc_tap_level_mask(orc_extractor*, int level, uint8_t* out);
int  orc_tap_candidates(orc_extractor*, int level, orc_keypoint* out, int cap); /* pre oct-tree, border-relative */
int  orc_tap_selected(orc_extractor*, int level, orc_keypoint* out, int cap);   /* post oct-tree+orientation, level coords */
/* E9: rays + grid */
void orc_rays(const orc_ocam* cam, const orc_keypoint* kps, int n, double* rays /* 3n */);
int  orc_pos_in_grid(const orc_ocam* cam, float x, float y, int* gx, int* gy);

/* ---- matcher (M1..M5) ---- */
int orc_dist64(const uint64_t* a, const uint64_t* b, int dim);
int orc_dist64_masked(const uint64_t* a, const uint64_t* b, const uint64_t* ma, const uint64_t* mb, int dim);
void orc_thresholds(int featDim, int havingMasks, int* th_high, int* th_low);
/* M4: SearchByBoW(KF,KF).  valid1/valid2: 1 = "has good map point".  match12[i] = idx2 or -1. returns nmatches */
int orc_search_kf_kf(const uint8_t* d1, const uint8_t* m1, const uint8_t* valid1, int n1,
                     const uint8_t* d2, const uint8_t* m2, const uint8_t* valid2, int n2,
                     int dim, int havingMasks, double nnratio, int* match12);
/* M4': SearchByBoW(KF,F) with the BoW restriction removed.  matchF[j] = idxKF or -1 (indexed by FRAME feature) */
int orc_search_kf_f(const uint8_t* dKF, const uint8_t* mKF, const uint8_t* validKF, int nKF,
                    const uint8_t* dF, const uint8_t* mF, int nF,
                    int dim, int havingMasks, double nnratio, int* matchF);
/* M5: SearchForTriangulationRaw. hasMP: 1 = already has a map point (skipped). E: nrCams*nrCams 3x3 row-major */
int orc_search_triangulation(const uint8_t* d1, const uint8_t* m1, const uint8_t* hasMP1, const int* cam1, const double* rays1, int n1,
                             const uint8_t* d2, const uint8_t* m2, const uint8_t* hasMP2, const int* cam2, const double* rays2, int n2,
                             const double* E, int nrCams, int dim, int havingMasks, int* match12);
int orc_check_epipolar(const double* ray1, const double* ray2, const double* E12, double thresh);
/* "next" row: cORBmatcher::SearchByProjection(cMultiFrame&, const vector<cMapPoint*>&, th) src/cORBmatcher.cpp:67-166
 * with cMultiFrame::GetFeaturesInArea src/cMultiFrame.cpp:272-340 and PosInGrid :342-353.  One "projection" = one (map point, camera)
 * pair with mbTrackInView, in the reference's visiting order.  assigned[] = F.mvpMapPoints[idx] != NULL (updated in place).
 * match[p] = frame feature index or -1.  returns nmatches. */
int orc_search_by_projection(const double* projx, const double* projy, const double* viewcos, const int* level, const int* pcam,
                             const uint8_t* pdesc, const uint8_t* pmask, int nproj,
                             const orc_keypoint* keys, const uint8_t* fdesc, const uint8_t* fmask, const int* fcam, uint8_t* assigned, int nfeat,
                             const int* width, const int* height, int nrCams, const double* scaleFactors, int nlevels,
                             double th, double nnratio, int dim, int havingMasks, int* match);

/* "next" row, the remaining grid-window matchers (src/cORBmatcher.cpp:326-726, 1990-2118) + the projection they consume
 * (src/cam_system_omni.cpp:92-133, src/cam_model_omni.cpp:163-178).  A frame view = the flat (all cameras) per-feature arrays. */
typedef struct orc_frame_view {
	const orc_keypoint* keys; const uint8_t* desc; const uint8_t* mask; const int32_t* cam; int32_t n; int32_t nrCams;
	const int32_t* width; const int32_t* height;
} orc_frame_view;
void orc_world_to_cam(const double* MtMc_inv, const orc_ocam* cams, const uint8_t* const* mirrorMasks, const double* pts3, const int* pcam, int n,
                      double* uv, uint8_t* flags);
int orc_window_search(const orc_frame_view* F1, const uint8_t* hasMP1, const orc_frame_view* F2, int windowSize, int minScaleLevel, int maxScaleLevel,
                      double nnratio, int dim, int havingMasks, int checkOri, int* match21);
int orc_search_by_projection_frames(const orc_frame_view* F1, const int* mp1, const uint8_t* bad1, const orc_frame_view* F2, const int* mp2,
                                    const double* uv, const uint8_t* inMask, int windowSize, double nnratio, int dim, int havingMasks, int* match21);
int orc_search_for_initialization(const orc_frame_view* F1, const orc_frame_view* F2, double* prevMatched, int windowSize, double nnratio, int dim,
                                  int havingMasks, int checkOri, int* match12);
int orc_search_by_projection_last(const orc_frame_view* Cur, uint8_t* curAssigned, const orc_frame_view* Last, const uint8_t* lastMP,
                                  const uint8_t* lastOutlier, const double* uv, const uint8_t* inMask, const double* scaleFactors, double th, int dim,
                                  int havingMasks, int checkOri, int* matchCur);

/* best-in-window search loop of Fuse / SearchBySim3 / SearchForTriangulationBetweenCameras / the relocalisation SearchByProjection */
int orc_window_best(const double* x, const double* y, const double* radius, const int* minLevel, const int* maxLevel, const int* pcam, const uint8_t* pdesc,
                    const uint8_t* pmask, int nprobes, const orc_frame_view* F, uint8_t* assigned, int maxDist, int skipTaken, int dim, int havingMasks,
                    int* match, int* dist);
/* "next" row 3: cMapPoint::ComputeDistinctiveDescriptors (src/cMapPoint.cpp:294-382): index of the chosen observation */
int orc_distinctive_descriptor(const uint8_t* desc, const uint8_t* mask, int N, int dim, int havingMasks);

/* mbCheckOrientation: rotation-consistency filter shared by the searches (four bin-arithmetic variants, see mcs_oracle.cpp) */
int orc_rotation_consistency(int variant, const float* angle_slot, const float* angle_partner, const int* accepted, int* match, int n, int swapped);

/* "next" row 4: DBoW2 vocabulary descent of cMultiFrame::ComputeBoW and the vocabulary-restricted SearchByBoW(KF,F) */
void orc_bow_transform(const uint8_t* node_desc, const int32_t* child_off, const int32_t* child_idx, int L, const uint8_t* desc, int n, int stride,
                       int levelsup, int32_t* leaf, int32_t* nid);
int orc_search_kf_f_bow(const uint8_t* dKF, const uint8_t* mKF, const uint8_t* validKF, const int* nodeKF, int nKF, const uint8_t* dF, const uint8_t* mF,
                        const int* nodeF, int nF, int dim, int havingMasks, double nnratio, int* matchF);

/* ---- timed CPU baseline helper: extract nimg images (OpenMP over images), returns total keypoints ---- */
long orc_extract_many(const orc_params* p, int nimg, const uint8_t* const* imgs, int w, int h, int stride,
                      const uint8_t* const* masks, const orc_ocam* cams, int threads,
                      orc_keypoint* kps, int cap, int* nkp, uint8_t* desc, uint8_t* dmask);
int orc_num_threads(void);
/* extract nframes multi-frames of ncam cameras (image f*ncam+c), then SearchByBoW(KF,KF) of every multi-frame f >= 1
 * against multi-frame f-1 (all features "have map points"); OpenMP over images / frames.  Returns total keypoints;
 * nmatch[f] = matches of frame f (nmatch[0] = 0); seconds[0] = extraction wall time, seconds[1] = matching wall time. */
long orc_extract_match_many(const orc_params* p, int nframes, int ncam, const uint8_t* const* imgs, int w, int h, int stride,
                            const uint8_t* const* masks, const orc_ocam* cams, int threads, double nnratio, int* nmatch, double* seconds);
/* the same pass, results handed out (cap = nfeatures + 4 * nlevels rows per image; out_match[f][ncam * cap] = m12 of frame f over its flattened valid rows) */
long orc_extract_match_many_out(const orc_params* p, int nframes, int ncam, const uint8_t* const* imgs, int w, int h, int stride,
                                const uint8_t* const* masks, const orc_ocam* cams, int threads, double nnratio, int* nmatch, double* seconds,
                                int* out_nkp, orc_keypoint* out_kps, uint8_t* out_desc, uint8_t* out_mask, int* out_match);

#ifdef __cplusplus
}
#endif
#endif

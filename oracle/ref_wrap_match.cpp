// ref_wrap_match.cpp — TEST INFRASTRUCTURE: drives the REFERENCE's own matcher side (src/cORBmatcher.cpp, cMultiFrame.cpp, cMultiKeyFrame.cpp,
// cMapPoint.cpp, cMap.cpp, cMultiKeyFrameDatabase.cpp, compiled unmodified against oracle/cvshim) on a small scene: real cMultiFrame /
// cMultiKeyFrame / cMapPoint objects, real cORBmatcher calls.  tests/test_oracle_vs_ref_match.py compares every result with the oracle.
#include <map>
#include "cORBmatcher.h"
#include "cMultiFrame.h"
#include "cMultiKeyFrame.h"
#include "cMultiKeyFrameDatabase.h"
#include "cMapPoint.h"
#include "cMap.h"
#include "cORBVocabulary.h"
#include "misc.h"

using namespace MultiColSLAM;
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void rs_crash(int sig) {   // test aid: where did a scene call die
	void* fr[64]; const int n = backtrace(fr, 64);
	const char msg[] = "rs_*: fatal signal, backtrace:\n"; (void)!write(2, msg, sizeof(msg) - 1);
	backtrace_symbols_fd(fr, n, 2);
	_exit(128 + sig);
}

extern "C" void ref_arena(int on);   // ref_wrap.cpp: bump allocation (monotone addresses) while a scene lives

namespace {
struct Scene {
	cMultiCamSys_ rig;
	std::vector<mdBRIEFextractorOct*> ex;
	ORBVocabulary* voc = nullptr;
	cMap* map = nullptr;
	cMultiKeyFrameDatabase* db = nullptr;
	std::vector<cMultiFrame*> frames;
	std::vector<cMultiKeyFrame*> kfs;
	std::map<cMapPoint*, int> idOf;   // the id the test gave a map point
	int dim = 32; bool masks = false;
};
cCamModelGeneral_ make_cam(const orc_ocam* cam, const uint8_t* mask) {
	cv::Mat_<double> poly(cam->p_deg, 1), invpoly(cam->invP_deg, 1);
	for (int i = 0; i < cam->p_deg; ++i) poly.at<double>(i, 0) = cam->p[i];
	for (int i = 0; i < cam->invP_deg; ++i) invpoly.at<double>(i, 0) = cam->invP[i];
	double cdeu0v0[5] = {cam->c, cam->d, cam->e, cam->u0, cam->v0};
	cCamModelGeneral_ m(cdeu0v0, poly, invpoly, cam->width, cam->height);
	std::vector<cv::Mat> masks;
	masks.push_back(cv::Mat(cam->height, cam->width, CV_8UC1, (void*)mask).clone());
	m.SetMirrorMasks(masks);
	return m;
}
int id_or_minus1(Scene* s, cMapPoint* p) { if (!p) return -1; auto it = s->idOf.find(p); return it == s->idOf.end() ? -2 : it->second; }
}  // namespace

extern "C" void* rs_create(const double* M_c, const orc_ocam* cams, const uint8_t* const* masks, int nrCams, const orc_params* p, const char* vocPath) {
	try {
		if (getenv("RS_BACKTRACE")) { signal(SIGSEGV, rs_crash); signal(SIGABRT, rs_crash); }
		ref_arena(1);
		Scene* s = new Scene();
		std::vector<cv::Matx44d> Mc(nrCams);
		std::vector<cCamModelGeneral_> models;
		for (int c = 0; c < nrCams; ++c) { std::memcpy(Mc[c].val, M_c + 16 * c, 128); models.push_back(make_cam(&cams[c], masks[c])); }
		s->rig = cMultiCamSys_(cv::Matx44d::eye(), Mc, models);
		for (int c = 0; c < nrCams; ++c)
			s->ex.push_back(new mdBRIEFextractorOct(p->nfeatures, p->scaleFactor, p->nlevels, p->edgeThreshold, p->firstLevel, p->scoreType, p->patchSize,
			                                        p->fastThreshold, p->useAgast != 0, p->fastAgastType, p->do_dBrief != 0, p->learnMasks != 0, p->descSize));
		s->dim = p->descSize; s->masks = p->learnMasks != 0;
		s->voc = new ORBVocabulary();
		if (vocPath && vocPath[0]) s->voc->load(std::string(vocPath));
		s->map = new cMap();
		s->db = new cMultiKeyFrameDatabase(*s->voc);
		return s;
	} catch (const std::exception& e) { std::cerr << "rs_create: " << e.what() << std::endl; return nullptr; }
	catch (const std::string& e) { std::cerr << "rs_create: " << e << std::endl; return nullptr; }
}
extern "C" void rs_destroy(void*) { ref_arena(0); }   // everything lived in the arena

extern "C" int rs_add_frame(void* h, const uint8_t* const* imgs, int w, int hgt, double ts, const double* M_t) {
	Scene* s = (Scene*)h;
	try {
		cv::Matx44d Mt; std::memcpy(Mt.val, M_t, 128);
		s->rig.Set_M_t(Mt);
		std::vector<cv::Mat> images;
		for (int c = 0; c < s->rig.GetNrCams(); ++c) images.push_back(cv::Mat(hgt, w, CV_8UC1, (void*)imgs[c]).clone());
		cMultiFrame* F = new cMultiFrame(images, ts, s->ex, s->voc, s->rig, (int)s->frames.size());
		if (!s->voc->empty()) F->ComputeBoW();
		s->frames.push_back(F);
		return (int)s->frames.size() - 1;
	} catch (const std::exception& e) { std::cerr << "rs_add_frame: " << e.what() << std::endl; return -1; }
}
extern "C" int rs_frame_total(void* h, int f) { return (int)((Scene*)h)->frames[f]->totalN; }
extern "C" void rs_frame_get(void* h, int f, orc_keypoint* keys, uint8_t* desc, uint8_t* mask, int* cam, double* rays, int* node, double* gridInv) {
	Scene* s = (Scene*)h; cMultiFrame* F = s->frames[f];
	for (size_t i = 0; i < F->totalN; ++i) {
		const cv::KeyPoint& k = F->mvKeys[i];
		keys[i].x = k.pt.x; keys[i].y = k.pt.y; keys[i].size = k.size; keys[i].angle = k.angle; keys[i].response = k.response; keys[i].octave = k.octave; keys[i].class_id = k.class_id;
		const int c = F->keypoint_to_cam.find(i)->second, l = F->cont_idx_to_local_cam_idx.find(i)->second;
		cam[i] = c;
		std::memcpy(desc + i * s->dim, F->mDescriptors[c].ptr<uchar>(l), s->dim);
		std::memcpy(mask + i * s->dim, F->mDescriptorMasks[c].ptr<uchar>(l), s->dim);
		for (int k3 = 0; k3 < 3; ++k3) rays[3 * i + k3] = F->mvKeysRays[i](k3);
		node[i] = -1;
	}
	for (auto& e : F->mFeatVec) for (unsigned i : e.second) node[i] = (int)e.first;
	for (int c = 0; c < s->rig.GetNrCams(); ++c) { gridInv[2 * c] = F->mfGridElementWidthInv[c]; gridInv[2 * c + 1] = F->mfGridElementHeightInv[c]; }
}
// the remaining fields the trackers read (include/cMultiFrame.h:69-162): per-camera counts, local indices, image bounds, scale tables, flags.
// ints: [N[c] x nrCams][minX maxX minY maxY x nrCams][cont_idx_to_local_cam_idx x totalN][mnScaleLevels masksLearned descDimension mdBRIEF(1) imgCnt nOutliers nMapPoints sizes-consistent]
// doubles: [mfScaleFactor][mvScaleFactors][mvLevelSigma2][mvInvLevelSigma2]   (levels = mnScaleLevels)
extern "C" void rs_frame_extra(void* h, int f, int* ints, double* dbl) {
	Scene* s = (Scene*)h; cMultiFrame* F = s->frames[f];
	const int nr = s->rig.GetNrCams();
	int* o = ints;
	for (int c = 0; c < nr; ++c) *o++ = F->N[c];
	for (int c = 0; c < nr; ++c) { *o++ = F->mnMinX[c]; *o++ = F->mnMaxX[c]; *o++ = F->mnMinY[c]; *o++ = F->mnMaxY[c]; }
	for (size_t i = 0; i < F->totalN; ++i) *o++ = F->cont_idx_to_local_cam_idx.find(i)->second;
	*o++ = F->mnScaleLevels; *o++ = F->HavingMasks() ? 1 : 0; *o++ = F->DescDims(); *o++ = 1; *o++ = F->GetImgCnt();
	int nout = 0, nmp = 0;
	for (size_t i = 0; i < F->mvbOutlier.size(); ++i) nout += F->mvbOutlier[i] ? 1 : 0;
	for (size_t i = 0; i < F->mvpMapPoints.size(); ++i) nmp += F->mvpMapPoints[i] ? 1 : 0;
	*o++ = nout; *o++ = nmp;
	*o++ = (F->mvKeys.size() == F->totalN && F->mvKeysRays.size() == F->totalN && F->mvbOutlier.size() == F->totalN && F->mvpMapPoints.size() == F->totalN &&
	        F->keypoint_to_cam.size() == F->totalN && F->cont_idx_to_local_cam_idx.size() == F->totalN && (int)F->mDescriptors.size() == nr && (int)F->mGrids.size() == nr) ? 1 : 0;
	double* d = dbl;
	*d++ = F->mfScaleFactor;
	for (int i = 0; i < F->mnScaleLevels; ++i) *d++ = F->mvScaleFactors[i];
	for (int i = 0; i < F->mnScaleLevels; ++i) *d++ = F->mvLevelSigma2[i];
	for (int i = 0; i < F->mnScaleLevels; ++i) *d++ = F->mvInvLevelSigma2[i];
}
// the 64x48 grid: cell of feature i (or -1) as the reference filled mGrids
extern "C" void rs_frame_grid(void* h, int f, int* cellOf) {
	Scene* s = (Scene*)h; cMultiFrame* F = s->frames[f];
	for (size_t i = 0; i < F->totalN; ++i) cellOf[i] = -1;
	for (int c = 0; c < s->rig.GetNrCams(); ++c)
		for (size_t x = 0; x < F->mGrids[c].size(); ++x)
			for (size_t y = 0; y < F->mGrids[c][x].size(); ++y)
				for (size_t i : F->mGrids[c][x][y]) cellOf[i] = (int)(x * 48 + y);
}
extern "C" int rs_make_keyframe(void* h, int f) {
	Scene* s = (Scene*)h;
	s->kfs.push_back(new cMultiKeyFrame(*s->frames[f], s->map, s->db));
	return (int)s->kfs.size() - 1;
}
// flag[i]: 0 no map point, 1 good map point, 2 bad map point (frames only: a keyframe drops a bad point).  pos: 3 doubles per feature (or NULL),
// share[i] >= 0: reuse the map point already created for feature share[i] of the same container.  ids are base + i.
extern "C" int rs_set_mappoints(void* h, int isKF, int idx, const uint8_t* flag, const double* pos, const int* share, int base, int refKF) {
	Scene* s = (Scene*)h;
	try {
		if (refKF < 0 || (size_t)refKF >= s->kfs.size()) { std::cerr << "rs_set_mappoints: reference keyframe " << refKF << " does not exist" << std::endl; return -1; }
		cMultiKeyFrame* ref = s->kfs[refKF];
		const size_t n = isKF ? s->kfs[idx]->GetKeyPoints().size() : s->frames[idx]->totalN;
		std::vector<cMapPoint*> made(n, nullptr);
		for (size_t i = 0; i < n; ++i) {
			cMapPoint* mp = nullptr;
			if (flag[i]) {
				if (share && share[i] >= 0) mp = made[share[i]];
				else {
					cv::Vec3d P(pos ? pos[3 * i] : 0.0, pos ? pos[3 * i + 1] : 0.0, pos ? pos[3 * i + 2] : 1.0);
					mp = new cMapPoint(P, ref, s->map);
					s->idOf[mp] = base + (int)i;
				}
				made[i] = mp;
			}
			if (isKF) {
				if (mp && flag[i] == 1) { mp->AddObservation(s->kfs[idx], i); s->kfs[idx]->AddMapPoint(mp, i); }
			} else s->frames[idx]->mvpMapPoints[i] = mp;
		}
		for (size_t i = 0; i < n; ++i) if (made[i] && flag[i] == 2 && !(share && share[i] >= 0)) made[i]->SetBadFlag();
		if (isKF) for (size_t i = 0; i < n; ++i) if (made[i] && flag[i] == 1) made[i]->ComputeDistinctiveDescriptors(s->masks);
		return 0;
	} catch (const std::exception& e) { std::cerr << "rs_set_mappoints: " << e.what() << std::endl; return -1; }
}
extern "C" void rs_set_outliers(void* h, int f, const uint8_t* out) { cMultiFrame* F = ((Scene*)h)->frames[f]; for (size_t i = 0; i < F->totalN; ++i) F->mvbOutlier[i] = out[i] != 0; }
extern "C" void rs_frame_mappoint_ids(void* h, int f, int* ids) { Scene* s = (Scene*)h; cMultiFrame* F = s->frames[f]; for (size_t i = 0; i < F->totalN; ++i) ids[i] = id_or_minus1(s, F->mvpMapPoints[i]); }

extern "C" int rs_bow_kf_kf(void* h, int k1, int k2, double nnratio, int* match12ids) {
	Scene* s = (Scene*)h;
	cORBmatcher m(nnratio, false, s->dim, s->masks);
	std::vector<cMapPoint*> v;
	const int n = m.SearchByBoW(s->kfs[k1], s->kfs[k2], v);
	for (size_t i = 0; i < v.size(); ++i) match12ids[i] = id_or_minus1(s, v[i]);
	return n;
}
extern "C" int rs_bow_kf_f(void* h, int k, int f, double nnratio, int checkOri, int* matchFids) {
	Scene* s = (Scene*)h;
	cORBmatcher m(nnratio, checkOri != 0, s->dim, s->masks);
	std::vector<cMapPoint*> v;
	const int n = m.SearchByBoW(s->kfs[k], *s->frames[f], v);
	for (size_t i = 0; i < v.size(); ++i) matchFids[i] = id_or_minus1(s, v[i]);
	return n;
}
extern "C" int rs_triangulation(void* h, int k1, int k2, int checkOri, int* match12, double* E) {
	Scene* s = (Scene*)h;
	cORBmatcher m(0.6, checkOri != 0, s->dim, s->masks);
	std::vector<cv::KeyPoint> a, b; std::vector<cv::Vec3d> ra, rb; std::vector<std::pair<size_t, size_t>> pairs;
	const int n = m.SearchForTriangulationRaw(s->kfs[k1], s->kfs[k2], a, ra, b, rb, pairs);
	const size_t n1 = s->kfs[k1]->GetKeyPoints().size();
	for (size_t i = 0; i < n1; ++i) match12[i] = -1;
	for (auto& p : pairs) match12[p.first] = (int)p.second;
	const int nc = s->rig.GetNrCams();   // the essential matrices exactly as :990-1003 forms them
	for (int i = 0; i < nc; ++i) for (int j = 0; j < nc; ++j) {
		cv::Matx33d E12 = ComputeE(s->kfs[k1]->camSystem.Get_MtMc_inv(i), s->kfs[k2]->camSystem.Get_MtMc(j));
		std::memcpy(E + 9 * (i * nc + j), E12.val, 72);
	}
	return n;
}
extern "C" int rs_window_search(void* h, int f1, int f2, int windowSize, int minLvl, int maxLvl, double nnratio, int checkOri, int* match2ids) {
	Scene* s = (Scene*)h;
	cORBmatcher m(nnratio, checkOri != 0, s->dim, s->masks);
	std::vector<cMapPoint*> v;
	const int n = m.WindowSearch(*s->frames[f1], *s->frames[f2], windowSize, v, minLvl, maxLvl);
	for (size_t i = 0; i < v.size(); ++i) match2ids[i] = id_or_minus1(s, v[i]);
	return n;
}
extern "C" int rs_search_init(void* h, int f1, int f2, double* prevMatched, int windowSize, double nnratio, int checkOri, int* match12) {
	Scene* s = (Scene*)h;
	cORBmatcher m(nnratio, checkOri != 0, s->dim, s->masks);
	cMultiFrame* F1 = s->frames[f1];
	std::vector<cv::Vec2d> prev(F1->totalN);
	for (size_t i = 0; i < F1->totalN; ++i) prev[i] = cv::Vec2d(prevMatched[2 * i], prevMatched[2 * i + 1]);
	std::vector<int> v;
	const int n = m.SearchForInitialization(*F1, *s->frames[f2], prev, v, windowSize);
	for (size_t i = 0; i < v.size(); ++i) { match12[i] = v[i]; prevMatched[2 * i] = prev[i](0); prevMatched[2 * i + 1] = prev[i](1); }
	return n;
}
// SearchByProjection(F, vpMapPoints, th): the map points are those of keyframe k in feature order (the ones that exist); their tracking fields
// (mbTrackInView, mTrackProjX/Y, mnTrackScaleLevel, mTrackViewCos), [nfeat][nrCams] each, are written as isInFrustum would have left them
extern "C" int rs_proj_mappoints(void* h, int f, int k, const uint8_t* inView, const double* px, const double* py, const int* lvl, const double* vcos, double th,
                                 double nnratio, int* matchFids) {
	Scene* s = (Scene*)h;
	cORBmatcher m(nnratio, false, s->dim, s->masks);
	std::vector<cMapPoint*> kfmp = s->kfs[k]->GetMapPointMatches(), list;
	const int nc = s->rig.GetNrCams();
	for (size_t i = 0; i < kfmp.size(); ++i) {
		cMapPoint* p = kfmp[i];
		if (!p) continue;
		for (int c = 0; c < nc; ++c) {
			p->mbTrackInView[c] = inView[i * nc + c] != 0; p->mTrackProjX[c] = px[i * nc + c]; p->mTrackProjY[c] = py[i * nc + c];
			p->mnTrackScaleLevel[c] = lvl[i * nc + c]; p->mTrackViewCos[c] = vcos[i * nc + c];
		}
		list.push_back(p);
	}
	const int n = m.SearchByProjection(*s->frames[f], list, th);
	cMultiFrame* F = s->frames[f];
	for (size_t i = 0; i < F->totalN; ++i) matchFids[i] = id_or_minus1(s, F->mvpMapPoints[i]);
	return n;
}
extern "C" int rs_proj_last(void* h, int cur, int last, double th, int checkOri, int* curIds) {
	Scene* s = (Scene*)h;
	cORBmatcher m(0.8, checkOri != 0, s->dim, s->masks);
	const int n = m.SearchByProjection(*s->frames[cur], *s->frames[last], th);
	cMultiFrame* F = s->frames[cur];
	for (size_t i = 0; i < F->totalN; ++i) curIds[i] = id_or_minus1(s, F->mvpMapPoints[i]);
	return n;
}
extern "C" int rs_proj_frames(void* h, int f1, int f2, int windowSize, double nnratio, int* match2ids) {
	Scene* s = (Scene*)h;
	cORBmatcher m(nnratio, false, s->dim, s->masks);
	std::vector<cMapPoint*> v;
	const int n = m.SearchByProjection(*s->frames[f1], *s->frames[f2], windowSize, v);
	for (size_t i = 0; i < v.size(); ++i) match2ids[i] = id_or_minus1(s, v[i]);
	return n;
}
// cMapPoint::ComputeDistinctiveDescriptors on a fresh point observed by features idx[0..n) of keyframe k -> the chosen descriptor (+ mask)
extern "C" int rs_distinctive(void* h, int k, const int* idx, int n, uint8_t* desc, uint8_t* mask) {
	Scene* s = (Scene*)h;
	cMapPoint* mp = new cMapPoint(cv::Vec3d(0, 0, 1), s->kfs[k], s->map);
	for (int i = 0; i < n; ++i) mp->AddObservation(s->kfs[k], (size_t)idx[i]);
	mp->ComputeDistinctiveDescriptors(s->masks);
	cv::Mat d = mp->GetDescriptor();
	if (d.empty()) return -1;
	std::memcpy(desc, d.ptr<uchar>(0), s->dim);
	if (s->masks) { cv::Mat mm = mp->GetDescriptorMask(); std::memcpy(mask, mm.ptr<uchar>(0), s->dim); }
	return 0;
}

// cORBmatcher::Fuse(pKF, curKF, vpMapPoints, th) (src/cORBmatcher.cpp:1265-1418), one fresh map point at a time against a keyframe that holds no
// map points: every accepted (point, camera) shows up as a new observation, which is recorded and removed again.  bestIdx: [n][nrCams], -1 = none.
// minmax: GetMinDistanceInvariance / GetMaxDistanceInvariance of each point after UpdateNormalAndDepth.
extern "C" int rs_fuse_probes(void* h, int kTarget, int kSource, const int* feat, const double* pos, int n, double th, int* bestIdx, double* minmax) {
	Scene* s = (Scene*)h;
	try {
		cORBmatcher m(0.8, false, s->dim, s->masks);
		cMultiKeyFrame* T = s->kfs[kTarget]; cMultiKeyFrame* S = s->kfs[kSource];
		const int nc = s->rig.GetNrCams();
		for (int i = 0; i < n; ++i) {
			for (int c = 0; c < nc; ++c) bestIdx[i * nc + c] = -1;
			cMapPoint* mp = new cMapPoint(cv::Vec3d(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), S, s->map);
			mp->AddObservation(S, (size_t)feat[i]);
			mp->ComputeDistinctiveDescriptors(s->masks);
			mp->UpdateNormalAndDepth();
			minmax[2 * i] = mp->GetMinDistanceInvariance(); minmax[2 * i + 1] = mp->GetMaxDistanceInvariance();
			std::vector<cMapPoint*> v(1, mp);
			m.Fuse(T, S, v, th);
			std::map<cMultiKeyFrame*, std::vector<size_t>> obs = mp->GetObservations();
			if (obs.count(T))
				for (size_t idx : obs[T]) { bestIdx[i * nc + T->keypoint_to_cam.find(idx)->second] = (int)idx; T->EraseMapPointMatch(idx); }
		}
		return 0;
	} catch (const std::exception& e) { std::cerr << "rs_fuse_probes: " << e.what() << std::endl; return -1; }
}

// The Fuse overloads the reference calls with whole lists (the Replace / AddObservation surgery runs against a target keyframe that holds map points):
//   variant 0  Fuse(pKF, curKF, vpMapPoints, th)   cLocalMapping::SearchInNeighbors (src/cLocalMapping.cpp:416-424): the source keyframe's map-point list
//   variant 1  Fuse(pKF, vpMapPoints, th)          (:450) the same list as fuse candidates
//   variant 2  Fuse(pKF, Scw, vpPoints, th)        cLoopClosing::SearchAndFuse (src/cLoopClosing.cpp:608): the non-NULL points of the source keyframe
// -> nFused; idsT / idsS = the map-point ids both keyframes hold afterwards, badS[i] = the point source feature i held BEFORE the call is bad now.
extern "C" int rs_fuse(void* h, int kTarget, int kSource, double th, int variant, const double* Scw, int* idsT, int* idsS, uint8_t* badS) {
	Scene* s = (Scene*)h;
	try {
		cORBmatcher m(0.8, false, s->dim, s->masks);
		cMultiKeyFrame* T = s->kfs[kTarget]; cMultiKeyFrame* S = s->kfs[kSource];
		std::vector<cMapPoint*> v = S->GetMapPointMatches();
		for (cMapPoint* p : v) if (p && !p->isBad()) p->UpdateNormalAndDepth();
		int n = 0;
		if (variant == 0) n = m.Fuse(T, S, v, th);
		else if (variant == 1) n = m.Fuse(T, v, th);
		else {
			std::vector<cMapPoint*> pts;
			for (cMapPoint* p : v) if (p) pts.push_back(p);
			cv::Matx44d M; std::memcpy(M.val, Scw, 128);
			n = m.Fuse(T, M, pts, th);
		}
		std::vector<cMapPoint*> a = T->GetMapPointMatches(), b = S->GetMapPointMatches();
		for (size_t i = 0; i < a.size(); ++i) idsT[i] = id_or_minus1(s, a[i]);
		for (size_t i = 0; i < b.size(); ++i) { idsS[i] = id_or_minus1(s, b[i]); badS[i] = v[i] && v[i]->isBad(); }
		return n;
	} catch (const std::exception& e) { std::cerr << "rs_fuse: " << e.what() << std::endl; return -1; }
}
// cLoopClosing::ComputeSim3's pair (src/cLoopClosing.cpp:281, :343): SearchByBoW(KF1, KF2) fills vpMatches12, SearchBySim3 adds to it.
// ids12[i] = id of vpMatches12[i] afterwards; nBow = what SearchByBoW returned
extern "C" int rs_sim3(void* h, int k1, int k2, double s12, const double* R12, const double* t12, double th, int* ids12, int* nBow) {
	Scene* s = (Scene*)h;
	try {
		cORBmatcher m(0.75, false, s->dim, s->masks);
		std::vector<cMapPoint*> v;
		*nBow = m.SearchByBoW(s->kfs[k1], s->kfs[k2], v);
		for (cMapPoint* p : s->kfs[k1]->GetMapPointMatches()) if (p && !p->isBad()) p->UpdateNormalAndDepth();
		for (cMapPoint* p : s->kfs[k2]->GetMapPointMatches()) if (p && !p->isBad()) p->UpdateNormalAndDepth();
		cv::Matx33d R; std::memcpy(R.val, R12, 72);
		cv::Vec3d t(t12[0], t12[1], t12[2]);
		const int n = m.SearchBySim3(s->kfs[k1], s->kfs[k2], v, s12, R, t, th);
		for (size_t i = 0; i < v.size(); ++i) ids12[i] = id_or_minus1(s, v[i]);
		return n;
	} catch (const std::exception& e) { std::cerr << "rs_sim3: " << e.what() << std::endl; return -1; }
}
// cLoopClosing::ComputeSim3's last step (src/cLoopClosing.cpp:378-403): SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (src/cORBmatcher.cpp:2265-2392).
// vpPoints = the first nList non-NULL map points of keyframe kSource (the caller keeps nList within the target's camera-0 feature count: the reference
// looks the LIST position up as a feature index and reads that camera's descriptor matrix with rig-wide row numbers, which is only in bounds for camera 0).
// preSlot[i] >= 0: vpMatched[i] holds list entry preSlot[i] on entry.  ids[i] = id of vpMatched[i] afterwards.
extern "C" int rs_proj_scw(void* h, int kTarget, int kSource, int nList, const double* Scw, int th, const int* preSlot, int* ids) {
	Scene* s = (Scene*)h;
	try {
		cORBmatcher m(0.75, false, s->dim, s->masks);
		cMultiKeyFrame* T = s->kfs[kTarget]; cMultiKeyFrame* S = s->kfs[kSource];
		std::vector<cMapPoint*> pts;
		for (cMapPoint* p : S->GetMapPointMatches()) if (p && (int)pts.size() < nList) { if (!p->isBad()) p->UpdateNormalAndDepth(); pts.push_back(p); }
		const size_t N = T->GetKeyPoints().size();
		std::vector<cMapPoint*> matched(N, static_cast<cMapPoint*>(NULL));
		for (size_t i = 0; i < N; ++i) if (preSlot[i] >= 0 && preSlot[i] < (int)pts.size()) matched[i] = pts[preSlot[i]];
		cv::Matx44d M; std::memcpy(M.val, Scw, 128);
		const int n = m.SearchByProjection(T, M, pts, matched, th);
		for (size_t i = 0; i < N; ++i) ids[i] = id_or_minus1(s, matched[i]);
		return n;
	} catch (const std::exception& e) { std::cerr << "rs_proj_scw: " << e.what() << std::endl; return -1; }
}
// SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (src/cORBmatcher.cpp:2120-2263, no caller in the reference).  sAlreadyFound = the map points of
// keyframe k at the feature indices found[0..nFound).  curIds[i] = id of the frame's mvpMapPoints[i] afterwards.  The caller keeps the scene inside what the reference
// defines (it looks every candidate feature index of the FRAME up in the KEYFRAME's index map and reads that row of the probe camera's matrix, :2196-2197).
extern "C" int rs_proj_kf(void* h, int f, int k, double th, int orbDist, int checkOri, const int* found, int nFound, int* curIds) {
	Scene* s = (Scene*)h;
	try {
		cORBmatcher m(0.8, checkOri != 0, s->dim, s->masks);
		std::vector<cMapPoint*> mps = s->kfs[k]->GetMapPointMatches();
		std::set<cMapPoint*> already;
		for (int i = 0; i < nFound; ++i) if (found[i] >= 0 && found[i] < (int)mps.size() && mps[found[i]]) already.insert(mps[found[i]]);
		const int n = m.SearchByProjection(*s->frames[f], s->kfs[k], already, th, orbDist);
		cMultiFrame* F = s->frames[f];
		for (size_t i = 0; i < F->totalN; ++i) curIds[i] = id_or_minus1(s, F->mvpMapPoints[i]);
		return n;
	} catch (const std::exception& e) { std::cerr << "rs_proj_kf: " << e.what() << std::endl; return -1; }
}
// SearchForTriangulationBetweenCameras(pKF, cam1, cam2, ...) (:1158-1263, no caller in the reference): match12[idx1] = idx2 or -1
extern "C" int rs_tri_between(void* h, int k, int cam1, int cam2, int* match12) {
	Scene* s = (Scene*)h;
	try {
		cORBmatcher m(0.8, false, s->dim, s->masks);
		std::vector<cv::KeyPoint> k1, k2; std::vector<cv::Vec3d> r1, r2; std::vector<std::pair<size_t, size_t>> pairs;
		const int n = m.SearchForTriangulationBetweenCameras(s->kfs[k], cam1, cam2, k1, r1, k2, r2, pairs);
		const size_t N = s->kfs[k]->GetKeyPoints().size();
		for (size_t i = 0; i < N; ++i) match12[i] = -1;
		for (auto& p : pairs) match12[p.first] = (int)p.second;
		return n;
	} catch (const std::exception& e) { std::cerr << "rs_tri_between: " << e.what() << std::endl; return -1; }
}

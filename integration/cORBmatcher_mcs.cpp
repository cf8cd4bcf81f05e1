// cORBmatcher_mcs.cpp — DROP-IN replacement for the reference's src/cORBmatcher.cpp.
//
// Same class, same header (the reference's own include/cORBmatcher.h, unmodified); the searches hand flat views of the reference's
// cMultiFrame / cMultiKeyFrame / cMapPoint objects to libmcs_hip.so (include/mcs_c.h) and write the results back into the reference's own
// containers.  What stays host code is exactly what the reference does between its distance loops: which features take part (map point present /
// bad / already found / level gate), and the map-point surgery of Fuse.
//   on the GPU      SearchByBoW (KF,KF) and (KF,F) incl. the vocabulary restriction, SearchForTriangulationRaw, SearchForTriangulationBetweenCameras,
//                   WindowSearch, SearchForInitialization, SearchByProjection (F, mapPoints) / (F1, F2, window) / (Current, Last), SearchBySim3,
//                   Fuse (pKF, curKF, points) / (pKF, points) / (pKF, Scw, points), and the mbCheckOrientation pass of each search that has one
//                   SearchByProjection(pKF, Scw, ...) (loop closing): GPU projection + window search, bounds-safe where the reference reads past
//                   its descriptor matrices (see that function)
//                   SearchByProjection(CurrentFrame, pKF, sAlreadyFound, ...): no caller in the reference; it indexes the keyframe's descriptors
//                   with the frame's feature indices — reproduced where that is defined (see that function)
//                   SearchForTriangulation and Fuse(curKF, neighKFs, map) are declared in the header but defined nowhere in the reference.
// tests/test_gpu_dropin.py builds the reference's cMultiFrame.cpp, cMultiKeyFrame.cpp, cMapPoint.cpp ... around this file and
// mdBRIEFextractorOct_mcs.cpp and compares every search with the all-reference build.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <set>
#include <stdexcept>
#include <string>

#include "cORBmatcher.h"
#include "misc.h"
#include "mcs_c.h"
#include "mcs_dropin.h"

using namespace std;

namespace MultiColSLAM
{
const int cORBmatcher::HISTO_LENGTH = 30;

namespace
{
	// The reference builds matchers on the stack of three threads (tracking, local mapping, loop closing) and runs them concurrently; an mcs_ctx owns one
	// HIP stream and its staging buffers and serves one thread at a time, so every thread gets its own (released when the thread ends).
	struct ThreadCtx
	{
		mcs_ctx* h = nullptr;
		~ThreadCtx() { if (h) mcs_ctx_destroy(h); }
	};
	mcs_ctx* ctx()
	{
		static thread_local ThreadCtx t;
		if (!t.h && mcs_ctx_create(0, nullptr, &t.h) != MCS_OK) throw std::runtime_error(std::string("mcs_ctx_create: ") + mcs_last_error());
		return t.h;
	}
	void check(int rc, const char* what) { if (rc != MCS_OK) throw std::runtime_error(std::string(what) + ": " + mcs_last_error()); }
	[[noreturn]] void not_replaced(const char* name) { throw std::logic_error(std::string("cORBmatcher::") + name + " is not replaced by the GPU drop-in (keep the reference's body)"); }

	// all cameras of a frame / keyframe concatenated in mvKeys order
	struct Flat
	{
		std::vector<mcs_keypoint> keys; std::vector<uint8_t> d, m; std::vector<int32_t> cam, w, h; std::vector<double> sf; int n = 0;
	};
	mcs_keypoint to_kp(const cv::KeyPoint& k) { mcs_keypoint o = { k.pt.x, k.pt.y, k.size, k.angle, k.response, k.octave, k.class_id }; return o; }
	void sizes(cMultiCamSys_& cs, Flat& f)
	{
		for (int c = 0; c < cs.GetNrCams(); ++c) { f.w.push_back((int)cs.GetCamModelObj(c).GetWidth()); f.h.push_back((int)cs.GetCamModelObj(c).GetHeight()); }
	}
	Flat flatten(cMultiFrame& F, int dim, bool masks)
	{
		Flat f; f.n = (int)F.totalN;
		f.keys.resize(f.n); f.d.resize((size_t)f.n * dim); f.m.resize(masks ? (size_t)f.n * dim : 0); f.cam.resize(f.n);
		for (int i = 0; i < f.n; ++i)
		{
			const int c = F.keypoint_to_cam.find(i)->second, l = F.cont_idx_to_local_cam_idx.find(i)->second;
			f.keys[i] = to_kp(F.mvKeys[i]); f.cam[i] = c;
			std::memcpy(&f.d[(size_t)i * dim], F.mDescriptors[c].ptr<uchar>(l), dim);
			if (masks) std::memcpy(&f.m[(size_t)i * dim], F.mDescriptorMasks[c].ptr<uchar>(l), dim);
		}
		sizes(F.camSystem, f);
		f.sf = F.mvScaleFactors;
		return f;
	}
	Flat flatten(cMultiKeyFrame* K, int dim, bool masks)
	{
		Flat f;
		std::vector<cv::KeyPoint> keys = K->GetKeyPoints();
		f.n = (int)keys.size();
		f.keys.resize(f.n); f.d.resize((size_t)f.n * dim); f.m.resize(masks ? (size_t)f.n * dim : 0); f.cam.resize(f.n);
		for (int i = 0; i < f.n; ++i)
		{
			const int c = K->keypoint_to_cam.find(i)->second, l = K->cont_idx_to_local_cam_idx.find(i)->second;
			f.keys[i] = to_kp(keys[i]); f.cam[i] = c;
			std::memcpy(&f.d[(size_t)i * dim], K->GetDescriptorRowPtr(c, l), dim);
			if (masks) std::memcpy(&f.m[(size_t)i * dim], K->GetDescriptorMaskRowPtr(c, l), dim);
		}
		sizes(K->camSystem, f);
		f.sf = K->GetScaleFactors();
		return f;
	}
	mcs_frame_view view(const Flat& f, uint8_t* assigned, int dim, bool masks)
	{
		mcs_frame_view v = { f.keys.data(), f.d.data(), masks ? f.m.data() : nullptr, f.cam.data(), assigned, f.n, dim, (int32_t)f.w.size(), f.w.data(), f.h.data(),
			f.sf.data(), (int32_t)f.sf.size() };
		return v;
	}
	// window probes: centre, radius, level range, camera and the descriptor (+mask) row of `from`
	struct Probes
	{
		std::vector<double> x, y, r; std::vector<int32_t> lo, hi, cam, src; std::vector<uint8_t> d, m;
		void add(double x_, double y_, double r_, int lo_, int hi_, int cam_, int src_) { x.push_back(x_); y.push_back(y_); r.push_back(r_); lo.push_back(lo_); hi.push_back(hi_); cam.push_back(cam_); src.push_back(src_); }
		void rows(const Flat& from, int dim, bool masks)
		{
			d.resize(std::max<size_t>(src.size(), 1) * dim); m.resize(masks ? d.size() : 0);
			for (size_t k = 0; k < src.size(); ++k)
			{
				std::memcpy(&d[k * dim], &from.d[(size_t)src[k] * dim], dim);
				if (masks) std::memcpy(&m[k * dim], &from.m[(size_t)src[k] * dim], dim);
			}
		}
		mcs_window_probes c(int dim, bool masks, int32_t* accepted = nullptr) const
		{
			mcs_window_probes p = { x.data(), y.data(), r.data(), lo.data(), hi.data(), cam.data(), d.data(), masks ? m.data() : nullptr, (int32_t)x.size(), dim, accepted };
			return p;
		}
	};
	// WorldToCamHom_fast + isPointInMirrorMask for many (point, camera) pairs of one camera system
	void project(cMultiCamSys_& cs, const std::vector<double>& pts, const std::vector<int32_t>& cam, std::vector<double>& uv, std::vector<uint8_t>& flags,
		bool cameraCoordinates = false)   // true: the points are in camera coordinates already (cCamModelGeneral_::WorldToImg alone; identity * p is exact)
	{
		const int nr = cs.GetNrCams(), n = (int)cam.size();
		std::vector<double> M((size_t)nr * 16);
		std::vector<mcs_ocam> oc(nr);
		std::vector<cv::Mat> keep(nr);
		std::vector<const uint8_t*> masks(nr);
		for (int c = 0; c < nr; ++c)
		{
			cv::Matx44d inv = cameraCoordinates ? cv::Matx44d::eye() : cs.Get_MtMc_inv(c);
			std::memcpy(&M[16 * (size_t)c], inv.val, 128);
			cCamModelGeneral_ cm = cs.GetCamModelObj(c);
			mcs_ocam& o = oc[c];
			std::memset(&o, 0, sizeof(o));
			o.c = cm.Get_c(); o.d = cm.Get_d(); o.e = cm.Get_e(); o.u0 = cm.Get_u0(); o.v0 = cm.Get_v0();
			cv::Mat_<double> P = cm.Get_P(), iP = cm.Get_invP();
			o.p_deg = cm.GetPolDeg(); o.invP_deg = cm.GetInvDeg();
			if (o.p_deg < 1 || o.p_deg > MCS_MAX_POLY || o.invP_deg < 1 || o.invP_deg > MCS_MAX_POLY) throw std::runtime_error("camera polynomial degree outside [1, MCS_MAX_POLY]");
			for (int i = 0; i < o.p_deg; ++i) o.p[i] = P.at<double>(i);
			for (int i = 0; i < o.invP_deg; ++i) o.invP[i] = iP.at<double>(i);
			o.width = (int)cm.GetWidth(); o.height = (int)cm.GetHeight();
			keep[c] = cm.GetMirrorMask(0);
			masks[c] = keep[c].data;   // level-0 masks are full, contiguous matrices
		}
		uv.assign(std::max<size_t>(n, 1) * 2, 0.0); flags.assign(std::max<size_t>(n, 1), 0);
		if (n) check(mcs_world_to_cam(ctx(), M.data(), oc.data(), nr, masks.data(), pts.data(), cam.data(), n, MCS_MEM_HOST, uv.data(), flags.data()), "mcs_world_to_cam");
	}
	bool good(cMapPoint* p) { return p && !p->isBad(); }
}

// --------------------------------------------------------------------------------------------------------------------------------
cORBmatcher::cORBmatcher(double nnratio, bool checkOri, const int featDim, bool havingMasks_) :
	mfNNratio(nnratio), mbCheckOrientation(checkOri), mbFeatDim(featDim), havingMasks(havingMasks_)
{
	if (havingMasks) { TH_HIGH_ = floor(1.5 * featDim); TH_LOW_ = floor(featDim); }
	else { TH_HIGH_ = 3 * featDim; TH_LOW_ = 2 * featDim; }
}

int DescriptorDistance64(const uint64_t* descr_i, const uint64_t* descr_j, const int& dim)   // scalar callers (cMapPoint) stay on the host
{
	uint64_t dist = 0;
	for (int d = 0; d < dim / 8; ++d) dist += __builtin_popcountll(descr_i[d] ^ descr_j[d]);
	return static_cast<int>(dist);
}

int DescriptorDistance64Masked(const uint64_t* descr_i, const uint64_t* descr_j, const uint64_t* mask_i, const uint64_t* mask_j, const int& dim)
{
	uint64_t dist = 0;
	for (int i = 0; i < dim / 8; ++i)
	{
		const uint64_t x = descr_i[i] ^ descr_j[i];
		dist += __builtin_popcountll(x & mask_i[i]);
		dist += __builtin_popcountll(x & mask_j[i]);
	}
	return static_cast<int>(dist / 2);
}

double cORBmatcher::RadiusByViewingCos(const double &viewCos) { return viewCos > 0.998 ? 2.5 : 4.0; }

void cORBmatcher::ComputeThreeMaxima(std::vector<int>*, const int, int&, int&, int&) { /* runs inside mcs_rotation_consistency */ }

int cORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { const int dim = a.cols; return DescriptorDistance64(a.ptr<uint64_t>(0), b.ptr<uint64_t>(0), dim); }

// ---- brute-force searches -------------------------------------------------------------------------------------------------------
int cORBmatcher::SearchByBoW(cMultiKeyFrame *pKF1, cMultiKeyFrame *pKF2, vector<cMapPoint*> &vpMatches12)
{
	vector<cMapPoint*> mp1 = pKF1->GetMapPointMatches(), mp2 = pKF2->GetMapPointMatches();
	Flat a = flatten(pKF1, mbFeatDim, havingMasks), b = flatten(pKF2, mbFeatDim, havingMasks);
	std::vector<uint8_t> v1(std::max(a.n, 1)), v2(std::max(b.n, 1));
	for (int i = 0; i < a.n; ++i) v1[i] = good(mp1[i]);
	for (int i = 0; i < b.n; ++i) v2[i] = good(mp2[i]);
	mcs_desc_set q = { a.d.data(), havingMasks ? a.m.data() : nullptr, v1.data(), nullptr, a.n, mbFeatDim };
	mcs_desc_set t = { b.d.data(), havingMasks ? b.m.data() : nullptr, v2.data(), nullptr, b.n, mbFeatDim };
	std::vector<int32_t> m12(std::max(a.n, 1), -1);
	int32_t n = 0;
	check(mcs_search_kf_kf(ctx(), 1, &q, 0, &t, 0, mbFeatDim, mfNNratio, 32, MCS_MEM_HOST, m12.data(), &n, nullptr), "mcs_search_kf_kf");
	vpMatches12 = vector<cMapPoint*>(mp1.size(), static_cast<cMapPoint*>(NULL));
	for (int i = 0; i < a.n; ++i) if (m12[i] >= 0) vpMatches12[i] = mp2[m12[i]];
	return n;
}

int cORBmatcher::SearchByBoW(cMultiKeyFrame* pKF, cMultiFrame &F, vector<cMapPoint*> &vpMapPointMatches)
{
	vector<cMapPoint*> mpKF = pKF->GetMapPointMatches();
	vpMapPointMatches = vector<cMapPoint*>(F.mvpMapPoints.size(), static_cast<cMapPoint*>(NULL));
	DBoW2::FeatureVector fvKF = pKF->GetFeatureVector();
	Flat a = flatten(pKF, mbFeatDim, havingMasks), b = flatten(F, mbFeatDim, havingMasks);
	// keyframe rows in FeatureVector order (node ascending, index ascending); the node id is the `group` both sides must share
	std::vector<int32_t> order, gk, gf(std::max(b.n, 1), -1);
	for (DBoW2::FeatureVector::iterator it = fvKF.begin(); it != fvKF.end(); ++it)
		for (size_t k = 0; k < it->second.size(); ++k) { order.push_back((int)it->second[k]); gk.push_back((int)it->first); }
	for (DBoW2::FeatureVector::iterator it = F.mFeatVec.begin(); it != F.mFeatVec.end(); ++it)
		for (size_t k = 0; k < it->second.size(); ++k) gf[it->second[k]] = (int)it->first;
	const int nq = (int)order.size();
	if (nq == 0 || b.n == 0) return 0;
	std::vector<uint8_t> qd((size_t)nq * mbFeatDim), qm(havingMasks ? qd.size() : 0), vq(nq), vf(b.n);
	for (int r = 0; r < nq; ++r)
	{
		std::memcpy(&qd[(size_t)r * mbFeatDim], &a.d[(size_t)order[r] * mbFeatDim], mbFeatDim);
		if (havingMasks) std::memcpy(&qm[(size_t)r * mbFeatDim], &a.m[(size_t)order[r] * mbFeatDim], mbFeatDim);
		vq[r] = good(mpKF[order[r]]);
	}
	for (int j = 0; j < b.n; ++j) vf[j] = gf[j] >= 0;
	mcs_desc_set q = { qd.data(), havingMasks ? qm.data() : nullptr, vq.data(), gk.data(), nq, mbFeatDim };
	mcs_desc_set t = { b.d.data(), havingMasks ? b.m.data() : nullptr, vf.data(), gf.data(), b.n, mbFeatDim };
	std::vector<int32_t> mF(b.n, -1);
	int32_t n = 0;
	check(mcs_search_kf_f(ctx(), 1, &q, 0, &t, 0, mbFeatDim, mfNNratio, 32, MCS_MEM_HOST, mF.data(), &n, nullptr), "mcs_search_kf_f");
	for (int j = 0; j < b.n; ++j) if (mF[j] >= 0) mF[j] = order[mF[j]];
	if (mbCheckOrientation)
	{
		int32_t removed = 0;
		check(mcs_rotation_consistency(ctx(), 0, &b.keys[0].angle, sizeof(mcs_keypoint), &a.keys[0].angle, sizeof(mcs_keypoint), nullptr, mF.data(), b.n, a.n, 1,
			MCS_MEM_HOST, &removed), "mcs_rotation_consistency");
		n -= removed;
	}
	for (int j = 0; j < b.n; ++j) if (mF[j] >= 0) vpMapPointMatches[j] = mpKF[mF[j]];
	return n;
}

int cORBmatcher::SearchForTriangulationRaw(cMultiKeyFrame *pKF1, cMultiKeyFrame *pKF2, std::vector<cv::KeyPoint> &vMatchedKeys1,
	std::vector<cv::Vec3d> &vMatchedKeysRays1, std::vector<cv::KeyPoint> &vMatchedKeys2, std::vector<cv::Vec3d> &vMatchedKeysRays2,
	std::vector<std::pair<size_t, size_t> > &vMatchedPairs)
{
	vector<cMapPoint*> mp1 = pKF1->GetMapPointMatches(), mp2 = pKF2->GetMapPointMatches();
	vector<cv::KeyPoint> k1 = pKF1->GetKeyPoints(), k2 = pKF2->GetKeyPoints();
	vector<cv::Vec3d> r1 = pKF1->GetKeyPointsRays(), r2 = pKF2->GetKeyPointsRays();
	const int nrCams = pKF1->camSystem.GetNrCams();
	std::vector<double> E((size_t)nrCams * nrCams * 9);
	for (int i = 0; i < nrCams; ++i)
		for (int j = 0; j < nrCams; ++j)
		{
			cv::Matx33d E12 = ComputeE(pKF1->camSystem.Get_MtMc_inv(i), pKF2->camSystem.Get_MtMc(j));
			std::memcpy(&E[9 * ((size_t)i * nrCams + j)], E12.val, 72);
		}
	Flat a = flatten(pKF1, mbFeatDim, havingMasks), b = flatten(pKF2, mbFeatDim, havingMasks);
	std::vector<uint8_t> v1(std::max(a.n, 1)), v2(std::max(b.n, 1));
	std::vector<double> ra((size_t)std::max(a.n, 1) * 3), rb((size_t)std::max(b.n, 1) * 3);
	for (int i = 0; i < a.n; ++i) { v1[i] = mp1[i] == NULL; for (int k = 0; k < 3; ++k) ra[3 * (size_t)i + k] = r1[i](k); }
	for (int i = 0; i < b.n; ++i) { v2[i] = mp2[i] == NULL; for (int k = 0; k < 3; ++k) rb[3 * (size_t)i + k] = r2[i](k); }
	mcs_desc_set q = { a.d.data(), havingMasks ? a.m.data() : nullptr, v1.data(), a.cam.data(), a.n, mbFeatDim };
	mcs_desc_set t = { b.d.data(), havingMasks ? b.m.data() : nullptr, v2.data(), b.cam.data(), b.n, mbFeatDim };
	std::vector<int32_t> m12(std::max(a.n, 1), -1);
	int32_t n = 0;
	check(mcs_search_triangulation(ctx(), 1, &q, 0, &t, 0, ra.data(), rb.data(), E.data(), nrCams, mbFeatDim, 32, MCS_MEM_HOST, m12.data(), &n, nullptr),
		"mcs_search_triangulation");
	if (mbCheckOrientation && a.n && b.n)
	{
		int32_t removed = 0;
		check(mcs_rotation_consistency(ctx(), 3, &a.keys[0].angle, sizeof(mcs_keypoint), &b.keys[0].angle, sizeof(mcs_keypoint), nullptr, m12.data(), a.n, b.n, 0,
			MCS_MEM_HOST, &removed), "mcs_rotation_consistency");
		n -= removed;
	}
	vMatchedKeys1.clear(); vMatchedKeysRays1.clear(); vMatchedKeys2.clear(); vMatchedKeysRays2.clear(); vMatchedPairs.clear();
	for (int i = 0; i < a.n; ++i)
	{
		if (m12[i] < 0) continue;
		vMatchedKeys1.push_back(k1[i]); vMatchedKeys2.push_back(k2[m12[i]]);
		vMatchedKeysRays1.push_back(r1[i]); vMatchedKeysRays2.push_back(r2[m12[i]]);
		vMatchedPairs.push_back(make_pair((size_t)i, (size_t)m12[i]));
	}
	return n;
}

// ---- grid-window searches -------------------------------------------------------------------------------------------------------
int cORBmatcher::WindowSearch(cMultiFrame &F1, cMultiFrame &F2, int windowSize, vector<cMapPoint *> &vpMapPointMatches2, int minScaleLevel, int maxScaleLevel)
{
	vpMapPointMatches2 = vector<cMapPoint*>(F2.mvpMapPoints.size(), static_cast<cMapPoint*>(NULL));
	Flat a = flatten(F1, mbFeatDim, havingMasks), b = flatten(F2, mbFeatDim, havingMasks);
	const bool bMinLevel = minScaleLevel > 0, bMaxLevel = maxScaleLevel < INT_MAX;
	Probes p;
	for (int i1 = 0; i1 < a.n; ++i1)
	{
		if (!good(F1.mvpMapPoints[i1])) continue;
		const int level1 = F1.mvKeys[i1].octave;
		if ((bMinLevel && level1 < minScaleLevel) || (bMaxLevel && level1 > maxScaleLevel)) continue;
		p.add(F1.mvKeys[i1].pt.x, F1.mvKeys[i1].pt.y, windowSize, -1, -1, a.cam[i1], i1);
	}
	if (p.x.empty() || b.n == 0) return 0;
	p.rows(a, mbFeatDim, havingMasks);
	std::vector<uint8_t> taken(b.n, 0);
	mcs_window_probes pr = p.c(mbFeatDim, havingMasks);
	mcs_frame_view fv = view(b, taken.data(), mbFeatDim, havingMasks);
	std::vector<int32_t> match(p.x.size(), -1), m21(b.n, -1);
	int32_t n = 0;
	check(mcs_window_match(ctx(), &pr, &fv, MCS_WINDOW_RATIO, mfNNratio, mbFeatDim, MCS_MEM_HOST, match.data(), &n), "mcs_window_match");
	for (size_t k = 0; k < match.size(); ++k) if (match[k] >= 0) m21[match[k]] = p.src[k];
	if (mbCheckOrientation)
	{
		int32_t removed = 0;
		check(mcs_rotation_consistency(ctx(), 1, &b.keys[0].angle, sizeof(mcs_keypoint), &a.keys[0].angle, sizeof(mcs_keypoint), nullptr, m21.data(), b.n, a.n, 1,
			MCS_MEM_HOST, &removed), "mcs_rotation_consistency");
		n -= removed;
	}
	for (int i2 = 0; i2 < b.n; ++i2) if (m21[i2] >= 0) vpMapPointMatches2[i2] = F1.mvpMapPoints[m21[i2]];
	return n;
}

int cORBmatcher::SearchByProjection(cMultiFrame &F1, cMultiFrame &F2, int windowSize, vector<cMapPoint *> &vpMapPointMatches2)
{
	vpMapPointMatches2 = F2.mvpMapPoints;
	set<cMapPoint*> found(vpMapPointMatches2.begin(), vpMapPointMatches2.end()), seen;
	Flat a = flatten(F1, mbFeatDim, havingMasks), b = flatten(F2, mbFeatDim, havingMasks);
	const int nr = F1.camSystem.GetNrCams();
	std::vector<double> pts; std::vector<int32_t> pc, owner;
	for (int i1 = 0; i1 < a.n; ++i1)
	{
		cMapPoint* pMP1 = F1.mvpMapPoints[i1];
		if (!pMP1 || pMP1->isBad() || found.count(pMP1) || seen.count(pMP1)) continue;
		seen.insert(pMP1);
		cv::Vec3d X = pMP1->GetWorldPos();
		for (int c = 0; c < nr; ++c) { pts.push_back(X(0)); pts.push_back(X(1)); pts.push_back(X(2)); pc.push_back(c); owner.push_back(i1); }
	}
	if (owner.empty() || b.n == 0) return 0;
	std::vector<double> uv; std::vector<uint8_t> fl;
	project(F2.camSystem, pts, pc, uv, fl);
	Probes p;
	for (size_t k = 0; k < owner.size(); ++k)
		if (fl[k] & 1) { const int l = F1.mvKeys[owner[k]].octave; p.add(uv[2 * k], uv[2 * k + 1], windowSize, l, l, pc[k], owner[k]); }
	if (p.x.empty()) return 0;
	p.rows(a, mbFeatDim, havingMasks);
	std::vector<uint8_t> taken(b.n);
	for (int i2 = 0; i2 < b.n; ++i2) taken[i2] = vpMapPointMatches2[i2] != NULL;
	mcs_window_probes pr = p.c(mbFeatDim, havingMasks);
	mcs_frame_view fv = view(b, taken.data(), mbFeatDim, havingMasks);
	std::vector<int32_t> match(p.x.size(), -1);
	int32_t n = 0;
	check(mcs_window_match(ctx(), &pr, &fv, MCS_WINDOW_RATIO, mfNNratio, mbFeatDim, MCS_MEM_HOST, match.data(), &n), "mcs_window_match");
	for (size_t k = 0; k < match.size(); ++k) if (match[k] >= 0) vpMapPointMatches2[match[k]] = F1.mvpMapPoints[p.src[k]];
	return n;
}

int cORBmatcher::SearchForInitialization(cMultiFrame &F1, cMultiFrame &F2, vector<cv::Vec2d> &vbPrevMatched, vector<int> &vnMatches12, int windowSize)
{
	Flat a = flatten(F1, mbFeatDim, havingMasks), b = flatten(F2, mbFeatDim, havingMasks);
	vnMatches12 = vector<int>(a.n, -1);
	if (a.n == 0 || b.n == 0) return 0;
	Probes p;
	for (int i1 = 0; i1 < a.n; ++i1) { const int l = F1.mvKeys[i1].octave; p.add(vbPrevMatched[i1](0), vbPrevMatched[i1](1), windowSize, l, l, a.cam[i1], i1); }
	p.rows(a, mbFeatDim, havingMasks);
	std::vector<int32_t> accepted(a.n, -1), match(a.n, -1);
	mcs_window_probes pr = p.c(mbFeatDim, havingMasks, accepted.data());
	mcs_frame_view fv = view(b, nullptr, mbFeatDim, havingMasks);
	int32_t n = 0;
	check(mcs_window_match(ctx(), &pr, &fv, MCS_WINDOW_INITIALIZE, mfNNratio, mbFeatDim, MCS_MEM_HOST, match.data(), &n), "mcs_window_match");
	if (mbCheckOrientation)
	{
		int32_t removed = 0;
		check(mcs_rotation_consistency(ctx(), 2, &a.keys[0].angle, sizeof(mcs_keypoint), &b.keys[0].angle, sizeof(mcs_keypoint), accepted.data(), match.data(), a.n, b.n, 0,
			MCS_MEM_HOST, &removed), "mcs_rotation_consistency");
		n -= removed;
	}
	for (int i1 = 0; i1 < a.n; ++i1)
	{
		vnMatches12[i1] = match[i1];
		if (match[i1] >= 0) vbPrevMatched[i1] = cv::Vec2d(F2.mvKeys[match[i1]].pt.x, F2.mvKeys[match[i1]].pt.y);
	}
	return n;
}

int cORBmatcher::SearchByProjection(cMultiFrame &F, const vector<cMapPoint*> &vpMapPoints, const double th)
{
	const int nr = F.camSystem.GetNrCams();
	Flat b = flatten(F, mbFeatDim, havingMasks);
	std::vector<double> px, py, vc; std::vector<int32_t> lv, pc; std::vector<uint8_t> d, m; std::vector<cMapPoint*> owner;
	for (size_t iMP = 0; iMP < vpMapPoints.size(); ++iMP)
	{
		cMapPoint* pMP = vpMapPoints[iMP];
		if (pMP->isBad()) continue;
		for (int cam = 0; cam < nr; ++cam)
		{
			if (!pMP->mbTrackInView[cam]) continue;
			px.push_back(pMP->mTrackProjX[cam]); py.push_back(pMP->mTrackProjY[cam]); vc.push_back(pMP->mTrackViewCos[cam]); lv.push_back(pMP->mnTrackScaleLevel[cam]);
			pc.push_back(cam); owner.push_back(pMP);
			const uchar* dp = (const uchar*)pMP->GetDescriptorPtr();
			d.insert(d.end(), dp, dp + mbFeatDim);
			if (havingMasks) { const uchar* mp = (const uchar*)pMP->GetDescriptorMaskPtr(); m.insert(m.end(), mp, mp + mbFeatDim); }
		}
	}
	if (owner.empty() || b.n == 0) return 0;
	std::vector<uint8_t> taken(b.n);
	for (int i = 0; i < b.n; ++i) taken[i] = F.mvpMapPoints[i] != NULL;
	mcs_projection_set ps = { px.data(), py.data(), vc.data(), lv.data(), pc.data(), d.data(), havingMasks ? m.data() : nullptr, (int32_t)owner.size(), mbFeatDim };
	mcs_frame_view fv = view(b, taken.data(), mbFeatDim, havingMasks);
	std::vector<int32_t> match(owner.size(), -1);
	int32_t n = 0;
	check(mcs_search_by_projection(ctx(), &ps, &fv, th, mfNNratio, mbFeatDim, MCS_MEM_HOST, match.data(), &n), "mcs_search_by_projection");
	for (size_t k = 0; k < match.size(); ++k) if (match[k] >= 0) F.mvpMapPoints[match[k]] = owner[k];
	return n;
}

int cORBmatcher::SearchByProjection(cMultiFrame &CurrentFrame, const cMultiFrame &LastFrame_, double th)
{
	cMultiFrame& LastFrame = const_cast<cMultiFrame&>(LastFrame_);
	Flat a = flatten(LastFrame, mbFeatDim, havingMasks), b = flatten(CurrentFrame, mbFeatDim, havingMasks);
	std::vector<double> pts; std::vector<int32_t> pc, owner;
	for (int i = 0; i < a.n; ++i)
	{
		cMapPoint* pMP = LastFrame.mvpMapPoints[i];
		if (!pMP || pMP->isBad() || LastFrame.mvbOutlier[i]) continue;
		cv::Vec3d X = pMP->GetWorldPos();
		pts.push_back(X(0)); pts.push_back(X(1)); pts.push_back(X(2)); pc.push_back(a.cam[i]); owner.push_back(i);
	}
	if (owner.empty() || b.n == 0) return 0;
	std::vector<double> uv; std::vector<uint8_t> fl;
	project(CurrentFrame.camSystem, pts, pc, uv, fl);
	Probes p;
	for (size_t k = 0; k < owner.size(); ++k)
		if (fl[k] & 1) { const int o = LastFrame.mvKeys[owner[k]].octave; p.add(uv[2 * k], uv[2 * k + 1], th * CurrentFrame.mvScaleFactors[o], o - 1, o + 1, pc[k], owner[k]); }
	if (p.x.empty()) return 0;
	p.rows(a, mbFeatDim, havingMasks);
	std::vector<uint8_t> taken(b.n);
	for (int i2 = 0; i2 < b.n; ++i2) taken[i2] = CurrentFrame.mvpMapPoints[i2] != NULL;
	mcs_window_probes pr = p.c(mbFeatDim, havingMasks);
	mcs_frame_view fv = view(b, taken.data(), mbFeatDim, havingMasks);
	std::vector<int32_t> match(p.x.size(), -1), mcur(b.n, -1);
	int32_t n = 0;
	check(mcs_window_match(ctx(), &pr, &fv, MCS_WINDOW_BEST, mfNNratio, mbFeatDim, MCS_MEM_HOST, match.data(), &n), "mcs_window_match");
	for (size_t k = 0; k < match.size(); ++k) if (match[k] >= 0) mcur[match[k]] = p.src[k];
	if (mbCheckOrientation)
	{
		int32_t removed = 0;
		check(mcs_rotation_consistency(ctx(), 0, &b.keys[0].angle, sizeof(mcs_keypoint), &a.keys[0].angle, sizeof(mcs_keypoint), nullptr, mcur.data(), b.n, a.n, 1,
			MCS_MEM_HOST, &removed), "mcs_rotation_consistency");
		n -= removed;
	}
	for (int i2 = 0; i2 < b.n; ++i2) if (mcur[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[mcur[i2]];
	return n;
}

// ---- Fuse: the search on the GPU (one mcs_window_best for all (map point, camera) pairs), the map-point surgery as in the reference -------------
namespace
{
	struct FuseRule
	{
		bool floatDepth;          // dist3D rounded to float (:1305, :1463) or kept in double (:1621)
		bool zeroDistance;        // Fuse(pKF, vpMapPoints, th) drops the value DescriptorDistance64 returns (:1513-1516): every distance is 0 there and the
		                          // first feature of the window inside the level range wins.  Reproduced: all-zero descriptors on both sides.
		bool inKeyFrameTest;      // skip test "isBad || IsInKeyFrame(pKF)" (re-evaluated per list entry) vs "isBad || in the snapshot set" (:1601)
		cMultiKeyFrame* curKF;    // the epipolar test of :1384-1395, or NULL
	};
	int fuse_common(cMultiKeyFrame* pKF, cMultiCamSys_& camSys, const cv::Vec3d& Ow, const std::vector<cMapPoint*>& vpMapPoints, const std::set<cMapPoint*>& snapshot,
		double th, const FuseRule& rule, int dim, bool havingMasks, int thLow)
	{
		const int nr = camSys.GetNrCams(), nMaxLevel = pKF->GetScaleLevels() - 1;
		vector<double> vfScaleFactors = pKF->GetScaleFactors();
		const bool masks = havingMasks && !rule.zeroDistance;
		Flat b = flatten(pKF, dim, masks);
		if (rule.zeroDistance) std::fill(b.d.begin(), b.d.end(), 0);
		std::vector<double> pts; std::vector<int32_t> pc, owner;
		for (size_t i = 0; i < vpMapPoints.size(); ++i)
		{
			cMapPoint* pMP = vpMapPoints[i];
			if (!pMP || pMP->isBad() || (rule.inKeyFrameTest ? pMP->IsInKeyFrame(pKF) : snapshot.count(pMP) != 0)) continue;
			cv::Vec3d X = pMP->GetWorldPos();
			for (int cam = 0; cam < nr; ++cam) { pts.push_back(X(0)); pts.push_back(X(1)); pts.push_back(X(2)); pc.push_back(cam); owner.push_back((int)i); }
		}
		if (owner.empty() || b.n == 0) return 0;
		std::vector<double> uv; std::vector<uint8_t> fl;
		project(camSys, pts, pc, uv, fl);
		Probes p;
		for (size_t k = 0; k < owner.size(); ++k)
		{
			if (!(fl[k] & 1)) continue;
			cMapPoint* pMP = vpMapPoints[owner[k]];
			const double maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
			cv::Vec3d PO = pMP->GetWorldPos() - Ow;
			double dist3D = cv::norm(PO);
			if (rule.floatDepth) dist3D = (float)dist3D;
			if (dist3D < minDistance || dist3D > maxDistance) continue;
			const double ratio = dist3D / minDistance;
			vector<double>::iterator it = std::lower_bound(vfScaleFactors.begin(), vfScaleFactors.end(), ratio);
			const int nPredictedLevel = std::min(static_cast<int>(it - vfScaleFactors.begin()), nMaxLevel);
			p.add(uv[2 * k], uv[2 * k + 1], th * vfScaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, pc[k], owner[k]);
			if (rule.zeroDistance) { p.d.insert(p.d.end(), dim, 0); continue; }
			const uchar* dp = (const uchar*)pMP->GetDescriptorPtr();
			p.d.insert(p.d.end(), dp, dp + dim);
			if (masks) { const uchar* mp = (const uchar*)pMP->GetDescriptorMaskPtr(); p.m.insert(p.m.end(), mp, mp + dim); }
		}
		if (p.x.empty()) return 0;
		mcs_window_probes pr = p.c(dim, masks);
		mcs_frame_view fv = view(b, nullptr, dim, masks);
		std::vector<int32_t> match(p.x.size(), -1);
		int32_t nfound = 0;
		check(mcs_window_best(ctx(), &pr, &fv, thLow, 0, dim, MCS_MEM_HOST, match.data(), nullptr, &nfound), "mcs_window_best");
		int nFused = 0, skipOwner = -1;
		for (size_t k = 0; k < match.size(); ++k)   // probes are in (map point, camera) order: the surgery below sees them like the reference's loop
		{
			cMapPoint* pMP = vpMapPoints[p.src[k]];
			// the reference evaluates its skip test when it reaches list entry i, i.e. after the surgery of the entries before it (a map point listed twice,
			// or replaced meanwhile, is skipped then); all cameras of one entry are searched before its own surgery starts
			if (k == 0 || p.src[k] != p.src[k - 1]) skipOwner = (pMP->isBad() || (rule.inKeyFrameTest && pMP->IsInKeyFrame(pKF))) ? p.src[k] : -1;
			if (p.src[k] == skipOwner || match[k] < 0) continue;
			const int bestIdx = match[k];
			cMapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
			if (!pMPinKF) { pMP->AddObservation(pKF, bestIdx); pKF->AddMapPoint(pMP, bestIdx); continue; }
			bool ok = !pMPinKF->isBad();
			if (rule.curKF)
			{
				cv::Vec3d ray1 = rule.curKF->GetKeyPointRay(p.src[k]), ray2 = pKF->GetKeyPointRay(bestIdx);
				const int camIdx1 = p.cam[k];
				cv::Matx33d E12 = ComputeE(rule.curKF->camSystem.Get_MtMc_inv(camIdx1) * pKF->camSystem.Get_MtMc(camIdx1));
				ok = ok && CheckDistEpipolarLine(ray1, ray2, E12, 1e-2);
			}
			if (ok) { pMP->Replace(pMPinKF); ++nFused; }
		}
		return nFused;
	}
	// Scw -> the camera system posed at the de-scaled Scw, and its centre (:1576-1585, :2279-2290)
	cv::Vec3d pose_from_sim3(cMultiCamSys_& camSys, const cv::Matx44d& Scw)
	{
		cv::Matx33d sRcw = cConverter::Hom2R(Scw);
		cv::Vec3d row1(sRcw(0, 0), sRcw(0, 1), sRcw(0, 2));
		const double inv_scw = 1.0 / std::sqrt(row1.dot(row1));   // (the reference writes cv::sqrt: the same IEEE square root)
		cv::Matx33d Rcw = inv_scw * sRcw;
		cv::Vec3d tcw = inv_scw * cConverter::Hom2T(Scw);
		camSys.Set_M_t(cConverter::invMat(cConverter::Rt2Hom(Rcw, tcw)));
		return cConverter::Hom2T(camSys.Get_M_t());
	}
}

int cORBmatcher::Fuse(cMultiKeyFrame *pKF, cMultiKeyFrame *curKF, vector<cMapPoint *> &vpMapPoints, double th)
{
	FuseRule rule = { true, false, true, curKF };
	return fuse_common(pKF, pKF->camSystem, pKF->GetCameraCenter(), vpMapPoints, std::set<cMapPoint*>(), th, rule, mbFeatDim, havingMasks, TH_LOW_);
}

int cORBmatcher::Fuse(cMultiKeyFrame* pKF, std::vector<cMapPoint*> &vpMapPoints, double th)
{
	FuseRule rule = { true, true, true, nullptr };
	return fuse_common(pKF, pKF->camSystem, pKF->GetCameraCenter(), vpMapPoints, std::set<cMapPoint*>(), th, rule, mbFeatDim, havingMasks, TH_LOW_);
}

int cORBmatcher::Fuse(cMultiKeyFrame *pKF, cv::Matx44d Scw, const vector<cMapPoint *> &vpPoints, double th)
{
	cMultiCamSys_ camSys = pKF->camSystem;
	cv::Vec3d Ow = pose_from_sim3(camSys, Scw);
	FuseRule rule = { false, false, false, nullptr };
	return fuse_common(pKF, camSys, Ow, vpPoints, pKF->GetMapPoints(), th, rule, mbFeatDim, havingMasks, TH_LOW_);
}

// ---- SearchBySim3: both directions as one mcs_window_best each, the agreement test on the host --------------------------------------------------------
int cORBmatcher::SearchBySim3(cMultiKeyFrame *pKF1, cMultiKeyFrame *pKF2, vector<cMapPoint*> &vpMatches12, const double &s12, const cv::Matx33d &R12,
	const cv::Vec3d &t12, double th)
{
	cv::Matx44d T1 = pKF1->GetPoseInverse(), T2 = pKF2->GetPoseInverse();
	cv::Matx33d R1w = cConverter::Hom2R(T1), R2w = cConverter::Hom2R(T2);
	cv::Vec3d t1w = cConverter::Hom2T(T1), t2w = cConverter::Hom2T(T2);
	cv::Matx33d sR12 = s12 * R12;
	cv::Matx33d sR21 = (1.0 / s12) * R12.t();
	cv::Vec3d t21 = -sR21 * t12;
	vector<cMapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
	const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
	vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
	for (int i = 0; i < N1; ++i)
	{
		cMapPoint* pMP = vpMatches12[i];
		if (!pMP) continue;
		vbAlreadyMatched1[i] = true;
		int idx2 = pMP->GetIndexInKeyFrame(pKF2)[0];
		if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
	}
	// one direction: the points of `from` (already in the rig frame of `to` after Rw/tw and the similarity) into camera keypoint_to_cam[i] of `to`
	struct Dir { cMultiKeyFrame* from; cMultiKeyFrame* to; const vector<cMapPoint*>* mps; const vector<bool>* done; cv::Matx33d Rw; cv::Vec3d tw; cv::Matx33d sR; cv::Vec3d t; };
	Dir dirs[2] = { { pKF1, pKF2, &vpMapPoints1, &vbAlreadyMatched1, R1w, t1w, sR21, t21 }, { pKF2, pKF1, &vpMapPoints2, &vbAlreadyMatched2, R2w, t2w, sR12, t12 } };
	std::vector<int> vnMatch[2] = { std::vector<int>(N1, -1), std::vector<int>(N2, -1) };
	for (int dI = 0; dI < 2; ++dI)
	{
		Dir& D = dirs[dI];
		const int nMaxLevel = D.to->GetScaleLevels() - 1;
		vector<double> sf = D.to->GetScaleFactors();
		std::vector<double> pts, d3; std::vector<int32_t> pc, owner;
		for (int i = 0; i < (int)D.mps->size(); ++i)
		{
			cMapPoint* pMP = (*D.mps)[i];
			if (!pMP || (*D.done)[i] || pMP->isBad()) continue;
			const int camIdx = D.from->keypoint_to_cam.find(i)->second;
			cv::Vec3d p3Dw = pMP->GetWorldPos();
			cv::Vec3d pa = D.Rw * p3Dw + D.tw;
			cv::Vec3d pb = D.sR * pa + D.t;
			cv::Vec4d p4 = cConverter::invMat(D.to->camSystem.Get_M_c(camIdx)) * cConverter::toVec4d(pb);
			if (pb(2) < 0.0) continue;
			pts.push_back(p4(0)); pts.push_back(p4(1)); pts.push_back(p4(2)); pc.push_back(camIdx); owner.push_back(i);
			d3.push_back(cv::norm(cv::Vec3d(p4(0), p4(1), p4(2))));
		}
		Flat b = flatten(D.to, mbFeatDim, havingMasks);
		if (owner.empty() || b.n == 0) continue;
		std::vector<double> uv; std::vector<uint8_t> fl;
		project(pKF2->camSystem, pts, pc, uv, fl, true);   // both directions use pKF2's camera models (:1787, :1877); the points are in camera coordinates already
		Probes p;
		for (size_t k = 0; k < owner.size(); ++k)
		{
			if (!(fl[k] & 1)) continue;
			cMapPoint* pMP = (*D.mps)[owner[k]];
			const double maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
			if (d3[k] < minDistance || d3[k] > maxDistance) continue;
			const double ratio = d3[k] / minDistance;
			vector<double>::iterator it = std::lower_bound(sf.begin(), sf.end(), ratio);
			const int lvl = std::min(static_cast<int>(it - sf.begin()), nMaxLevel);
			p.add(uv[2 * k], uv[2 * k + 1], th * sf[lvl], lvl - 1, lvl, pc[k], owner[k]);
			const uchar* dp = (const uchar*)pMP->GetDescriptorPtr();
			p.d.insert(p.d.end(), dp, dp + mbFeatDim);
			if (havingMasks) { const uchar* mp = (const uchar*)pMP->GetDescriptorMaskPtr(); p.m.insert(p.m.end(), mp, mp + mbFeatDim); }
		}
		if (p.x.empty()) continue;
		mcs_window_probes pr = p.c(mbFeatDim, havingMasks);
		mcs_frame_view fv = view(b, nullptr, mbFeatDim, havingMasks);
		std::vector<int32_t> match(p.x.size(), -1);
		int32_t nfound = 0;
		check(mcs_window_best(ctx(), &pr, &fv, TH_HIGH_, 0, mbFeatDim, MCS_MEM_HOST, match.data(), nullptr, &nfound), "mcs_window_best");
		for (size_t k = 0; k < match.size(); ++k) vnMatch[dI][p.src[k]] = match[k];
	}
	int nFound = 0;
	for (int i1 = 0; i1 < N1; ++i1)
	{
		const int idx2 = vnMatch[0][i1];
		if (idx2 >= 0 && vnMatch[1][idx2] == i1) { vpMatches12[i1] = vpMapPoints2[idx2]; ++nFound; }
	}
	return nFound;
}

// ---- SearchForTriangulationBetweenCameras: features of cam1 without a map point into cam2 of the same keyframe -----------------------------------------
int cORBmatcher::SearchForTriangulationBetweenCameras(cMultiKeyFrame *pKF1, const int cam1, const int cam2, std::vector<cv::KeyPoint> &vMatchedKeys1,
	std::vector<cv::Vec3d> &vMatchedKeysRays1, std::vector<cv::KeyPoint> &vMatchedKeys2, std::vector<cv::Vec3d> &vMatchedKeysRays2,
	std::vector<std::pair<size_t, size_t> > &vMatchedPairs)
{
	cv::Matx44d RelOri = cConverter::invMat(pKF1->camSystem.Get_M_c(cam1)) * pKF1->camSystem.Get_M_c(cam2);
	cv::Matx33d E12 = ComputeE(RelOri);
	cv::Matx33d Rrel = cConverter::Hom2R(RelOri).t();
	cv::Vec3d trel = -Rrel * cConverter::Hom2T(RelOri);
	vector<cMapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
	vector<cv::KeyPoint> vKeys1 = pKF1->GetKeyPoints();
	vector<cv::Vec3d> vKeysRays1 = pKF1->GetKeyPointsRays();
	Flat a = flatten(pKF1, mbFeatDim, havingMasks);
	std::vector<double> pts; std::vector<int32_t> pc, owner;
	for (size_t idx1 = 0; idx1 < vpMapPoints1.size(); ++idx1)
	{
		if (vpMapPoints1[idx1] || a.cam[idx1] != cam1) continue;
		cv::Vec3d rotPoint = Rrel * vKeysRays1[idx1] + trel;
		rotPoint /= norm(rotPoint);
		pts.push_back(rotPoint(0)); pts.push_back(rotPoint(1)); pts.push_back(rotPoint(2)); pc.push_back(cam2); owner.push_back((int)idx1);
	}
	if (owner.empty()) return 0;
	std::vector<double> uv; std::vector<uint8_t> fl;
	project(pKF1->camSystem, pts, pc, uv, fl, true);
	Probes p;
	for (size_t k = 0; k < owner.size(); ++k) if (fl[k] & 1) p.add(uv[2 * k], uv[2 * k + 1], 40, -1, -1, cam2, owner[k]);
	if (p.x.empty()) return 0;
	p.rows(a, mbFeatDim, havingMasks);
	mcs_window_probes pr = p.c(mbFeatDim, havingMasks);
	mcs_frame_view fv = view(a, nullptr, mbFeatDim, havingMasks);
	std::vector<int32_t> match(p.x.size(), -1);
	int32_t nfound = 0;
	check(mcs_window_best(ctx(), &pr, &fv, 100, 0, mbFeatDim, MCS_MEM_HOST, match.data(), nullptr, &nfound), "mcs_window_best");
	int nmatches = 0;
	for (size_t k = 0; k < match.size(); ++k)
	{
		if (match[k] < 0) continue;
		const int idx1 = p.src[k], bestIdx2 = match[k];
		if (!CheckDistEpipolarLine(vKeysRays1[idx1], pKF1->GetKeyPointRay(bestIdx2), E12, 1e-2)) continue;
		vMatchedKeys1.push_back(vKeys1[idx1]); vMatchedKeys2.push_back(vKeys1[bestIdx2]);
		vMatchedKeysRays1.push_back(vKeysRays1[idx1]); vMatchedKeysRays2.push_back(vKeysRays1[bestIdx2]);
		vMatchedPairs.push_back(make_pair((size_t)idx1, (size_t)bestIdx2));
		++nmatches;
	}
	return nmatches;
}

// ---- SearchByProjection(pKF, Scw, vpPoints, vpMatched, th): the loop-closing matcher (:2265-2392, called from cLoopClosing.cpp:401) ---------------------
// Projection, mirror mask, depth gate and predicted level as in the reference; the window search is mcs_window_best with skip_taken = 1 (features
// with vpMatched[idx] set are skipped, an accepted feature is taken at once).  Two peculiarities of the reference body are kept:
//   * the camera of list entry iMP is keypoint_to_cam[iMP] (:2308) — the LIST position looked up as a feature index.  Where iMP is no feature
//     index of the keyframe the reference dereferences end(); such entries are skipped here.
//   * the descriptor compared for rig-wide feature idx of camera c is row idx of mDescriptors[c] (:2364-2372), not the feature's own row.  That row
//     exists for every feature of camera 0 (there idx IS the local row) and for those idx < rows(c) of the other cameras — reproduced literally.
//     Beyond the matrix the reference reads foreign memory; there the feature's own row is used (the evident intent).  Bounds-safe, and identical to
//     the reference wherever the reference is defined.
//   * `bestIdx > 0` (:2386): feature 0 can be the best candidate but is never accepted.  The device routine has no such rule, so from the first probe
//     that would take feature 0 on, the loop is finished on the host (same arithmetic: DescriptorDistance64[Masked] over the same rows).
int cORBmatcher::SearchByProjection(cMultiKeyFrame* pKF, cv::Matx44d Scw, const std::vector<cMapPoint*>& vpPoints, std::vector<cMapPoint*>& vpMatched, int th)
{
	cMultiCamSys_ camSys = pKF->camSystem;
	const cv::Vec3d Ow = pose_from_sim3(camSys, Scw);
	const int nMaxLevel = pKF->GetScaleLevels() - 1;
	vector<double> vfScaleFactors = pKF->GetScaleFactors();
	const int dim = mbFeatDim;
	const bool masks = havingMasks;
	set<cMapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
	spAlreadyFound.erase(static_cast<cMapPoint*>(NULL));

	std::vector<double> pts; std::vector<int32_t> pc, owner;
	for (int iMP = 0, iend = (int)vpPoints.size(); iMP < iend; ++iMP)
	{
		cMapPoint* pMP = vpPoints[iMP];
		if (!pMP || pMP->isBad() || spAlreadyFound.count(pMP)) continue;
		std::unordered_map<size_t, int>::const_iterator kc = pKF->keypoint_to_cam.find(iMP);
		if (kc == pKF->keypoint_to_cam.end()) continue;
		cv::Vec3d p3Dw = pMP->GetWorldPos();
		pts.push_back(p3Dw(0)); pts.push_back(p3Dw(1)); pts.push_back(p3Dw(2)); pc.push_back(kc->second); owner.push_back(iMP);
	}
	if (owner.empty()) return 0;
	std::vector<double> uv; std::vector<uint8_t> fl;
	project(camSys, pts, pc, uv, fl);
	Probes p;
	for (size_t k = 0; k < owner.size(); ++k)
	{
		if (!(fl[k] & 1)) continue;
		cMapPoint* pMP = vpPoints[owner[k]];
		const double maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
		cv::Vec3d PO = pMP->GetWorldPos() - Ow;
		const double dist3D = cv::norm(PO);
		if (dist3D < minDistance || dist3D > maxDistance) continue;
		const double ratio = dist3D / minDistance;
		vector<double>::iterator it = std::lower_bound(vfScaleFactors.begin(), vfScaleFactors.end(), ratio);
		const int nPredictedLevel = std::min(static_cast<int>(it - vfScaleFactors.begin()), nMaxLevel);
		p.add(uv[2 * k], uv[2 * k + 1], th * pKF->GetScaleFactor(nPredictedLevel), nPredictedLevel - 1, nPredictedLevel, pc[k], owner[k]);
		const uchar* dp = (const uchar*)pMP->GetDescriptorPtr();
		p.d.insert(p.d.end(), dp, dp + dim);
		if (masks) { const uchar* mp = (const uchar*)pMP->GetDescriptorMaskPtr(); p.m.insert(p.m.end(), mp, mp + dim); }
	}
	if (p.x.empty()) return 0;

	// the keyframe's features with the descriptor row the reference compares for each of them (see above)
	Flat b = flatten(pKF, dim, masks);
	std::vector<int> rowsOfCam(camSys.GetNrCams(), 0);
	for (int i = 0; i < b.n; ++i) ++rowsOfCam[b.cam[i]];
	for (int i = 0; i < b.n; ++i)
	{
		const int c = b.cam[i];
		if (i >= rowsOfCam[c]) continue;   // past the end of mDescriptors[c]: the feature's own row stays
		std::memcpy(&b.d[(size_t)i * dim], pKF->GetDescriptorRowPtr(c, i), dim);
		if (masks) std::memcpy(&b.m[(size_t)i * dim], pKF->GetDescriptorMaskRowPtr(c, i), dim);
	}
	std::vector<uint8_t> taken(std::max(b.n, 1), 0);
	for (int i = 0; i < b.n && i < (int)vpMatched.size(); ++i) taken[i] = vpMatched[i] != NULL;
	mcs_window_probes pr = p.c(dim, masks);
	mcs_frame_view fv = view(b, taken.data(), dim, masks);
	std::vector<int32_t> match(p.x.size(), -1);
	int32_t nfound = 0;
	check(mcs_window_best(ctx(), &pr, &fv, TH_LOW_, 1, dim, MCS_MEM_HOST, match.data(), nullptr, &nfound), "mcs_window_best");

	size_t hostFrom = match.size();
	for (size_t k = 0; k < match.size(); ++k) if (match[k] == 0) { hostFrom = k; break; }
	int nmatches = 0;
	for (size_t k = 0; k < hostFrom; ++k)
		if (match[k] > 0) { vpMatched[match[k]] = vpPoints[p.src[k]]; ++nmatches; }
	// the tail after a probe that preferred feature 0: the reference's loop itself, on the rows gathered above
	for (size_t k = hostFrom; k < match.size(); ++k)
	{
		vector<size_t> vIndices = pKF->GetFeaturesInArea(p.cam[k], p.x[k], p.y[k], p.r[k]);
		const uint64_t* dMP = (const uint64_t*)&p.d[k * dim];
		const uint64_t* mMP = masks ? (const uint64_t*)&p.m[k * dim] : nullptr;
		int bestDist = INT_MAX, bestIdx = -1;
		for (size_t idx : vIndices)
		{
			if (vpMatched[idx]) continue;
			const int kpLevel = b.keys[idx].octave;
			if (kpLevel < p.lo[k] || kpLevel > p.hi[k]) continue;
			const uint64_t* dKF = (const uint64_t*)&b.d[idx * dim];
			const int dist = masks ? DescriptorDistance64Masked(dMP, dKF, mMP, (const uint64_t*)&b.m[idx * dim], dim) : DescriptorDistance64(dMP, dKF, dim);
			if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
		}
		if (bestDist <= TH_LOW_ && bestIdx > 0) { vpMatched[bestIdx] = vpPoints[p.src[k]]; ++nmatches; }
	}
	return nmatches;
}

// ---- SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (:2120-2263): no caller in the reference (cTracking::Relocalisation's calls are commented out) ----
// Every good map point of the keyframe that is not in sAlreadyFound is projected into EVERY camera of the current frame's rig (pose of the frame); inside the mirror
// mask it gets a window of radius th * scale(predicted level) over the levels predicted - 1 .. predicted + 1, features that already hold a map point are skipped, the
// nearest one wins if its distance is <= ORBdist and is taken at once: mcs_window_best with skip_taken = 1, probes in the reference's order (point, then camera).
// One peculiarity of the reference body is kept: the descriptor compared for feature i2 of the CURRENT FRAME is the KEYFRAME's row
// pKF->GetDescriptorRowPtr(cam, pKF->cont_idx_to_local_cam_idx[i2]) (:2196-2197) — the frame's feature index looked up in the keyframe's index map, the row taken from the
// probe camera's matrix.  Where that is defined (i2 is a feature index of the keyframe and the local index is a row of that camera's matrix) it is reproduced literally; where
// the reference dereferences end() or reads past the matrix, the frame feature's own row is used (the evident intent).  mbCheckOrientation through mcs_rotation_consistency.
int cORBmatcher::SearchByProjection(cMultiFrame& CurrentFrame, cMultiKeyFrame* pKF, const std::set<cMapPoint*>& sAlreadyFound, double th, int ORBdist)
{
	cMultiCamSys_& camSys = CurrentFrame.camSystem;
	cv::Matx44d Tcurr = CurrentFrame.GetPose();
	const cv::Matx33d Rcw = Tcurr.get_minor<3, 3>(0, 0);
	const cv::Vec3d tcw(Tcurr(0, 3), Tcurr(1, 3), Tcurr(2, 3));
	const cv::Vec3d Ow = -Rcw.t() * tcw;
	const int dim = mbFeatDim, nr = camSys.GetNrCams();
	const bool masks = havingMasks;
	Flat b = flatten(CurrentFrame, dim, masks);
	vector<cMapPoint*> vpMPs = pKF->GetMapPointMatches();
	std::vector<double> pts; std::vector<int32_t> pc, owner;
	for (int i = 0, iend = (int)vpMPs.size(); i < iend; ++i)
	{
		cMapPoint* pMP = vpMPs[i];
		if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
		cv::Vec3d X = pMP->GetWorldPos();
		for (int cam = 0; cam < nr; ++cam) { pts.push_back(X(0)); pts.push_back(X(1)); pts.push_back(X(2)); pc.push_back(cam); owner.push_back(i); }
	}
	if (owner.empty() || b.n == 0) return 0;
	std::vector<double> uv; std::vector<uint8_t> fl;
	project(camSys, pts, pc, uv, fl);
	Probes p;
	for (size_t k = 0; k < owner.size(); ++k)
	{
		if (!(fl[k] & 1)) continue;
		cMapPoint* pMP = vpMPs[owner[k]];
		const double minDistance = pMP->GetMinDistanceInvariance();
		cv::Vec3d PO = pMP->GetWorldPos() - Ow;
		const double dist3D = cv::norm(PO);
		const double ratio = dist3D / minDistance;
		vector<double>::iterator it = std::lower_bound(CurrentFrame.mvScaleFactors.begin(), CurrentFrame.mvScaleFactors.end(), ratio);
		const int nPredictedLevel = std::min(static_cast<int>(it - CurrentFrame.mvScaleFactors.begin()), CurrentFrame.mnScaleLevels - 1);
		p.add(uv[2 * k], uv[2 * k + 1], th * CurrentFrame.mvScaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel + 1, pc[k], owner[k]);
		const uchar* dp = (const uchar*)pMP->GetDescriptorPtr();
		p.d.insert(p.d.end(), dp, dp + dim);
		if (masks) { const uchar* mp = (const uchar*)pMP->GetDescriptorMaskPtr(); p.m.insert(p.m.end(), mp, mp + dim); }
	}
	if (p.x.empty()) return 0;
	// the frame's features with the row the reference compares for each of them (see above)
	{
		Flat kf = flatten(pKF, dim, masks);
		int nrKF = nr;   // the keyframe may carry camera indices the current frame does not have (undefined in the reference; bounds-safe here)
		for (int i = 0; i < kf.n; ++i) nrKF = std::max(nrKF, kf.cam[i] + 1);
		std::vector<int> rowsOfCam(nrKF, 0);
		for (int i = 0; i < kf.n; ++i) if (kf.cam[i] >= 0) ++rowsOfCam[kf.cam[i]];
		for (int i2 = 0; i2 < b.n; ++i2)
		{
			std::unordered_map<size_t, int>::const_iterator it = pKF->cont_idx_to_local_cam_idx.find(i2);
			const int c = b.cam[i2];
			if (it == pKF->cont_idx_to_local_cam_idx.end() || c < 0 || c >= nrKF || it->second < 0 || it->second >= rowsOfCam[c]) continue;   // undefined in the reference: the feature's own row stays
			std::memcpy(&b.d[(size_t)i2 * dim], pKF->GetDescriptorRowPtr(c, it->second), dim);
			if (masks) std::memcpy(&b.m[(size_t)i2 * dim], pKF->GetDescriptorMaskRowPtr(c, it->second), dim);
		}
	}
	std::vector<uint8_t> taken(b.n);
	for (int i2 = 0; i2 < b.n; ++i2) taken[i2] = CurrentFrame.mvpMapPoints[i2] != NULL;
	mcs_window_probes pr = p.c(dim, masks);
	mcs_frame_view fv = view(b, taken.data(), dim, masks);
	std::vector<int32_t> match(p.x.size(), -1), mcur(b.n, -1);
	int32_t n = 0;
	check(mcs_window_best(ctx(), &pr, &fv, ORBdist, 1, dim, MCS_MEM_HOST, match.data(), nullptr, &n), "mcs_window_best");
	for (size_t k = 0; k < match.size(); ++k) if (match[k] >= 0) mcur[match[k]] = p.src[k];
	if (mbCheckOrientation)
	{
		Flat kf = flatten(pKF, dim, false);
		int32_t removed = 0;
		check(mcs_rotation_consistency(ctx(), 0, &b.keys[0].angle, sizeof(mcs_keypoint), &kf.keys[0].angle, sizeof(mcs_keypoint), nullptr, mcur.data(), b.n, kf.n, 1,
			MCS_MEM_HOST, &removed), "mcs_rotation_consistency");
		n -= removed;
	}
	for (int i2 = 0; i2 < b.n; ++i2) if (mcur[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = vpMPs[mcur[i2]];
	return n;
}
}

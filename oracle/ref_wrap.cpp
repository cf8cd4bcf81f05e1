// ref_wrap.cpp — TEST INFRASTRUCTURE: C entry point around the REFERENCE's own mdBRIEFextractorOct (compiled from /root/reference/src, unmodified,
// against oracle/cvshim).  Used by tests/test_oracle_vs_ref.py and tools/gen_golden_ref.py to pin the oracle's restatement.
#include "mdBRIEFextractorOct.h"
#include "cam_model_omni.h"

using namespace MultiColSLAM;

// The reference's DistributeOctTree breaks ties between equally full nodes by comparing the HEAP ADDRESSES of its std::list nodes
// (sort of pair<int, ExtractorNode*>, src/mdBRIEFextractorOct.cpp:745-760), so its output depends on the allocator.  Inside this library every
// allocation made during a ref_extract call comes from a bump arena (monotonically increasing addresses, nothing reused), which makes "pointer
// order" = "creation order" — the order the oracle (and the GPU oct-tree) use for the same tie.  Linked with -Bsymbolic: only this .so is affected.
#include <atomic>
#include <cstdlib>
#include <new>
#include <sys/mman.h>
namespace {
// The arena serves only the thread that switched it on (the test's calling thread).  When this library is loaded together with libmcs_hip.so (the
// drop-in builds), the HIP / HSA runtime's own threads can resolve operator new to the definitions below; they must get plain malloc memory, and the
// bump pointer must not race with them.
char* g_arena = nullptr; size_t g_cap = 0; std::atomic<size_t> g_used{0}; thread_local bool g_on = false;
void* bump(size_t n) {
	n = (n + 15) & ~size_t(15);
	if (!g_on) return std::malloc(n);
	const size_t at = g_used.fetch_add(n);
	if (at + n > g_cap) return std::malloc(n);
	return g_arena + at;
}
bool in_arena(void* p) { return (char*)p >= g_arena && (char*)p < g_arena + g_cap; }
}  // namespace
// The arena is never rewound: function-local statics of the reference / libstdc++ that were first touched while it was on must stay valid.
// 64 GiB of address space are reserved (MAP_NORESERVE), only touched pages cost memory.
static void arena_on() {
	if (!g_arena) {
		g_cap = size_t(64) << 30;
		void* p = mmap(nullptr, g_cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
		if (p == MAP_FAILED) { g_cap = size_t(2) << 30; p = std::malloc(g_cap); }
		g_arena = (char*)p;
	}
	g_on = true;
}
extern "C" void ref_arena(int on) { if (on) arena_on(); else g_on = false; }   // scenes of ref_wrap_match.cpp keep it on for their whole life
namespace {
struct ArenaScope {
	ArenaScope() { arena_on(); }
	~ArenaScope() { g_on = false; }
};
}  // namespace
void* operator new(size_t n) { void* p = bump(n); if (!p) throw std::bad_alloc(); return p; }
void* operator new[](size_t n) { void* p = bump(n); if (!p) throw std::bad_alloc(); return p; }
void operator delete(void* p) noexcept { if (p && !in_arena(p)) std::free(p); }
void operator delete[](void* p) noexcept { if (p && !in_arena(p)) std::free(p); }
void operator delete(void* p, size_t) noexcept { if (p && !in_arena(p)) std::free(p); }
void operator delete[](void* p, size_t) noexcept { if (p && !in_arena(p)) std::free(p); }

extern "C" int ref_extract(const orc_params* p, const uint8_t* img, int w, int h, int stride, const uint8_t* mask, int mstride, const orc_ocam* cam,
                           orc_keypoint* kps, int cap, uint8_t* desc, uint8_t* dmask) {
	try {
		ArenaScope arena;   // everything below (and all its temporaries) dies before this scope ends
		cv::Mat_<double> poly(cam->p_deg, 1), invpoly(cam->invP_deg, 1);
		for (int i = 0; i < cam->p_deg; ++i) poly.at<double>(i, 0) = cam->p[i];
		for (int i = 0; i < cam->invP_deg; ++i) invpoly.at<double>(i, 0) = cam->invP[i];
		double cdeu0v0[5] = {cam->c, cam->d, cam->e, cam->u0, cam->v0};
		cCamModelGeneral_ camModel(cdeu0v0, poly, invpoly, cam->width, cam->height);
		mdBRIEFextractorOct ex(p->nfeatures, p->scaleFactor, p->nlevels, p->edgeThreshold, p->firstLevel, p->scoreType, p->patchSize, p->fastThreshold,
		                       p->useAgast != 0, p->fastAgastType, p->do_dBrief != 0, p->learnMasks != 0, p->descSize);
		cv::Mat image(h, w, CV_8UC1, (void*)img, (size_t)stride), m;
		if (mask) m = cv::Mat(h, w, CV_8UC1, (void*)mask, (size_t)mstride);
		std::vector<cv::KeyPoint> keys;
		cv::Mat descriptors, descriptorMasks;
		ex(image, m, keys, camModel, descriptors, descriptorMasks);
		const int n = (int)keys.size();
		if (n > cap) return -2;
		for (int i = 0; i < n; ++i) {
			kps[i].x = keys[i].pt.x; kps[i].y = keys[i].pt.y; kps[i].size = keys[i].size; kps[i].angle = keys[i].angle; kps[i].response = keys[i].response;
			kps[i].octave = keys[i].octave; kps[i].class_id = keys[i].class_id;
			std::memcpy(desc + (size_t)i * p->descSize, descriptors.ptr<uchar>(i), p->descSize);
			std::memcpy(dmask + (size_t)i * p->descSize, descriptorMasks.ptr<uchar>(i), p->descSize);
		}
		return n;
	} catch (const std::exception& e) {
		std::cerr << "ref_extract: " << e.what() << std::endl;
		return -1;
	}
}

extern "C" void ref_world2img(const orc_ocam* cam, double x, double y, double z, double* u, double* v) {
	cv::Mat_<double> poly(cam->p_deg, 1), invpoly(cam->invP_deg, 1);
	for (int i = 0; i < cam->p_deg; ++i) poly.at<double>(i, 0) = cam->p[i];
	for (int i = 0; i < cam->invP_deg; ++i) invpoly.at<double>(i, 0) = cam->invP[i];
	double cdeu0v0[5] = {cam->c, cam->d, cam->e, cam->u0, cam->v0};
	cCamModelGeneral_ camModel(cdeu0v0, poly, invpoly, cam->width, cam->height);
	camModel.WorldToImg(x, y, z, *u, *v);
}

extern "C" void ref_img2world(const orc_ocam* cam, double u, double v, double* x, double* y, double* z) {
	cv::Mat_<double> poly(cam->p_deg, 1), invpoly(cam->invP_deg, 1);
	for (int i = 0; i < cam->p_deg; ++i) poly.at<double>(i, 0) = cam->p[i];
	for (int i = 0; i < cam->invP_deg; ++i) invpoly.at<double>(i, 0) = cam->invP[i];
	double cdeu0v0[5] = {cam->c, cam->d, cam->e, cam->u0, cam->v0};
	cCamModelGeneral_ camModel(cdeu0v0, poly, invpoly, cam->width, cam->height);
	camModel.ImgToWorld(*x, *y, *z, u, v);
}

// ---------------------------------------------------------------- DBoW2 (ThirdParty/DBoW2, compiled unmodified): vocabulary load + transform
#include "cORBVocabulary.h"

// descriptors: n rows of 32 bytes.  Outputs: node id per feature (-1 if the feature is not in the FeatureVector), the BowVector as (ids, values)
// up to cap entries (returns its size), vocabulary size in *nnodes_words[0..1].
extern "C" int ref_bow_transform(const char* vocPath, const uint8_t* desc, int n, int levelsup, int* node, int* bowIds, double* bowVals, int cap, int* info) {
	try {
		ORBVocabulary voc;
		voc.load(std::string(vocPath));
		std::vector<cv::Mat> feats;
		for (int i = 0; i < n; ++i) feats.push_back(cv::Mat(1, 32, CV_8U, (void*)(desc + 32 * (size_t)i)));
		DBoW2::BowVector bv; DBoW2::FeatureVector fv;
		voc.transform(feats, bv, fv, levelsup);
		for (int i = 0; i < n; ++i) node[i] = -1;
		for (auto& e : fv) for (unsigned i : e.second) node[i] = (int)e.first;
		int k = 0;
		for (auto& e : bv) { if (k < cap) { bowIds[k] = (int)e.first; bowVals[k] = e.second; } ++k; }
		info[0] = (int)voc.size(); info[1] = voc.getBranchingFactor(); info[2] = voc.getDepthLevels();
		return k;
	} catch (const std::exception& e) { std::cerr << "ref_bow_transform: " << e.what() << std::endl; return -1; }
	catch (const std::string& e) { std::cerr << "ref_bow_transform: " << e << std::endl; return -1; }
}

// ---------------------------------------------------------------- camera system, converter, misc (src/cam_system_omni.cpp, cConverter.cpp, misc.cpp)
#include "cam_system_omni.h"
#include "cConverter.h"
#include "misc.h"

static cCamModelGeneral_ make_cam(const orc_ocam* cam, const uint8_t* mask) {
	cv::Mat_<double> poly(cam->p_deg, 1), invpoly(cam->invP_deg, 1);
	for (int i = 0; i < cam->p_deg; ++i) poly.at<double>(i, 0) = cam->p[i];
	for (int i = 0; i < cam->invP_deg; ++i) invpoly.at<double>(i, 0) = cam->invP[i];
	double cdeu0v0[5] = {cam->c, cam->d, cam->e, cam->u0, cam->v0};
	cCamModelGeneral_ m(cdeu0v0, poly, invpoly, cam->width, cam->height);
	std::vector<cv::Mat> masks;
	if (mask) masks.push_back(cv::Mat(cam->height, cam->width, CV_8UC1, (void*)mask).clone());
	else masks.push_back(cv::Mat::ones(cv::Size(cam->width, cam->height), CV_8UC1));
	m.SetMirrorMasks(masks);
	return m;
}

// cMultiCamSys_(M_t, M_c, camModels) -> for every point: WorldToCamHom_fast(cam, Vec4d) (returns z <= 0) + isPointInMirrorMask(u, v, 0);
// also returns the MtMc_inv matrices the reference computed (invMat(M_t * M_c[c])).
extern "C" int ref_world_to_cam(const double* M_t, const double* M_c, const orc_ocam* cams, const uint8_t* const* masks, int nrCams, const double* pts3,
                                const int* pcam, int n, double* uv, uint8_t* flags, double* MtMc_inv_out) {
	try {
		cv::Matx44d Mt; std::memcpy(Mt.val, M_t, 128);
		std::vector<cv::Matx44d> Mc(nrCams);
		std::vector<cCamModelGeneral_> models;
		for (int c = 0; c < nrCams; ++c) { std::memcpy(Mc[c].val, M_c + 16 * c, 128); models.push_back(make_cam(&cams[c], masks ? masks[c] : nullptr)); }
		cMultiCamSys_ sys(Mt, Mc, models);
		for (int c = 0; c < nrCams; ++c) { cv::Matx44d inv = sys.Get_MtMc_inv(c); std::memcpy(MtMc_inv_out + 16 * c, inv.val, 128); }
		for (int i = 0; i < n; ++i) {
			cv::Vec4d p4(pts3[3 * i], pts3[3 * i + 1], pts3[3 * i + 2], 1.0);
			cv::Vec2d px(0.0, 0.0);
			const bool behind = sys.WorldToCamHom_fast(pcam[i], p4, px);
			cv::Vec3d p3(pts3[3 * i], pts3[3 * i + 1], pts3[3 * i + 2]);
			cv::Vec2d px3(0.0, 0.0);
			sys.WorldToCamHom_fast(pcam[i], p3, px3);   // the Vec3d overload must agree with the Vec4d one
			if (px3(0) != px(0) && !(px3(0) != px3(0) && px(0) != px(0))) return -3;
			uv[2 * i] = px(0); uv[2 * i + 1] = px(1);
			uint8_t fl = behind ? 2 : 0;
			if (sys.GetCamModelObj(pcam[i]).isPointInMirrorMask(px(0), px(1), 0)) fl |= 1;
			flags[i] = fl;
		}
		return 0;
	} catch (const std::exception& e) { std::cerr << "ref_world_to_cam: " << e.what() << std::endl; return -1; }
}

extern "C" int ref_check_epipolar(const double* ray1, const double* ray2, const double* E, double thresh) {
	cv::Vec3d r1(ray1[0], ray1[1], ray1[2]), r2(ray2[0], ray2[1], ray2[2]);
	cv::Matx33d E12; std::memcpy(E12.val, E, 72);
	return CheckDistEpipolarLine(r1, r2, E12, thresh) ? 1 : 0;
}

extern "C" int ref_median_int(const int* v, int n) { std::vector<int> x(v, v + n); return median(x); }

extern "C" void ref_cayley2hom(const double* c6, double* M) {
	cv::Matx61d c; std::memcpy(c.val, c6, 48);
	cv::Matx44d H = cayley2hom<double>(c);
	std::memcpy(M, H.val, 128);
}

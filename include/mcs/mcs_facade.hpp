// mcs_facade.hpp — header-only C++ host facade over the C ABI (include/mcs_c.h) with the reference's class names, so
// that cTracking / cLocalMapping / cLoopClosing compile against it unchanged in spirit:
//
//   MultiColSLAM::mdBRIEFextractorOct   include/mdBRIEFextractorOct.h:335-421 (13-argument ctor, operator(), getters)
//   MultiColSLAM::ORBextractor          include/cORBextractor.h:48-67 (the ORB-mode extractor behind its own constructor / operator())
//   MultiColSLAM::cORBmatcher           include/cORBmatcher.h:43-133 (brute-force and grid-window searches on flat views)
//   MultiColSLAM::cMultiCamSys_ / LoadMCS / cORBVocabulary / ComputeDistinctiveDescriptors   the callers either side of the path
//   MultiColSLAM::DescriptorDistance64[_Masked]   src/cORBmatcher.cpp:2438-2474
//
// OpenCV is not a dependency: minimal layout-compatible PODs stand in for cv::KeyPoint / cv::Mat / cv::Vec3d.  With
// -DMCS_WITH_OPENCV the adapters at the bottom accept the real types (cv::KeyPoint is 28 bytes, same layout).
// Link with -lmcs_hip.  No CPU fallback: every call needs a HIP device and throws std::runtime_error otherwise.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../mcs_c.h"

namespace MultiColSLAM {

struct KeyPoint { float ptx, pty, size, angle, response; int32_t octave, class_id; };   // == cv::KeyPoint, 28 B
static_assert(sizeof(KeyPoint) == 28 && sizeof(mcs_keypoint) == 28, "cv::KeyPoint layout");
struct Vec3d { double v[3]; };

struct Mat8u {   // CV_8UC1 matrix view / owner (rows x cols, step bytes per row)
	int rows = 0, cols = 0, step = 0;
	std::vector<uint8_t> store;
	uint8_t* data = nullptr;
	void create(int r, int c) { rows = r; cols = c; step = c; store.assign((size_t)r * c, 0); data = store.data(); }
	void release() { rows = cols = step = 0; store.clear(); data = nullptr; }
	bool empty() const { return rows == 0 || cols == 0; }
	const uint64_t* ptr64(int r) const { return reinterpret_cast<const uint64_t*>(data + (size_t)r * step); }
};

inline void mcs_throw(int rc) { if (rc != MCS_OK) throw std::runtime_error(std::string("libmcs_hip: ") + mcs_last_error()); }

class Context {
public:
	explicit Context(int device = 0, void* hipStream = nullptr) { mcs_throw(mcs_ctx_create(device, hipStream, &h)); }
	~Context() { mcs_ctx_destroy(h); }
	Context(const Context&) = delete;
	Context& operator=(const Context&) = delete;
	// A pipelined host (include/mcs_c.h: "Long transfers beside the step"): results leave on resultStream() through copyNarrow(), images arrive by hipMemcpyAsync on
	// transferStream(), a stream the library has probed to share a hardware queue with none of the streams the extraction runs on.
	void* resultStream() const { void* s = nullptr; mcs_throw(mcs_ctx_result_stream(h, &s)); return s; }
	void* transferStream(unsigned* conflicts = nullptr) const { void* s = nullptr; mcs_throw(mcs_ctx_transfer_stream(h, &s, conflicts)); return s; }
	unsigned streamConflicts(void* hipStream) const { unsigned m = 0; mcs_throw(mcs_ctx_stream_conflicts(h, hipStream, &m)); return m; }
	void copyNarrow(void* dst, const void* src, size_t bytes, void* hipStream, int workgroups = 2) const { mcs_throw(mcs_copy_narrow(h, dst, src, bytes, workgroups, hipStream)); }
	void synchronize() const { mcs_throw(mcs_ctx_synchronize(h)); }
	mcs_ctx* h = nullptr;
};

// cCamModelGeneral_ (include/cam_model_omni.h): calibration + level-0 mirror mask
struct cCamModelGeneral_ {
	mcs_ocam ocam{};
	Mat8u mirrorMask0;
	double GetWidth() const { return ocam.width; }
	double GetHeight() const { return ocam.height; }
	const Mat8u& GetMirrorMask(int) const { return mirrorMask0; }
};

class mdBRIEFextractorOct {
public:
	enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
	mdBRIEFextractorOct(Context& ctx, int _nfeatures = 1000, float _scaleFactor = 1.2f, int _nlevels = 8, int _edgeThreshold = 25,
	                    int _firstLevel = 0, int _scoreType = HARRIS_SCORE, int _patchSize = 32, int _fastThreshold = 20, bool _useAgast = false,
	                    int _fastAgastType = 2, bool _do_dBrief = false, bool _learnMasks = false, int _descSize = 32)
	    : ctx_(ctx), p_{_nfeatures, _scaleFactor, _nlevels, _edgeThreshold, _firstLevel, _scoreType, _patchSize, _fastThreshold, _useAgast ? 1 : 0,
	                    _fastAgastType, _do_dBrief ? 1 : 0, _learnMasks ? 1 : 0, _descSize} {}
	~mdBRIEFextractorOct() { if (ex_) mcs_extractor_destroy(ex_); }

	// operator()(image, mask, keypoints, camModel, descriptors, descriptorMasks)  (src/mdBRIEFextractorOct.cpp:1244-1337)
	void operator()(const Mat8u& image, const Mat8u& mask, std::vector<KeyPoint>& keypoints, cCamModelGeneral_& camModel, Mat8u& descriptors,
	                Mat8u& descriptorMasks) {
		extract(image, mask, keypoints, &camModel.ocam, descriptors, descriptorMasks);
	}
	// the same with an optional camera model (ORB mode reads none: ORBextractor below)
	void extract(const Mat8u& image, const Mat8u& mask, std::vector<KeyPoint>& keypoints, const mcs_ocam* ocam, Mat8u& descriptors, Mat8u& descriptorMasks) {
		if (image.empty()) return;   // :1252-1253
		ensure(image.cols, image.rows, 1);
		std::vector<mcs_keypoint> kps(cap_);
		std::vector<uint8_t> d((size_t)cap_ * p_.descSize), m((size_t)cap_ * p_.descSize);
		int32_t n = 0;
		mcs_throw(mcs_extract_batch(ex_, 1, image.data, 0, image.step, mask.empty() ? nullptr : mask.data, 0, mask.step, ocam,
		                            MCS_MEM_HOST, &n, kps.data(), d.data(), m.data(), nullptr));
		keypoints.resize(n);
		if (n) std::memcpy(keypoints.data(), kps.data(), (size_t)n * sizeof(KeyPoint));
		if (n == 0) { descriptors.release(); descriptorMasks.release(); return; }   // :1270-1274
		descriptors.create(n, p_.descSize);
		descriptorMasks.create(n, p_.descSize);
		std::memcpy(descriptors.data, d.data(), (size_t)n * p_.descSize);
		std::memcpy(descriptorMasks.data, m.data(), (size_t)n * p_.descSize);
	}

	// All cameras of a rig in one device batch (the GPU analogue of `#pragma omp parallel for num_threads(nrCams)`,
	// src/cMultiFrame.cpp:128); also returns the rays of src/cMultiFrame.cpp:146-152.
	void extractRig(const std::vector<const uint8_t*>& images, int width, int height, int stride, const std::vector<const uint8_t*>& masks,
	                const std::vector<mcs_ocam>& cams, std::vector<std::vector<KeyPoint>>& keys, std::vector<Mat8u>& desc, std::vector<Mat8u>& dmask,
	                std::vector<std::vector<Vec3d>>& rays) {
		const int n = (int)images.size();
		ensure(width, height, n);
		std::vector<uint8_t> img((size_t)n * stride * height), msk;
		for (int i = 0; i < n; ++i) std::memcpy(img.data() + (size_t)i * stride * height, images[i], (size_t)stride * height);
		if (!masks.empty()) {
			msk.resize(img.size());
			for (int i = 0; i < n; ++i) std::memcpy(msk.data() + (size_t)i * stride * height, masks[i], (size_t)stride * height);
		}
		std::vector<int32_t> nkp(n);
		std::vector<mcs_keypoint> kps((size_t)n * cap_);
		std::vector<uint8_t> d((size_t)n * cap_ * p_.descSize), m(d.size());
		std::vector<double> r((size_t)n * cap_ * 3);
		mcs_throw(mcs_extract_batch(ex_, n, img.data(), (size_t)stride * height, stride, msk.empty() ? nullptr : msk.data(), (size_t)stride * height,
		                            stride, cams.data(), MCS_MEM_HOST, nkp.data(), kps.data(), d.data(), m.data(), r.data()));
		keys.assign(n, {}); desc.assign(n, {}); dmask.assign(n, {}); rays.assign(n, {});
		for (int i = 0; i < n; ++i) {
			const int k = nkp[i];
			keys[i].resize(k); rays[i].resize(k);
			if (!k) continue;
			std::memcpy(keys[i].data(), kps.data() + (size_t)i * cap_, (size_t)k * sizeof(KeyPoint));
			std::memcpy(rays[i].data(), r.data() + (size_t)i * cap_ * 3, (size_t)k * sizeof(Vec3d));
			desc[i].create(k, p_.descSize); dmask[i].create(k, p_.descSize);
			std::memcpy(desc[i].data, d.data() + (size_t)i * cap_ * p_.descSize, (size_t)k * p_.descSize);
			std::memcpy(dmask[i].data, m.data() + (size_t)i * cap_ * p_.descSize, (size_t)k * p_.descSize);
		}
	}

	int GetLevels() { return p_.nlevels; }
	double GetScaleFactor() { return (double)p_.scaleFactor; }   // member is the double of the FLOAT ctor argument
	bool GetMasksLearned() { return p_.learnMasks != 0; }
	int GetDescriptorSize() { return p_.descSize; }

private:
	void ensure(int w, int h, int batch) {
		if (ex_ && w == w_ && h == h_ && batch <= batch_) return;
		if (ex_) mcs_extractor_destroy(ex_);
		ex_ = nullptr;
		mcs_throw(mcs_extractor_create(ctx_.h, &p_, w, h, batch, &ex_));
		mcs_throw(mcs_extractor_kp_capacity(ex_, &cap_));
		w_ = w; h_ = h; batch_ = batch;
	}
	Context& ctx_;
	mcs_extractor_params p_;
	mcs_extractor* ex_ = nullptr;
	int w_ = 0, h_ = 0, batch_ = 0, cap_ = 0;
};

// ---------------------------------------------------------------------------------------------- settings / calibration files
// The flat `key: value` subset of OpenCV FileStorage YAML the reference's Examples use (`%YAML:1.0` header, `#` comments).
inline std::map<std::string, std::string> ReadSettings(const std::string& path) {
	std::ifstream f(path);
	if (!f) throw std::runtime_error("cannot open " + path);
	std::map<std::string, std::string> out;
	std::string line;
	while (std::getline(f, line)) {
		const size_t h = line.find('#');
		if (h != std::string::npos) line.erase(h);
		const size_t c = line.find(':');
		if (line.empty() || line[0] == '%' || c == std::string::npos) continue;
		auto trim = [](std::string v) { const size_t a = v.find_first_not_of(" \t\r\""), b = v.find_last_not_of(" \t\r\""); return a == std::string::npos ? std::string() : v.substr(a, b - a + 1); };
		const std::string k = trim(line.substr(0, c)), v = trim(line.substr(c + 1));
		if (!k.empty() && !v.empty()) out[k] = v;
	}
	return out;
}
inline double SettingD(const std::map<std::string, std::string>& s, const std::string& k) {
	auto it = s.find(k);
	if (it == s.end()) throw std::runtime_error("missing setting " + k);
	return std::strtod(it->second.c_str(), nullptr);
}
inline int SettingI(const std::map<std::string, std::string>& s, const std::string& k) { return (int)SettingD(s, k); }

// CreateMirrorMask (src/cam_model_omni.cpp:181-220), level 0: float arithmetic, the principal-point names swapped like the reference
inline void CreateMirrorMask0(const mcs_ocam& cam, Mat8u& mask) {
	const int w = cam.width, h = cam.height;
	const float u0 = (float)cam.v0, v0 = (float)cam.u0;
	mask.create(h, w);
	for (int i = 0; i < h; ++i)
		for (int j = 0; j < w; ++j) {
			const float ans = std::sqrt((float)std::pow((double)(i - u0), 2) + (float)std::pow((double)(j - v0), 2));
			mask.data[(size_t)i * w + j] = ans < (u0 + 22.0f) ? 255 : 0;
		}
}

using Matx44d = std::array<double, 16>;   // row-major
inline Matx44d MatMul(const Matx44d& A, const Matx44d& B) {   // cv::Matx product: s = 0; s += a(i,k) * b(k,j)
	Matx44d C{};
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j]; C[4 * i + j] = s; }
	return C;
}
inline Matx44d InvMat(const Matx44d& M) {   // cConverter::invMat (src/cConverter.cpp:31-44): rigid inverse, t = -(R^T) t
	Matx44d o{};
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[4 * i + j] = M[4 * j + i];
	for (int i = 0; i < 3; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += -o[4 * i + k] * M[4 * k + 3]; o[4 * i + 3] = s; }
	o[15] = 1.0;
	return o;
}
inline Matx44d Cayley2Hom(const double c[6]) {   // include/misc.h:132-160, 211-224
	const double c1 = c[0], c2 = c[1], c3 = c[2], a = c1 * c1, b = c2 * c2, g = c3 * c3, sc = 1.0 / (1.0 + a + b + g);
	const double R[9] = {1 + a - b - g, 2 * (c1 * c2 - c3), 2 * (c1 * c3 + c2), 2 * (c1 * c2 + c3), 1 - a + b - g, 2 * (c2 * c3 - c1),
	                     2 * (c1 * c3 - c2), 2 * (c2 * c3 + c1), 1 - a - b + g};
	Matx44d M{};
	for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) M[4 * i + j] = sc * R[3 * i + j]; M[4 * i + 3] = c[3 + i]; }
	M[15] = 1.0;
	return M;
}

// ORBextractor (include/cORBextractor.h:48-67; the reference declares the class and never compiles an implementation): the extractor in ORB mode behind the
// five-argument constructor and the four-argument operator() — no camera model, no descriptor masks.  scaleFactor is a double in this header (a float in
// mdBRIEFextractorOct's): it is narrowed to the float the extraction tables are built from, and GetScaleFactor() returns what was passed.
class ORBextractor {
public:
	enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
	ORBextractor(Context& ctx, int nfeatures = 1000, double scaleFactor = 1.2, int nlevels = 8, int scoreType = FAST_SCORE, int fastTh = 20)
	    : impl_(ctx, nfeatures, (float)scaleFactor, nlevels, 25, 0, scoreType, 32, fastTh, false, 2, false, false, 32), nlevels_(nlevels), scaleFactor_(scaleFactor) {}
	// operator()(image, mask, keypoints, descriptors)   (include/cORBextractor.h:61-63)
	void operator()(const Mat8u& image, const Mat8u& mask, std::vector<KeyPoint>& keypoints, Mat8u& descriptors) {
		Mat8u unusedMasks;
		impl_.extract(image, mask, keypoints, nullptr, descriptors, unusedMasks);   // ORB mode reads no camera model
	}
	int GetLevels() { return nlevels_; }
	double GetScaleFactor() { return scaleFactor_; }

private:
	mdBRIEFextractorOct impl_;
	int nlevels_;
	double scaleFactor_;
};

// cMultiCamSys_ (include/cam_system_omni.h): calibrations + poses; the projection of many points runs on the GPU in one call
class cMultiCamSys_ {
public:
	std::vector<cCamModelGeneral_> camModels;
	std::vector<Matx44d> M_c, MtMc, MtMc_inv;
	Matx44d M_t{};
	int GetNrCams() const { return (int)camModels.size(); }
	cCamModelGeneral_& GetCamModelObj(int c) { return camModels[c]; }
	void Set_M_t(const Matx44d& M) {   // src/cam_system_omni.cpp:184-198
		M_t = M;
		MtMc.resize(M_c.size()); MtMc_inv.resize(M_c.size());
		for (size_t c = 0; c < M_c.size(); ++c) { MtMc[c] = MatMul(M_t, M_c[c]); MtMc_inv[c] = InvMat(MtMc[c]); }
	}
	// WorldToCamHom_fast + isPointInMirrorMask(u, v, 0) for n points (point i into camera cam[i]): uv = 2 doubles per point,
	// flags bit0 = inside the mirror mask, bit1 = behind the camera (src/cam_system_omni.cpp:92-133, src/cam_model_omni.cpp:163-178)
	void WorldToCamHom_fast(Context& ctx, const double* pts3, const int32_t* cam, int n, double* uv, uint8_t* flags) {
		std::vector<mcs_ocam> oc;
		std::vector<const uint8_t*> masks;
		bool any = false;
		for (auto& m : camModels) { oc.push_back(m.ocam); masks.push_back(m.mirrorMask0.empty() ? nullptr : m.mirrorMask0.data); any = any || !m.mirrorMask0.empty(); }
		std::vector<double> M((size_t)GetNrCams() * 16);
		for (int c = 0; c < GetNrCams(); ++c) std::memcpy(&M[16 * (size_t)c], MtMc_inv[c].data(), 128);
		mcs_throw(mcs_world_to_cam(ctx.h, M.data(), oc.data(), GetNrCams(), any ? masks.data() : nullptr, pts3, cam, n, MCS_MEM_HOST, uv, flags));
	}
};

// cSystem::LoadMCS (src/cSystem.cpp:125-180): MultiCamSys_Calibration.yaml + InteriorOrientationFisheye<c>.yaml, M_t = identity
inline void LoadMCS(const std::string& path2calibrations, cMultiCamSys_& camSystem) {
	const auto mcs = ReadSettings(path2calibrations + "/MultiCamSys_Calibration.yaml");
	const int nrCams = SettingI(mcs, "CameraSystem.nrCams");
	camSystem.camModels.assign(nrCams, {});
	camSystem.M_c.assign(nrCams, {});
	for (int c = 0; c < nrCams; ++c) {
		double cay[6];
		for (int p = 1; p < 7; ++p) cay[p - 1] = SettingD(mcs, "CameraSystem.cam" + std::to_string(c + 1) + "_" + std::to_string(p));
		camSystem.M_c[c] = Cayley2Hom(cay);
		const auto fs = ReadSettings(path2calibrations + "/InteriorOrientationFisheye" + std::to_string(c) + ".yaml");
		mcs_ocam& o = camSystem.camModels[c].ocam;
		std::memset(&o, 0, sizeof(o));
		const int nrpol = SettingI(fs, "Camera.nrpol"), nrinvpol = SettingI(fs, "Camera.nrinvpol");
		if (nrpol > MCS_MAX_POLY || nrinvpol > MCS_MAX_POLY) throw std::runtime_error("polynomial degree above MCS_MAX_POLY");
		for (int i = 0; i < nrpol; ++i) o.p[i] = SettingD(fs, "Camera.a" + std::to_string(i));
		for (int i = 0; i < nrinvpol; ++i) o.invP[i] = SettingD(fs, "Camera.pol" + std::to_string(i));
		o.p_deg = nrpol > 5 ? nrpol : 5; o.invP_deg = nrinvpol > 12 ? nrinvpol : 12;   // cv::Mat::zeros(5,1) / zeros(12,1) in the reference
		o.width = SettingI(fs, "Camera.Iw"); o.height = SettingI(fs, "Camera.Ih");
		o.c = SettingD(fs, "Camera.c"); o.d = SettingD(fs, "Camera.d"); o.e = SettingD(fs, "Camera.e"); o.u0 = SettingD(fs, "Camera.u0"); o.v0 = SettingD(fs, "Camera.v0");
		if (SettingI(fs, "Camera.mirrorMask") == 1) CreateMirrorMask0(o, camSystem.camModels[c].mirrorMask0);
		else { camSystem.camModels[c].mirrorMask0.create(o.height, o.width); std::memset(camSystem.camModels[c].mirrorMask0.data, 1, (size_t)o.width * o.height); }
	}
	Matx44d I{}; I[0] = I[5] = I[10] = I[15] = 1.0;
	camSystem.Set_M_t(I);
}

// ORBVocabulary (include/cORBVocabulary.h = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) as far as cMultiFrame::ComputeBoW uses it
class cORBVocabulary {
public:
	using BowVector = std::map<unsigned, double>;                       // DBoW2::BowVector
	using FeatureVector = std::map<unsigned, std::vector<unsigned>>;    // DBoW2::FeatureVector
	cORBVocabulary(Context& ctx) : ctx_(ctx) {}
	~cORBVocabulary() { if (h_) mcs_vocabulary_destroy(h_); }
	cORBVocabulary(const cORBVocabulary&) = delete;
	cORBVocabulary& operator=(const cORBVocabulary&) = delete;
	// TemplatedVocabulary::load from the OpenCV-YAML vocabulary file (ThirdParty/DBoW2/DBoW2/TemplatedVocabulary.h:1573-1622)
	void load(const std::string& path) {
		std::ifstream f(path);
		if (!f) throw std::runtime_error("cannot open " + path);
		std::stringstream ss; ss << f.rdbuf();
		const std::string t = ss.str();
		auto intAfter = [&](const std::string& key, size_t from, size_t& pos) { pos = t.find(key, from); return pos == std::string::npos ? -1L : std::strtol(t.c_str() + pos + key.size(), nullptr, 10); };
		size_t p;
		m_k = (int)intAfter(" k:", 0, p); m_L = (int)intAfter(" L:", 0, p);   // (" L:" — the header line "%YAML:1.0" also contains "L:")
		if (intAfter("scoringType:", 0, p) != 0 || intAfter("weightingType:", 0, p) != 0) throw std::runtime_error("only TF_IDF / L1 vocabularies are mirrored");
		struct N { int id, parent; double w; uint8_t d[32]; };
		std::vector<N> nodes;
		size_t pos = t.find("nodes:");
		const size_t wordsAt = t.find("words:");
		while (true) {
			size_t q;
			N n{};
			n.id = (int)intAfter("nodeId:", pos, q);
			if (q == std::string::npos || q > wordsAt) break;
			n.parent = (int)intAfter("parentId:", q, p);
			size_t wq = t.find("weight:", q);
			n.w = std::strtod(t.c_str() + wq + 7, nullptr);
			size_t dq = t.find('"', wq) + 1;
			const char* c = t.c_str() + dq;
			for (int i = 0; i < 32; ++i) { char* e; n.d[i] = (uint8_t)std::strtol(c, &e, 10); c = e; }
			nodes.push_back(n);
			pos = t.find('}', dq);
		}
		const int nn = (int)nodes.size() + 1;
		nodeDesc.assign((size_t)nn * 32, 0); weight.assign(nn, 0.0); wordId.assign(nn, -1);
		std::vector<std::vector<int>> ch(nn);
		for (auto& n : nodes) { std::memcpy(&nodeDesc[(size_t)n.id * 32], n.d, 32); weight[n.id] = n.w; ch[n.parent].push_back(n.id); }
		childOff.assign(nn + 1, 0); childIdx.clear();
		for (int i = 0; i < nn; ++i) { childOff[i + 1] = childOff[i] + (int)ch[i].size(); childIdx.insert(childIdx.end(), ch[i].begin(), ch[i].end()); }
		pos = wordsAt;
		while (true) {
			size_t q;
			const long wid = intAfter("wordId:", pos, q);
			if (q == std::string::npos) break;
			wordId[intAfter("nodeId:", q, p)] = (int)wid;
			pos = q + 7;
		}
		if (h_) { mcs_vocabulary_destroy(h_); h_ = nullptr; }
		mcs_throw(mcs_vocabulary_create(ctx_.h, nn, nodeDesc.data(), childOff.data(), childIdx.data(), m_L, &h_));
	}
	// transform(features, v, fv, levelsup) (:1127-1205, TF_IDF weighting, L1 scoring): descriptors = n rows of `stride` >= 32 bytes
	void transform(const uint8_t* descriptors, int n, int stride, BowVector& v, FeatureVector& fv, int levelsup) {
		v.clear(); fv.clear();
		if (!h_ || n <= 0) return;
		std::vector<int32_t> leaf(n), nid(n);
		mcs_throw(mcs_bow_transform(h_, descriptors, n, stride, levelsup, MCS_MEM_HOST, leaf.data(), nid.data()));
		for (int i = 0; i < n; ++i) {
			const double w = weight[leaf[i]];
			if (w > 0) { v[(unsigned)wordId[leaf[i]]] += w; fv[(unsigned)nid[i]].push_back((unsigned)i); }
		}
		double norm = 0.0;
		for (auto& e : v) norm += std::fabs(e.second);
		if (norm > 0.0) for (auto& e : v) e.second /= norm;
	}
	int m_k = 0, m_L = 0;
	std::vector<uint8_t> nodeDesc; std::vector<int32_t> childOff, childIdx, wordId; std::vector<double> weight;
private:
	Context& ctx_;
	mcs_vocabulary* h_ = nullptr;
};

// cMapPoint::ComputeDistinctiveDescriptors (src/cMapPoint.cpp:294-382) for a batch: offsets = CSR rows of the observed descriptors
inline void ComputeDistinctiveDescriptors(Context& ctx, const uint8_t* desc, const uint8_t* mask, int stride, int dim, const std::vector<int32_t>& offsets,
                                          std::vector<int32_t>& bestIdx) {
	const int np = (int)offsets.size() - 1;
	bestIdx.assign(np > 0 ? np : 0, -1);
	if (np > 0) mcs_throw(mcs_distinctive_descriptors(ctx.h, desc, mask, stride, dim, offsets.data(), np, MCS_MEM_HOST, bestIdx.data()));
}

// flat view of a (multi-)keyframe / frame as the brute-force searches see it: all cameras concatenated in mvKeys order
struct FeatureSetView {
	const uint8_t* descriptors = nullptr;   // n x dim
	const uint8_t* masks = nullptr;         // n x dim or nullptr
	std::vector<uint8_t> flag;              // meaning depends on the search (see cORBmatcher)
	std::vector<int32_t> cam;               // keypoint_to_cam
	const double* rays = nullptr;           // n x 3
	int n = 0;
};

class cORBmatcher {
public:
	cORBmatcher(Context& ctx, double nnratio = 0.6, bool checkOri = true, int featDim = 32, bool havingMasks_ = false, int K = 32)
	    : ctx_(ctx), mfNNratio(nnratio), mbFeatDim(featDim), havingMasks(havingMasks_), K_(K) {
		mbCheckOrientation = checkOri;   // the searches below return unfiltered matches; RotationConsistency() is the separate pass of the reference's flag
	}
	// mbCheckOrientation pass (ComputeThreeMaxima, :2394-2436) on a match array (slot -> partner or -1); variant / swapped per search are listed at
	// mcs_rotation_consistency in mcs_c.h.  Returns the number of matches removed; a no-op unless the matcher was built with checkOri = true.
	int RotationConsistency(int variant, const KeyPoint* slotKeys, const KeyPoint* partnerKeys, int nPartner, std::vector<int>& match, bool swapped,
	                        const std::vector<int>* accepted = nullptr) {
		if (!mbCheckOrientation || match.empty() || nPartner <= 0) return 0;
		int32_t removed = 0;
		mcs_throw(mcs_rotation_consistency(ctx_.h, variant, &slotKeys[0].angle, (int)sizeof(KeyPoint), &partnerKeys[0].angle, (int)sizeof(KeyPoint),
		                                   accepted ? accepted->data() : nullptr, match.data(), (int)match.size(), nPartner, swapped ? 1 : 0, MCS_MEM_HOST, &removed));
		return removed;
	}
	// SearchByBoW(pKF1, pKF2, vpMatches12): flag = "has a good map point".  match12[i] = index in kf2 or -1.  (:885-966)
	int SearchByBoW(const FeatureSetView& kf1, const FeatureSetView& kf2, std::vector<int>& match12) {
		return run(0, kf1, kf2, nullptr, 0, match12, kf1.n);
	}
	// SearchByBoW(pKF, F, vpMapPointMatches) without the vocabulary restriction: matchF[j] = keyframe feature or -1.  (:179-323)
	int SearchByBoWFrame(const FeatureSetView& kf, const FeatureSetView& frame, std::vector<int>& matchF) {
		return run(1, kf, frame, nullptr, 0, matchF, frame.n);
	}
	// SearchForTriangulationRaw: flag = "has NO map point", cam + rays required, E = nrCams*nrCams 3x3 row-major.  (:968-1155)
	int SearchForTriangulationRaw(const FeatureSetView& kf1, const FeatureSetView& kf2, const double* E, int nrCams,
	                              std::vector<std::pair<size_t, size_t>>& vMatchedPairs) {
		std::vector<int> m12;
		const int n = run(2, kf1, kf2, E, nrCams, m12, kf1.n);
		vMatchedPairs.clear();
		for (size_t i = 0; i < m12.size(); ++i) if (m12[i] >= 0) vMatchedPairs.emplace_back(i, (size_t)m12[i]);
		return n;
	}

	// ---- grid-window searches (src/cORBmatcher.cpp:326-726, 1990-2118).  A FrameGridView is the flat cMultiFrame the windows are opened in.
	struct FrameGridView {
		const KeyPoint* mvKeys = nullptr; const uint8_t* descriptors = nullptr; const uint8_t* masks = nullptr;
		std::vector<int32_t> keypoint_to_cam; std::vector<uint8_t> hasMapPoint;   // hasMapPoint[i] = mvpMapPoints[i] != NULL (updated by the searches)
		std::vector<int32_t> width, height; std::vector<double> mvScaleFactors; int n = 0;
	};
	// WindowSearch(F1, F2, windowSize, vpMapPointMatches2, minScaleLevel, maxScaleLevel): good1[i1] = map point non-NULL && !isBad();
	// vnMatches21[i2] = i1 or -1.  (:326-473)
	int WindowSearch(const FrameGridView& F1, const std::vector<uint8_t>& good1, FrameGridView& F2, int windowSize, std::vector<int>& vnMatches21,
	                 int minScaleLevel = 0, int maxScaleLevel = INT32_MAX) {
		Probes p;
		for (int i1 = 0; i1 < F1.n; ++i1) {
			const int lvl = F1.mvKeys[i1].octave;
			if (!good1[i1] || (minScaleLevel > 0 && lvl < minScaleLevel) || (maxScaleLevel < INT32_MAX && lvl > maxScaleLevel)) continue;
			p.add(F1.mvKeys[i1].ptx, F1.mvKeys[i1].pty, windowSize, -1, -1, F1.keypoint_to_cam[i1], i1);
		}
		std::vector<uint8_t> none(F2.n > 0 ? F2.n : 1, 0);   // vpMapPointMatches2 starts all-NULL
		std::swap(none, F2.hasMapPoint);
		std::vector<int> m;
		const int nm = window(MCS_WINDOW_RATIO, p, F1, F2, m);
		std::swap(none, F2.hasMapPoint);
		vnMatches21.assign(F2.n, -1);
		for (size_t k = 0; k < m.size(); ++k) if (m[k] >= 0) vnMatches21[m[k]] = p.src[k];
		return nm;
	}
	// SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize): vbPrevMatched = 2 doubles per F1 feature, updated.  (:579-726)
	int SearchForInitialization(const FrameGridView& F1, FrameGridView& F2, std::vector<double>& vbPrevMatched, std::vector<int>& vnMatches12,
	                            int windowSize = 10) {
		Probes p;
		for (int i1 = 0; i1 < F1.n; ++i1)
			p.add(vbPrevMatched[2 * i1], vbPrevMatched[2 * i1 + 1], windowSize, F1.mvKeys[i1].octave, F1.mvKeys[i1].octave, F1.keypoint_to_cam[i1], i1);
		const int nm = window(MCS_WINDOW_INITIALIZE, p, F1, F2, vnMatches12);
		vnMatches12.resize(F1.n);
		for (int i1 = 0; i1 < F1.n; ++i1)
			if (vnMatches12[i1] >= 0) { vbPrevMatched[2 * i1] = F2.mvKeys[vnMatches12[i1]].ptx; vbPrevMatched[2 * i1 + 1] = F2.mvKeys[vnMatches12[i1]].pty; }
		return nm;
	}
	// SearchByProjection(CurrentFrame, LastFrame, th): search1[i] = map point good && !mvbOutlier[i] && projection inside the mirror mask;
	// uv = the projections (mcs_world_to_cam).  matchCur[i2] = index in LastFrame or -1; CurrentFrame.hasMapPoint is updated.  (:1990-2118)
	int SearchByProjection(FrameGridView& CurrentFrame, const FrameGridView& LastFrame, const std::vector<uint8_t>& search1, const double* uv, double th,
	                       std::vector<int>& matchCur) {
		Probes p;
		for (int i = 0; i < LastFrame.n; ++i) {
			if (!search1[i]) continue;
			const int o = LastFrame.mvKeys[i].octave;
			p.add(uv[2 * i], uv[2 * i + 1], th * CurrentFrame.mvScaleFactors[o], o - 1, o + 1, LastFrame.keypoint_to_cam[i], i);
		}
		std::vector<int> m;
		const int nm = window(MCS_WINDOW_BEST, p, LastFrame, CurrentFrame, m);
		matchCur.assign(CurrentFrame.n, -1);
		for (size_t k = 0; k < m.size(); ++k) if (m[k] >= 0) matchCur[m[k]] = p.src[k];
		return nm;
	}

	// SearchByProjection(F, vpMapPoints, th) (:67-166): one projection per (map point, camera) with mbTrackInView, in the reference's order;
	// match[p] = frame feature or -1; F.hasMapPoint is updated.
	struct Projections { std::vector<double> x, y, viewCos; std::vector<int32_t> level, cam; const uint8_t* desc = nullptr; const uint8_t* mask = nullptr; };
	int SearchByProjection(FrameGridView& F, const Projections& mp, double th, std::vector<int>& match) {
		const int n = (int)mp.x.size();
		mcs_projection_set ps{mp.x.data(), mp.y.data(), mp.viewCos.data(), mp.level.data(), mp.cam.data(), mp.desc, havingMasks ? mp.mask : nullptr, n, mbFeatDim};
		mcs_frame_view fv{reinterpret_cast<const mcs_keypoint*>(F.mvKeys), F.descriptors, havingMasks ? F.masks : nullptr, F.keypoint_to_cam.data(),
		                  F.hasMapPoint.data(), F.n, mbFeatDim, (int32_t)F.width.size(), F.width.data(), F.height.data(), F.mvScaleFactors.data(),
		                  (int32_t)F.mvScaleFactors.size()};
		match.assign(n > 0 ? n : 1, -1);
		int32_t nm = 0;
		mcs_throw(mcs_search_by_projection(ctx_.h, &ps, &fv, th, mfNNratio, mbFeatDim, MCS_MEM_HOST, match.data(), &nm));
		match.resize(n);
		return nm;
	}
	// The search loop of Fuse / SearchBySim3 / SearchForTriangulationBetweenCameras (skipTaken = false) and of the relocalisation
	// SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (skipTaken = true): closest feature per window, accepted if <= maxDist.
	struct Windows { std::vector<double> x, y, r; std::vector<int32_t> lo, hi, cam; const uint8_t* desc = nullptr; const uint8_t* mask = nullptr; };
	int BestInWindows(FrameGridView& F, const Windows& w, int maxDist, bool skipTaken, std::vector<int>& match, std::vector<int>& dist) {
		const int n = (int)w.x.size();
		mcs_window_probes pr{w.x.data(), w.y.data(), w.r.data(), w.lo.data(), w.hi.data(), w.cam.data(), w.desc, havingMasks ? w.mask : nullptr, n, mbFeatDim};
		mcs_frame_view fv{reinterpret_cast<const mcs_keypoint*>(F.mvKeys), F.descriptors, havingMasks ? F.masks : nullptr, F.keypoint_to_cam.data(),
		                  skipTaken ? F.hasMapPoint.data() : nullptr, F.n, mbFeatDim, (int32_t)F.width.size(), F.width.data(), F.height.data(),
		                  F.mvScaleFactors.data(), (int32_t)F.mvScaleFactors.size()};
		match.assign(n > 0 ? n : 1, -1); dist.assign(n > 0 ? n : 1, 0);
		int32_t nm = 0;
		mcs_throw(mcs_window_best(ctx_.h, &pr, &fv, maxDist, skipTaken ? 1 : 0, mbFeatDim, MCS_MEM_HOST, match.data(), dist.data(), &nm));
		match.resize(n); dist.resize(n);
		return nm;
	}
	// SearchByBoW(pKF, F, ...) WITH the vocabulary (:179-323): kf rows must be given in FeatureVector order (node ascending, index ascending),
	// `cam` of both views carries the FeatureVector node id (frame features of stopped words: flag 0).  matchF[j] = kf row or -1.
	int SearchByBoWFrameVocabulary(const FeatureSetView& kfInNodeOrder, const FeatureSetView& frame, std::vector<int>& matchF) {
		mcs_desc_set q{kfInNodeOrder.descriptors, havingMasks ? kfInNodeOrder.masks : nullptr, kfInNodeOrder.flag.data(), kfInNodeOrder.cam.data(), kfInNodeOrder.n, mbFeatDim};
		mcs_desc_set t{frame.descriptors, havingMasks ? frame.masks : nullptr, frame.flag.empty() ? nullptr : frame.flag.data(), frame.cam.data(), frame.n, mbFeatDim};
		matchF.assign(frame.n > 0 ? frame.n : 1, -1);
		int32_t nm = 0, fb = 0;
		mcs_throw(mcs_search_kf_f(ctx_.h, 1, &q, 0, &t, 0, mbFeatDim, mfNNratio, K_, MCS_MEM_HOST, matchF.data(), &nm, &fb));
		matchF.resize(frame.n);
		return nm;
	}

private:
	struct Probes {
		std::vector<double> x, y, r; std::vector<int32_t> lo, hi, cam, src;
		void add(double x_, double y_, double r_, int lo_, int hi_, int cam_, int src_) {
			x.push_back(x_); y.push_back(y_); r.push_back(r_); lo.push_back(lo_); hi.push_back(hi_); cam.push_back(cam_); src.push_back(src_);
		}
	};
	int window(mcs_window_rule rule, const Probes& p, const FrameGridView& from, FrameGridView& in, std::vector<int>& match) {
		const int n = (int)p.x.size();
		std::vector<uint8_t> d((size_t)(n > 0 ? n : 1) * mbFeatDim), m(havingMasks ? d.size() : 0);
		for (int k = 0; k < n; ++k) {
			std::memcpy(&d[(size_t)k * mbFeatDim], from.descriptors + (size_t)p.src[k] * mbFeatDim, mbFeatDim);
			if (havingMasks) std::memcpy(&m[(size_t)k * mbFeatDim], from.masks + (size_t)p.src[k] * mbFeatDim, mbFeatDim);
		}
		mcs_window_probes pr{p.x.data(), p.y.data(), p.r.data(), p.lo.data(), p.hi.data(), p.cam.data(), d.data(), havingMasks ? m.data() : nullptr, n, mbFeatDim};
		mcs_frame_view fv{reinterpret_cast<const mcs_keypoint*>(in.mvKeys), in.descriptors, havingMasks ? in.masks : nullptr, in.keypoint_to_cam.data(),
		                  rule == MCS_WINDOW_INITIALIZE ? nullptr : in.hasMapPoint.data(), in.n, mbFeatDim, (int32_t)in.width.size(), in.width.data(),
		                  in.height.data(), in.mvScaleFactors.data(), (int32_t)in.mvScaleFactors.size()};
		match.assign(n > 0 ? n : 1, -1);
		int32_t nm = 0;
		mcs_throw(mcs_window_match(ctx_.h, &pr, &fv, rule, mfNNratio, mbFeatDim, MCS_MEM_HOST, match.data(), &nm));
		match.resize(n);
		return nm;
	}
	int run(int mode, const FeatureSetView& a, const FeatureSetView& b, const double* E, int nrCams, std::vector<int>& out, int outN) {
		mcs_desc_set q{a.descriptors, havingMasks ? a.masks : nullptr, a.flag.empty() ? nullptr : a.flag.data(), mode == 2 ? a.cam.data() : nullptr, a.n, mbFeatDim};
		mcs_desc_set t{b.descriptors, havingMasks ? b.masks : nullptr, (mode == 1 || b.flag.empty()) ? nullptr : b.flag.data(), mode == 2 ? b.cam.data() : nullptr, b.n, mbFeatDim};
		out.assign(outN > 0 ? outN : 1, -1);
		int32_t nm = 0, fb = 0;
		int rc;
		if (mode == 0) rc = mcs_search_kf_kf(ctx_.h, 1, &q, 0, &t, 0, mbFeatDim, mfNNratio, K_, MCS_MEM_HOST, out.data(), &nm, &fb);
		else if (mode == 1) rc = mcs_search_kf_f(ctx_.h, 1, &q, 0, &t, 0, mbFeatDim, mfNNratio, K_, MCS_MEM_HOST, out.data(), &nm, &fb);
		else rc = mcs_search_triangulation(ctx_.h, 1, &q, 0, &t, 0, a.rays, b.rays, E, nrCams, mbFeatDim, K_, MCS_MEM_HOST, out.data(), &nm, &fb);
		mcs_throw(rc);
		out.resize(outN);
		return nm;
	}
	Context& ctx_;
	double mfNNratio;
	int mbFeatDim;
	bool havingMasks;
	int K_;
	bool mbCheckOrientation = false;
};

inline int DescriptorDistance64(Context& c, const uint64_t* descr_i, const uint64_t* descr_j, const int& dim) {
	int out = 0;
	mcs_throw(mcs_descriptor_distance(c.h, (const uint8_t*)descr_i, (const uint8_t*)descr_j, dim, &out));
	return out;
}
inline int DescriptorDistance64Masked(Context& c, const uint64_t* descr_i, const uint64_t* descr_j, const uint64_t* mask_i, const uint64_t* mask_j,
                                      const int& dim) {
	int out = 0;
	mcs_throw(mcs_descriptor_distance_masked(c.h, (const uint8_t*)descr_i, (const uint8_t*)descr_j, (const uint8_t*)mask_i, (const uint8_t*)mask_j, dim, &out));
	return out;
}

}  // namespace MultiColSLAM

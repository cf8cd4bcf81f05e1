// mdBRIEFextractorOct_mcs.cpp — DROP-IN replacement for the reference's src/mdBRIEFextractorOct.cpp.
//
// Same class, same header (the reference's own include/mdBRIEFextractorOct.h, unmodified): the constructor keeps the public state the rest of the
// system reads (GetLevels, GetScaleFactor, GetMasksLearned, GetDescriptorSize), operator() hands the image to libmcs_hip.so through its C ABI
// (include/mcs_c.h) and returns the same keypoints / descriptors / masks.  cMultiFrame, cTracking ... compile and link against it unchanged:
// replace the one source file in the reference's CMake target and add `-lmcs_hip`.  The header cannot grow members (and its inline destructor
// gives no hook), so the device extractors live in a process-wide pool keyed by EVERY constructor parameter plus the image size: objects with equal
// parameters share one device extractor (calls are serialised on the pool's stream anyway), an object built at a recycled address with other
// parameters can never pick up a stale one, and nothing is leaked per destroyed object.  tests/test_gpu_dropin.py builds the reference's cMultiFrame.cpp around this file and compares the
// resulting cMultiFrame with the one the reference's own extractor produces.
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

#include "mdBRIEFextractorOct.h"
#include "mcs_c.h"
#include "mcs_dropin.h"

namespace MultiColSLAM
{
namespace
{
	struct Device { mcs_extractor* ex = nullptr; int cap = 0; };
	struct Key   // all 13 constructor arguments + the image size (plain ints / one float, compared bytewise)
	{
		mcs_extractor_params p; int32_t w, h;
		bool operator<(const Key& o) const { return std::memcmp(this, &o, sizeof(Key)) < 0; }
	};
	std::mutex g_mutex;
	mcs_ctx* g_ctx = nullptr;                                   // one context (device 0, its own stream) for all extractors of the process
	std::map<Key, Device> g_devices;

	void check(int rc, const char* what) { mcs_dropin::check(rc, what); }
}
namespace mcs_dropin
{
	void check(int rc, const char* what)
	{
		if (rc != MCS_OK) throw std::runtime_error(std::string(what) + ": " + mcs_last_error());
	}
	std::mutex& mutex() { return g_mutex; }
	mcs_ctx* context()   // call with mutex() held
	{
		if (!g_ctx) check(mcs_ctx_create(0, nullptr, &g_ctx), "mcs_ctx_create");
		return g_ctx;
	}
}

mdBRIEFextractorOct::mdBRIEFextractorOct(int _nfeatures, float _scaleFactor, int _nlevels, int _edgeThreshold, int _firstLevel, int _scoreType,
	int _patchSize, int _fastThreshold, bool _useAgast, int _fastAgastType, bool _do_dBrief, bool _learnMasks, int _descSize) :
	nfeatures(_nfeatures), scaleFactor(_scaleFactor), numlevels(_nlevels), edgeThreshold(_edgeThreshold), firstLevel(_firstLevel),
	scoreType(_scoreType), patchSize(_patchSize), fastThreshold(_fastThreshold), useAgast(_useAgast), fastAgastType(_fastAgastType),
	learnMasks(_learnMasks), descSize(_descSize), do_dBrief(_do_dBrief)
{
	// the scale tables other classes copy from the extractor (cMultiFrame reads GetScaleFactor() / GetLevels() only, but keep the vectors valid)
	mvScaleFactor.resize(numlevels);
	mvInvScaleFactor.resize(numlevels);
	mvScaleFactor[0] = 1; mvInvScaleFactor[0] = 1;
	for (int i = 1; i < numlevels; i++) { mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor; mvInvScaleFactor[i] = mvInvScaleFactor[i - 1] * (1.0 / scaleFactor); }
}

void mdBRIEFextractorOct::operator()(cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint>& _keypoints, cCamModelGeneral_& camModel,
	cv::OutputArray _descriptors, cv::OutputArray _descriptorMasks)
{
	if (_image.empty())
		return;
	cv::Mat image = _image.getMat(), mask = _mask.getMat();
	Device dev;
	{
		std::lock_guard<std::mutex> lock(g_mutex);
		mcs_dropin::context();
		Key key;
		std::memset(&key, 0, sizeof(key));   // padding bytes take part in the comparison
		const mcs_extractor_params p = { nfeatures, (float)scaleFactor, numlevels, edgeThreshold, firstLevel, scoreType, patchSize, fastThreshold,
			useAgast ? 1 : 0, fastAgastType, do_dBrief ? 1 : 0, learnMasks ? 1 : 0, descSize };
		key.p = p; key.w = image.cols; key.h = image.rows;
		Device& d = g_devices[key];
		if (!d.ex)
		{
			check(mcs_extractor_create(g_ctx, &p, image.cols, image.rows, 1, &d.ex), "mcs_extractor_create");
			check(mcs_extractor_kp_capacity(d.ex, &d.cap), "mcs_extractor_kp_capacity");
		}
		dev = d;
	}
	mcs_ocam cam = {};
	cam.c = camModel.Get_c(); cam.d = camModel.Get_d(); cam.e = camModel.Get_e(); cam.u0 = camModel.Get_u0(); cam.v0 = camModel.Get_v0();
	cv::Mat_<double> P = camModel.Get_P(), invP = camModel.Get_invP();
	cam.p_deg = camModel.GetPolDeg(); cam.invP_deg = camModel.GetInvDeg();
	if (cam.p_deg > MCS_MAX_POLY || cam.invP_deg > MCS_MAX_POLY) throw std::runtime_error("camera polynomial degree above MCS_MAX_POLY");
	for (int i = 0; i < cam.p_deg; ++i) cam.p[i] = P.at<double>(i);
	for (int i = 0; i < cam.invP_deg; ++i) cam.invP[i] = invP.at<double>(i);
	cam.width = image.cols; cam.height = image.rows;

	std::vector<mcs_keypoint> kps(dev.cap);
	std::vector<uint8_t> desc((size_t)dev.cap * descSize), dmask((size_t)dev.cap * descSize);
	int32_t n = 0;
	{
		std::lock_guard<std::mutex> lock(g_mutex);   // cMultiFrame calls the per-camera extractors from an OpenMP loop; one stream serves them in turn
		check(mcs_extract_batch(dev.ex, 1, image.data, 0, (int)image.step, mask.empty() ? nullptr : mask.data, 0, mask.empty() ? 0 : (int)mask.step, &cam,
			MCS_MEM_HOST, &n, kps.data(), desc.data(), dmask.data(), nullptr), "mcs_extract_batch");
	}
	_keypoints.clear();
	_keypoints.reserve(n);
	for (int i = 0; i < n; ++i)
		_keypoints.push_back(cv::KeyPoint(kps[i].x, kps[i].y, kps[i].size, kps[i].angle, kps[i].response, kps[i].octave, kps[i].class_id));
	if (n == 0)
	{
		_descriptors.release();
		_descriptorMasks.release();
		return;
	}
	_descriptors.create(n, descSize, CV_8U);
	_descriptorMasks.create(n, descSize, CV_8U);
	cv::Mat d = _descriptors.getMat(), m = _descriptorMasks.getMat();
	for (int i = 0; i < n; ++i)
	{
		std::memcpy(d.ptr<uchar>(i), &desc[(size_t)i * descSize], descSize);
		std::memcpy(m.ptr<uchar>(i), &dmask[(size_t)i * descSize], descSize);
	}
}
}

// mcs_dropin.h — shared by the drop-in translation units (integration/*_mcs.cpp): ONE libmcs_hip context per process and the lock that serialises its use.
// Not part of the reference's headers and not installed; the reference's own headers stay unmodified.
//
// THE OPENCV SURFACE OF THE DROP-IN FILES.  integration/*_mcs.cpp have so far only been compiled against oracle/cvshim (a stand-in for OpenCV's types written
// for this repository: no OpenCV in the build image).  To keep "exchange two source files"  true against a genuine OpenCV 3.x, the files use
// ONLY the following cv:: symbols and members, each in its documented 3.x form (opencv2/core/mat.hpp, core/types.hpp, core/matx.hpp, core/base.hpp);
// tests/test_integration_cv_surface.py fails when a file uses anything outside this list, and checks the shim declares every entry the way the list states it:
//   cv::Mat            int rows, cols; uchar* data; step ONLY as (int)m.step / (size_t)m.step (a MatStep in 3.x); bool empty() const; bool isContinuous() const; int type() const;
//                      template<typename _Tp> _Tp* ptr(int i0 = 0); template<typename _Tp> _Tp& at(int i0, int i1) and at(int i0) (single row / column); void create(int rows, int cols, int type);
//   cv::Mat_<double>   (as returned by the reference's cCamModelGeneral_::Get_P / Get_invP) rows, at<double>(i, j) / operator()(i, j)
//   cv::InputArray     (const _InputArray&)   Mat getMat(int idx = -1) const; bool empty() const;
//   cv::OutputArray    (const _OutputArray&)  void create(int rows, int cols, int type, ...) const; void release() const; Mat getMat(int idx = -1) const;
//   cv::KeyPoint       KeyPoint(float x, float y, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1);
//                      Point2f pt; float size, angle, response; int octave, class_id   (28 bytes, asserted below: mcs_keypoint mirrors it field for field)
//   cv::Vec2d / Vec3d / Vec4d, cv::Matx33d / Matx44d   operator()(i[, j]), operator*, operator-, .t(), .dot(); cv::norm(Vec3d); (cv::sqrt is not used: std::sqrt)
// Anything else — Mat::step[i], Mat::clone, Mat::copyTo, Mat::row, MatExpr, cv::Range, cv::Rect ROIs ... — is deliberately absent from the drop-in files.
#pragma once
#include <cstddef>
#include <mutex>
#include <opencv2/core/core.hpp>
#include "mcs_c.h"

// the keypoint records cross the C ABI as mcs_keypoint and are handed to the reference as cv::KeyPoint: same seven fields, same offsets
static_assert(sizeof(cv::KeyPoint) == 28 && sizeof(mcs_keypoint) == 28, "cv::KeyPoint / mcs_keypoint must be the 28-byte record of OpenCV 3.x");
static_assert(offsetof(cv::KeyPoint, pt) == 0 && offsetof(cv::KeyPoint, size) == 8 && offsetof(cv::KeyPoint, angle) == 12 && offsetof(cv::KeyPoint, response) == 16 &&
              offsetof(cv::KeyPoint, octave) == 20 && offsetof(cv::KeyPoint, class_id) == 24, "cv::KeyPoint field offsets differ from OpenCV 3.x");
static_assert(offsetof(mcs_keypoint, x) == 0 && offsetof(mcs_keypoint, y) == 4 && offsetof(mcs_keypoint, size) == 8 && offsetof(mcs_keypoint, angle) == 12 &&
              offsetof(mcs_keypoint, response) == 16 && offsetof(mcs_keypoint, octave) == 20 && offsetof(mcs_keypoint, class_id) == 24, "mcs_keypoint layout");
static_assert(sizeof(cv::Vec3d) == 24 && sizeof(cv::Matx33d) == 72 && sizeof(cv::Matx44d) == 128, "cv::Vec / cv::Matx must be plain arrays of doubles");

namespace MultiColSLAM
{
namespace mcs_dropin
{
	mcs_ctx* context();        // created on first use (device 0, its own stream); throws std::runtime_error when there is no HIP device
	std::mutex& mutex();       // every call into the context's extractors happens under this lock (one stream serves them in turn)
	void check(int rc, const char* what);
}
}

// cvshim.hpp — TEST INFRASTRUCTURE.  A minimal stand-in for the OpenCV 3 types the reference's extractor and camera-model sources use, so that
// /root/reference/src/mdBRIEFextractorOct.cpp and cam_model_omni.cpp compile UNMODIFIED into oracle/_ref (see oracle/Makefile, target ref).
// OpenCV itself is not installed in this environment.  The types (Mat with ROIs, Mat_, Matx, Vec, Point, KeyPoint ...) are re-implemented
// here only as far as those two files need them; the five IMAGE PRIMITIVES they call (resize, copyMakeBorder, FAST detect, boxFilter,
// fastAtan2) forward to the oracle's own restatements of OpenCV (oracle/mcs_oracle.cpp).  So oracle/_ref pins everything
// that is the REFERENCE's code (pyramid loop, cell grid, DistributeOctTree, IC_Angle, pattern rotation / distortion, the three descriptor
// variants, masks, operator() glue, the omni camera model) and leaves exactly the OpenCV primitives unpinned.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <Eigen/Dense>

#include "../mcs_oracle.h"

typedef unsigned char uchar;
typedef unsigned short ushort;

#define CV_8U 0
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 0
#define CV_64FC1 6
#define CV_PI 3.1415926535897932384626433832795
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert failed: " #expr); } while (0)

inline int cvRound(double v) { return orc_cvRound(v); }
inline int cvRound(float v) { return orc_cvRound((double)v); }
inline int cvRound(int v) { return v; }
inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }

namespace cv {

template <class T> inline T sqrt(T v) { return std::sqrt(v); }   // a template, so that unqualified sqrt(double) under `using namespace cv, std` stays unambiguous
enum { NORM_HAMMING = 6 };
template <class T> struct DataType { enum { type = CV_8U }; };
template <> struct DataType<int> { enum { type = CV_32S }; };
template <> struct DataType<float> { enum { type = CV_32F }; };
template <> struct DataType<double> { enum { type = CV_64F }; };
template <class T> struct AutoBuffer { std::vector<T> v; explicit AutoBuffer(size_t n) : v(n) {} operator T*() { return v.data(); } };
template <class T, int cn> struct Vec;
inline float fastAtan2(float y, float x) { return orc_fastAtan2(y, x); }

template <class T> struct Point_ {
	T x, y;
	Point_() : x(0), y(0) {}
	Point_(T x_, T y_) : x(x_), y(y_) {}
	template <class U> Point_(const Point_<U>& o) : x((T)o.x), y((T)o.y) {}
	Point_(const Vec<T, 2>& v);
	Point_& operator*=(T s) { x *= s; y *= s; return *this; }
	Point_& operator+=(const Point_& o) { x += o.x; y += o.y; return *this; }
};
template <class T> Point_<T> operator*(const Point_<T>& p, T s) { return Point_<T>(p.x * s, p.y * s); }
template <class T> Point_<T> operator+(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(a.x + b.x, a.y + b.y); }
template <class T> Point_<T> operator-(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(a.x - b.x, a.y - b.y); }
typedef Point_<int> Point2i; typedef Point2i Point; typedef Point_<float> Point2f; typedef Point_<double> Point2d;
template <class T> struct Point3_ { T x, y, z; Point3_() : x(0), y(0), z(0) {} Point3_(T a, T b, T c) : x(a), y(b), z(c) {} };
template <class T> struct Size_ { T width, height; Size_() : width(0), height(0) {} Size_(T w, T h) : width(w), height(h) {} };
typedef Size_<int> Size;
template <class T> struct Rect_ { T x, y, width, height; Rect_() : x(0), y(0), width(0), height(0) {} Rect_(T a, T b, T c, T d) : x(a), y(b), width(c), height(d) {} };
typedef Rect_<int> Rect;
struct Range { int start, end; Range(int s, int e) : start(s), end(e) {} };

template <class T, int m, int n> struct Matx {
	T val[m * n];
	Matx() { for (int i = 0; i < m * n; ++i) val[i] = T(0); }
	Matx(T a) : Matx() { val[0] = a; }
	Matx(T a, T b) : Matx() { val[0] = a; val[1] = b; }
	Matx(T a, T b, T c) : Matx() { val[0] = a; val[1] = b; val[2] = c; }
	Matx(T a, T b, T c, T d) : Matx() { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
	Matx(T a, T b, T c, T d, T e, T f) : Matx() { T v[6] = {a, b, c, d, e, f}; for (int i = 0; i < 6; ++i) val[i] = v[i]; }
	Matx(T a, T b, T c, T d, T e, T f, T g, T h, T i_) : Matx() { T v[9] = {a, b, c, d, e, f, g, h, i_}; for (int i = 0; i < 9; ++i) val[i] = v[i]; }
	Matx(T a, T b, T c, T d, T e, T f, T g, T h, T i_, T j, T k, T l, T m_, T n_, T o, T p) : Matx() {
		T v[16] = {a, b, c, d, e, f, g, h, i_, j, k, l, m_, n_, o, p}; for (int i = 0; i < 16; ++i) val[i] = v[i]; }
	static Matx zeros() { return Matx(); }
	static Matx all(T v) { Matx r; for (int i = 0; i < m * n; ++i) r.val[i] = v; return r; }
	static Matx ones() { return all(T(1)); }
	T dot(const Matx& o) const { T s = 0; for (int i = 0; i < m * n; ++i) s += val[i] * o.val[i]; return s; }
	Matx inv() const {   // Gauss-Jordan with partial pivoting (square matrices; only small calibration algebra reaches this)
		static_assert(m == n, "inv of a square matrix");
		T a[m][2 * m];
		for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { a[i][j] = val[i * n + j]; a[i][m + j] = i == j ? T(1) : T(0); }
		for (int c = 0; c < m; ++c) {
			int p = c; for (int r = c + 1; r < m; ++r) if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
			for (int j = 0; j < 2 * m; ++j) std::swap(a[c][j], a[p][j]);
			const T d = a[c][c];
			for (int j = 0; j < 2 * m; ++j) a[c][j] /= d;
			for (int r = 0; r < m; ++r) if (r != c) { const T f = a[r][c]; for (int j = 0; j < 2 * m; ++j) a[r][j] -= f * a[c][j]; }
		}
		Matx r; for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) r.val[i * n + j] = a[i][m + j];
		return r;
	}
	static Matx eye() { Matx r; for (int i = 0; i < (m < n ? m : n); ++i) r.val[i * n + i] = T(1); return r; }
	T& operator()(int i, int j) { return val[i * n + j]; }
	const T& operator()(int i, int j) const { return val[i * n + j]; }
	T& operator()(int i) { return val[i]; }
	const T& operator()(int i) const { return val[i]; }
	Matx<T, n, m> t() const { Matx<T, n, m> r; for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) r.val[j * m + i] = val[i * n + j]; return r; }
	template <int m1, int n1> Matx<T, m1, n1> get_minor(int i0, int j0) const { Matx<T, m1, n1> r; for (int i = 0; i < m1; ++i) for (int j = 0; j < n1; ++j) r.val[i * n1 + j] = val[(i0 + i) * n + j0 + j]; return r; }
};
template <class T, int m, int k, int n> Matx<T, m, n> operator*(const Matx<T, m, k>& a, const Matx<T, k, n>& b) {
	Matx<T, m, n> r; for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) { T s = 0; for (int q = 0; q < k; ++q) s += a(i, q) * b(q, j); r(i, j) = s; } return r; }
template <class T, int m, int n> Matx<T, m, n> operator*(T s, const Matx<T, m, n>& a) { Matx<T, m, n> r; for (int i = 0; i < m * n; ++i) r.val[i] = s * a.val[i]; return r; }
template <class T, int m, int n> Matx<T, m, n> operator+(const Matx<T, m, n>& a, const Matx<T, m, n>& b) { Matx<T, m, n> r; for (int i = 0; i < m * n; ++i) r.val[i] = a.val[i] + b.val[i]; return r; }
template <class T, int m, int n> Matx<T, m, n> operator-(const Matx<T, m, n>& a, const Matx<T, m, n>& b) { Matx<T, m, n> r; for (int i = 0; i < m * n; ++i) r.val[i] = a.val[i] - b.val[i]; return r; }
template <class T, int m, int n> Matx<T, m, n> operator-(const Matx<T, m, n>& a) { Matx<T, m, n> r; for (int i = 0; i < m * n; ++i) r.val[i] = -a.val[i]; return r; }
typedef Matx<double, 2, 2> Matx22d; typedef Matx<double, 3, 4> Matx34d; typedef Matx<double, 2, 1> Matx21d; typedef Matx<double, 4, 1> Matx41d;
typedef Matx<double, 3, 3> Matx33d; typedef Matx<double, 4, 4> Matx44d; typedef Matx<double, 3, 1> Matx31d; typedef Matx<double, 6, 1> Matx61d;

template <class T, int cn> struct Vec : Matx<T, cn, 1> {
	Vec() {}
	Vec(T a, T b) { this->val[0] = a; this->val[1] = b; }
	Vec(T a, T b, T c) { this->val[0] = a; this->val[1] = b; this->val[2] = c; }
	Vec(T a, T b, T c, T d) { this->val[0] = a; this->val[1] = b; this->val[2] = c; this->val[3] = d; }
	Vec(const Matx<T, cn, 1>& o) { for (int i = 0; i < cn; ++i) this->val[i] = o.val[i]; }
	T& operator[](int i) { return this->val[i]; }
	const T& operator[](int i) const { return this->val[i]; }
	Vec& operator/=(T s) { for (int i = 0; i < cn; ++i) this->val[i] /= s; return *this; }
	Vec cross(const Vec& o) const { static_assert(cn == 3, "cross"); return Vec(this->val[1] * o.val[2] - this->val[2] * o.val[1], this->val[2] * o.val[0] - this->val[0] * o.val[2], this->val[0] * o.val[1] - this->val[1] * o.val[0]); }
	Vec& operator=(T s) { for (int i = 0; i < cn; ++i) this->val[i] = s; return *this; }
};
template <class T> Point_<T>::Point_(const Vec<T, 2>& v) : x(v.val[0]), y(v.val[1]) {}
typedef Vec<double, 2> Vec2d; typedef Vec<double, 3> Vec3d; typedef Vec<double, 4> Vec4d; typedef Vec<float, 2> Vec2f;
template <class T, int cn> double norm(const Vec<T, cn>& v) { double s = 0; for (int i = 0; i < cn; ++i) s += (double)v.val[i] * v.val[i]; return std::sqrt(s); }
template <class T, int cn> Vec<T, cn> operator*(const Vec<T, cn>& v, T s) { Vec<T, cn> r; for (int i = 0; i < cn; ++i) r.val[i] = v.val[i] * s; return r; }
template <class T, int cn> Vec<T, cn> operator*(T s, const Vec<T, cn>& v) { return v * s; }
template <class T, int cn> Vec<T, cn> operator+(const Vec<T, cn>& a, const Vec<T, cn>& b) { Vec<T, cn> r; for (int i = 0; i < cn; ++i) r.val[i] = a.val[i] + b.val[i]; return r; }
template <class T, int cn> Vec<T, cn> operator-(const Vec<T, cn>& a, const Vec<T, cn>& b) { Vec<T, cn> r; for (int i = 0; i < cn; ++i) r.val[i] = a.val[i] - b.val[i]; return r; }
template <class T, int cn> Vec<T, cn> operator-(const Vec<T, cn>& a) { Vec<T, cn> r; for (int i = 0; i < cn; ++i) r.val[i] = -a.val[i]; return r; }
template <class T, int m, int n> Vec<T, m> operator*(const Matx<T, m, n>& a, const Vec<T, n>& b) { Vec<T, m> r; for (int i = 0; i < m; ++i) { T s = 0; for (int q = 0; q < n; ++q) s += a(i, q) * b.val[q]; r.val[i] = s; } return r; }
template <class T, int m, int n> double norm(const Matx<T, m, n>& v) { double s = 0; for (int i = 0; i < m * n; ++i) s += (double)v.val[i] * v.val[i]; return std::sqrt(s); }
template <class T, int cn> Vec<T, cn> operator/(const Vec<T, cn>& v, int s) { Vec<T, cn> r; for (int i = 0; i < cn; ++i) r.val[i] = v.val[i] / s; return r; }
template <class T, int cn> Vec<T, cn> operator/(const Vec<T, cn>& v, T s) { Vec<T, cn> r; for (int i = 0; i < cn; ++i) r.val[i] = v.val[i] / s; return r; }

struct KeyPoint {
	Point2f pt; float size, angle, response; int octave, class_id;
	KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
	KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
	KeyPoint(Point2f p, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(p), size(s), angle(a), response(r), octave(o), class_id(c) {}
};

// ---------------------------------------------------------------------------------------------- Mat (8U and 64F, single channel, 2-D, ROIs)
struct MatExprZeros { int r, c, type; };   // Mat::zeros(): assigning it to a Mat of the same shape fills THAT buffer, like cv::MatExpr
struct MatStep { size_t v; MatStep(size_t s = 0) : v(s) {} operator size_t() const { return v; } };
class Mat {
public:
	int flags_type = CV_8UC1, rows = 0, cols = 0;
	MatStep step;
	uchar* data = nullptr;
	std::shared_ptr<std::vector<uchar>> store;
	int wholeRows = 0, wholeCols = 0, ofsY = 0, ofsX = 0;   // position inside the parent allocation (locateROI)

	Mat() {}
	Mat(int r, int c, int type) { create(r, c, type); }
	Mat(Size sz, int type) { create(sz.height, sz.width, type); }
	template <class T, int m, int n> Mat(const Matx<T, m, n>& M) { create(m, n, CV_64F); for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) at<double>(i, j) = (double)M(i, j); }
	Mat(int r, int c, int type, void* ext, size_t st = 0) : flags_type(type), rows(r), cols(c), step(st ? st : (size_t)c * esz(type)), data((uchar*)ext), wholeRows(r), wholeCols(c) {}
	static size_t esz(int type) { return type == CV_64F ? 8 : ((type == CV_32F || type == CV_32S) ? 4 : 1); }
	void create(int r, int c, int type) {
		if (data && rows == r && cols == c && flags_type == type) return;   // cv::Mat::create keeps a matching allocation (copyMakeBorder relies on it)
		flags_type = type; rows = r; cols = c; step = (size_t)c * esz(type);
		store = std::make_shared<std::vector<uchar>>((size_t)r * (size_t)step + 64, 0);
		data = store->data(); wholeRows = r; wholeCols = c; ofsY = ofsX = 0;
	}
	void create(Size sz, int type) { create(sz.height, sz.width, type); }
	static MatExprZeros zeros(int r, int c, int type) { return MatExprZeros{r, c, type}; }
	static MatExprZeros zeros(Size sz, int type) { return MatExprZeros{sz.height, sz.width, type}; }
	Mat(const MatExprZeros& e) { *this = e; }
	Mat& operator=(const MatExprZeros& e) {
		create(e.r, e.c, e.type);   // keeps a matching allocation (also a ROI of one): cv::MatExpr::assign writes in place
		for (int i = 0; i < rows; ++i) std::memset(data + (size_t)i * step, 0, (size_t)cols * elemSize());
		return *this;
	}
	static Mat ones(Size sz, int type) { Mat m = zeros(sz, type); for (int i = 0; i < m.rows; ++i) for (int j = 0; j < m.cols; ++j) { if (type == CV_64F) m.at<double>(i, j) = 1.0; else m.at<uchar>(i, j) = 1; } return m; }
	int type() const { return flags_type; }
	int depth() const { return flags_type; }
	int channels() const { return 1; }
	size_t elemSize() const { return esz(flags_type); }
	size_t elemSize1() const { return esz(flags_type); }
	size_t step1() const { return (size_t)step / elemSize1(); }
	bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
	bool isContinuous() const { return (size_t)step == (size_t)cols * elemSize(); }
	Size size() const { return Size(cols, rows); }
	void release() { *this = Mat(); }
	Mat clone() const { Mat m; if (empty()) return m; m.create(rows, cols, flags_type); for (int i = 0; i < rows; ++i) std::memcpy(m.data + (size_t)i * m.step, data + (size_t)i * step, (size_t)cols * elemSize()); return m; }
	void copyTo(Mat& o) const { o = clone(); }
	Mat roi(int y, int x, int h, int w) const {
		if (y < 0 || x < 0 || h < 0 || w < 0 || y + h > rows || x + w > cols) throw std::runtime_error("cvshim: ROI outside the matrix (cv::Mat would CV_Assert)");
		Mat m = *this; m.rows = h; m.cols = w; m.data = data + (size_t)y * step + (size_t)x * elemSize(); m.ofsY = ofsY + y; m.ofsX = ofsX + x; return m; }
	Mat operator()(const Rect& r) const { return roi(r.y, r.x, r.height, r.width); }
	Mat rowRange(int a, int b) const { return roi(a, 0, b - a, cols); }
	Mat colRange(int a, int b) const { return roi(0, a, rows, b - a); }
	Mat rowRange(double a, double b) const { return rowRange((int)a, (int)b); }
	Mat colRange(double a, double b) const { return colRange((int)a, (int)b); }
	Mat row(int i) const { return roi(i, 0, 1, cols); }
	template <class T> T* ptr(int i = 0) { return reinterpret_cast<T*>(data + (size_t)i * step); }
	template <class T> const T* ptr(int i = 0) const { return reinterpret_cast<const T*>(data + (size_t)i * step); }
	uchar* ptr(int i = 0) { return data + (size_t)i * step; }
	const uchar* ptr(int i = 0) const { return data + (size_t)i * step; }
	template <class T> T& at(int i, int j) { return reinterpret_cast<T*>(data + (size_t)i * step)[j]; }
	template <class T> const T& at(int i, int j) const { return reinterpret_cast<const T*>(data + (size_t)i * step)[j]; }
	template <class T> T& at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
	template <class T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
	template <class T> T& at(Point p) { return at<T>(p.y, p.x); }
};

template <class T> class Mat_;
template <class T> struct MatCommaInitializer_ {
	Mat_<T>* m; int idx;
	MatCommaInitializer_(Mat_<T>* m_) : m(m_), idx(0) {}
	template <class U> MatCommaInitializer_& operator,(U v);
	operator Mat_<T>() const;
};
template <class T> class Mat_ : public Mat {
public:
	Mat_() { flags_type = CV_64F; }
	Mat_(int r, int c) { create(r, c, CV_64F); std::memset(data, 0, (size_t)r * step); }
	Mat_(const Mat& o) : Mat(o) {}
	T& operator()(int i, int j) { return this->template at<T>(i, j); }
	const T& operator()(int i, int j) const { return this->template at<T>(i, j); }
};
template <class T> template <class U> MatCommaInitializer_<T>& MatCommaInitializer_<T>::operator,(U v) { m->template at<T>(idx / m->cols, idx % m->cols) = (T)v; ++idx; return *this; }
template <class T> MatCommaInitializer_<T>::operator Mat_<T>() const { return *m; }
template <class T, class U> MatCommaInitializer_<T> operator<<(const Mat_<T>& m, U v) {
	Mat_<T>* keep = new Mat_<T>(m);   // shares the buffer with m (tiny calibration matrices; leaked on purpose: test infrastructure)
	MatCommaInitializer_<T> ci(keep);
	ci.operator,(v);
	return ci;
}

struct _InputArray {
	const Mat* m; Mat dummy;
	_InputArray() : m(&dummy) {}
	_InputArray(const Mat& mm) : m(&mm) {}
	Mat getMat() const { return *m; }
	bool empty() const { return m->empty(); }
};
struct _OutputArray {
	Mat* m;
	_OutputArray(Mat& mm) : m(&mm) {}
	void create(int r, int c, int type) const { m->create(r, c, type); std::memset(m->data, 0, (size_t)r * m->step); }
	Mat getMat() const { return *m; }
	void release() const { m->release(); }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
inline _InputArray noArray() { return _InputArray(); }

template <class T> using Ptr = std::shared_ptr<T>;

enum { INTER_NEAREST = 0, INTER_LINEAR = 1 };
enum { BORDER_CONSTANT = 0, BORDER_REFLECT_101 = 4, BORDER_DEFAULT = 4, BORDER_ISOLATED = 16 };

// ---------------------------------------------------------------------------------------------- image primitives -> the oracle's restatements
inline void resize(const Mat& src, Mat& dst, Size dsize, double, double, int interpolation) {
	if (dst.empty() || dst.rows != dsize.height || dst.cols != dsize.width) dst.create(dsize.height, dsize.width, src.type());
	if (interpolation == INTER_LINEAR) orc_resize_linear(src.data, src.cols, src.rows, (int)src.step, dst.data, dst.cols, dst.rows, (int)dst.step);
	else orc_resize_nearest(src.data, src.cols, src.rows, (int)src.step, dst.data, dst.cols, dst.rows, (int)dst.step);
}
inline void copyMakeBorder(const Mat& src, Mat& dst, int top, int bottom, int left, int right, int borderType) {
	// used by the reference only with equal borders and either src = the interior ROI of dst itself or a free-standing image
	assert(top == bottom && top == left && top == right);
	dst.create(src.rows + 2 * top, src.cols + 2 * top, src.type());
	uchar* interior = dst.data + (size_t)top * dst.step + top;
	if (src.data != interior) for (int i = 0; i < src.rows; ++i) std::memmove(interior + (size_t)i * dst.step, src.data + (size_t)i * src.step, src.cols);
	if ((borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101) orc_border_reflect101(dst.data, src.cols, src.rows, (int)dst.step, top);
	else {
		for (int i = 0; i < dst.rows; ++i) {
			uchar* r = dst.data + (size_t)i * dst.step;
			if (i < top || i >= top + src.rows) std::memset(r, 0, dst.cols);
			else { std::memset(r, 0, top); std::memset(r + top + src.cols, 0, top); }
		}
	}
}
inline void boxFilter(const Mat& src, Mat& dst, int, Size ksize, Point, bool normalize, int borderType) {
	// the reference: 5x5, normalised, in place on a pyramid ROI, BORDER_REFLECT_101 WITHOUT BORDER_ISOLATED (reads the 25-px frame around the ROI)
	if (!(src.data == dst.data && ksize.width == 5 && ksize.height == 5 && normalize && borderType == BORDER_REFLECT_101 && src.ofsX >= 2 && src.ofsY >= 2))
		throw std::runtime_error("cvshim::boxFilter: only the reference's call shape is provided");
	orc_box5_inplace(dst.data, dst.cols, dst.rows, (int)dst.step);
}
inline void GaussianBlur(const Mat&, Mat&, Size, double, double, int) { throw std::runtime_error("cvshim: GaussianBlur is not on the reference's active path"); }
inline void buildPyramid(const Mat& src, std::vector<Mat>& dst, int maxlevel) {   // sizes only (CreateMirrorMask uses the level sizes)
	dst.clear(); dst.push_back(src);
	for (int i = 1; i <= maxlevel; ++i) dst.push_back(Mat(Mat::zeros((dst.back().rows + 1) / 2, (dst.back().cols + 1) / 2, src.type())));
}

class FastFeatureDetector {
public:
	enum { TYPE_5_8 = 0, TYPE_7_12 = 1, TYPE_9_16 = 2 };
	int threshold; bool nonmax; int type;
	static Ptr<FastFeatureDetector> create(int threshold = 10, bool nonmaxSuppression = true, int type = TYPE_9_16) {
		auto p = std::make_shared<FastFeatureDetector>(); p->threshold = threshold; p->nonmax = nonmaxSuppression; p->type = type; return p; }
	void setThreshold(int t) { threshold = t; }
	void detect(const Mat& image, std::vector<KeyPoint>& keypoints, const Mat& mask = Mat()) {
		if (type < TYPE_5_8 || type > TYPE_9_16 || !nonmax) throw std::runtime_error("cvshim: FAST is restated with non-max suppression only");
		std::vector<orc_keypoint> out((size_t)image.rows * image.cols + 1);
		const int n = orc_fast_type(type, image.data, image.cols, image.rows, (int)image.step, mask.empty() ? nullptr : mask.data, mask.empty() ? 0 : (int)mask.step, threshold,
		                            out.data(), (int)out.size());
		keypoints.clear();
		for (int i = 0; i < n; ++i) keypoints.push_back(KeyPoint(out[i].x, out[i].y, out[i].size, out[i].angle, out[i].response, out[i].octave, out[i].class_id));
	}
};
class AgastFeatureDetector {
public:
	enum { AGAST_5_8 = 0, AGAST_7_12d = 1, AGAST_7_12s = 2, OAST_9_16 = 3 };
	int threshold; bool nonmax; int type;
	static Ptr<AgastFeatureDetector> create(int threshold = 10, bool nonmaxSuppression = true, int type = OAST_9_16) {
		auto p = std::make_shared<AgastFeatureDetector>(); p->threshold = threshold; p->nonmax = nonmaxSuppression; p->type = type; return p; }
	void setThreshold(int t) { threshold = t; }
	void detect(const Mat& image, std::vector<KeyPoint>& keypoints, const Mat& mask = Mat()) {
		if (type < AGAST_5_8 || type > OAST_9_16 || !nonmax) throw std::runtime_error("cvshim: AGAST is restated with non-max suppression only");
		std::vector<orc_keypoint> out((size_t)image.rows * image.cols + 1);
		const int n = orc_agast_type(type, image.data, image.cols, image.rows, (int)image.step, mask.empty() ? nullptr : mask.data, mask.empty() ? 0 : (int)mask.step, threshold,
		                             out.data(), (int)out.size());
		keypoints.clear();
		for (int i = 0; i < n; ++i) keypoints.push_back(KeyPoint(out[i].x, out[i].y, out[i].size, out[i].angle, out[i].response, out[i].octave, out[i].class_id));
	}
};
inline void FAST(const Mat& image, std::vector<KeyPoint>& keypoints, int threshold, bool nonmax = true) { FastFeatureDetector::create(threshold, nonmax)->detect(image, keypoints); }
struct KeyPointsFilter {
	static void retainBest(std::vector<KeyPoint>& k, int n) {   // only reached from the reference's unused ComputeKeyPointsOld path
		if (n >= 0 && (int)k.size() > n) { std::stable_sort(k.begin(), k.end(), [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; }); k.resize(n); } }
};

// Eigen <-> cv (opencv2/core/eigen.hpp): only dense double matrices
template <class T, int R, int Cc, int Opt, int MR, int MC> void cv2eigen(const Mat& src, Eigen::Matrix<T, R, Cc, Opt, MR, MC>& dst) {
	for (int i = 0; i < src.rows; ++i) for (int j = 0; j < src.cols; ++j) dst(i, j) = (T)src.at<double>(i, j); }
template <class T, int m, int n, class E> void cv2eigen(const Matx<T, m, n>& src, E& dst) { for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) dst(i, j) = src(i, j); }
template <class E, class T, int m, int n> void eigen2cv(const E& src, Matx<T, m, n>& dst) { for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) dst(i, j) = src(i, j); }
template <class E> void eigen2cv(const E& src, Mat& dst) { dst.create((int)src.rows(), (int)src.cols(), CV_64F); for (int i = 0; i < dst.rows; ++i) for (int j = 0; j < dst.cols; ++j) dst.at<double>(i, j) = src(i, j); }

// ---------------------------------------------------------------------------------------------- FileStorage (READ of the DBoW2 vocabulary layout)
// Just enough for DBoW2::TemplatedVocabulary::load (ThirdParty/DBoW2/DBoW2/TemplatedVocabulary.h:1573-1622): a top-level map with scalar
// entries and two sequences of one-line / two-line flow maps ("- { key:value, key:value, key:\"string\" }").  Writing is a no-op.
class FileNode {
public:
	struct Data { std::string scalar; std::vector<std::pair<std::string, FileNode>> map; std::vector<FileNode> seq; };
	std::shared_ptr<Data> d;
	FileNode() : d(std::make_shared<Data>()) {}
	FileNode operator[](const std::string& k) const { for (auto& e : d->map) if (e.first == k) return e.second; return FileNode(); }
	FileNode operator[](const char* k) const { return (*this)[std::string(k)]; }
	FileNode operator[](int i) const { return d->seq[i]; }
	size_t size() const { return d->seq.size(); }
	operator int() const { return (int)std::strtol(d->scalar.c_str(), nullptr, 10); }
	operator float() const { return (float)std::strtod(d->scalar.c_str(), nullptr); }
	operator double() const { return std::strtod(d->scalar.c_str(), nullptr); }
	operator std::string() const { return d->scalar; }
};
class FileStorage {
public:
	enum { READ = 0, WRITE = 1 };
	FileNode root; bool opened = false;
	FileStorage() {}
	FileStorage(const std::string& filename, int flags) { if (flags == READ) opened = parse(filename); }
	bool isOpened() const { return opened; }
	void release() {}
	FileNode operator[](const std::string& k) const { return root[k]; }
	FileNode operator[](const char* k) const { return root[std::string(k)]; }
	template <class T> FileStorage& operator<<(const T&) { return *this; }
private:
	static std::string trim(const std::string& v) { const size_t a = v.find_first_not_of(" \t\r\n"), b = v.find_last_not_of(" \t\r\n"); return a == std::string::npos ? std::string() : v.substr(a, b - a + 1); }
	static FileNode flow_map(const std::string& body) {   // key:value pairs separated by commas outside quotes
		FileNode n;
		size_t i = 0;
		while (i < body.size()) {
			const size_t c = body.find(':', i);
			if (c == std::string::npos) break;
			const std::string key = trim(body.substr(i, c - i));
			size_t j = c + 1;
			while (j < body.size() && (body[j] == ' ' || body[j] == '\n' || body[j] == '\r' || body[j] == '\t')) ++j;
			std::string val;
			if (j < body.size() && body[j] == '"') { const size_t e = body.find('"', j + 1); val = body.substr(j + 1, e - j - 1); j = body.find(',', e); }
			else { const size_t e = body.find(',', j); val = trim(body.substr(j, (e == std::string::npos ? body.size() : e) - j)); j = e; }
			FileNode s; s.d->scalar = val;
			n.d->map.emplace_back(key, s);
			if (j == std::string::npos) break;
			i = j + 1;
		}
		return n;
	}
	bool parse(const std::string& filename) {
		std::ifstream f(filename);
		if (!f) return false;
		std::stringstream ss; ss << f.rdbuf();
		const std::string t = ss.str();
		// top-level "name:" line, then "   key: value" scalars, then "   key:" sequences of "- { ... }"
		size_t pos = t.find('\n');   // skip %YAML header
		FileNode top, *cur = nullptr;
		std::string topName;
		std::vector<FileNode>* seq = nullptr;
		while (pos != std::string::npos && pos < t.size()) {
			size_t e = t.find('\n', pos + 1);
			std::string line = t.substr(pos + 1, (e == std::string::npos ? t.size() : e) - pos - 1);
			const std::string tl = trim(line);
			if (tl.empty() || tl[0] == '#') { pos = e; continue; }
			if (tl[0] == '-') {   // sequence element, may span lines until the closing brace
				size_t ob = t.find('{', pos), cb = t.find('}', ob);
				if (seq) seq->push_back(flow_map(t.substr(ob + 1, cb - ob - 1)));
				pos = t.find('\n', cb);
				continue;
			}
			const size_t c = tl.find(':');
			const std::string key = trim(tl.substr(0, c)), val = trim(tl.substr(c + 1));
			if (line[0] != ' ') { topName = key; cur = &top; seq = nullptr; }
			else if (val.empty()) { FileNode s; cur->d->map.emplace_back(key, s); seq = &cur->d->map.back().second.d->seq; }
			else { FileNode s; s.d->scalar = val; cur->d->map.emplace_back(key, s); seq = nullptr; }
			pos = e;
		}
		root.d->map.emplace_back(topName, top);
		return true;
	}
};

}  // namespace cv

"""The OpenCV surface of the drop-in translation units (integration/*_mcs.cpp), pinned on CPU.

The files replace src/mdBRIEFextractorOct.cpp, src/cORBmatcher.cpp and src/cMultiFrame.cpp of the reference (interfaces: include/mdBRIEFextractorOct.h:355-361,
include/cORBmatcher.h, include/cMultiFrame.h) and must compile against a genuine OpenCV 3.x — but this image has none: they have only ever met oracle/cvshim, a
stand-in written for this repository.  A method the shim has and OpenCV spells differently would break "exchange two source files" unnoticed.  So the surface is a
LIST (integration/mcs_dropin.h) and this test enforces it:
  1. every cv:: symbol the files name is on the list;
  2. every member called on an object is either on the list of cv members or a standard-library / reference-class member named here — anything new fails;
  3. the shim declares each listed member with the documented OpenCV 3.x shape (parameter count, defaults, constness), so code that compiles against the shim
     uses the members the way 3.x offers them;
  4. the layout assertions (cv::KeyPoint = 28 bytes, field offsets) are in the header every drop-in file includes.
Not a proof — only a build against real headers is — but it pins what the shim was allowed to invent."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = [os.path.join(ROOT, "integration", f) for f in ("mdBRIEFextractorOct_mcs.cpp", "cORBmatcher_mcs.cpp", "cMultiFrame_mcs.cpp", "mcs_dropin.h")]
SHIM = os.path.join(ROOT, "oracle", "cvshim", "cvshim.hpp")

CV_SYMBOLS = {"Mat", "Mat_", "InputArray", "OutputArray", "KeyPoint", "Vec2d", "Vec3d", "Vec4d", "Matx33d", "Matx44d", "norm"}
# members of cv objects the files may call / read (OpenCV 3.x: core/mat.hpp, core/types.hpp, core/matx.hpp)
CV_MEMBERS = {"empty", "isContinuous", "type", "ptr", "at", "create", "getMat", "release", "t", "dot", "rows", "cols", "data", "pt", "size", "angle", "response", "octave",
              "class_id", "x", "y", "val"}


def strip(src):
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r'"(?:\\.|[^"\\])*"', '""', src)


def test_only_listed_cv_symbols_are_named():
    for f in FILES:
        used = set(re.findall(r"\bcv::([A-Za-z_][A-Za-z_0-9]*)", strip(open(f).read())))
        assert used <= CV_SYMBOLS, (os.path.basename(f), sorted(used - CV_SYMBOLS))
    hdr = open(FILES[-1]).read()
    for s in CV_SYMBOLS:   # ... and the list in the header names each of them
        assert re.search(r"cv::%s\b|\b%s\b" % (s, s), hdr), s


def test_no_unlisted_cv_member_is_called():
    """members that exist on cv::Mat / _InputArray / _OutputArray in 3.x but are NOT on the list must not appear at all (as `.name(` or `.name` on anything): the
    drop-in files have no use for them, and each is a place where the shim and OpenCV could differ"""
    banned = ["step1", "clone", "copyTo", "row", "col", "rowRange", "colRange", "reshape", "convertTo", "setTo", "total", "channels", "depth", "elemSize",
              "elemSize1", "zeros", "ones", "eye", "mul", "inv", "cross", "diag", "push_back", "locateROI", "adjustROI", "isSubmatrix", "checkVector", "getMatRef",
              "needed", "fixedSize", "fixedType", "kind", "getUMat", "assign", "u", "datastart", "dataend", "allocator", "refcount", "flags", "dims"]
    std_ok = {"push_back", "assign", "data", "size", "flags"}   # the same names on std::vector / std::string / std::ios are fine: checked by receiver below
    for f in FILES[:3]:
        src = strip(open(f).read())
        for name in banned:
            for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9\]\)]*)\s*(\.|->)\s*%s\b" % name, src):
                recv = m.group(1)
                if name in std_ok:
                    continue
                # receivers that are cv objects in these files: anything ending in Mat-like names
                assert not re.search(r"(?i)(mat|image|mask|desc|img|_descriptors|_image|_mask)$", recv), (os.path.basename(f), m.group(0))


def test_mat_step_is_only_read_through_a_cast():
    """Mat::step is a MatStep (operator size_t() const) in 3.x and in the shim: `(int)m.step` / `(size_t)m.step` is the one spelling valid for both"""
    for f in FILES[:3]:
        src = strip(open(f).read())
        for m in re.finditer(r"(.{0,12})\.step\b(.{0,3})", src):
            assert re.search(r"\((int|size_t)\)\s*\w+$", m.group(1)) and not m.group(2).lstrip().startswith(("[", ".", "(")), (os.path.basename(f), m.group(0))


def _shim():
    return open(SHIM).read()


def test_shim_declares_the_listed_members_in_their_3x_form():
    s = _shim()
    mat = s[s.index("class Mat {"):]
    mat = mat[:mat.index("\n};")]
    # Mat::create(int rows, int cols, int type)
    assert re.search(r"void create\(int \w+, int \w+, int \w+\)", mat)
    # template<typename _Tp> _Tp* ptr(int i0 = 0)  (+ const form)
    assert re.search(r"template <class T> T\* ptr\(int \w+ = 0\)", mat) and re.search(r"template <class T> const T\* ptr\(int \w+ = 0\) const", mat)
    # template<typename _Tp> _Tp& at(int i0, int i1)
    assert re.search(r"template <class T> T& at\(int \w+, int \w+\)", mat)
    for decl in (r"bool empty\(\) const", r"bool isContinuous\(\) const", r"int type\(\) const"):
        assert re.search(decl, mat), decl
    # public data members rows, cols, data (uchar*)
    assert re.search(r"\bint [^;]*\brows\b[^;]*\bcols\b", mat) and re.search(r"uchar\* data", mat) and re.search(r"MatStep step;", mat)
    # _InputArray::getMat(int idx = -1) const — the shim's has no parameter: callers therefore never pass one, which 3.x accepts
    assert len(re.findall(r"Mat getMat\(\) const", s)) >= 2
    # _OutputArray::create(int rows, int cols, int type, ...) const, release() const
    assert re.search(r"void create\(int \w+, int \w+, int \w+\) const", s) and re.search(r"void release\(\) const", s)
    # KeyPoint(float x, float y, float _size, float _angle=-1, float _response=0, int _octave=0, int _class_id=-1) and the seven fields in order
    assert re.search(r"KeyPoint\(float \w+, float \w+, float \w+, float \w+ = -1, float \w+ = 0, int \w+ = 0, int \w+ = -1\)", s)
    kp = s[s.index("struct KeyPoint {"):]
    kp = kp[:kp.index("\n};")]
    fields = re.findall(r"\b(pt|size|angle|response|octave|class_id)\b\s*[;,]", kp)
    order = []
    for x in fields:
        if x not in order:
            order.append(x)
    assert order == ["pt", "size", "angle", "response", "octave", "class_id"], order


def test_calls_match_the_listed_arities():
    """how the files CALL the listed members: create with exactly (rows, cols, type); getMat / release / empty / isContinuous / type without arguments;
    ptr<T> with at most one index, at<T> with one (single row / column) or two"""
    for f in FILES[:3]:
        src = strip(open(f).read())
        for m in re.finditer(r"\.(getMat|release|isContinuous|type)\(([^()]*)\)", src):
            assert m.group(2).strip() == "", (os.path.basename(f), m.group(0))
        for m in re.finditer(r"(\w+)\.create\(([^;]*?)\);", src):
            assert m.group(2).count(",") == 2, (os.path.basename(f), m.group(0))
        for m in re.finditer(r"\.ptr<\w+>\(([^()]*)\)", src):
            assert m.group(1).count(",") == 0, (os.path.basename(f), m.group(0))
        for m in re.finditer(r"\.at<\w+>\(([^()]*)\)", src):
            assert m.group(1).count(",") <= 1, (os.path.basename(f), m.group(0))   # at(i0, i1), or at(i0) on a single-row / single-column matrix (both documented)


def test_layout_assertions_are_compiled_into_every_drop_in_file():
    hdr = open(FILES[-1]).read()
    assert "static_assert(sizeof(cv::KeyPoint) == 28" in hdr and "offsetof(cv::KeyPoint, class_id) == 24" in hdr and "offsetof(mcs_keypoint, class_id) == 24" in hdr
    for f in FILES[:3]:
        assert '#include "mcs_dropin.h"' in open(f).read(), os.path.basename(f)

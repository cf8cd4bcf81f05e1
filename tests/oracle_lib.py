"""ctypes binding of the CPU oracle (oracle/libmcs_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "oracle", "libmcs_oracle.so")


class KeyPoint(C.Structure):  # cv::KeyPoint, 28 B
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int), ("class_id", C.c_int)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28 and C.sizeof(KeyPoint) == 28


class Ocam(C.Structure):
    _fields_ = [("c", C.c_double), ("d", C.c_double), ("e", C.c_double), ("u0", C.c_double), ("v0", C.c_double),
                ("p", C.c_double * 16), ("p_deg", C.c_int), ("invP", C.c_double * 16), ("invP_deg", C.c_int),
                ("width", C.c_int), ("height", C.c_int)]


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scaleFactor", C.c_float), ("nlevels", C.c_int), ("edgeThreshold", C.c_int),
                ("firstLevel", C.c_int), ("scoreType", C.c_int), ("patchSize", C.c_int), ("fastThreshold", C.c_int),
                ("useAgast", C.c_int), ("fastAgastType", C.c_int), ("do_dBrief", C.c_int), ("learnMasks", C.c_int),
                ("descSize", C.c_int)]


def make_params(nfeatures=1000, scaleFactor=1.2, nlevels=8, fastThreshold=20, do_dBrief=0, learnMasks=0, descSize=32, fastAgastType=2, useAgast=0):
    return Params(nfeatures, scaleFactor, nlevels, 25, 0, 0, 32, fastThreshold, useAgast, fastAgastType, do_dBrief, learnMasks, descSize)


def make_ocam(cam):
    o = Ocam()
    o.c, o.d, o.e, o.u0, o.v0 = cam["c"], cam["d"], cam["e"], cam["u0"], cam["v0"]
    for i, v in enumerate(cam["p"]):
        o.p[i] = v
    o.p_deg = len(cam["p"])
    for i, v in enumerate(cam["invP"]):
        o.invP[i] = v
    o.invP_deg = len(cam["invP"])
    o.width, o.height = cam["width"], cam["height"]
    return o


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        u8p, i32p, f64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int), C.POINTER(C.c_double)
        L.orc_fastAtan2.restype = C.c_float
        L.orc_fastAtan2.argtypes = [C.c_float, C.c_float]
        L.orc_ic_angle.restype = C.c_float
        L.orc_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
        L.orc_cvRound.argtypes = [C.c_double]
        L.orc_extractor_create.restype = C.c_void_p
        L.orc_extractor_create.argtypes = [C.POINTER(Params)]
        L.orc_extractor_destroy.argtypes = [C.c_void_p]
        L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(Ocam),
                                  C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_extract_many.restype = C.c_long
        L.orc_extract_many.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("orc_tap_level_image", "orc_tap_level_mask"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int] + ([C.c_int] if name.endswith("image") else []) + [C.c_void_p]
        L.orc_tap_level_size.argtypes = [C.c_void_p, C.c_int, i32p, i32p]
        L.orc_tap_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_tap_selected.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_world2img.argtypes = [C.POINTER(Ocam), C.c_double, C.c_double, C.c_double, f64p, f64p]
        L.orc_img2world.argtypes = [C.POINTER(Ocam), C.c_double, C.c_double, f64p, f64p, f64p]
        L.orc_mirror_mask.argtypes = [C.POINTER(Ocam), C.c_void_p]
        L.orc_rays.argtypes = [C.POINTER(Ocam), C.c_void_p, C.c_int, C.c_void_p]
        L.orc_pos_in_grid.argtypes = [C.POINTER(Ocam), C.c_float, C.c_float, i32p, i32p]
        L.orc_dist64.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_dist64_masked.argtypes = [C.c_void_p] * 4 + [C.c_int]
        L.orc_search_kf_kf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.orc_search_kf_f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.orc_search_triangulation.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p, C.c_int,
                                                                                                           C.c_int, C.c_int, C.c_void_p]
        L.orc_check_epipolar.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        L.orc_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_score.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_fast_type.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_score_type.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_agast_type.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_agast_corners.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_agast_score_type.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_box5_inplace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_resize_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_resize_nearest.argtypes = L.orc_resize_linear.argtypes
        L.orc_border_reflect101.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_distribute_octtree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Extractor:
    """Oracle-side mdBRIEFextractorOct (one stateful instance per camera)."""

    def __init__(self, **kw):
        self.params = make_params(**kw)
        self.h = lib().orc_extractor_create(C.byref(self.params))
        assert self.h, "oracle rejected the parameters"
        self.cap = self.params.nfeatures + 4 * self.params.nlevels

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_extractor_destroy(self.h)
            self.h = None

    def __call__(self, img, mask, ocam):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        kps = np.zeros(self.cap, dtype=KP_DTYPE)
        ds = self.params.descSize
        desc = np.zeros((self.cap, ds), np.uint8)
        dmask = np.zeros((self.cap, ds), np.uint8)
        n = lib().orc_extract(self.h, ptr(img), w, h, w, ptr(mask), w, C.byref(ocam) if ocam is not None else None,
                              ptr(kps), self.cap, ptr(desc), ptr(dmask))
        assert n >= 0, n
        return kps[:n].copy(), desc[:n].copy(), dmask[:n].copy()

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        assert lib().orc_tap_level_size(self.h, level, C.byref(w), C.byref(h)) == 0
        return w.value, h.value

    def level_image(self, level, blurred=False):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        lib().orc_tap_level_image(self.h, level, int(blurred), ptr(out))
        return out

    def level_mask(self, level):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        assert lib().orc_tap_level_mask(self.h, level, ptr(out)) == 0
        return out

    def candidates(self, level):
        buf = np.zeros(1 << 16, dtype=KP_DTYPE)
        n = lib().orc_tap_candidates(self.h, level, ptr(buf), len(buf))
        return buf[:n].copy()

    def selected(self, level):
        buf = np.zeros(self.cap, dtype=KP_DTYPE)
        n = lib().orc_tap_selected(self.h, level, ptr(buf), len(buf))
        return buf[:n].copy()


def mirror_mask(ocam):
    m = np.zeros((ocam.height, ocam.width), np.uint8)
    lib().orc_mirror_mask(C.byref(ocam), ptr(m))
    return m


def search_kf_kf(d1, m1, v1, d2, m2, v2, masks, ratio):
    n1, dim = d1.shape
    out = np.full(n1, -1, np.int32)
    n = lib().orc_search_kf_kf(ptr(d1), ptr(m1), ptr(v1), n1, ptr(d2), ptr(m2), ptr(v2), d2.shape[0], dim, int(masks), ratio, ptr(out))
    return n, out


def search_kf_f(dk, mk, vk, df, mf, masks, ratio):
    nk, dim = dk.shape
    out = np.full(df.shape[0], -1, np.int32)
    n = lib().orc_search_kf_f(ptr(dk), ptr(mk), ptr(vk), nk, ptr(df), ptr(mf), df.shape[0], dim, int(masks), ratio, ptr(out))
    return n, out


def search_triangulation(d1, m1, mp1, cam1, rays1, d2, m2, mp2, cam2, rays2, E, nr_cams, masks):
    n1, dim = d1.shape
    out = np.full(n1, -1, np.int32)
    n = lib().orc_search_triangulation(ptr(d1), ptr(m1), ptr(mp1), ptr(cam1), ptr(rays1), n1, ptr(d2), ptr(m2), ptr(mp2), ptr(cam2),
                                       ptr(rays2), d2.shape[0], ptr(E), nr_cams, dim, int(masks), ptr(out))
    return n, out


def search_by_projection(px, py, vc, lv, pc, pd, pm, keys, fd, fm, fc, assigned, width, height, scales, th, ratio, masks):
    L = lib()
    L.orc_search_by_projection.argtypes = [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                                                              C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p]
    match = np.full(len(px), -1, np.int32)
    n = L.orc_search_by_projection(ptr(px), ptr(py), ptr(vc), ptr(lv), ptr(pc), ptr(pd), ptr(pm), len(px), ptr(keys), ptr(fd), ptr(fm), ptr(fc),
                                   ptr(assigned), len(keys), ptr(width), ptr(height), len(width), ptr(scales), len(scales), th, ratio, pd.shape[1],
                                   int(masks), ptr(match))
    return n, match


# ---- "next" row: grid-window matchers + projection (oracle/mcs_oracle.h: orc_frame_view ...)
class FrameViewO(C.Structure):
    _fields_ = [("keys", C.c_void_p), ("desc", C.c_void_p), ("mask", C.c_void_p), ("cam", C.c_void_p), ("n", C.c_int32), ("nrCams", C.c_int32),
                ("width", C.c_void_p), ("height", C.c_void_p)]


def frame_view(keys, desc, mask, cam, width, height):
    """-> (FrameViewO, keepalive).  keys: KP_DTYPE array; desc/mask: [n, dim] uint8 (mask may be None); cam: int32 [n]."""
    keep = [np.ascontiguousarray(keys), np.ascontiguousarray(desc, np.uint8), None if mask is None else np.ascontiguousarray(mask, np.uint8),
            np.ascontiguousarray(cam, np.int32), np.ascontiguousarray(width, np.int32), np.ascontiguousarray(height, np.int32)]
    fv = FrameViewO(ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), len(keep[0]), len(keep[4]), ptr(keep[4]), ptr(keep[5]))
    return fv, keep


def world_to_cam(MtMc_inv, cams, masks, pts3, pcam):
    L = lib()
    L.orc_world_to_cam.restype = None
    L.orc_world_to_cam.argtypes = [C.c_void_p, C.POINTER(Ocam), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    nr = len(cams)
    M = np.ascontiguousarray(np.asarray(MtMc_inv, np.float64).reshape(nr, 16))
    ocs = (Ocam * nr)(*[make_ocam(c) for c in cams])
    keep = [None if m is None else np.ascontiguousarray(m, np.uint8) for m in (masks or [None] * nr)]
    mp = (C.c_void_p * nr)(*[None if m is None else m.ctypes.data for m in keep])
    pts3 = np.ascontiguousarray(pts3, np.float64).reshape(-1, 3)
    pcam = np.ascontiguousarray(pcam, np.int32)
    n = len(pts3)
    uv, fl = np.zeros((max(n, 1), 2)), np.zeros(max(n, 1), np.uint8)
    L.orc_world_to_cam(ptr(M), ocs, mp if masks else None, ptr(pts3), ptr(pcam), n, ptr(uv), ptr(fl))
    return uv[:n], fl[:n]


def window_search(F1, hasMP1, F2, windowSize, minLevel, maxLevel, ratio, dim, masks, checkOri=0):
    L = lib()
    L.orc_window_search.argtypes = [C.POINTER(FrameViewO), C.c_void_p, C.POINTER(FrameViewO), C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p]
    hasMP1 = np.ascontiguousarray(hasMP1, np.uint8)
    m21 = np.full(max(F2.n, 1), -1, np.int32)
    n = L.orc_window_search(C.byref(F1), ptr(hasMP1), C.byref(F2), windowSize, minLevel, maxLevel, ratio, dim, int(masks), checkOri, ptr(m21))
    return n, m21[:F2.n]


def search_by_projection_frames(F1, mp1, bad1, F2, mp2, uv, inMask, windowSize, ratio, dim, masks):
    L = lib()
    L.orc_search_by_projection_frames.argtypes = [C.POINTER(FrameViewO), C.c_void_p, C.c_void_p, C.POINTER(FrameViewO), C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p]
    mp1, mp2 = np.ascontiguousarray(mp1, np.int32), np.ascontiguousarray(mp2, np.int32)
    bad1, inMask = np.ascontiguousarray(bad1, np.uint8), np.ascontiguousarray(inMask, np.uint8)
    uv = np.ascontiguousarray(uv, np.float64)
    m21 = np.full(max(F2.n, 1), -1, np.int32)
    n = L.orc_search_by_projection_frames(C.byref(F1), ptr(mp1), ptr(bad1), C.byref(F2), ptr(mp2), ptr(uv), ptr(inMask), windowSize, ratio, dim, int(masks),
                                          ptr(m21))
    return n, m21[:F2.n]


def search_for_initialization(F1, F2, prevMatched, windowSize, ratio, dim, masks, checkOri=0):
    L = lib()
    L.orc_search_for_initialization.argtypes = [C.POINTER(FrameViewO), C.POINTER(FrameViewO), C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                                C.c_void_p]
    pm = np.ascontiguousarray(prevMatched, np.float64).copy()
    m12 = np.full(max(F1.n, 1), -1, np.int32)
    n = L.orc_search_for_initialization(C.byref(F1), C.byref(F2), ptr(pm), windowSize, ratio, dim, int(masks), checkOri, ptr(m12))
    return n, m12[:F1.n], pm


def search_by_projection_last(Cur, curAssigned, Last, lastMP, lastOutlier, uv, inMask, scales, th, dim, masks, checkOri=0):
    L = lib()
    L.orc_search_by_projection_last.argtypes = [C.POINTER(FrameViewO), C.c_void_p, C.POINTER(FrameViewO), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p]
    ca = np.ascontiguousarray(curAssigned, np.uint8).copy()
    lastMP, lastOutlier, inMask = (np.ascontiguousarray(v, np.uint8) for v in (lastMP, lastOutlier, inMask))
    uv, scales = np.ascontiguousarray(uv, np.float64), np.ascontiguousarray(scales, np.float64)
    mc = np.full(max(Cur.n, 1), -1, np.int32)
    n = L.orc_search_by_projection_last(C.byref(Cur), ptr(ca), C.byref(Last), ptr(lastMP), ptr(lastOutlier), ptr(uv), ptr(inMask), ptr(scales), th, dim,
                                        int(masks), checkOri, ptr(mc))
    return n, mc[:Cur.n], ca


def distinctive_descriptor(desc, mask):
    """oracle cMapPoint::ComputeDistinctiveDescriptors: index of the chosen row (-1 for no rows)"""
    L = lib()
    L.orc_distinctive_descriptor.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    desc = np.ascontiguousarray(desc, np.uint8)
    mask = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    if len(desc) == 0:
        return -1
    return L.orc_distinctive_descriptor(ptr(desc), ptr(mask), len(desc), desc.shape[1], int(mask is not None))


def window_best(x, y, r, lo, hi, cam, pdesc, pmask, F, assigned, max_dist, skip_taken, dim, masks):
    L = lib()
    L.orc_window_best.argtypes = [C.c_void_p] * 8 + [C.c_int, C.POINTER(FrameViewO), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    x, y, r = (np.ascontiguousarray(v, np.float64) for v in (x, y, r))
    lo, hi, cam = (np.ascontiguousarray(v, np.int32) for v in (lo, hi, cam))
    pdesc = np.ascontiguousarray(pdesc, np.uint8)
    pmask = None if pmask is None else np.ascontiguousarray(pmask, np.uint8)
    asg = np.ascontiguousarray(assigned if assigned is not None else np.zeros(max(F.n, 1)), np.uint8).copy()
    n = len(x)
    match, dist = np.full(max(n, 1), -1, np.int32), np.zeros(max(n, 1), np.int32)
    nm = L.orc_window_best(ptr(x), ptr(y), ptr(r), ptr(lo), ptr(hi), ptr(cam), ptr(pdesc), ptr(pmask), n, C.byref(F), ptr(asg), max_dist, int(skip_taken), dim,
                           int(masks), ptr(match), ptr(dist))
    return nm, match[:n], dist[:n], asg


def bow_transform(voc, desc, levelsup):
    """oracle DBoW2 descent: voc = dict of multicol-slam_amd.io.load_vocabulary -> (leaf node, node at level L - levelsup) per row"""
    L = lib()
    L.orc_bow_transform.restype = None
    L.orc_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    nd = np.ascontiguousarray(voc["node_desc"], np.uint8)
    co, ci = np.ascontiguousarray(voc["child_off"], np.int32), np.ascontiguousarray(voc["child_idx"], np.int32)
    d = np.ascontiguousarray(desc, np.uint8)
    n = len(d)
    leaf, nid = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    L.orc_bow_transform(ptr(nd), ptr(co), ptr(ci), voc["L"], ptr(d), n, d.shape[1], levelsup, ptr(leaf), ptr(nid))
    return leaf[:n], nid[:n]


def search_kf_f_bow(dk, mk, vk, nodek, df, mf, nodef, masks, ratio):
    L = lib()
    L.orc_search_kf_f_bow.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    dk, df = np.ascontiguousarray(dk, np.uint8), np.ascontiguousarray(df, np.uint8)
    mk = None if mk is None else np.ascontiguousarray(mk, np.uint8)
    mf = None if mf is None else np.ascontiguousarray(mf, np.uint8)
    vk = np.ascontiguousarray(vk, np.uint8)
    nodek, nodef = np.ascontiguousarray(nodek, np.int32), np.ascontiguousarray(nodef, np.int32)
    out = np.full(max(len(df), 1), -1, np.int32)
    n = L.orc_search_kf_f_bow(ptr(dk), ptr(mk), ptr(vk), ptr(nodek), len(dk), ptr(df), ptr(mf), ptr(nodef), len(df), dk.shape[1], int(masks), ratio, ptr(out))
    return n, out[:len(df)]

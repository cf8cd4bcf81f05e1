"""ctypes driver of oracle/_ref's scene API (oracle/ref_wrap_match.cpp): real cMultiFrame / cMultiKeyFrame / cMapPoint / cORBmatcher objects of the
reference, compiled unmodified against oracle/cvshim.  TEST INFRASTRUCTURE."""
import ctypes as C
import os

import numpy as np

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmcs_ref.so")
vp = C.c_void_p


def P(a):
    return None if a is None else a.ctypes.data_as(vp)


class RefScene:
    def __init__(self, cams, masks, M_c, voc_path=None, so_path=None, **params):
        self.L = C.CDLL(so_path or REF_SO)
        L = self.L
        L.rs_create.restype = vp
        L.rs_create.argtypes = [vp, C.POINTER(O.Ocam), C.POINTER(vp), C.c_int, C.POINTER(O.Params), C.c_char_p]
        L.rs_add_frame.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_double, vp]
        for name in ("rs_frame_total", "rs_make_keyframe"):
            getattr(L, name).argtypes = [vp, C.c_int]
        L.rs_frame_get.argtypes = [vp, C.c_int] + [vp] * 7
        L.rs_frame_grid.argtypes = [vp, C.c_int, vp]
        L.rs_frame_extra.argtypes = [vp, C.c_int, vp, vp]
        L.rs_set_mappoints.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int]
        L.rs_set_outliers.argtypes = [vp, C.c_int, vp]
        L.rs_frame_mappoint_ids.argtypes = [vp, C.c_int, vp]
        L.rs_bow_kf_kf.argtypes = [vp, C.c_int, C.c_int, C.c_double, vp]
        L.rs_bow_kf_f.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_int, vp]
        L.rs_triangulation.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
        L.rs_window_search.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp]
        L.rs_search_init.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_double, C.c_int, vp]
        L.rs_proj_mappoints.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_double, C.c_double, vp]
        L.rs_proj_last.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_int, vp]
        L.rs_proj_frames.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, vp]
        L.rs_distinctive.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp]
        L.rs_fuse_probes.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_double, vp, vp]
        if hasattr(L, "rs_fuse"):
            L.rs_fuse.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, vp, vp]
            L.rs_sim3.argtypes = [vp, C.c_int, C.c_int, C.c_double, vp, vp, C.c_double, vp, vp]
            L.rs_tri_between.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
            L.rs_proj_scw.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp]
            if hasattr(L, "rs_proj_kf"):
                L.rs_proj_kf.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, vp, C.c_int, vp]
        L.rs_destroy.argtypes = [vp]
        self.nr = len(cams)
        self.prm = O.make_params(**params)
        self.dim = self.prm.descSize
        self._keep = [np.ascontiguousarray(m, np.uint8) for m in masks]
        oc = (O.Ocam * self.nr)(*[O.make_ocam(c) for c in cams])
        mp = (vp * self.nr)(*[m.ctypes.data for m in self._keep])
        Mc = np.ascontiguousarray(np.stack(M_c), np.float64)
        self.h = L.rs_create(P(Mc), oc, mp, self.nr, C.byref(self.prm), (voc_path or "").encode())
        assert self.h, "rs_create failed"

    def close(self):
        if self.h:
            self.L.rs_destroy(self.h)
            self.h = None

    def add_frame(self, imgs, ts, M_t):
        keep = [np.ascontiguousarray(i, np.uint8) for i in imgs]
        ip = (vp * self.nr)(*[k.ctypes.data for k in keep])
        Mt = np.ascontiguousarray(M_t, np.float64)
        f = self.L.rs_add_frame(self.h, ip, keep[0].shape[1], keep[0].shape[0], ts, P(Mt))
        assert f >= 0
        return f

    def frame(self, f):
        n = self.L.rs_frame_total(self.h, f)
        keys, d, m = np.zeros(n, O.KP_DTYPE), np.zeros((n, self.dim), np.uint8), np.zeros((n, self.dim), np.uint8)
        cam, rays, node, ginv, cell = np.zeros(n, np.int32), np.zeros((n, 3)), np.zeros(n, np.int32), np.zeros((self.nr, 2)), np.zeros(n, np.int32)
        self.L.rs_frame_get(self.h, f, P(keys), P(d), P(m), P(cam), P(rays), P(node), P(ginv))
        self.L.rs_frame_grid(self.h, f, P(cell))
        return dict(n=n, keys=keys, desc=d, mask=m, cam=cam, rays=rays, node=node, grid_inv=ginv, cell=cell)

    def frame_extra(self, f, levels=8):
        """every other field of the cMultiFrame the trackers read (rs_frame_extra): per-camera counts, image bounds, local indices, flags; the scale tables"""
        n = self.L.rs_frame_total(self.h, f)
        ints, dbl = np.zeros(5 * self.nr + n + 8, np.int32), np.zeros(1 + 3 * 16)
        self.L.rs_frame_extra(self.h, f, P(ints), P(dbl))
        return ints, dbl[:1 + 3 * levels]

    def make_keyframe(self, f):
        return self.L.rs_make_keyframe(self.h, f)

    def set_mappoints(self, is_kf, idx, flag, pos=None, share=None, base=0, ref_kf=0):
        flag = np.ascontiguousarray(flag, np.uint8)
        pos = None if pos is None else np.ascontiguousarray(pos, np.float64)
        share = None if share is None else np.ascontiguousarray(share, np.int32)
        assert self.L.rs_set_mappoints(self.h, int(is_kf), idx, P(flag), P(pos), P(share), base, ref_kf) == 0

    def set_outliers(self, f, out):
        out = np.ascontiguousarray(out, np.uint8)
        self.L.rs_set_outliers(self.h, f, P(out))

    def _ids(self, fn, n, *args):
        out = np.full(max(n, 1), -9, np.int32)
        cnt = fn(self.h, *args, P(out))
        return cnt, out[:n]

"""Helpers shared by the -m gpu parity tests: run the HIP path (through the C ABI) and the CPU oracle on the same inputs."""
import importlib

import numpy as np

import oracle_lib as O

mcs = importlib.import_module("multicol-slam_amd")
synth = importlib.import_module("multicol-slam_amd.synth")

_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        _ctx = mcs.Context(0)
    return _ctx


def cams3():
    return synth.lafida_cameras()


def frame_inputs(frame, ncam=3, cams=None):
    cams = cams or cams3()
    imgs = [synth.synth_image(frame, c, cams[c % len(cams)]) for c in range(ncam)]
    masks = [synth.mirror_mask(cams[c % len(cams)]) for c in range(ncam)]
    return imgs, masks, [cams[c % len(cams)] for c in range(ncam)]


def oracle_extract(img, mask, cam, **kw):
    ex = O.Extractor(**kw)
    oc = O.make_ocam(cam)
    kps, d, dm = ex(img, mask, oc)
    rays = np.zeros((len(kps), 3))
    if len(kps):
        O.lib().orc_rays(oc, O.ptr(kps), len(kps), O.ptr(rays))
    return ex, kps, d, dm, rays


def first_diff(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return "shape %s vs %s" % (a.shape, b.shape)
    bad = np.argwhere(a != b)
    if len(bad) == 0:
        return None
    i = tuple(bad[0])
    return "%d mismatches, first at %s: %s vs %s" % (len(bad), i, a[i], b[i])


# ---- raw device buffers through the HIP runtime the library itself uses (no torch: importing torch after libmcs_hip.so would bring a
# second HIP runtime into the process)
_hip = None


def hip():
    global _hip
    if _hip is None:
        import ctypes as C
        mcs.lib()
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    return _hip


class DevBuf:
    """device copy of a numpy array (hipMalloc + hipMemcpy H2D); .read() copies it back with the same dtype / shape"""

    def __init__(self, arr):
        import ctypes as C
        self.arr = np.ascontiguousarray(arr)
        self.ptr = C.c_void_p()
        assert hip().hipMalloc(C.byref(self.ptr), max(self.arr.nbytes, 1)) == 0
        if self.arr.nbytes:
            assert hip().hipMemcpy(self.ptr, self.arr.ctypes.data_as(C.c_void_p), self.arr.nbytes, 1) == 0

    def read(self):
        import ctypes as C
        out = np.empty_like(self.arr)
        assert hip().hipDeviceSynchronize() == 0
        if out.nbytes:
            assert hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, 2) == 0
        return out

    def zero(self):
        assert hip().hipMemset(self.ptr, 0, max(self.arr.nbytes, 1)) == 0

    def __del__(self):
        try:
            hip().hipFree(self.ptr)
        except Exception:
            pass

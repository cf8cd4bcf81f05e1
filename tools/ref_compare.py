#!/usr/bin/env python3
"""Compare the reference's own extractor (oracle/_ref, compiled from /root/reference against oracle/cvshim) with the oracle restatement."""
import ctypes as C, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oracle_lib as O
synth = importlib.import_module("multicol-slam_amd.synth")
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libmcs_ref.so"))
ref.ref_extract.argtypes = [C.POINTER(O.Params), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(O.Ocam), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]


def run_ref(img, mask, cam, **kw):
    prm = O.make_params(**kw)
    cap = prm.nfeatures + 64
    kps = np.zeros(cap, O.KP_DTYPE); d = np.zeros((cap, prm.descSize), np.uint8); m = np.zeros_like(d)
    oc = O.make_ocam(cam)
    img = np.ascontiguousarray(img)
    n = ref.ref_extract(C.byref(prm), img.ctypes.data, img.shape[1], img.shape[0], img.strides[0], None if mask is None else mask.ctypes.data,
                        0 if mask is None else mask.strides[0], C.byref(oc), kps.ctypes.data, cap, d.ctypes.data, m.ctypes.data)
    assert n >= 0, n
    return kps[:n], d[:n], m[:n]


if __name__ == "__main__":
    cams = synth.lafida_cameras()
    for mode, (db, lm) in {"orb": (0, 0), "dbrief": (1, 0), "mdbrief": (1, 1)}.items():
        for f, c in ((0, 0), (1, 2)):
            img = synth.synth_image(f, c, cams[c]); mask = np.ascontiguousarray(synth.mirror_mask(cams[c]))
            t = time.time(); k, d, m = run_ref(img, mask, cams[c], nfeatures=1000, do_dBrief=db, learnMasks=lm); tr = time.time() - t
            ok, od, om = O.Extractor(nfeatures=1000, do_dBrief=db, learnMasks=lm)(img, mask, O.make_ocam(cams[c]))
            if len(k) != len(ok):
                print(mode, f, c, "COUNT DIFFERS", len(k), len(ok)); continue
            kd = [(f_, int((k[f_] != ok[f_]).sum())) for f_ in k.dtype.names]
            print(mode, f, c, "n", len(k), "kp field diffs", [x for x in kd if x[1]], "desc diff rows", int((d != od).any(1).sum()), "mask diff rows", int((m != om).any(1).sum()), "ref %.2fs" % tr)

"""CPU: the oracle pinned against the REFERENCE's own extractor code.

oracle/_ref/libmcs_ref.so is /root/reference/src/mdBRIEFextractorOct.cpp + cam_model_omni.cpp compiled unmodified against oracle/cvshim
(OpenCV types re-implemented as far as those files need them; the image primitives resize / copyMakeBorder / FAST / boxFilter / fastAtan2
forward to the oracle's restatements).  Where that library exists the oracle must reproduce it bit for bit; everywhere it must reproduce the
golden vectors tools/gen_golden_ref.py stored from it."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_lib as O

synth = importlib.import_module("multicol-slam_amd.synth")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmcs_ref.so")
have_ref = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs the reference checkout)")


def oracle_case(frame, ci, w, h, nf, sf, nl, th, db, lm, ds, um):
    import gen_golden_ref as GG
    cam, img, mask = GG.case_inputs(frame, ci, w, h, um)
    k, d, m = O.Extractor(nfeatures=nf, scaleFactor=sf, nlevels=nl, fastThreshold=th, do_dBrief=db, learnMasks=lm, descSize=ds)(img, mask, O.make_ocam(cam))
    return k, d, m


def test_oracle_reproduces_reference_golden_vectors():
    import gen_golden_ref as GG
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_extract.npz"))
    for case in GG.CASES:
        name = case[0]
        k, d, m = oracle_case(*case[1:])
        gk, gd, gm = g[name + "_kps"], g[name + "_desc"], g[name + "_mask"]
        assert len(k) == len(gk) and len(gk) > 100, name
        for f in k.dtype.names:
            assert np.array_equal(k[f], gk[f]), (name, f)
        assert np.array_equal(d, gd) and np.array_equal(m, gm), name


@have_ref
@pytest.mark.parametrize("mode", ["orb", "dbrief", "mdbrief"])
def test_oracle_equals_reference_code_full_size(mode):
    import ref_compare as R
    db, lm = {"orb": (0, 0), "dbrief": (1, 0), "mdbrief": (1, 1)}[mode]
    cams = synth.lafida_cameras()
    for f, c, nf, th in ((0, 0, 1000, 20), (5, 1, 400, 20), (2, 2, 800, 5)):   # tracking and initialisation settings of src/cTracking.cpp:152-158
        img = synth.synth_image(f, c, cams[c])
        mask = np.ascontiguousarray(synth.mirror_mask(cams[c]))
        k, d, m = R.run_ref(img, mask, cams[c], nfeatures=nf, fastThreshold=th, do_dBrief=db, learnMasks=lm)
        ok, od, om = O.Extractor(nfeatures=nf, fastThreshold=th, do_dBrief=db, learnMasks=lm)(img, mask, O.make_ocam(cams[c]))
        assert len(k) == len(ok) and all(np.array_equal(k[x], ok[x]) for x in k.dtype.names), (mode, f, c)
        assert np.array_equal(d, od) and np.array_equal(m, om), (mode, f, c)


@have_ref
def test_camera_model_equals_reference_code():
    ref = C.CDLL(REF_SO)
    dbl, dp = C.c_double, C.POINTER(C.c_double)
    ref.ref_world2img.argtypes = [C.POINTER(O.Ocam), dbl, dbl, dbl, dp, dp]
    ref.ref_img2world.argtypes = [C.POINTER(O.Ocam), dbl, dbl, dp, dp, dp]
    L = O.lib()
    L.orc_world2img.argtypes = ref.ref_world2img.argtypes
    L.orc_img2world.argtypes = ref.ref_img2world.argtypes
    rng = np.random.default_rng(0)
    for cam in synth.lafida_cameras():
        oc = O.make_ocam(cam)
        for _ in range(2000):
            x, y, z = rng.normal(0, 2, 3)
            a, b, c, d = dbl(), dbl(), dbl(), dbl()
            ref.ref_world2img(C.byref(oc), x, y, z, C.byref(a), C.byref(b))
            L.orc_world2img(C.byref(oc), x, y, z, C.byref(c), C.byref(d))
            assert a.value == c.value and b.value == d.value
            u, v = rng.uniform(0, cam["width"]), rng.uniform(0, cam["height"])
            r = [dbl() for _ in range(6)]
            ref.ref_img2world(C.byref(oc), u, v, *[C.byref(q) for q in r[:3]])
            L.orc_img2world(C.byref(oc), u, v, *[C.byref(q) for q in r[3:]])
            assert [q.value for q in r[:3]] == [q.value for q in r[3:]]

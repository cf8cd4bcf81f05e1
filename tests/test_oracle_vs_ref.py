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


def oracle_case(frame, ci, w, h, nf, sf, nl, th, db, lm, ds, um, ft=2, ag=0):
    import gen_golden_ref as GG
    cam, img, mask = GG.case_inputs(frame, ci, w, h, um)
    k, d, m = O.Extractor(nfeatures=nf, scaleFactor=sf, nlevels=nl, fastThreshold=th, do_dBrief=db, learnMasks=lm, descSize=ds, fastAgastType=ft, useAgast=ag)(img, mask, O.make_ocam(cam))
    return k, d, m


def test_oracle_reproduces_reference_golden_vectors():
    import gen_golden_ref as GG
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_extract.npz"))
    for case in GG.CASES + GG.CASES_DET:   # CASES_DET: the small FAST rings and the four AGAST types
        name = case[0]
        k, d, m = oracle_case(*case[1:])
        gk, gd, gm = g[name + "_kps"], g[name + "_desc"], g[name + "_mask"]
        assert len(k) == len(gk) and len(gk) > (30 if name.startswith("fast5_8") else 100), name   # (the 8-pixel FAST ring needs the whole ring darker or brighter: few corners)
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
@pytest.mark.parametrize("case", [
    dict(scaleFactor=1.1, nlevels=8, nfeatures=600, descSize=32, do_dBrief=1, learnMasks=1),
    dict(scaleFactor=1.5, nlevels=5, nfeatures=500, descSize=32, do_dBrief=1, learnMasks=1),
    dict(scaleFactor=1.1, nlevels=12, nfeatures=1500, descSize=32, do_dBrief=0, learnMasks=0),
    dict(scaleFactor=1.2, nlevels=8, nfeatures=700, descSize=16, do_dBrief=1, learnMasks=1),
    dict(scaleFactor=1.2, nlevels=8, nfeatures=700, descSize=64, do_dBrief=1, learnMasks=1),
    dict(scaleFactor=1.2, nlevels=8, nfeatures=300, descSize=64, do_dBrief=1, learnMasks=0),
    dict(scaleFactor=1.3, nlevels=3, nfeatures=2000, descSize=32, do_dBrief=1, learnMasks=1, fastThreshold=9),
    # the two small FAST rings (extractor.fastAgastType 1 = TYPE_7_12, 0 = TYPE_5_8, src/mdBRIEFextractorOct.cpp:869-872): the detector itself is the
    # oracle's restatement on both sides (oracle/cvshim), what the reference's own code adds is the cell loop, the oct-tree and everything downstream
    dict(scaleFactor=1.2, nlevels=8, nfeatures=400, descSize=32, do_dBrief=1, learnMasks=1, fastThreshold=8, fastAgastType=1),
    dict(scaleFactor=1.2, nlevels=8, nfeatures=300, descSize=32, do_dBrief=0, learnMasks=0, fastThreshold=4, fastAgastType=0),
    # the AGAST branch of the reference's cell loop (useAgast, :869-870, 912-914), every type: again the detector is the oracle's restatement on both sides;
    # the reference's code adds the views, the offsets, the corners reported twice by overlapping views, the oct-tree over them and everything downstream
    dict(scaleFactor=1.2, nlevels=8, nfeatures=500, descSize=32, do_dBrief=1, learnMasks=1, fastThreshold=20, fastAgastType=0, useAgast=1),
    dict(scaleFactor=1.2, nlevels=8, nfeatures=500, descSize=32, do_dBrief=0, learnMasks=0, fastThreshold=20, fastAgastType=1, useAgast=1),
    dict(scaleFactor=1.2, nlevels=8, nfeatures=400, descSize=32, do_dBrief=1, learnMasks=1, fastThreshold=20, fastAgastType=2, useAgast=1),   # the type the shipped settings name
    dict(scaleFactor=1.3, nlevels=5, nfeatures=700, descSize=32, do_dBrief=1, learnMasks=0, fastThreshold=9, fastAgastType=3, useAgast=1),
])
def test_oracle_equals_reference_code_over_the_parameter_space(case):
    """pyramid geometry (scale factor, level count), feature budget, descriptor size and mode away from the shipped settings, on the Lafida sensor
    size and on the 1280x800 rig of BASELINE configs 4-5 (scaled calibration)"""
    import ref_compare as R
    cams = synth.lafida_cameras()
    big = synth.scaled_camera(cams[1], 1280, 800)
    for f, cam in ((3, cams[2]), (1, big)):
        img = synth.synth_image(f, 1, cam)
        mask = np.ascontiguousarray(synth.mirror_mask(cam))
        k, d, m = R.run_ref(img, mask, cam, **case)
        ok, od, om = O.Extractor(**case)(img, mask, O.make_ocam(cam))
        assert len(k) == len(ok) and len(k) > (100 if case.get("fastAgastType", 2) == 2 or case.get("useAgast") else 20) and all(np.array_equal(k[x], ok[x]) for x in k.dtype.names), (case, cam["width"], len(k))
        assert np.array_equal(d, od) and np.array_equal(m, om), (case, cam["width"])


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


def _expected_maps(vd, desc, levelsup):
    """BowVector / FeatureVector from the ORACLE's descent with DBoW2's map arithmetic (TF_IDF + L1)"""
    leaf, nid = O.bow_transform(vd, desc, levelsup)
    bow, node = {}, np.full(len(desc), -1, np.int32)
    for i, (lf, nd) in enumerate(zip(leaf, nid)):
        w = float(vd["weight"][lf])
        if w > 0:
            wid = int(vd["word_id"][lf])
            bow[wid] = bow.get(wid, 0.0) + w
            node[i] = nd
    norm = 0.0
    for k in sorted(bow):
        norm += abs(bow[k])
    return node, [(k, bow[k] / norm) for k in sorted(bow)]


@have_ref
@pytest.mark.parametrize("which", ["synthetic", "shipped"])
def test_bow_transform_equals_dbow2_code(which, tmp_path):
    """DBoW2 (ThirdParty/DBoW2, compiled unmodified into oracle/_ref): TemplatedVocabulary::load + transform vs the vocabulary loader of
    multicol-slam_amd.io + the oracle's descent + the host-side map arithmetic."""
    import vocab_synth
    io = importlib.import_module("multicol-slam_amd.io")
    if which == "shipped":
        path = "/root/reference/Examples/small_orb_omni_voc_9_6.yml"
        if not os.path.exists(path):
            pytest.skip("reference checkout not present")
    else:
        path = str(tmp_path / "voc.yml")
        vocab_synth.write_vocabulary(path, k=9, L=5, seed=3)
    vd = io.load_vocabulary(path)
    rng = np.random.default_rng(4)
    desc = rng.integers(0, 256, (3000, 32), dtype=np.uint8)
    desc[:500] = vd["node_desc"][1:501]          # exact node descriptors (distance 0 / ties)
    ref = C.CDLL(REF_SO)
    ref.ref_bow_transform.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    # levels at or above the shallowest leaf only: below it the reference leaves `nid` UNINITIALISED when a feature's path ends early
    # (NodeId nid; in the caller's loop, TemplatedVocabulary.h:1147-1158 -> the previous feature's value in practice); oracle and GPU return 0 there
    for levelsup in (4, 3, 7):
        node = np.zeros(len(desc), np.int32)
        ids, vals, info = np.zeros(8000, np.int32), np.zeros(8000, np.float64), np.zeros(3, np.int32)
        k = ref.ref_bow_transform(path.encode(), desc.ctypes.data, len(desc), levelsup, node.ctypes.data, ids.ctypes.data, vals.ctypes.data, 8000, info.ctypes.data)
        assert k > 0 and list(info) == [vd["n_words"], vd["k"], vd["L"]]
        enode, ebow = _expected_maps(vd, desc, levelsup)
        assert np.array_equal(node, enode)
        assert k == len(ebow) and list(ids[:k]) == [b[0] for b in ebow] and np.array_equal(vals[:k], np.array([b[1] for b in ebow]))


@have_ref
def test_camera_system_pose_and_projection_equal_reference_code():
    """src/cam_system_omni.cpp + cConverter.cpp compiled unmodified: cMultiCamSys_(M_t, M_c, camModels) -> MtMc_inv, WorldToCamHom_fast,
    isPointInMirrorMask vs the host mirror (frontend._matx_mul / _inv_mat, io.cayley2hom) and the oracle's orc_world_to_cam."""
    FE = importlib.import_module("multicol-slam_amd.frontend")
    io = importlib.import_module("multicol-slam_amd.io")
    import test_io_formats as T
    ref = C.CDLL(REF_SO)
    cams = synth.lafida_cameras()
    masks = [np.ascontiguousarray(synth.mirror_mask(c)) for c in cams]
    # Cayley -> homogeneous (include/misc.h) through the reference's template
    ref.ref_cayley2hom.argtypes = [C.c_void_p, C.c_void_p]
    M_c = []
    for cay in T.CAYLEY:
        c6, M = np.array(cay, np.float64), np.zeros((4, 4))
        ref.ref_cayley2hom(c6.ctypes.data, M.ctypes.data)
        assert np.array_equal(M, io.cayley2hom(cay))
        M_c.append(M)
    a = np.deg2rad(7.0)
    M_t = np.eye(4)
    M_t[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    M_t[:3, 3] = [0.3, -0.2, 0.1]
    rig = FE.cMultiCamSys_([FE.cCamModelGeneral_.from_dict(c, m) for c, m in zip(cams, masks)], M_c, M_t)
    rng = np.random.default_rng(2)
    n = 5000
    pts = rng.normal(0, 3.0, (n, 3))
    pc = rng.integers(0, 3, n).astype(np.int32)
    uv, fl, inv = np.zeros((n, 2)), np.zeros(n, np.uint8), np.zeros((3, 4, 4))
    oc = (O.Ocam * 3)(*[O.make_ocam(c) for c in cams])
    mp = (C.c_void_p * 3)(*[m.ctypes.data for m in masks])
    ref.ref_world_to_cam.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(O.Ocam), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    rc = ref.ref_world_to_cam(np.ascontiguousarray(M_t).ctypes.data, np.ascontiguousarray(np.stack(M_c)).ctypes.data, oc, mp, 3, pts.ctypes.data, pc.ctypes.data,
                              n, uv.ctypes.data, fl.ctypes.data, inv.ctypes.data)
    assert rc == 0
    for c in range(3):
        assert np.array_equal(inv[c], rig.MtMc_inv[c])          # invMat(M_t * M_c[c]) with the Matx summation order
    euv, efl = O.world_to_cam(np.stack(rig.MtMc_inv), cams, masks, pts, pc)
    fin = np.isfinite(euv).all(axis=1)
    assert np.array_equal(uv[fin], euv[fin]) and np.array_equal(fl[fin], efl[fin])
    assert 0.05 < (fl & 1).mean() < 0.9 and 0.2 < (fl >> 1).mean() < 0.8


@have_ref
def test_epipolar_check_and_median_equal_reference_code():
    """src/misc.cpp: CheckDistEpipolarLine (SearchForTriangulationRaw) and median() (ComputeDistinctiveDescriptors) vs the oracle"""
    ref = C.CDLL(REF_SO)
    ref.ref_check_epipolar.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    ref.ref_median_int.argtypes = [C.c_void_p, C.c_int]
    L = O.lib()
    L.orc_check_epipolar.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    rng = np.random.default_rng(9)
    agree = 0
    for i in range(20000):
        r1, r2 = rng.normal(0, 1, 3), rng.normal(0, 1, 3)
        r1 /= np.linalg.norm(r1); r2 /= np.linalg.norm(r2)
        E = rng.normal(0, 1, (3, 3)) if i % 50 else np.zeros((3, 3))
        th = float(rng.choice([1e-2, 1e-1, 0.5]))
        a = ref.ref_check_epipolar(r1.ctypes.data, r2.ctypes.data, np.ascontiguousarray(E).ctypes.data, th)
        b = L.orc_check_epipolar(r1.ctypes.data, r2.ctypes.data, np.ascontiguousarray(E).ctypes.data, th)
        assert a == b
        agree += a
    assert 1000 < agree < 19000
    for n in (1, 2, 3, 4, 7, 10, 33):
        v = rng.integers(0, 50, n).astype(np.int32)
        assert ref.ref_median_int(v.ctypes.data, n) == sorted(v.tolist())[n // 2]
    # the distinctive-descriptor rule built on that median: rows j > i, position size/2 (oracle restatement vs a direct numpy evaluation)
    for n in (3, 5, 8, 21):
        d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        dist = np.unpackbits(d[:, None, :] ^ d[None, :, :], axis=2).sum(2)
        med = [sorted(dist[i, i + 1:].tolist())[(n - 1 - i) // 2] for i in range(n - 1)]
        assert O.distinctive_descriptor(d, None) == int(np.argmin(med))


@have_ref
def test_gpu_golden_test_restates_the_generators_case_tables():
    """tests/test_gpu_golden.py cannot import the generator on the GPU box (it loads the reference library): its copy of the case tables must be the generator's"""
    import gen_golden_ref as GG
    src = open(os.path.join(ROOT, "tests", "test_gpu_golden.py")).read()
    ns = {}
    exec("cases =" + src[src.index("cases = [") + 7:src.index("    assert sorted")], ns)
    want = [tuple(t[:12]) + (2, 0) for t in GG.CASES] + [tuple(t[:12]) + tuple(t[13:]) for t in GG.CASES_DET]
    assert ns["cases"] == want

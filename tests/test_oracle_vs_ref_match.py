"""CPU: the oracle's matcher side pinned against the REFERENCE's own code — src/cORBmatcher.cpp, cMultiFrame.cpp, cMultiKeyFrame.cpp, cMapPoint.cpp
(+ cMap, cMultiKeyFrameDatabase, DBoW2) compiled unmodified against oracle/cvshim and driven through real cMultiFrame / cMultiKeyFrame /
cMapPoint / cORBmatcher objects (oracle/ref_wrap_match.cpp).  Needs oracle/_ref (built where the reference checkout exists; it travels)."""
import importlib
import os

import numpy as np
import pytest

import oracle_lib as O
import vocab_synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmcs_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs the reference checkout)")
synth = importlib.import_module("multicol-slam_amd.synth")
io = importlib.import_module("multicol-slam_amd.io")
W, H, NC = 754, 480, 3


def motion(rz_deg, t):
    a = np.deg2rad(rz_deg)
    M = np.eye(4)
    M[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    M[:3, 3] = t
    return M


def flat_vocabulary(path):
    """one leaf under the root: every feature falls into FeatureVector node 0, so the vocabulary-restricted loops become brute force"""
    with open(path, "w") as f:
        f.write("%YAML:1.0\nvocabulary:\n   k: 1\n   L: 1\n   scoringType: 0\n   weightingType: 0\n   nodes:\n")
        f.write("      - { nodeId:1, parentId:0, weight:1.0000000000000000e+000,\n          descriptor:\"%s \" }\n" % " ".join(["0"] * 32))
        f.write("   words:\n      - { wordId:0, nodeId:1 }\n")


@pytest.fixture(scope="module", params=["mdbrief_tree", "orb_flat"])
def scene(request, tmp_path_factory):
    import ref_scene
    import test_io_formats as T
    d = tmp_path_factory.mktemp("voc")
    tree = request.param == "mdbrief_tree"
    voc_path = str(d / "voc.yml")
    if tree:
        vocab_synth.write_vocabulary(voc_path, k=9, L=5, seed=3)
    else:
        flat_vocabulary(voc_path)
    cams = synth.lafida_cameras()
    masks = [np.ascontiguousarray(synth.mirror_mask(c)) for c in cams]
    M_c = [io.cayley2hom(c) for c in T.CAYLEY]
    params = dict(nfeatures=400, do_dBrief=1 if tree else 0, learnMasks=1 if tree else 0)
    S = ref_scene.RefScene(cams, masks, M_c, voc_path, **params)
    poses = [np.eye(4), motion(0.4, [0.02, -0.01, 0.015])]
    imgs = [synth.synth_multiframe(f, cams) for f in range(2)]
    frames = [S.frame(S.add_frame(imgs[f], 0.04 * f, poses[f])) for f in range(2)]
    yield dict(S=S, cams=cams, masks=masks, M_c=M_c, poses=poses, imgs=imgs, fr=frames, tree=tree, voc=io.load_vocabulary(voc_path), params=params,
               having=bool(params["learnMasks"]))
    S.close()


def view(fr, having, cams):
    return O.frame_view(fr["keys"], fr["desc"], fr["mask"] if having else None, fr["cam"], [c["width"] for c in cams], [c["height"] for c in cams])


def test_multiframe_constructor_fields(scene):
    """cMultiFrame::cMultiFrame (src/cMultiFrame.cpp:92-216) + ComputeBoW: per-camera extraction, flattening order, rays, grid, FeatureVector"""
    for f, fr in enumerate(scene["fr"]):
        s = 0
        for c in range(NC):
            ok, od, om = O.Extractor(**scene["params"])(scene["imgs"][f][c], scene["masks"][c], O.make_ocam(scene["cams"][c]))
            n = int((fr["cam"] == c).sum())
            assert n == len(ok) and (fr["cam"][s:s + n] == c).all()
            assert all(np.array_equal(fr["keys"][x][s:s + n], ok[x]) for x in ok.dtype.names)
            assert np.array_equal(fr["desc"][s:s + n], od) and np.array_equal(fr["mask"][s:s + n], om)
            rays = np.zeros((n, 3))
            oc = O.make_ocam(scene["cams"][c])
            O.lib().orc_rays(oc, O.ptr(ok), n, O.ptr(rays))
            assert np.array_equal(fr["rays"][s:s + n], rays)
            import ctypes as C
            for i in range(s, s + n, 7):
                gx, gy = C.c_int(), C.c_int()
                inside = O.lib().orc_pos_in_grid(C.byref(oc), C.c_float(fr["keys"]["x"][i]), C.c_float(fr["keys"]["y"][i]), C.byref(gx), C.byref(gy))
                assert fr["cell"][i] == (gx.value * 48 + gy.value if inside else -1)
            s += n
        leaf, nid = O.bow_transform(scene["voc"], fr["desc"], 4)
        assert np.array_equal(fr["node"], np.where(scene["voc"]["weight"][leaf] > 0, nid, -1))


def test_search_by_bow_keyframes_and_frame(scene):
    S, fr, having = scene["S"], scene["fr"], scene["having"]
    rng = np.random.default_rng(1)
    n0, n1 = fr[0]["n"], fr[1]["n"]
    k0, k1 = S.make_keyframe(0), S.make_keyframe(1)
    f0 = (rng.random(n0) < 0.8).astype(np.uint8)
    f1 = (rng.random(n1) < 0.7).astype(np.uint8)
    S.set_mappoints(True, k0, f0, base=0, ref_kf=k0)
    S.set_mappoints(True, k1, f1, base=100000, ref_kf=k1)
    scene["kfs"] = (k0, k1, f0, f1)
    # SearchByBoW(KF, KF) :885-966
    cnt, ids = S._ids(S.L.rs_bow_kf_kf, n0, k0, k1, 0.8)
    en, e12 = O.search_kf_kf(fr[0]["desc"], fr[0]["mask"], f0, fr[1]["desc"], fr[1]["mask"], f1, having, 0.8)
    assert cnt == en and np.array_equal(ids, np.where(e12 >= 0, 100000 + e12, -1)) and cnt > 50
    # SearchByBoW(KF, F) :179-323 — vocabulary-restricted with the tree, brute force with the one-leaf vocabulary
    cnt, ids = S._ids(S.L.rs_bow_kf_f, n1, k0, 1, 0.9, 0)
    if scene["tree"]:
        en, eF = O.search_kf_f_bow(fr[0]["desc"], fr[0]["mask"] if having else None, f0, fr[0]["node"], fr[1]["desc"], fr[1]["mask"] if having else None,
                                   fr[1]["node"], having, 0.9)
    else:
        en, eF = O.search_kf_f(fr[0]["desc"], fr[0]["mask"], f0, fr[1]["desc"], fr[1]["mask"], having, 0.9)
    assert cnt == en and np.array_equal(ids, eF) and cnt > 30


def test_search_for_triangulation_raw(scene):
    S, fr, having = scene["S"], scene["fr"], scene["having"]
    if "kfs" not in scene:
        pytest.skip("needs the keyframes of the previous test")
    k0, k1, f0, f1 = scene["kfs"]
    n0 = fr[0]["n"]
    m12, E = np.zeros(n0, np.int32), np.zeros((NC * NC, 9))
    cnt = S.L.rs_triangulation(S.h, k0, k1, 0, m12.ctypes.data, E.ctypes.data)
    # the essential matrices of :990-1003 from the host-side pose algebra (frontend._matx_mul / _inv_mat mirror cv::Matx)
    FE = importlib.import_module("multicol-slam_amd.frontend")
    r0 = FE.cMultiCamSys_([FE.cCamModelGeneral_.from_dict(c, None) for c in scene["cams"]], scene["M_c"], scene["poses"][0])
    r1 = FE.cMultiCamSys_([FE.cCamModelGeneral_.from_dict(c, None) for c in scene["cams"]], scene["M_c"], scene["poses"][1])
    for i in range(NC):
        for j in range(NC):
            assert np.allclose(E[i * NC + j].reshape(3, 3), compute_E(r0.MtMc_inv[i], r1.MtMc[j]), rtol=0, atol=1e-12)
    en, e12 = O.search_triangulation(fr[0]["desc"], fr[0]["mask"], f0, fr[0]["cam"], fr[0]["rays"], fr[1]["desc"], fr[1]["mask"], f1, fr[1]["cam"], fr[1]["rays"],
                                     E, NC, having)
    assert cnt == en and np.array_equal(m12, e12)


def compute_E(T1, T2):   # src/misc.cpp:71-85
    R1, R2, t1, t2 = T1[:3, :3], T2[:3, :3], T1[:3, 3], T2[:3, 3]
    R12 = R1 @ R2.T
    t12 = -R1 @ R2.T @ t2 + t1
    t12 = t12 / np.linalg.norm(t12)
    sk = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    return sk @ R12


def test_window_search_and_initialization(scene):
    S, fr, having, cams = scene["S"], scene["fr"], scene["having"], scene["cams"]
    rng = np.random.default_rng(2)
    n0, n1 = fr[0]["n"], fr[1]["n"]
    flag = rng.choice([0, 1, 2], n0, p=[0.25, 0.7, 0.05]).astype(np.uint8)
    if "kfs" not in scene and "kf_any" not in scene:   # map points need a reference keyframe; run alone, this test has to make one
        scene["kf_any"] = S.make_keyframe(0)
    S.set_mappoints(False, 0, flag, base=200000, ref_kf=0)
    v0, _a = view(fr[0], having, cams)
    v1, _b = view(fr[1], having, cams)
    for window, lo, hi in ((60, 0, 2**31 - 1), (50, 3, 2**31 - 1), (40, 1, 5)):
        cnt, ids = S._ids(S.L.rs_window_search, n1, 0, 1, window, lo, hi, 0.8, 0)
        en, e21 = O.window_search(v0, (flag == 1).astype(np.uint8), v1, window, lo, hi if hi < 2**31 - 1 else -1, 0.8, 32, having)
        assert cnt == en and np.array_equal(ids, np.where(e21 >= 0, 200000 + e21, -1)) and cnt > 20, (window, lo, hi)
    for window in (50, 100):
        prev = np.stack([fr[0]["keys"]["x"], fr[0]["keys"]["y"]], axis=1).astype(np.float64)
        prev[:3] = [[-50, 10], [2000, 10], [377, 240]]
        p = prev.copy()
        m12 = np.zeros(n0, np.int32)
        cnt = S.L.rs_search_init(S.h, 0, 1, p.ctypes.data, window, 0.9, 0, m12.ctypes.data)
        en, e12, ep = O.search_for_initialization(v0, v1, prev, window, 0.9, 32, having)
        assert cnt == en and np.array_equal(m12, e12) and np.array_equal(p, ep) and cnt > 20


def test_projection_searches(scene):
    S, fr, having, cams = scene["S"], scene["fr"], scene["having"], scene["cams"]
    if "kfs" not in scene:
        pytest.skip("needs the keyframes of an earlier test")
    k0, k1, f0, f1 = scene["kfs"]
    FE = importlib.import_module("multicol-slam_amd.frontend")
    rng = np.random.default_rng(3)
    n0, n1 = fr[0]["n"], fr[1]["n"]
    rig0 = FE.cMultiCamSys_([FE.cCamModelGeneral_.from_dict(c, None) for c in cams], scene["M_c"], scene["poses"][0])
    rig1 = FE.cMultiCamSys_([FE.cCamModelGeneral_.from_dict(c, None) for c in cams], scene["M_c"], scene["poses"][1])
    pos = np.stack([(rig0.MtMc[int(c)] @ np.append(r * rng.uniform(1.5, 6.0), 1.0))[:3] for c, r in zip(fr[0]["cam"], fr[0]["rays"])])
    sc = np.cumprod([1.0] + [float(np.float32(1.2))] * 7)
    v1, _b = view(fr[1], having, cams)
    v0, _a = view(fr[0], having, cams)
    # --- SearchByProjection(F, vpMapPoints, th) :67-166 — the map points of keyframe 0, with tracking fields as isInFrustum leaves them
    inview = np.zeros((n0, NC), np.uint8); px = np.zeros((n0, NC)); py = np.zeros((n0, NC)); lv = np.zeros((n0, NC), np.int32); vc = np.ones((n0, NC))
    for i in range(n0):
        c = int(fr[0]["cam"][i])
        for cc in ([c] if rng.random() < 0.9 else [c, (c + 1) % 3]):
            inview[i, cc] = 1
            px[i, cc], py[i, cc] = fr[0]["keys"]["x"][i] + 3.0 + rng.normal(0, 1.5), fr[0]["keys"]["y"][i] + 1.0 + rng.normal(0, 1.5)
            lv[i, cc] = int(np.clip(fr[0]["keys"]["octave"][i] + rng.integers(-1, 2), 0, 7))
            vc[i, cc] = float(rng.choice([0.9995, 0.99, 0.5]))
    pre = (rng.random(n1) < 0.1).astype(np.uint8)
    S.set_mappoints(False, 1, pre, base=300000, ref_kf=k0)
    cnt, ids = S._ids(S.L.rs_proj_mappoints, n1, 1, k0, inview.ctypes.data, px.ctypes.data, py.ctypes.data, lv.ctypes.data, vc.ctypes.data, 3.0, 0.8)
    rows = [(i, c) for i in range(n0) if f0[i] for c in range(NC) if inview[i, c]]
    ii = np.array([r[0] for r in rows]); cc = np.array([r[1] for r in rows], np.int32)
    asg = pre.copy()
    en, em = O.search_by_projection(px[ii, cc], py[ii, cc], vc[ii, cc], lv[ii, cc], cc, fr[0]["desc"][ii], fr[0]["mask"][ii], np.ascontiguousarray(fr[1]["keys"]),
                                    fr[1]["desc"], fr[1]["mask"], fr[1]["cam"], asg, np.array([W] * NC, np.int32), np.array([H] * NC, np.int32), sc, 3.0, 0.8, having)
    exp = np.where(pre == 1, 300000 + np.arange(n1), -1)
    for p, j in enumerate(em):
        if j >= 0:
            exp[j] = ii[p]
    assert cnt == en and np.array_equal(ids, exp) and cnt > 30
    # --- SearchByProjection(CurrentFrame, LastFrame, th) :1990-2118
    flag = rng.choice([0, 1, 2], n0, p=[0.2, 0.75, 0.05]).astype(np.uint8)
    outl = (rng.random(n0) < 0.1).astype(np.uint8)
    S.set_mappoints(False, 0, flag, pos=pos, base=400000, ref_kf=k0)
    S.set_outliers(0, outl)
    pre = (rng.random(n1) < 0.1).astype(np.uint8)
    S.set_mappoints(False, 1, pre, base=500000, ref_kf=k0)
    cnt, ids = S._ids(S.L.rs_proj_last, n1, 1, 0, 15.0, 0)
    euv, efl = O.world_to_cam(np.stack(rig1.MtMc_inv), cams, scene["masks"], pos, fr[0]["cam"])
    en, ecur, _ = O.search_by_projection_last(v1, pre, v0, (flag == 1).astype(np.uint8), outl, euv, efl & 1, sc, 15.0, 32, having)
    exp = np.where(pre == 1, 500000 + np.arange(n1), -1)
    exp[ecur >= 0] = 400000 + ecur[ecur >= 0]
    assert cnt == en and np.array_equal(ids, exp) and cnt > 20
    S.set_outliers(0, np.zeros(n0, np.uint8))
    # --- SearchByProjection(F1, F2, windowSize, vpMapPointMatches2) :476-577 (duplicated observations, points F2 already holds)
    flag = rng.choice([0, 1, 2], n0, p=[0.3, 0.65, 0.05]).astype(np.uint8)
    share = np.full(n0, -1, np.int32)
    owners = np.flatnonzero(flag == 1)
    for i in np.flatnonzero(flag == 0)[:30]:
        o = int(owners[owners < i][-1]) if (owners < i).any() else -1
        if o >= 0:
            flag[i], share[i] = 1, o
    S.set_mappoints(False, 0, flag, pos=pos, share=share, base=600000, ref_kf=k0)
    pre = (rng.random(n1) < 0.1).astype(np.uint8)
    S.set_mappoints(False, 1, pre, base=700000, ref_kf=k0)
    cnt, ids = S._ids(S.L.rs_proj_frames, n1, 0, 1, 40, 0.8)
    mp1 = np.where(flag > 0, np.where(share >= 0, 600000 + share, 600000 + np.arange(n0)), -1).astype(np.int32)
    Pp = np.repeat(np.where(share[:, None] >= 0, pos[np.maximum(share, 0)], pos), NC, axis=0)
    euv, efl = O.world_to_cam(np.stack(rig1.MtMc_inv), cams, scene["masks"], Pp, np.tile(np.arange(NC, dtype=np.int32), n0))
    en, e21 = O.search_by_projection_frames(v0, mp1, (flag == 2).astype(np.uint8), v1, np.where(pre == 1, 700000 + np.arange(n1), -1).astype(np.int32),
                                            euv, efl & 1, 40, 0.8, 32, having)
    exp = np.where(pre == 1, 700000 + np.arange(n1), -1)
    exp[e21 >= 0] = mp1[e21[e21 >= 0]]
    assert cnt == en and np.array_equal(ids, exp) and cnt > 10


def test_compute_distinctive_descriptors(scene):
    S, fr, having = scene["S"], scene["fr"], scene["having"]
    if "kfs" not in scene:
        pytest.skip("needs the keyframes of an earlier test")
    k0 = scene["kfs"][0]
    rng = np.random.default_rng(4)
    for n in (1, 2, 3, 4, 7, 12, 25):
        idx = np.sort(rng.choice(fr[0]["n"], n, replace=False)).astype(np.int32)
        d, m = np.zeros(32, np.uint8), np.zeros(32, np.uint8)
        assert S.L.rs_distinctive(S.h, k0, idx.ctypes.data, n, d.ctypes.data, m.ctypes.data) == 0
        b = O.distinctive_descriptor(fr[0]["desc"][idx], fr[0]["mask"][idx] if having else None)
        assert np.array_equal(d, fr[0]["desc"][idx[b]])
        if having:
            assert np.array_equal(m, fr[0]["mask"][idx[b]])


def rotation_filter(variant, angle_slot, angle_partner, match, swapped, accepted=None):
    import ctypes as C
    L = O.lib()
    L.orc_rotation_consistency.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    a, b = np.ascontiguousarray(angle_slot, np.float32), np.ascontiguousarray(angle_partner, np.float32)
    m = np.ascontiguousarray(match, np.int32).copy()
    acc = None if accepted is None else np.ascontiguousarray(accepted, np.int32)
    removed = L.orc_rotation_consistency(variant, O.ptr(a), O.ptr(b), O.ptr(acc), O.ptr(m), len(m), swapped)
    return removed, m


def test_rotation_consistency_with_check_orientation(scene):
    """mbCheckOrientation = true (never used by the reference's callers, but a constructor argument): the reference's searches with the flag vs the
    oracle — embedded histograms for the window searches, and the plain search + orc_rotation_consistency for the brute-force ones."""
    S, fr, having, cams = scene["S"], scene["fr"], scene["having"], scene["cams"]
    if "kfs" not in scene:
        pytest.skip("needs the keyframes of an earlier test")
    k0, k1, f0, f1 = scene["kfs"]
    rng = np.random.default_rng(5)
    n0, n1 = fr[0]["n"], fr[1]["n"]
    a0, a1 = fr[0]["keys"]["angle"], fr[1]["keys"]["angle"]
    v0, _a = view(fr[0], having, cams)
    v1, _b = view(fr[1], having, cams)
    # SearchByBoW(KF, F): matchF[j] = kf feature; rot = KF angle - frame angle (variant 0, partner minus slot)
    cnt, ids = S._ids(S.L.rs_bow_kf_f, n1, k0, 1, 0.9, 1)
    if scene["tree"]:
        en, eF = O.search_kf_f_bow(fr[0]["desc"], fr[0]["mask"] if having else None, f0, fr[0]["node"], fr[1]["desc"], fr[1]["mask"] if having else None,
                                   fr[1]["node"], having, 0.9)
    else:
        en, eF = O.search_kf_f(fr[0]["desc"], fr[0]["mask"], f0, fr[1]["desc"], fr[1]["mask"], having, 0.9)
    rem, eF2 = rotation_filter(0, a1, a0, eF, 1)
    assert cnt == en - rem and np.array_equal(ids, eF2) and rem > 0
    # SearchForTriangulationRaw: match12[idx1] = idx2; rot = kp1 - kp2 (variant 3, slot minus partner)
    m12, E = np.zeros(n0, np.int32), np.zeros((NC * NC, 9))
    cnt = S.L.rs_triangulation(S.h, k0, k1, 1, m12.ctypes.data, E.ctypes.data)
    en, e12 = O.search_triangulation(fr[0]["desc"], fr[0]["mask"], f0, fr[0]["cam"], fr[0]["rays"], fr[1]["desc"], fr[1]["mask"], f1, fr[1]["cam"], fr[1]["rays"],
                                     E, NC, having)
    rem, e12f = rotation_filter(3, a0, a1, e12, 0)
    assert cnt == en - rem and np.array_equal(m12, e12f)
    # WindowSearch / SearchForInitialization: the oracle's embedded histograms, and the generic filter on top of the unfiltered result
    flag = rng.choice([0, 1], n0, p=[0.25, 0.75]).astype(np.uint8)
    S.set_mappoints(False, 0, flag, base=800000, ref_kf=k0)
    cnt, ids = S._ids(S.L.rs_window_search, n1, 0, 1, 60, 0, 2**31 - 1, 0.8, 1)
    en, e21 = O.window_search(v0, flag, v1, 60, 0, -1, 0.8, 32, having, checkOri=1)
    assert cnt == en and np.array_equal(ids, np.where(e21 >= 0, 800000 + e21, -1))
    un, u21 = O.window_search(v0, flag, v1, 60, 0, -1, 0.8, 32, having, checkOri=0)
    rem, g21 = rotation_filter(1, a1, a0, u21, 1)
    assert np.array_equal(g21, e21) and un - rem == en and rem > 0
    prev = np.stack([fr[0]["keys"]["x"], fr[0]["keys"]["y"]], axis=1).astype(np.float64)
    p = prev.copy()
    m12 = np.zeros(n0, np.int32)
    cnt = S.L.rs_search_init(S.h, 0, 1, p.ctypes.data, 100, 0.9, 1, m12.ctypes.data)
    en, e12, ep = O.search_for_initialization(v0, v1, prev, 100, 0.9, 32, having, checkOri=1)
    assert cnt == en and np.array_equal(m12, e12) and np.array_equal(p, ep)
    # SearchByProjection(Cur, Last): variant 0, rot = Last angle - Cur angle (partner minus slot)
    FE = importlib.import_module("multicol-slam_amd.frontend")
    rig0 = FE.cMultiCamSys_([FE.cCamModelGeneral_.from_dict(c, None) for c in cams], scene["M_c"], scene["poses"][0])
    rig1 = FE.cMultiCamSys_([FE.cCamModelGeneral_.from_dict(c, None) for c in cams], scene["M_c"], scene["poses"][1])
    pos = np.stack([(rig0.MtMc[int(c)] @ np.append(r * rng.uniform(1.5, 6.0), 1.0))[:3] for c, r in zip(fr[0]["cam"], fr[0]["rays"])])
    sc = np.cumprod([1.0] + [float(np.float32(1.2))] * 7)
    S.set_mappoints(False, 0, flag, pos=pos, base=900000, ref_kf=k0)
    S.set_mappoints(False, 1, np.zeros(n1, np.uint8), base=0, ref_kf=k0)
    cnt, ids = S._ids(S.L.rs_proj_last, n1, 1, 0, 15.0, 1)
    euv, efl = O.world_to_cam(np.stack(rig1.MtMc_inv), cams, scene["masks"], pos, fr[0]["cam"])
    en, ecur, _ = O.search_by_projection_last(v1, np.zeros(n1, np.uint8), v0, flag, np.zeros(n0, np.uint8), euv, efl & 1, sc, 15.0, 32, having, checkOri=1)
    assert cnt == en and np.array_equal(ids, np.where(ecur >= 0, 900000 + ecur, -1))


def test_fuse_search_loop(scene):
    """cORBmatcher::Fuse(pKF, curKF, vpMapPoints, th) :1265-1418 — projection into every camera of the target keyframe, depth gate, predicted
    level, radius th*scale, best descriptor inside the window with levels [l-1, l], accept <= TH_LOW — vs orc_world_to_cam + orc_window_best
    (the loop mcs_window_best implements).  The reference is run one fresh map point at a time against a keyframe without map points."""
    S, fr, having, cams = scene["S"], scene["fr"], scene["having"], scene["cams"]
    FE = importlib.import_module("multicol-slam_amd.frontend")
    kS = S.make_keyframe(0)
    S.set_mappoints(False, 1, np.zeros(fr[1]["n"], np.uint8), base=0, ref_kf=kS)    # frame 1 without points -> a clean target keyframe
    kT = S.make_keyframe(1)
    rng = np.random.default_rng(6)
    n0 = fr[0]["n"]
    rig0 = FE.cMultiCamSys_([FE.cCamModelGeneral_.from_dict(c, None) for c in cams], scene["M_c"], scene["poses"][0])
    rig1 = FE.cMultiCamSys_([FE.cCamModelGeneral_.from_dict(c, None) for c in cams], scene["M_c"], scene["poses"][1])
    feat = np.sort(rng.choice(n0, 400, replace=False)).astype(np.int32)
    pos = np.stack([(rig0.MtMc[int(fr[0]["cam"][i])] @ np.append(fr[0]["rays"][i] * rng.uniform(1.0, 8.0), 1.0))[:3] for i in feat])
    for th in (3.0, 10.0):
        best, mm = np.zeros((len(feat), NC), np.int32), np.zeros((len(feat), 2))
        assert S.L.rs_fuse_probes(S.h, kT, kS, feat.ctypes.data, np.ascontiguousarray(pos).ctypes.data, len(feat), th, best.ctypes.data, mm.ctypes.data) == 0
        # expectation from the oracle
        sf = np.cumprod([1.0] + [float(np.float32(1.2))] * 7)
        Ow = scene["poses"][1][:3, 3]
        P3 = np.repeat(pos, NC, axis=0)
        pc = np.tile(np.arange(NC, dtype=np.int32), len(feat))
        uv, fl = O.world_to_cam(np.stack(rig1.MtMc_inv), cams, scene["masks"], P3, pc)
        PO = P3 - Ow
        dist3D = np.sqrt(PO[:, 0] * PO[:, 0] + PO[:, 1] * PO[:, 1] + PO[:, 2] * PO[:, 2]).astype(np.float32).astype(np.float64)
        minD, maxD = np.repeat(mm[:, 0], NC), np.repeat(mm[:, 1], NC)
        ok = ((fl & 1) == 1) & ~((dist3D < minD) | (dist3D > maxD))
        lvl = np.minimum(np.searchsorted(sf, dist3D / minD, side="left"), 7).astype(np.int32)
        v1, _k = view(fr[1], having, cams)
        rows = np.repeat(feat, NC)
        sel = np.flatnonzero(ok)
        th_low = 32 if having else 64
        en, em, ed, _ = O.window_best(uv[sel, 0], uv[sel, 1], th * sf[lvl[sel]], lvl[sel] - 1, lvl[sel], pc[sel], fr[0]["desc"][rows[sel]],
                                      fr[0]["mask"][rows[sel]] if having else None, v1, None, th_low, False, 32, having)
        exp = np.full(len(feat) * NC, -1, np.int32)
        exp[sel] = em
        assert np.array_equal(best.reshape(-1), exp), (th, int((best.reshape(-1) != exp).sum()))
        assert (exp >= 0).sum() > 40

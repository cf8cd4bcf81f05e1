"""-m gpu: the grid-window matchers of cTracking through the reference-named host classes and the C ABI vs the oracle.

WindowSearch, SearchByProjection(F1,F2,windowSize,..), SearchByProjection(Cur,Last,th), SearchForInitialization
(src/cORBmatcher.cpp:326-726, 1990-2118) + cMultiCamSys_::WorldToCamHom_fast / isPointInMirrorMask.  Match indices bit-exact.
"""
import ctypes as C
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_common
    return gpu_common


@pytest.fixture(scope="module")
def FE():
    return importlib.import_module("multicol-slam_amd.frontend")


class MP:   # stand-in for cMapPoint
    def __init__(self, i, pos=None, bad=False):
        self.i, self.pos, self.bad = i, pos, bad

    def isBad(self):
        return self.bad

    def GetWorldPos(self):
        return self.pos


def rot_y(deg):
    a = np.deg2rad(deg)
    M = np.eye(4)
    M[0, 0], M[0, 2], M[2, 0], M[2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
    return M


def small_motion(rx, ry, rz, t):
    ax, ay, az = np.deg2rad([rx, ry, rz])
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    M = np.eye(4)
    M[:3, :3] = Rz @ Ry @ Rx
    M[:3, 3] = t
    return M


@pytest.fixture(scope="module")
def frames(G, FE):
    cams = G.cams3()
    models = [FE.cCamModelGeneral_.from_dict(c, G.synth.mirror_mask(c)) for c in cams]
    M_c = []
    for c in range(3):
        M = rot_y(120.0 * c)
        M[:3, 3] = [0.1 * np.cos(c * 2.1), 0.02 * c, 0.1 * np.sin(c * 2.1)]
        M_c.append(M)
    ex = FE.mdBRIEFextractorOct(1000, 1.2, 8, 25, 0, 0, 32, 20, False, 2, True, True, 32, ctx=G.ctx())
    out = []
    for f in range(2):
        rig = FE.cMultiCamSys_(models, M_c, small_motion(0.2 * f, -0.3 * f, 0.1 * f, [0.01 * f, 0.0, 0.02 * f]))
        imgs = G.synth.synth_multiframe(f, cams)
        out.append(FE.cMultiFrame(imgs, 0.04 * f, [ex] * 3, None, rig, f))
    return cams, out


def oview(G, F, masks):
    return G.O.frame_view(F.mvKeys, F.all_descriptors(), F.all_masks() if masks else None, F.keypoint_to_cam, F.mnMaxX, F.mnMaxY)


def world_points(F, rng, idx):
    """map points 'seen' by features idx of frame F: bearing ray x depth in the camera frame, moved to the world by MtMc[c]"""
    pts = {}
    for i in idx:
        c = int(F.keypoint_to_cam[i])
        ray = F.mvKeysRays[i]
        pc = np.append(ray * rng.uniform(1.5, 6.0), 1.0)
        pts[int(i)] = (F.camSystem.MtMc[c] @ pc)[:3]
    return pts


def test_world_to_cam_matches_oracle(G, FE, frames):
    cams, fr = frames
    F = fr[1]
    rng = np.random.default_rng(5)
    n = 4000
    pts = rng.normal(0, 3.0, (n, 3))
    pts[:5] = [[0, 0, 1], [0, 0, -1], [1e-300, 0, 1], [0, 0, 0], [100, 100, 0.001]]   # on-axis (norm == 0 branch), origin, grazing
    pc = rng.integers(0, 3, n).astype(np.int32)
    masks = [G.synth.mirror_mask(c) for c in cams]
    uv, fl = F.camSystem.world_to_cam(pts, pc, G.ctx())
    euv, efl = G.O.world_to_cam(np.stack(F.camSystem.MtMc_inv), cams, masks, pts, pc)
    # the omni projection goes through atan(): device libm and glibc may differ in the last place, everything else is exact
    fin = np.isfinite(euv).all(axis=1)
    assert np.array_equal(np.isfinite(uv).all(axis=1), fin)
    assert np.allclose(uv[fin], euv[fin], rtol=0, atol=1e-9), np.abs(uv[fin] - euv[fin]).max()
    assert (uv[fin] == euv[fin]).mean() > 0.95
    assert np.array_equal(fl[fin], efl[fin])
    assert 0.05 < (fl & 1).mean() < 0.9 and 0.2 < ((fl >> 1) & 1).mean() < 0.8
    # no mask images: bounds test only
    models = [FE.cCamModelGeneral_.from_dict(c, None) for c in cams]
    rig2 = FE.cMultiCamSys_(models, F.camSystem.M_c, F.camSystem.M_t)
    uv2, fl2 = rig2.world_to_cam(pts, pc, G.ctx())
    _, efl2 = G.O.world_to_cam(np.stack(rig2.MtMc_inv), cams, None, pts, pc)
    assert np.array_equal(uv2[fin], uv[fin]) and np.array_equal(fl2[fin], efl2[fin]) and (fl2 & 1).sum() > (fl & 1).sum()
    assert np.array_equal(rig2.WorldToCamHom_fast(int(pc[7]), pts[7]), uv[7])


@pytest.mark.parametrize("masks,window,minlvl,maxlvl", [(True, 60, 0, 2**31 - 1), (True, 50, 4, 2**31 - 1), (False, 60, 0, 5), (False, 200, 2, 6)])
def test_window_search(G, FE, frames, masks, window, minlvl, maxlvl):
    _, fr = frames
    Fa, Fb = fr
    rng = np.random.default_rng(window + masks)
    Fa.mvpMapPoints = [MP(i, bad=rng.random() < 0.05) if rng.random() < 0.7 else None for i in range(Fa.totalN)]
    has = np.array([m is not None and not m.bad for m in Fa.mvpMapPoints], np.uint8)
    m = FE.cORBmatcher(0.8, False, 32, masks, ctx=G.ctx())
    n, out = m.WindowSearch(Fa, Fb, window, minlvl, maxlvl)
    v1, k1 = oview(G, Fa, masks)
    v2, k2 = oview(G, Fb, masks)
    en, e21 = G.O.window_search(v1, has, v2, window, minlvl, maxlvl if maxlvl < 2**31 - 1 else -1, 0.8, 32, masks)
    assert n == en and np.array_equal(m.last_matches21, e21), (n, en)
    assert [None if o is None else o.i for o in out] == [None if j < 0 else int(j) for j in e21]
    assert n > 100


@pytest.mark.parametrize("masks,window", [(True, 50), (False, 100), (True, 10)])
def test_search_for_initialization(G, FE, frames, masks, window):
    _, fr = frames
    Fa, Fb = fr
    prev = np.stack([Fa.mvKeys["x"], Fa.mvKeys["y"]], axis=1).astype(np.float64)
    prev[:7] = [[-100, 5], [5, -100], [3000, 10], [10, 3000], [0, 0], [753.9, 479.9], [377, 240]]
    m = FE.cORBmatcher(0.9, False, 32, masks, ctx=G.ctx())
    got_prev = prev.copy()
    n, m12 = m.SearchForInitialization(Fa, Fb, got_prev, window)
    v1, k1 = oview(G, Fa, masks)
    v2, k2 = oview(G, Fb, masks)
    en, e12, eprev = G.O.search_for_initialization(v1, v2, prev, window, 0.9, 32, masks)
    assert n == en and np.array_equal(m12, e12), (n, en, int((m12 != e12).sum()))
    assert np.array_equal(got_prev, eprev)
    assert n > (100 if window >= 50 else 20)


def test_search_for_initialization_steals(G, FE, frames):
    """every F1 feature probes the same spot: later, closer descriptors must steal the match (vnMatches21 / vMatchedDistance, :674-681)"""
    _, fr = frames
    Fa, Fb = fr
    target = int(np.argmax(Fb.mvKeys["octave"] == 0))
    prev = np.tile([[float(Fb.mvKeys[target]["x"]), float(Fb.mvKeys[target]["y"])]], (Fa.totalN, 1))
    # all F1 features claim to be in the target's camera so that every one of them opens the same window
    keep_cam = Fa.keypoint_to_cam.copy()
    Fa.keypoint_to_cam = np.full(Fa.totalN, Fb.keypoint_to_cam[target], np.int32)
    try:
        for masks in (False, True):
            m = FE.cORBmatcher(0.95, False, 32, masks, ctx=G.ctx())
            p = prev.copy()
            n, m12 = m.SearchForInitialization(Fa, Fb, p, 30)
            v1, _k1 = oview(G, Fa, masks)
            v2, _k2 = oview(G, Fb, masks)
            en, e12, ep = G.O.search_for_initialization(v1, v2, prev, 30, 0.95, 32, masks)
            assert n == en and np.array_equal(m12, e12) and np.array_equal(p, ep)
            assert n >= 1
    finally:
        Fa.keypoint_to_cam = keep_cam


@pytest.mark.parametrize("masks,th", [(True, 50.0), (False, 15.0), (True, 7.0)])
def test_search_by_projection_current_last(G, FE, frames, masks, th):
    cams, fr = frames
    Last, Cur = fr
    rng = np.random.default_rng(int(th) + masks)
    idx = rng.permutation(Last.totalN)[:int(0.8 * Last.totalN)]
    pts = world_points(Last, rng, idx)
    Last.mvpMapPoints = [MP(i, pts[i], bad=rng.random() < 0.04) if i in pts else None for i in range(Last.totalN)]
    Last.mvbOutlier = [bool(rng.random() < 0.1) for _ in range(Last.totalN)]
    Cur.mvpMapPoints = [MP(-1) if rng.random() < 0.1 else None for _ in range(Cur.totalN)]
    pre = np.array([mp is not None for mp in Cur.mvpMapPoints], np.uint8)
    try:
        m = FE.cORBmatcher(0.8, False, 32, masks, ctx=G.ctx())
        n = m.SearchByProjection(Cur, Last, th)
        # oracle: project every feature's map point (zeros where there is none)
        P = np.zeros((Last.totalN, 3))
        for i, p in pts.items():
            P[i] = p
        euv, efl = G.O.world_to_cam(np.stack(Cur.camSystem.MtMc_inv), cams, [G.synth.mirror_mask(c) for c in cams], P, Last.keypoint_to_cam)
        lastMP = np.array([mp is not None and not mp.bad for mp in Last.mvpMapPoints], np.uint8)
        vc, _kc = oview(G, Cur, masks)
        vl, _kl = oview(G, Last, masks)
        en, ecur, eas = G.O.search_by_projection_last(vc, pre, vl, lastMP, np.array(Last.mvbOutlier, np.uint8), euv, efl & 1, Cur.mvScaleFactors, th, 32, masks)
        got = np.array([mp.i if (mp is not None and not pre[j]) else -1 for j, mp in enumerate(Cur.mvpMapPoints)], np.int32)
        assert n == en and np.array_equal(got, ecur), (n, en, int((got != ecur).sum()))
        assert np.array_equal(np.array([mp is not None for mp in Cur.mvpMapPoints], np.uint8), eas)
        assert n > 50
    finally:
        Last.mvbOutlier = [False] * Last.totalN
        Last.mvpMapPoints = [None] * Last.totalN
        Cur.mvpMapPoints = [None] * Cur.totalN


@pytest.mark.parametrize("masks,window", [(True, 40), (False, 25)])
def test_search_by_projection_two_frames(G, FE, frames, masks, window):
    cams, fr = frames
    F1, F2 = fr
    rng = np.random.default_rng(window)
    idx = rng.permutation(F1.totalN)[:int(0.7 * F1.totalN)]
    pts = world_points(F1, rng, idx)
    mps = {i: MP(i, pts[i], bad=rng.random() < 0.04) for i in pts}
    F1.mvpMapPoints = [mps.get(i) for i in range(F1.totalN)]
    dup = [i for i in range(F1.totalN) if F1.mvpMapPoints[i] is None][:40]        # the same map point observed by a second feature
    for k, i in enumerate(dup):
        F1.mvpMapPoints[i] = mps[int(idx[k])]
    F2.mvpMapPoints = [None] * F2.totalN
    for k, j in enumerate(rng.permutation(F2.totalN)[:200]):                       # F2 already holds some points, a few of them F1's
        F2.mvpMapPoints[int(j)] = mps[int(idx[-1 - k])] if k < 60 else MP(100000 + k, np.zeros(3))
    pre = [mp for mp in F2.mvpMapPoints]
    try:
        m = FE.cORBmatcher(0.8, False, 32, masks, ctx=G.ctx())
        res = []
        n = m.SearchByProjection(F1, F2, window, res)
        mp1 = np.array([-1 if mp is None else mp.i for mp in F1.mvpMapPoints], np.int32)
        bad1 = np.array([mp is not None and mp.bad for mp in F1.mvpMapPoints], np.uint8)
        mp2 = np.array([-1 if mp is None else mp.i for mp in pre], np.int32)
        P = np.zeros((F1.totalN, 3, 3))
        for i in range(F1.totalN):
            if F1.mvpMapPoints[i] is not None:
                P[i, :] = F1.mvpMapPoints[i].pos
        euv, efl = G.O.world_to_cam(np.stack(F2.camSystem.MtMc_inv), cams, [G.synth.mirror_mask(c) for c in cams], P.reshape(-1, 3),
                                    np.tile(np.arange(3, dtype=np.int32), F1.totalN))
        v1, _k1 = oview(G, F1, masks)
        v2, _k2 = oview(G, F2, masks)
        en, e21 = G.O.search_by_projection_frames(v1, mp1, bad1, v2, mp2, euv, efl & 1, window, 0.8, 32, masks)
        got = np.array([res[j].i if (res[j] is not None and pre[j] is None) else -1 for j in range(F2.totalN)], np.int32)
        exp = np.array([mp1[i] if i >= 0 else -1 for i in e21], np.int32)
        assert n == en and np.array_equal(got, exp), (n, en, int((got != exp).sum()))
        assert all(res[j] is pre[j] for j in range(F2.totalN) if pre[j] is not None)
        assert n > 30
    finally:
        F1.mvpMapPoints = [None] * F1.totalN
        F2.mvpMapPoints = [None] * F2.totalN


def test_window_match_device_pointers_and_errors(G, FE, frames):
    """mcs_window_match with MCS_MEM_DEVICE buffers equals the host-kind call; argument validation fails loudly."""
    mcs = G.mcs
    cap = importlib.import_module("multicol-slam_amd._capi")
    _, fr = frames
    Fa, Fb = fr
    n = Fa.totalN
    lv = Fa.mvKeys["octave"].astype(np.int32)
    host = dict(x=Fa.mvKeys["x"].astype(np.float64), y=Fa.mvKeys["y"].astype(np.float64), r=np.full(n, 45.0), lo=lv - 1, hi=lv + 1,
                cam=Fa.keypoint_to_cam.astype(np.int32), d=Fa.all_descriptors(), m=Fa.all_masks())
    fh = dict(keys=np.ascontiguousarray(Fb.mvKeys), d=Fb.all_descriptors(), m=Fb.all_masks(), cam=Fb.keypoint_to_cam.astype(np.int32),
              asg=np.zeros(Fb.totalN, np.uint8), w=np.array(Fb.mnMaxX, np.int32), h=np.array(Fb.mnMaxY, np.int32), sc=np.array(Fb.mvScaleFactors))
    dp = {k: G.DevBuf(v) for k, v in host.items()}
    df = {k: G.DevBuf(v) for k, v in fh.items()}
    vp = lambda b: b.ptr
    for rule in (cap.WINDOW_RATIO, cap.WINDOW_BEST, cap.WINDOW_INITIALIZE):
        pr = cap.WindowProbes(vp(dp["x"]), vp(dp["y"]), vp(dp["r"]), vp(dp["lo"]), vp(dp["hi"]), vp(dp["cam"]), vp(dp["d"]), vp(dp["m"]), n, 32)
        df["asg"].zero()
        fv = cap.FrameView(vp(df["keys"]), vp(df["d"]), vp(df["m"]), vp(df["cam"]), vp(df["asg"]), Fb.totalN, 32, 3, vp(df["w"]), vp(df["h"]), vp(df["sc"]), 8)
        dmatch = G.DevBuf(np.full(n, -7, np.int32))
        dn = G.DevBuf(np.zeros(1, np.int32))
        mcs.check(mcs.lib().mcs_window_match(G.ctx().h, C.byref(pr), C.byref(fv), rule, 0.8, 32, mcs.MEM_DEVICE, vp(dmatch), vp(dn)))
        m = FE.cORBmatcher(0.8, False, 32, True, ctx=G.ctx())
        asg = np.zeros(Fb.totalN, np.uint8)
        hm, hn = m._window_match(rule, host["x"], host["y"], host["r"], host["lo"], host["hi"], host["cam"], np.arange(n), Fa, Fb,
                                 None if rule == cap.WINDOW_INITIALIZE else asg)
        assert hn == int(dn.read()[0]) and np.array_equal(hm, dmatch.read()) and hn > 100
        if rule != cap.WINDOW_INITIALIZE:
            assert np.array_equal(asg, df["asg"].read()) and asg.sum() == hn
    # loud failures: masks on one side only, unknown rule
    pr = cap.WindowProbes(vp(dp["x"]), vp(dp["y"]), vp(dp["r"]), vp(dp["lo"]), vp(dp["hi"]), vp(dp["cam"]), vp(dp["d"]), None, n, 32)
    fv = cap.FrameView(vp(df["keys"]), vp(df["d"]), vp(df["m"]), vp(df["cam"]), vp(df["asg"]), Fb.totalN, 32, 3, vp(df["w"]), vp(df["h"]), vp(df["sc"]), 8)
    dmatch = G.DevBuf(np.zeros(n, np.int32))
    dn = G.DevBuf(np.zeros(1, np.int32))
    for bad_rule, p in ((1, pr), (9, None)):
        with pytest.raises(mcs.McsError):
            q = p or cap.WindowProbes(vp(dp["x"]), vp(dp["y"]), vp(dp["r"]), vp(dp["lo"]), vp(dp["hi"]), vp(dp["cam"]), vp(dp["d"]), vp(dp["m"]), n, 32)
            mcs.check(mcs.lib().mcs_window_match(G.ctx().h, C.byref(q), C.byref(fv), bad_rule, 0.8, 32, mcs.MEM_DEVICE, vp(dmatch), vp(dn)))


def test_back_to_back_device_calls_share_the_scratch(G, FE, frames):
    """Device-kind calls only enqueue: three window searches in a row (different radii and rules) reuse the context's persistent scratch buffer while the
    earlier ones may still be running — stream order must keep them apart.  Results are read after all three and compared with host-kind calls."""
    mcs = G.mcs
    cap = importlib.import_module("multicol-slam_amd._capi")
    _, fr = frames
    Fa, Fb = fr
    n = Fa.totalN
    lv = Fa.mvKeys["octave"].astype(np.int32)
    base = dict(x=Fa.mvKeys["x"].astype(np.float64), y=Fa.mvKeys["y"].astype(np.float64), lo=lv - 1, hi=lv + 1, cam=Fa.keypoint_to_cam.astype(np.int32),
                d=Fa.all_descriptors(), m=Fa.all_masks())
    fh = dict(keys=np.ascontiguousarray(Fb.mvKeys), d=Fb.all_descriptors(), m=Fb.all_masks(), cam=Fb.keypoint_to_cam.astype(np.int32),
              w=np.array(Fb.mnMaxX, np.int32), h=np.array(Fb.mnMaxY, np.int32), sc=np.array(Fb.mvScaleFactors))
    dp = {k: G.DevBuf(v) for k, v in base.items()}
    df = {k: G.DevBuf(v) for k, v in fh.items()}
    vp = lambda b: b.ptr
    jobs = [(cap.WINDOW_RATIO, 30.0), (cap.WINDOW_BEST, 120.0), (cap.WINDOW_RATIO, 8.0)]
    outs = []
    for rule, rad in jobs:
        r = G.DevBuf(np.full(n, rad)); asg = G.DevBuf(np.zeros(Fb.totalN, np.uint8)); dm = G.DevBuf(np.full(n, -7, np.int32)); dn = G.DevBuf(np.zeros(1, np.int32))
        pr = cap.WindowProbes(vp(dp["x"]), vp(dp["y"]), vp(r), vp(dp["lo"]), vp(dp["hi"]), vp(dp["cam"]), vp(dp["d"]), vp(dp["m"]), n, 32)
        fv = cap.FrameView(vp(df["keys"]), vp(df["d"]), vp(df["m"]), vp(df["cam"]), vp(asg), Fb.totalN, 32, 3, vp(df["w"]), vp(df["h"]), vp(df["sc"]), 8)
        mcs.check(mcs.lib().mcs_window_match(G.ctx().h, C.byref(pr), C.byref(fv), rule, 0.8, 32, mcs.MEM_DEVICE, vp(dm), vp(dn)))
        outs.append((r, asg, dm, dn))
    G.ctx().synchronize()
    m = FE.cORBmatcher(0.8, False, 32, True, ctx=G.ctx())
    for (rule, rad), (r, asg, dm, dn) in zip(jobs, outs):
        hasg = np.zeros(Fb.totalN, np.uint8)
        hm, hn = m._window_match(rule, base["x"], base["y"], np.full(n, rad), base["lo"], base["hi"], base["cam"], np.arange(n), Fa, Fb, hasg)
        assert hn == int(dn.read()[0]) and np.array_equal(hm, dm.read()) and np.array_equal(hasg, asg.read()), (rule, rad)
    assert int(outs[1][3].read()[0]) > 100


def test_check_orientation_filters(G, FE, frames):
    """cORBmatcher(checkOri = True): the rotation-consistency pass (mcs_rotation_consistency, four bin-arithmetic variants) on the GPU vs the oracle's
    searches with the flag (embedded histograms) resp. search + orc_rotation_consistency."""
    import ctypes as C
    cams, fr = frames
    Fa, Fb = fr
    rng = np.random.default_rng(77)
    L = G.O.lib()
    L.orc_rotation_consistency.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]

    def ofilter(variant, a_slot, a_partner, match, swapped):
        a, b = np.ascontiguousarray(a_slot, np.float32), np.ascontiguousarray(a_partner, np.float32)
        m = np.ascontiguousarray(match, np.int32).copy()
        rem = L.orc_rotation_consistency(variant, G.O.ptr(a), G.O.ptr(b), None, G.O.ptr(m), len(m), swapped)
        return rem, m

    for masks in (True, False):
        m = FE.cORBmatcher(0.9, True, 32, masks, ctx=G.ctx())
        v1, _k1 = oview(G, Fa, masks)
        v2, _k2 = oview(G, Fb, masks)
        # WindowSearch (variant 1)
        Fa.mvpMapPoints = [MP(i) if rng.random() < 0.8 else None for i in range(Fa.totalN)]
        has = np.array([mp is not None for mp in Fa.mvpMapPoints], np.uint8)
        n, out = m.WindowSearch(Fa, Fb, 60)
        en, e21 = G.O.window_search(v1, has, v2, 60, 0, -1, 0.9, 32, masks, checkOri=1)
        un, _u = G.O.window_search(v1, has, v2, 60, 0, -1, 0.9, 32, masks, checkOri=0)
        assert n == en and np.array_equal(m.last_matches21, e21) and un > en
        # SearchForInitialization (variant 2, histogram over every acceptance)
        prev = np.stack([Fa.mvKeys["x"], Fa.mvKeys["y"]], axis=1).astype(np.float64)
        p = prev.copy()
        n, m12 = m.SearchForInitialization(Fa, Fb, p, 100)
        en, e12, ep = G.O.search_for_initialization(v1, v2, prev, 100, 0.9, 32, masks, checkOri=1)
        assert n == en and np.array_equal(m12, e12) and np.array_equal(p, ep)
        # SearchByBoW(KF, F) brute force (variant 0) and SearchForTriangulationRaw (variant 3)
        kf = FE.cMultiKeyFrame(Fa)
        n, outF = m.SearchByBoW(kf, Fb)
        en, eF = G.O.search_kf_f(Fa.all_descriptors(), Fa.all_masks(), has, Fb.all_descriptors(), Fb.all_masks(), masks, 0.9)
        rem, eF2 = ofilter(0, Fb.mvKeys["angle"], Fa.mvKeys["angle"], eF, 1)
        assert n == en - rem and [(-1 if o is None else o.i) for o in outF] == eF2.tolist() and rem > 0
        Fa.mvpMapPoints = [None] * Fa.totalN
    # SearchByProjection(Cur, Last) (variant 0)
    Last, Cur = fr
    idx = rng.permutation(Last.totalN)[:int(0.8 * Last.totalN)]
    pts = world_points(Last, rng, idx)
    Last.mvpMapPoints = [MP(i, pts[i]) if i in pts else None for i in range(Last.totalN)]
    try:
        m = FE.cORBmatcher(0.8, True, 32, True, ctx=G.ctx())
        n = m.SearchByProjection(Cur, Last, 15.0)
        P = np.zeros((Last.totalN, 3))
        for i, pp in pts.items():
            P[i] = pp
        euv, efl = G.O.world_to_cam(np.stack(Cur.camSystem.MtMc_inv), cams, [G.synth.mirror_mask(c) for c in cams], P, Last.keypoint_to_cam)
        lastMP = np.array([mp is not None for mp in Last.mvpMapPoints], np.uint8)
        vc, _kc = oview(G, Cur, True)
        vl, _kl = oview(G, Last, True)
        en, ecur, _ = G.O.search_by_projection_last(vc, np.zeros(Cur.totalN, np.uint8), vl, lastMP, np.zeros(Last.totalN, np.uint8), euv, efl & 1, Cur.mvScaleFactors,
                                                   15.0, 32, True, checkOri=1)
        got = np.array([mp.i if mp is not None else -1 for mp in Cur.mvpMapPoints], np.int32)
        assert n == en and np.array_equal(got, ecur)
    finally:
        Last.mvpMapPoints = [None] * Last.totalN
        Cur.mvpMapPoints = [None] * Cur.totalN


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_rotation_consistency_kernel_random(G, variant):
    """mcs_rotation_consistency on random matches / angles (all four bin arithmetics, both directions, with and without an acceptance record,
    keypoint-strided and plain float angle arrays) vs orc_rotation_consistency"""
    import ctypes as C
    L = G.O.lib()
    L.orc_rotation_consistency.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    rng = np.random.default_rng(variant)
    for trial in range(6):
        n, npart = int(rng.integers(1, 4000)), int(rng.integers(1, 4000))
        peak = rng.uniform(0, 360)
        a = np.where(rng.random(n) < 0.6, peak + rng.normal(0, 8, n), rng.uniform(0, 360, n)).astype(np.float32) % np.float32(360)
        b = rng.uniform(0, 360, npart).astype(np.float32)
        b[rng.integers(0, npart, npart // 3)] = np.float32(0.0)
        match = np.where(rng.random(n) < 0.7, rng.integers(0, npart, n), -1).astype(np.int32)
        accepted = None
        if trial % 2:
            accepted = match.copy()
            stolen = rng.random(n) < 0.1
            accepted[stolen & (match < 0)] = rng.integers(0, npart, int((stolen & (match < 0)).sum()))
        for swapped in (0, 1):
            em = match.copy()
            erem = L.orc_rotation_consistency(variant, G.O.ptr(a), G.O.ptr(b), G.O.ptr(accepted), G.O.ptr(em), n, swapped)
            gm, rem = match.copy(), np.zeros(1, np.int32)
            if trial % 3 == 0:   # angles inside keypoint records
                ka, kb = np.zeros(n, G.O.KP_DTYPE), np.zeros(npart, G.O.KP_DTYPE)
                ka["angle"], kb["angle"] = a, b
                pa, pb, st = C.c_void_p(ka.ctypes.data + 12), C.c_void_p(kb.ctypes.data + 12), 28
            else:
                pa, pb, st = C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), 4
            G.mcs.check(G.mcs.lib().mcs_rotation_consistency(G.ctx().h, variant, pa, st, pb, st, G.O.ptr(accepted), G.O.ptr(gm), n, npart, swapped, G.mcs.MEM_HOST,
                                                             G.O.ptr(rem)))
            assert int(rem[0]) == erem and np.array_equal(gm, em), (variant, trial, swapped)

"""-m gpu: the reference-named host classes (cMultiFrame / cORBmatcher) over the C ABI vs the CPU oracle.  Bit-exact."""
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


class MP:   # stand-in for cMapPoint: only isBad() matters to the searches
    def __init__(self, i, bad=False):
        self.i, self.bad = i, bad

    def isBad(self):
        return self.bad


@pytest.fixture(scope="module")
def frames(G, FE):
    cams = G.cams3()
    models = [FE.cCamModelGeneral_.from_dict(c, G.synth.mirror_mask(c)) for c in cams]
    rig = FE.cMultiCamSys_(models)
    ex = FE.mdBRIEFextractorOct(1000, 1.2, 8, 25, 0, 0, 32, 20, False, 2, True, True, 32, ctx=G.ctx())
    out = []
    for f in range(2):
        imgs = G.synth.synth_multiframe(f, cams)
        out.append((imgs, FE.cMultiFrame(imgs, 0.04 * f, [ex] * 3, None, rig, f)))
    return cams, rig, out


def test_multiframe_fields_match_oracle(G, FE, frames):
    cams, rig, fr = frames
    imgs, F = fr[0]
    off = 0
    for c in range(3):
        _, kps, d, dm, rays = G.oracle_extract(imgs[c], G.synth.mirror_mask(cams[c]), cams[c], do_dBrief=1, learnMasks=1)
        n = len(kps)
        assert F.N[c] == n
        assert G.first_diff(F.mvKeys[off:off + n], kps) is None
        assert G.first_diff(F.mDescriptors[c], d) is None and G.first_diff(F.mDescriptorMasks[c], dm) is None
        assert G.first_diff(F.mvKeysRays[off:off + n].view(np.uint64), rays.view(np.uint64)) is None
        assert (F.keypoint_to_cam[off:off + n] == c).all() and (F.cont_idx_to_local_cam_idx[off:off + n] == np.arange(n)).all()
        oc = G.O.make_ocam(cams[c])
        import ctypes as C
        gx, gy = C.c_int(), C.c_int()
        for i in range(0, n, 37):
            ok = G.O.lib().orc_pos_in_grid(C.byref(oc), float(kps["x"][i]), float(kps["y"][i]), C.byref(gx), C.byref(gy))
            ok2, px, py = F.PosInGrid(c, F.mvKeys[off + i])
            assert bool(ok) == ok2 and (not ok or (gx.value, gy.value) == (px, py))
            if ok:
                assert (off + i) in F.mGrids[c][px][py]
        off += n
    assert F.totalN == off and len(F.mvpMapPoints) == off
    assert F.mvScaleFactors[1] == float(np.float32(1.2)) and F.masksLearned and F.descDimension == 32


@pytest.mark.parametrize("K", [8, 2, 1])
def test_search_by_bow_kf_kf(G, FE, frames, K):
    _, _, fr = frames
    F1, F2 = fr[0][1], fr[1][1]
    rng = np.random.default_rng(K)
    F1.mvpMapPoints = [MP(i, bad=rng.random() < 0.05) if rng.random() < 0.8 else None for i in range(F1.totalN)]
    F2.mvpMapPoints = [MP(i, bad=rng.random() < 0.05) if rng.random() < 0.8 else None for i in range(F2.totalN)]
    K1, K2 = FE.cMultiKeyFrame(F1), FE.cMultiKeyFrame(F2)
    for ratio, masks in [(0.9, True), (0.6, True), (0.9, False)]:
        m = FE.cORBmatcher(ratio, False, 32, masks, ctx=G.ctx(), K=K)
        n, vp = m.SearchByBoW(K1, K2)
        v1 = np.array([FE._good(x) for x in K1.mvpMapPoints], np.uint8)
        v2 = np.array([FE._good(x) for x in K2.mvpMapPoints], np.uint8)
        en, e12 = G.O.search_kf_kf(K1._d, K1._m, v1, K2._d, K2._m, v2, masks, ratio)
        got = np.array([x.i if x is not None else -1 for x in vp])
        assert n == en and G.first_diff(got, e12) is None, (K, ratio, masks, m.last_fallbacks)
        assert n > 300 or not masks


@pytest.mark.parametrize("K", [8, 1])
def test_search_by_bow_kf_frame(G, FE, frames, K):
    _, _, fr = frames
    F1, F2 = fr[0][1], fr[1][1]
    F1.mvpMapPoints = [MP(i) if i % 5 else None for i in range(F1.totalN)]
    K1 = FE.cMultiKeyFrame(F1)
    for ratio, masks in [(0.9, True), (0.75, False)]:
        m = FE.cORBmatcher(ratio, False, 32, masks, ctx=G.ctx(), K=K)
        n, vp = m.SearchByBoW(K1, F2)
        v1 = np.array([x is not None for x in K1.mvpMapPoints], np.uint8)
        en, eF = G.O.search_kf_f(K1._d, K1._m, v1, F2.all_descriptors(), F2.all_masks(), masks, ratio)
        got = np.array([x.i if x is not None else -1 for x in vp])
        assert n == en and G.first_diff(got, eF) is None, (K, ratio, masks)


def test_greedy_with_many_duplicates_forces_rescans(G, FE):
    """repeated texture: many identical descriptors -> top-K lists saturate with already-taken rows -> exact rescans."""
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (12, 32)).astype(np.uint8)
    d2 = base[rng.integers(0, 12, 600)] ^ (rng.integers(0, 256, (600, 32)) & rng.integers(0, 256, (600, 32)) & rng.integers(0, 256, (600, 32)) &
                                         rng.integers(0, 256, (600, 32)) & rng.integers(0, 256, (600, 32))).astype(np.uint8)
    d1 = base[rng.integers(0, 12, 500)]
    ones1, ones2 = np.full_like(d1, 255), np.full_like(d2, 255)
    v1, v2 = np.ones(500, np.uint8), np.ones(600, np.uint8)
    import ctypes as C
    mcs = G.mcs
    for K in (1, 4, 8):
        for ratio in (0.9, 1.1):
            q = mcs.DescSet(mcs.np_ptr(d1), None, mcs.np_ptr(v1), None, 500, 32)
            t = mcs.DescSet(mcs.np_ptr(d2), None, mcs.np_ptr(v2), None, 600, 32)
            m12, nm, fb = np.full(500, -1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.int32)
            mcs.check(mcs.lib().mcs_search_kf_kf(G.ctx().h, 1, C.byref(q), 0, C.byref(t), 0, 32, ratio, K, 0, mcs.np_ptr(m12), mcs.np_ptr(nm), mcs.np_ptr(fb)))
            en, e12 = G.O.search_kf_kf(d1, ones1, v1, d2, ones2, v2, False, ratio)
            assert nm[0] == en and G.first_diff(m12, e12) is None, (K, ratio)
            mF = np.full(600, -1, np.int32)
            mcs.check(mcs.lib().mcs_search_kf_f(G.ctx().h, 1, C.byref(q), 0, C.byref(t), 0, 32, ratio, K, 0, mcs.np_ptr(mF), mcs.np_ptr(nm), mcs.np_ptr(fb)))
            en, eF = G.O.search_kf_f(d1, ones1, v1, d2, ones2, False, ratio)
            assert nm[0] == en and G.first_diff(mF, eF) is None, (K, ratio)
    assert fb[0] >= 0


def test_search_for_triangulation_raw(G, FE, frames):
    _, rig, fr = frames
    F1, F2 = fr[0][1], fr[1][1]
    rng = np.random.default_rng(5)
    F1.mvpMapPoints = [MP(i) if rng.random() < 0.5 else None for i in range(F1.totalN)]
    F2.mvpMapPoints = [MP(i) if rng.random() < 0.5 else None for i in range(F2.totalN)]
    K1, K2 = FE.cMultiKeyFrame(F1), FE.cMultiKeyFrame(F2)
    Es = rng.normal(size=(3, 3, 3, 3))
    for i in range(3):   # same-camera pairs: a plausible essential matrix [t]x for a small sideways motion -> many rays pass
        t = np.array([0.05, 0.01 * i, 0.0])
        Es[i, i] = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    for masks, K in [(True, 8), (False, 16)]:
        m = FE.cORBmatcher(0.9, False, 32, masks, ctx=G.ctx(), K=K)
        n, k1, r1, k2, r2, pairs = m.SearchForTriangulationRaw(K1, K2, Es)
        hp1 = np.array([x is not None for x in K1.mvpMapPoints], np.uint8)
        hp2 = np.array([x is not None for x in K2.mvpMapPoints], np.uint8)
        en, e12 = G.O.search_triangulation(K1._d, K1._m, hp1, np.ascontiguousarray(K1.keypoint_to_cam), np.ascontiguousarray(K1.mvKeysRays),
                                           K2._d, K2._m, hp2, np.ascontiguousarray(K2.keypoint_to_cam), np.ascontiguousarray(K2.mvKeysRays),
                                           np.ascontiguousarray(Es.reshape(9, 9)), 3, masks)
        exp_pairs = [(i, int(j)) for i, j in enumerate(e12) if j >= 0]
        assert n == en and pairs == exp_pairs, (masks, K, n, en)
        assert len(k1) == n and len(r2) == n
    assert en > 20


def test_extractor_facade_modes(G, FE):
    cam = G.cams3()[2]
    model = FE.cCamModelGeneral_.from_dict(cam, G.synth.mirror_mask(cam))
    img = G.synth.synth_image(5, 2, cam)
    orb = FE.ORBextractor(1000, 1.2, 8, 1, 20, ctx=G.ctx())
    kps, d = orb(img, model.GetMirrorMask(0))
    _, ek, ed, _, _ = G.oracle_extract(img, G.synth.mirror_mask(cam), cam)
    assert G.first_diff(kps, ek) is None and G.first_diff(d, ed) is None
    assert orb.GetLevels() == 8 and orb.GetScaleFactor() == float(np.float32(1.2)) and not orb.GetMasksLearned() and orb.GetDescriptorSize() == 32
    # (usemdBRIEF=0, masks=1): the reference runs compute_mdBRIEF with all undistorted keypoints = (0,0)
    odd = FE.mdBRIEFextractorOct(400, 1.2, 8, 25, 0, 0, 32, 20, False, 2, False, True, 32, ctx=G.ctx())
    kps, d, dm = odd(img, model.GetMirrorMask(0), model)
    _, ek, ed, edm, _ = G.oracle_extract(img, G.synth.mirror_mask(cam), cam, nfeatures=400, do_dBrief=0, learnMasks=1)
    assert G.first_diff(kps, ek) is None and G.first_diff(d, ed) is None and G.first_diff(dm, edm) is None
    k0, d0, m0 = odd(np.zeros((0, 0), np.uint8), None, model)
    assert len(k0) == 0 and d0 is None


class TrackedMP(MP):
    """cMapPoint stand-in carrying what cMultiFrame::isInFrustum leaves on it (src/cMultiFrame.cpp:218-270)."""

    def __init__(self, i, desc, mask, nr_cams):
        super().__init__(i)
        self.desc, self.mask = desc, mask
        self.mbTrackInView = [False] * nr_cams
        self.mTrackProjX, self.mTrackProjY = [0.0] * nr_cams, [0.0] * nr_cams
        self.mnTrackScaleLevel, self.mTrackViewCos = [0] * nr_cams, [1.0] * nr_cams

    def GetDescriptor(self):
        return self.desc

    def GetDescriptorMask(self):
        return self.mask


@pytest.mark.parametrize("th,masks", [(1.0, True), (3.0, True), (15.0, False)])
def test_search_by_projection_window_matcher(G, FE, frames, th, masks):
    """SearchByProjection(F, mapPoints, th) incl. GetFeaturesInArea / PosInGrid, vs the oracle."""
    _, rig, fr = frames
    Fa, Fb = fr[0][1], fr[1][1]
    rng = np.random.default_rng(int(th * 10) + masks)
    # map points = features of frame a (descriptor + where the motion model would project them in frame b: shifted by (3,1) + noise)
    mps = []
    da, ma = Fa.all_descriptors(), Fa.all_masks()
    for i in rng.permutation(Fa.totalN)[:1500]:
        kp = Fa.mvKeys[i]
        cam = int(Fa.keypoint_to_cam[i])
        mp = TrackedMP(int(i), da[i], ma[i], 3)
        mp.bad = rng.random() < 0.03
        for c in ([cam] if rng.random() < 0.9 else [cam, (cam + 1) % 3]):   # a few points are "in view" of two cameras
            mp.mbTrackInView[c] = True
            mp.mTrackProjX[c] = float(kp["x"]) + 3.0 + rng.normal(0, 1.5)
            mp.mTrackProjY[c] = float(kp["y"]) + 1.0 + rng.normal(0, 1.5)
            mp.mnTrackScaleLevel[c] = int(np.clip(kp["octave"] + rng.integers(-1, 2), 0, 7))
            mp.mTrackViewCos[c] = float(rng.choice([0.9995, 0.99, 0.5]))
        mps.append(mp)
    # a few projections far outside the image / at the borders (empty windows, clamped cell ranges)
    for k, (x, y) in enumerate([(-500.0, 10.0), (2000.0, 100.0), (5.0, 5.0), (750.0, 478.0), (377.0, -90.0)]):
        mp = TrackedMP(10000 + k, da[k], ma[k], 3)
        mp.mbTrackInView[0] = True
        mp.mTrackProjX[0], mp.mTrackProjY[0], mp.mnTrackScaleLevel[0], mp.mTrackViewCos[0] = x, y, k % 8, 0.999
        mps.append(mp)
    Fb.mvpMapPoints = [MP(-1) if rng.random() < 0.1 else None for _ in range(Fb.totalN)]   # some features already have a point
    pre = [m is not None for m in Fb.mvpMapPoints]
    m = FE.cORBmatcher(0.8, False, 32, masks, ctx=G.ctx())
    # oracle on the same flat inputs (built in the reference's visiting order)
    px, py, vc, lv, pc, pd, pm, owner = [], [], [], [], [], [], [], []
    for mp in mps:
        if mp.isBad():
            continue
        for c in range(3):
            if mp.mbTrackInView[c]:
                px.append(mp.mTrackProjX[c]); py.append(mp.mTrackProjY[c]); vc.append(mp.mTrackViewCos[c]); lv.append(mp.mnTrackScaleLevel[c])
                pc.append(c); pd.append(mp.desc); pm.append(mp.mask); owner.append(mp.i)
    arr = lambda v, t: np.ascontiguousarray(v, t)
    assigned = np.array(pre, np.uint8)
    en, ematch = G.O.search_by_projection(arr(px, np.float64), arr(py, np.float64), arr(vc, np.float64), arr(lv, np.int32), arr(pc, np.int32),
                                          arr(np.stack(pd), np.uint8), arr(np.stack(pm), np.uint8), arr(Fb.mvKeys, Fb.mvKeys.dtype),
                                          arr(Fb.all_descriptors(), np.uint8), arr(Fb.all_masks(), np.uint8), arr(Fb.keypoint_to_cam, np.int32), assigned,
                                          arr(Fb.mnMaxX, np.int32), arr(Fb.mnMaxY, np.int32), arr(Fb.mvScaleFactors, np.float64), th, 0.8, masks)
    n = m.SearchByProjection(Fb, mps, th)
    exp = {int(j): owner[p] for p, j in enumerate(ematch) if j >= 0}
    got = {j: mp.i for j, mp in enumerate(Fb.mvpMapPoints) if mp is not None and not pre[j]}
    assert n == en and got == exp, (th, masks, n, en)
    assert n > (300 if th <= 3 else 100)

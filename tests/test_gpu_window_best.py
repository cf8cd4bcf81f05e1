"""-m gpu: mcs_window_best — the search loop of Fuse / SearchBySim3 / SearchForTriangulationBetweenCameras / relocalisation SearchByProjection
(src/cORBmatcher.cpp:1158-1988, 2120-2392) vs the oracle.  Bit-exact indices and distances."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def two_frames():
    import gpu_common as G
    FE = importlib.import_module("multicol-slam_amd.frontend")
    cams = G.cams3()
    models = [FE.cCamModelGeneral_.from_dict(c, G.synth.mirror_mask(c)) for c in cams]
    rig = FE.cMultiCamSys_(models)
    ex = FE.mdBRIEFextractorOct(1000, 1.2, 8, 25, 0, 0, 32, 20, False, 2, True, True, 32, ctx=G.ctx())
    return G, FE, [FE.cMultiFrame(G.synth.synth_multiframe(f, cams), 0.04 * f, [ex] * 3, None, rig, f) for f in range(2)]


@pytest.mark.parametrize("masks,skip,maxd,th", [(True, False, 32, 3.0), (False, False, 96, 6.0), (True, True, 48, 10.0), (False, True, 100, 40.0),
                                                (True, False, 0, 3.0)])
def test_window_best(two_frames, masks, skip, maxd, th):
    G, FE, (Fa, Fb) = two_frames
    rng = np.random.default_rng(maxd + skip)
    idx = rng.permutation(Fa.totalN)[:2000]
    k = Fa.mvKeys[idx]
    x = k["x"].astype(np.float64) + 2.0 + rng.normal(0, 1.0, len(idx))
    y = k["y"].astype(np.float64) + 1.0 + rng.normal(0, 1.0, len(idx))
    lvl = np.clip(k["octave"] + rng.integers(-1, 2, len(idx)), 0, 7).astype(np.int32)
    r = th * np.asarray(Fb.mvScaleFactors)[lvl]
    x[:4], y[:4] = [-300, 5000, 3, 377], [10, 10, 3, 2000]                  # empty / clamped windows
    cam = Fa.keypoint_to_cam[idx].astype(np.int32)
    cam[5:40] = (cam[5:40] + 1) % 3                                          # Fuse probes every camera, not only the observing one
    d, m = Fa.all_descriptors()[idx], Fa.all_masks()[idx]
    assigned = (rng.random(Fb.totalN) < 0.15).astype(np.uint8)
    matcher = FE.cORBmatcher(0.8, False, 32, masks, ctx=G.ctx())
    asg = assigned.copy()
    match, dist, n = matcher.BestInWindows(x, y, r, lvl - 1, lvl, cam, d, m, Fb, maxd, skip, asg)
    v, _keep = G.O.frame_view(Fb.mvKeys, Fb.all_descriptors(), Fb.all_masks() if masks else None, Fb.keypoint_to_cam, Fb.mnMaxX, Fb.mnMaxY)
    en, ematch, edist, easg = G.O.window_best(x, y, r, lvl - 1, lvl, cam, d, m if masks else None, v, assigned, maxd, skip, 32, masks)
    assert n == en and np.array_equal(match, ematch), (n, en, int((match != ematch).sum()))
    assert np.array_equal(dist, edist)
    if skip:
        assert np.array_equal(asg, easg) and asg.sum() == assigned.sum() + n
        assert not (assigned[match[match >= 0]]).any()
    else:
        assert np.array_equal(asg, assigned)
    assert n > (300 if maxd >= 32 else -1)


def test_device_kind_of_the_mapping_side_entry_points(two_frames):
    """MCS_MEM_DEVICE variants of mcs_world_to_cam, mcs_window_best, mcs_distinctive_descriptors and mcs_bow_transform equal their host-kind calls"""
    import ctypes as C
    import vocab_synth, tempfile, os
    G, FE, (Fa, Fb) = two_frames
    mcs, cap = G.mcs, importlib.import_module("multicol-slam_amd._capi")
    io = importlib.import_module("multicol-slam_amd.io")
    L = mcs.lib()
    rng = np.random.default_rng(0)
    n = Fa.totalN
    # --- world_to_cam (matrices / calibrations / masks are host values for either kind; points and outputs on the device)
    pts = rng.normal(0, 3, (n, 3))
    pc = Fa.keypoint_to_cam.astype(np.int32)
    uv, fl = Fa.camSystem.world_to_cam(pts, pc, G.ctx())
    nr = 3
    M = np.ascontiguousarray(np.stack(Fa.camSystem.MtMc_inv).reshape(nr, 16))
    ocs = (cap.Ocam * nr)(*[cm.ocam for cm in Fa.camSystem.cams])
    dmask = [G.DevBuf(cm.GetMirrorMask(0)) for cm in Fa.camSystem.cams]
    mp = (C.c_void_p * nr)(*[b.ptr.value for b in dmask])
    dpts, dpc, duv, dfl = G.DevBuf(pts), G.DevBuf(pc), G.DevBuf(np.zeros((n, 2))), G.DevBuf(np.zeros(n, np.uint8))
    mcs.check(L.mcs_world_to_cam(G.ctx().h, M.ctypes.data, ocs, nr, mp, dpts.ptr, dpc.ptr, n, mcs.MEM_DEVICE, duv.ptr, dfl.ptr))
    assert np.array_equal(duv.read(), uv) and np.array_equal(dfl.read(), fl)
    # --- window_best
    k = Fa.mvKeys
    x, y = k["x"].astype(np.float64) + 2.0, k["y"].astype(np.float64) + 1.0
    lvl = k["octave"].astype(np.int32)
    r = 6.0 * np.asarray(Fb.mvScaleFactors)[lvl]
    m = FE.cORBmatcher(0.8, False, 32, True, ctx=G.ctx())
    hm, hd, hn = m.BestInWindows(x, y, r, lvl - 1, lvl, pc, Fa.all_descriptors(), Fa.all_masks(), Fb, 40)
    bufs = [G.DevBuf(a) for a in (x, y, r, lvl - 1, lvl, pc, Fa.all_descriptors(), Fa.all_masks())]
    pr = cap.WindowProbes(*[b.ptr for b in bufs], n, 32)
    fb = [G.DevBuf(a) for a in (np.ascontiguousarray(Fb.mvKeys), Fb.all_descriptors(), Fb.all_masks(), Fb.keypoint_to_cam.astype(np.int32),
                                np.array(Fb.mnMaxX, np.int32), np.array(Fb.mnMaxY, np.int32), np.array(Fb.mvScaleFactors))]
    fv = cap.FrameView(fb[0].ptr, fb[1].ptr, fb[2].ptr, fb[3].ptr, None, Fb.totalN, 32, 3, fb[4].ptr, fb[5].ptr, fb[6].ptr, 8)
    dm, dd, dn = G.DevBuf(np.zeros(n, np.int32)), G.DevBuf(np.zeros(n, np.int32)), G.DevBuf(np.zeros(1, np.int32))
    mcs.check(L.mcs_window_best(G.ctx().h, C.byref(pr), C.byref(fv), 40, 0, 32, mcs.MEM_DEVICE, dm.ptr, dd.ptr, dn.ptr))
    assert np.array_equal(dm.read(), hm) and np.array_equal(dd.read(), hd) and int(dn.read()[0]) == hn and hn > 300
    # --- distinctive descriptors
    off = np.arange(0, n + 1, 5, dtype=np.int32)
    off[-1] = n
    best = FE.ComputeDistinctiveDescriptorsBatch([(Fa.all_descriptors()[a:b], Fa.all_masks()[a:b]) for a, b in zip(off[:-1], off[1:])], 32, G.ctx())
    doff, dbest = G.DevBuf(off), G.DevBuf(np.zeros(len(off) - 1, np.int32))
    mcs.check(L.mcs_distinctive_descriptors(G.ctx().h, bufs[6].ptr, bufs[7].ptr, 32, 32, doff.ptr, len(off) - 1, mcs.MEM_DEVICE, dbest.ptr))
    assert np.array_equal(dbest.read(), best)
    # --- bow transform
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "voc.yml")
        vocab_synth.write_vocabulary(p, k=9, L=5, seed=3)
        voc = FE.cORBVocabulary(io.load_vocabulary(p), ctx=G.ctx())
    leaf, nid = voc.descend(Fa.all_descriptors(), 4)
    dl, dn2 = G.DevBuf(np.zeros(n, np.int32)), G.DevBuf(np.zeros(n, np.int32))
    mcs.check(L.mcs_bow_transform(voc.h, bufs[6].ptr, n, 32, 4, mcs.MEM_DEVICE, dl.ptr, dn2.ptr))
    assert np.array_equal(dl.read(), leaf) and np.array_equal(dn2.read(), nid)

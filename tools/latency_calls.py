"""Latency of the matcher entry points at live-tracker sizes (one 3-camera multi-frame vs one keyframe, ~3000 features each, host buffers):
wall time per call (median of 10 after warm-up) and the kernel times the library's own events report (match = top-K lists, greedy = resolution)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mcs = importlib.import_module("multicol-slam_amd")
synth = importlib.import_module("multicol-slam_amd.synth")
FE = importlib.import_module("multicol-slam_amd.frontend")


class MP:
    def __init__(self, i):
        self.i = i

    def isBad(self):
        return False


ctx = mcs.Context(0)
cams = synth.lafida_cameras()
rig = FE.cMultiCamSys_([FE.cCamModelGeneral_.from_dict(c, synth.mirror_mask(c)) for c in cams])
ex = FE.mdBRIEFextractorOct(1000, 1.2, 8, 25, 0, 0, 32, 20, False, 2, True, True, 32, ctx=ctx)
F = [FE.cMultiFrame(synth.synth_multiframe(f, cams), 0.04 * f, [ex] * 3, None, rig, f) for f in range(2)]
rng = np.random.default_rng(5)


def run(name, fn, reps=12):
    ts, km, kg = [], [], []
    for i in range(reps):
        ctx.enable_timing(True)
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
        for lst, k in ((km, "match"), (kg, "greedy")):
            try:
                lst.append(ctx.kernel_ms(k))
            except mcs.McsError:
                lst.append(0.0)
        ctx.enable_timing(False)
    print("%-44s wall %7.3f ms   kernels: match %6.3f  greedy %6.3f" % (name, 1e3 * np.median(ts[2:]), np.median(km[2:]), np.median(kg[2:])))


for masks in (True, False):
    for K in (32, 8):
        m = FE.cORBmatcher(0.9, False, 32, masks, ctx=ctx, K=K)
        for f in F:
            f.mvpMapPoints = [MP(i) for i in range(f.totalN)]
        K1, K2 = FE.cMultiKeyFrame(F[0]), FE.cMultiKeyFrame(F[1])
        tag = "masks=%d K=%d " % (masks, K)
        run(tag + "SearchByBoW(KF,KF) all points", lambda: m.SearchByBoW(K1, K2))
        F[1].mFeatVec = None
        run(tag + "SearchByBoW(KF,F) brute force", lambda: m.SearchByBoW(K1, F[1]))
        for f in F:
            f.mvpMapPoints = [MP(i) if rng.random() < 0.5 else None for i in range(f.totalN)]
        K1, K2 = FE.cMultiKeyFrame(F[0]), FE.cMultiKeyFrame(F[1])
        Es = rng.normal(size=(3, 3, 3, 3))
        for i in range(3):
            t = np.array([0.05, 0.01 * i, 0.0])
            Es[i, i] = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        run(tag + "SearchForTriangulationRaw (half free)", lambda: m.SearchForTriangulationRaw(K1, K2, Es))
        print("      matches %d, exact rescans %d of %d queries" % (m.SearchForTriangulationRaw(K1, K2, Es)[0], m.last_fallbacks, sum(x is None for x in K1.mvpMapPoints)))


def run_w(name, fn, reps=10):
    ts, kc, kg = [], [], []
    for i in range(reps):
        ctx.enable_timing(True)
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
        for lst, k in ((kc, "win_candidates"), (kg, "win_greedy")):
            try:
                lst.append(ctx.kernel_ms(k))
            except mcs.McsError:
                lst.append(0.0)
        ctx.enable_timing(False)
    print("%-58s wall %7.3f ms   kernels: candidates %6.3f  greedy %6.3f" % (name, 1e3 * np.median(ts[2:]), np.median(kc[2:]), np.median(kg[2:])))


for masks in (True, False):
    m = FE.cORBmatcher(0.8, False, 32, masks, ctx=ctx)
    F[0].mvpMapPoints = [MP(i) if rng.random() < 0.7 else None for i in range(F[0].totalN)]
    F[1].mvpMapPoints = [None] * F[1].totalN
    tag = "masks=%d " % masks
    for w in (40, 60, 100, 200):
        run_w(tag + "WindowSearch(window %d), %d probes" % (w, sum(x is not None for x in F[0].mvpMapPoints)), lambda: m.WindowSearch(F[0], F[1], w, 0, 2**31 - 1))
    for w in (10, 50, 100):
        prev = np.stack([F[0].mvKeys["x"], F[0].mvKeys["y"]], axis=1).astype(np.float64)
        run_w(tag + "SearchForInitialization(window %d), %d probes" % (w, F[0].totalN), lambda: m.SearchForInitialization(F[0], F[1], prev.copy(), w))
sys.stdout.flush()
os._exit(0)   # skip interpreter teardown (extractors would be released after their context)

"""Latency of ONE three-camera multi-frame through the host-memory entry points (the shape a live tracker calls): extraction of 3 images, then
SearchByBoW(KF,KF) against the previous multi-frame.  Prints ms per call (median of 30 after warm-up)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mcs = importlib.import_module("multicol-slam_amd")
synth = importlib.import_module("multicol-slam_amd.synth")
FE = importlib.import_module("multicol-slam_amd.frontend")

cams = synth.lafida_cameras()
masks = [synth.mirror_mask(c) for c in cams]
oc = [mcs.make_ocam(c) for c in cams]
ctx = mcs.Context(0)
frames = [synth.synth_multiframe(f, cams) for f in range(4)]
for mode, kw in (("mdbrief", dict(do_dBrief=1, learnMasks=1)), ("orb", dict(do_dBrief=0, learnMasks=0))):
    ex = mcs.Extractor(ctx, 754, 480, max_batch=3, **kw)
    ts = []
    for i in range(40):
        t = time.perf_counter()
        res = ex.extract_host(frames[i % 4], masks, oc)
        ts.append(time.perf_counter() - t)
    ts1 = []
    for i in range(40):
        t = time.perf_counter()
        for c in range(3):
            ex.extract_host([frames[i % 4][c]], [masks[c]], [oc[c]])
        ts1.append(time.perf_counter() - t)
    a = [ex.extract_host(frames[f], masks, oc) for f in range(2)]
    d = [np.concatenate([r[1] for r in a[f]]) for f in range(2)]
    m = [np.concatenate([r[2] for r in a[f]]) for f in range(2)]
    tm = []
    for i in range(40):
        t = time.perf_counter()
        ctx.match_topk(d[0], d[1], 32, 64, m[0] if mode == "mdbrief" else None, m[1] if mode == "mdbrief" else None)
        tm.append(time.perf_counter() - t)
    print("%s: extract 3 images in one call %.2f ms, as 3 calls %.2f ms, top-32 lists %dx%d %.2f ms  (host buffers in and out, %d keypoints)"
          % (mode, 1e3 * np.median(ts[10:]), 1e3 * np.median(ts1[10:]), len(d[0]), len(d[1]), 1e3 * np.median(tm[10:]), len(d[0])))
    ex.close()

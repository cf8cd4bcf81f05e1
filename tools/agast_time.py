"""FAST against the four AGAST types on the bench stream's images (192 images of 754 x 480, mdBRIEF, 1000 features): the detector kernel's time from the library's
own HIP events (mcs_ctx_enable_timing), candidates and keypoints per image.  python tools/agast_time.py  (on the GPU box)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib

mcs = importlib.import_module("multicol-slam_amd")
synth = importlib.import_module("multicol-slam_amd.synth")

cams = synth.lafida_cameras()
frames = 16
imgs = [synth.synth_image(f, c, cams[c]) for c in range(3) for f in range(frames)] * 4
masks = [synth.mirror_mask(cams[c]) for c in range(3) for f in range(frames)] * 4
ocams = [mcs.make_ocam(cams[c]) for c in range(3) for f in range(frames)] * 4
ctx = mcs.Context(0)
out = {}
for name, kw in (("FAST_9_16", dict(fastAgastType=2)), ("AGAST_5_8", dict(useAgast=1, fastAgastType=0)), ("AGAST_7_12d", dict(useAgast=1, fastAgastType=1)),
                 ("AGAST_7_12s", dict(useAgast=1, fastAgastType=2)), ("OAST_9_16", dict(useAgast=1, fastAgastType=3))):
    ex = mcs.Extractor(ctx, 754, 480, max_batch=len(imgs), nfeatures=1000, fastThreshold=20, do_dBrief=1, learnMasks=1, **kw)
    ex.extract_host(imgs, masks, ocams)
    ctx.enable_timing(True)
    ms = []
    for _ in range(5):
        res = ex.extract_host(imgs, masks, ocams)
        ms.append({k: ctx.kernel_ms(k) for k in ("fast", "octree")})
    ctx.enable_timing(False)
    cand = sum(len(ex.tap_candidates(0, l)[0]) for l in range(8))
    out[name] = {"fast_ms": round(float(np.median([m["fast"] for m in ms])), 4), "octree_ms": round(float(np.median([m["octree"] for m in ms])), 4),
                 "candidates_image0": cand, "keypoints_image0": len(res[0][0]), "images": len(imgs)}
    ex.close()
print(json.dumps(out))

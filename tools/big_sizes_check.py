# one-off parity check at large image sizes / feature budgets on the GPU box (1920x1080 .. 3840x2160, up to 8000 features): extraction vs the oracle, bit for bit
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import gpu_common as G
O = G.O; mcs = G.mcs; synth = G.synth
for (W, H, nf, nl, sf, n) in [(1920, 1080, 2000, 8, 1.2, 3), (2048, 1536, 3000, 8, 1.2, 2), (1280, 800, 1000, 8, 1.2, 5), (3840, 2160, 4000, 8, 1.2, 1), (640, 480, 8000, 8, 1.2, 2)]:
    cam = synth.scaled_camera(synth.lafida_cameras()[0], W, H)
    kw = dict(nfeatures=nf, scaleFactor=sf, nlevels=nl, do_dBrief=1, learnMasks=1)
    try:
        ex = mcs.Extractor(G.ctx(), W, H, max_batch=n, **kw)
    except Exception as e:
        print(W, H, nf, "refused:", str(e)[:150]); continue
    rng = np.random.default_rng(W)
    imgs = [synth.synth_image(f, 0, cam) if f % 2 == 0 else rng.integers(0, 256, (H, W)).astype(np.uint8) for f in range(n)]
    m = synth.mirror_mask(cam)
    res = ex.extract_host(imgs, [m] * n, [mcs.make_ocam(cam)] * n)
    ok = True; tot = 0
    for i in range(n):
        oex = O.Extractor(**kw); oex.cap = max(oex.cap, ex.cap)
        kps, d, dm = oex(imgs[i], m, O.make_ocam(cam))
        gk, gd, gm, gr = res[i]
        e = (None if len(gk) == len(kps) else "count %d vs %d" % (len(gk), len(kps))) or G.first_diff(gk, kps) or G.first_diff(gd, d) or G.first_diff(gm, dm)
        tot += len(kps)
        if e: ok = False; print(W, H, "image", i, e)
    print(W, H, nf, "batch", n, "keypoints", tot, "OK" if ok else "MISMATCH")
    ex.close()

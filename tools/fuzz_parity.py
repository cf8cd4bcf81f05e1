#!/usr/bin/env python3
"""Randomised parity run on the GPU box (not a test: `python tools/fuzz_parity.py [seconds] [seed]`): random image sizes, pyramid shapes, feature budgets,
detector settings, descriptor modes, image statistics and masks through the C ABI (host-kind extraction of a small batch) against the oracle, bit for bit —
keypoints, descriptors, masks, rays — and the two frames of every case through mcs_search_kf_kf / mcs_search_kf_f against the oracle's sequential loops.
Prints one line per failing case with the arguments that reproduce it; exit code 1 if any case failed.  The oracle is the checker here, as in tests/."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_common as G   # noqa: E402

O = G.O
mcs, synth = G.mcs, G.synth
cap_mod = importlib.import_module("multicol-slam_amd._capi")
REFUSED = []
MODES = {"orb": dict(do_dBrief=0, learnMasks=0), "dbrief": dict(do_dBrief=1, learnMasks=0), "mdbrief": dict(do_dBrief=1, learnMasks=1)}


def image(rng, kind, cam, frame, ci):
    w, h = cam["width"], cam["height"]
    if kind == "scene":
        img = synth.synth_image(frame, ci, cam)
    elif kind == "noise":
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == "blocks":   # saturated checker of random block size: corners everywhere, ties in the scores
        b = int(rng.integers(3, 17))
        yy, xx = np.mgrid[0:h, 0:w]
        img = (((yy // b + xx // b) & 1) * int(rng.integers(60, 256))).astype(np.uint8)
    elif kind == "flat":     # (almost) nothing to find
        img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        img[rng.integers(0, h, 40), rng.integers(0, w, 40)] = 255
    else:                    # smooth gradient + sparse speckle
        yy, xx = np.mgrid[0:h, 0:w]
        img = ((xx * 255 // max(w - 1, 1) + yy * 255 // max(h - 1, 1)) // 2).astype(np.uint8)
        n = w * h // 50
        img[rng.integers(0, h, n), rng.integers(0, w, n)] = rng.integers(0, 256, n)
    return np.ascontiguousarray(img)


def mask_of(rng, kind, cam):
    w, h = cam["width"], cam["height"]
    if kind == "none":
        return None
    if kind == "mirror":
        return synth.mirror_mask(cam)
    m = np.full((h, w), 255, np.uint8)
    for _ in range(int(rng.integers(1, 6))):   # random rectangles masked out
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        m[y0:y0 + int(rng.integers(8, h // 2 + 9)), x0:x0 + int(rng.integers(8, w // 2 + 9))] = 0
    return m


def one_case(rng, idx):
    W = int(rng.integers(200, 1001)); H = int(rng.integers(160, 721))
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.25, 1.3, 1.5, 2.0]))
    nl = int(rng.integers(1, 9))
    while nl > 1 and min(W, H) / sf ** (nl - 1) < 90:
        nl -= 1
    nf = int(rng.choice([30, 100, 250, 500, 1000, 1000, 1500, 2500]))
    th = int(rng.choice([7, 12, 20, 20, 30]))
    mode = str(rng.choice(list(MODES)))
    agast = int(rng.random() < 0.2); atype = int(rng.integers(0, 4 if agast else 3))   # (FAST: TYPE_5_8 / 7_12 / 9_16; AGAST: four types)
    ikind = str(rng.choice(["scene", "scene", "scene", "noise", "blocks", "flat", "gradient"]))
    mkind = str(rng.choice(["none", "mirror", "rects"]))
    base = synth.lafida_cameras()
    ci = int(rng.integers(0, len(base)))
    cam = synth.scaled_camera(base[ci], W, H)
    if rng.random() < 0.3:   # the principal point off the scaled one
        cam["u0"] += float(rng.uniform(-20, 20)); cam["v0"] += float(rng.uniform(-20, 20))
    ds = int(rng.choice([32, 32, 32, 16, 64]))
    kw = dict(nfeatures=nf, scaleFactor=sf, nlevels=nl, fastThreshold=th, useAgast=agast, fastAgastType=atype, descSize=ds, **MODES[mode])
    desc = "case %d: %dx%d sf=%.2f nl=%d nf=%d th=%d mode=%s ds=%d agast=%d/%d img=%s mask=%s cam=%d" % (idx, W, H, sf, nl, nf, th, mode, ds, agast, atype, ikind, mkind, ci)
    imgs = [image(rng, ikind, cam, f, ci) for f in (0, 1)]
    try:
        ex = mcs.Extractor(G.ctx(), W, H, max_batch=2, **kw)
    except Exception as e:   # a geometry the library refuses: say what the oracle does with it (the reference's DistributeOctTree has nIni = round(width / height) initial
        try:                 # nodes: 0 for portrait levels — hX = width / 0, no node at all)
            oe = O.Extractor(**kw)
            oe.cap = 4 * (nf + 64 * nl)
            kps, _, _ = oe(imgs[0], None, O.make_ocam(cam))
            REFUSED.append("%dx%d nl=%d nf=%d: oracle finds %d keypoints" % (W, H, nl, nf, len(kps)))
        except AssertionError as oe_err:
            REFUSED.append("%dx%d nl=%d nf=%d: oracle fails too (%s)" % (W, H, nl, nf, oe_err))
        return None, 0
    knobs = ""
    if mode != "orb":   # the descriptor stage's switches: every setting must give the same bits
        if rng.random() < 0.25:
            ex.set_describe(exact_only=True); knobs += " exact_only"
        elif rng.random() < 0.3:
            ge = float(rng.choice([1e-6, 1e-4, 1e-2])); ex.set_describe(exact_only=False, guard_eps=ge); knobs += " guard=%g" % ge
        if rng.random() < 0.3:
            tb = float(rng.choice([2e-4, 0.5, -1.0])); ex.set_tie_band(tb); knobs += " tie_band=%g" % tb
    desc += knobs
    m = mask_of(rng, mkind, cam)
    masks = [m, m]
    oc = mcs.make_ocam(cam)
    res = ex.extract_host(imgs, masks if m is not None else None, [oc, oc])
    sets = []
    nkp = 0
    for i in range(2):
        oex = O.Extractor(**kw)
        oex.cap = max(oex.cap, ex.cap)   # (the wrapper's default capacity is nfeatures + 4 per level; the oct-tree may return up to 4 * nIni per level)
        ocam = O.make_ocam(cam)
        kps, d, dm = oex(imgs[i], masks[i], ocam)
        rays = np.zeros((len(kps), 3))
        if len(kps):
            O.lib().orc_rays(ocam, O.ptr(kps), len(kps), O.ptr(rays))
        gk, gd, gm, gr = res[i]
        if len(gk) != len(kps):
            return desc + " -> image %d: %d keypoints, oracle %d" % (i, len(gk), len(kps)), 0
        for f in ("x", "y", "size", "angle", "response"):
            e = G.first_diff(gk[f].view(np.uint32), kps[f].view(np.uint32))
            if e:
                return desc + " -> image %d keypoint.%s: %s" % (i, f, e), 0
        if not (gk["octave"] == kps["octave"]).all():
            return desc + " -> image %d octaves" % i, 0
        for name, a, b in (("descriptors", gd, d), ("masks", gm, dm), ("rays", gr.view(np.uint64), rays.view(np.uint64))):
            e = G.first_diff(a, b)
            if e:
                return desc + " -> image %d %s: %s" % (i, name, e), 0
        sets.append((np.ascontiguousarray(d), np.ascontiguousarray(dm)))
        nkp += len(kps)
    # the same two images as two device-kind batches back to back (strided rows as the rig's exchange blocks, a capture ring of two slots, the first batch patched when
    # the pyramid already holds the second): rows, counts and keypoints must be the host-kind call's, i.e. the oracle's
    if rng.random() < 0.5:
        cap_rows = ex.cap
        pitch_rows = cap_rows + int(rng.integers(0, 3)); stride = 2 * ds + int(rng.choice([0, 0, 16]))
        ex.set_tie_capture(2, 4096)
        dimg = [G.DevBuf(im[None]) for im in imgs]
        dmsk = G.DevBuf(m[None]) if m is not None else None
        outs = []
        for i in range(2):
            o = dict(nkp=G.DevBuf(np.zeros(1, np.int32)), kps=G.DevBuf(np.zeros((1, cap_rows), cap_mod.KP_DTYPE)), rows=G.DevBuf(np.zeros((pitch_rows, stride), np.uint8)),
                     rays=G.DevBuf(np.zeros((1, cap_rows, 3), np.float64)))
            ex.extract_strided(1, dimg[i].ptr.value, W * H, W, dmsk.ptr.value if dmsk else 0, W * H, W, [oc], o["nkp"].ptr.value, o["kps"].ptr.value, o["rows"].ptr.value,
                               o["rows"].ptr.value + ds, o["rays"].ptr.value, pitch_rows, stride)
            outs.append(o)
        G.ctx().synchronize()
        for back, i in ((1, 0), (0, 1)):
            ex.patch_ties(back)
            gk, gd, gm, gr = res[i]
            n = int(outs[i]["nkp"].read()[0])
            rws = outs[i]["rows"].read()
            if n != len(gk):
                return desc + " -> device-kind batch %d: %d keypoints, host-kind %d" % (i, n, len(gk)), 0
            e = G.first_diff(rws[:n, :ds], gd) or G.first_diff(rws[:n, ds:2 * ds], gm) or G.first_diff(outs[i]["kps"].read()[0, :n], gk) \
                or G.first_diff(outs[i]["rays"].read()[0, :n].view(np.uint64), gr.view(np.uint64))
            if e:
                return desc + " -> device-kind batch %d (pitch %d, stride %d): %s" % (i, pitch_rows, stride, e), 0
    ex.close()
    # the two frames against each other: SearchByBoW(KF,KF) with a random eligibility and ratio, SearchByBoW(KF,F)
    (dq, mq), (dt, mt) = sets
    nq, nt = len(dq), len(dt)
    if nq and nt:
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        lib, ctx = mcs.lib(), G.ctx()
        masked = mode == "mdbrief"
        if not masked:
            mq, mt = np.full_like(dq, 255), np.full_like(dt, 255)
        vq, vt = (rng.random(nq) < 0.85).astype(np.uint8), (rng.random(nt) < 0.85).astype(np.uint8)
        ratio = float(rng.choice([0.6, 0.75, 0.9, 1.0]))
        K = int(rng.choice([32, 32, 8, 2]))
        q = cap_mod.DescSet(P(dq), P(mq) if masked else None, P(vq), None, nq, ds)
        t = cap_mod.DescSet(P(dt), P(mt) if masked else None, P(vt), None, nt, ds)
        m12 = np.full(nq, -7, np.int32); nm = np.zeros(1, np.int32); fb = np.zeros(1, np.int32)
        cap_mod.check(lib.mcs_search_kf_kf(ctx.h, 1, C.byref(q), 0, C.byref(t), 0, ds, ratio, K, cap_mod.MEM_HOST, P(m12), P(nm), P(fb)))
        en, e12 = O.search_kf_kf(dq, mq, vq, dt, mt, vt, masked, ratio)
        if int(nm[0]) != en or not np.array_equal(m12, e12):
            return desc + " -> search_kf_kf ratio=%.2f K=%d: %d matches, oracle %d" % (ratio, K, int(nm[0]), en), nkp
        t2 = cap_mod.DescSet(P(dt), P(mt) if masked else None, None, None, nt, ds)
        out = np.full(nt, -7, np.int32)
        cap_mod.check(lib.mcs_search_kf_f(ctx.h, 1, C.byref(q), 0, C.byref(t2), 0, ds, ratio, K, cap_mod.MEM_HOST, P(out), P(nm), P(fb)))
        en, eo = O.search_kf_f(dq, mq, vq, dt, mt, masked, ratio)
        if int(nm[0]) != en or not np.array_equal(out, eo):
            return desc + " -> search_kf_f ratio=%.2f K=%d: %d matches, oracle %d" % (ratio, K, int(nm[0]), en), nkp
    return None, nkp


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    n = bad = feats = 0
    while time.time() - t0 < budget:
        err, k = one_case(rng, n)
        n += 1
        feats += k
        if err:
            bad += 1
            print("FAIL", err, flush=True)
    for r in REFUSED[:12]:
        print("refused by the library:", r)
    print("fuzz_parity: seed %d, %d cases, %d keypoints compared, %d failures, %.0f s" % (seed, n, feats, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

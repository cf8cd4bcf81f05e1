"""Known-answer tests pinning the CPU oracle.  The reference ships no tests or golden
vectors for this path ("parity unpinned"); every expected value below is derived by hand from the
reference source (file:line cited) or from the published OpenCV 3.x algorithm."""
import ctypes as C

import numpy as np
import pytest


def test_features_per_level(oracle):
    # mdBRIEFextractorOct.cpp:167-179
    out = (C.c_int * 8)()
    oracle.lib().orc_features_per_level(1000, C.c_float(1.2), 8, out)
    assert list(out) == [217, 181, 151, 126, 105, 87, 73, 60]
    oracle.lib().orc_features_per_level(400, C.c_float(1.2), 8, out)
    assert list(out) == [87, 72, 60, 50, 42, 35, 29, 25]
    oracle.lib().orc_features_per_level(2000, C.c_float(1.2), 8, out)
    assert list(out) == [434, 362, 302, 251, 209, 175, 145, 122]


def test_level_sizes(oracle):
    # :1164-1165 with scaleFactor = (double)1.2f
    w, h = (C.c_int * 8)(), (C.c_int * 8)()
    oracle.lib().orc_level_sizes(754, 480, C.c_float(1.2), 8, w, h)
    assert list(zip(w, h)) == [(754, 480), (628, 400), (524, 333), (436, 278), (364, 231), (303, 193), (253, 161), (210, 134)]
    assert sum(a * b for a, b in zip(w, h)) == 1120256
    oracle.lib().orc_level_sizes(1280, 800, C.c_float(1.2), 8, w, h)
    assert (w[7], h[7]) == (357, 223)
    assert sum(a * b for a, b in zip(w, h)) == 3171309


def test_umax_and_patch(oracle):
    # :187-202, HALF_PATCH_SIZE = 16
    um = (C.c_int * 17)()
    oracle.lib().orc_umax(um)
    assert list(um) == [16, 16, 16, 16, 15, 15, 15, 14, 14, 13, 12, 12, 11, 9, 8, 6, 3]
    assert (2 * um[0] + 1) + 2 * sum(2 * u + 1 for u in list(um)[1:]) == 845


def test_pattern(oracle):
    xy = (C.c_int * 2048)()
    assert oracle.lib().orc_pattern(64, xy) == 1024
    v = np.array(xy[:], np.int32)
    assert v.min() == -14 and v.max() == 15
    assert [tuple(v[2 * i:2 * i + 2]) for i in range(4)] == [(-12, -1), (11, 0), (11, 1), (11, 15)]
    assert oracle.lib().orc_pattern(32, xy) == 512
    r = np.hypot(v[0::2], v[1::2]).max()
    assert abs(r - np.hypot(15, 15)) < 1e-9


def test_thresholds(oracle):
    # cORBmatcher.cpp:46-65
    hi, lo = C.c_int(), C.c_int()
    oracle.lib().orc_thresholds(32, 0, C.byref(hi), C.byref(lo))
    assert (hi.value, lo.value) == (96, 64)
    oracle.lib().orc_thresholds(32, 1, C.byref(hi), C.byref(lo))
    assert (hi.value, lo.value) == (48, 32)


def test_descriptor_distance(oracle):
    # :2438-2474
    L = oracle.lib()
    z = np.zeros(4, np.uint64)
    o = np.full(4, 0xFFFFFFFFFFFFFFFF, np.uint64)
    rng = np.random.default_rng(1)
    x = rng.integers(0, 2**63, 4).astype(np.uint64)
    y = rng.integers(0, 2**63, 4).astype(np.uint64)
    p = oracle.ptr
    assert L.orc_dist64(p(x), p(x), 32) == 0
    assert L.orc_dist64(p(z), p(o), 32) == 256
    ref = sum(bin(int(a) ^ int(b)).count("1") for a, b in zip(x, y))
    assert L.orc_dist64(p(x), p(y), 32) == ref
    assert L.orc_dist64_masked(p(x), p(y), p(o), p(o), 32) == ref       # full masks == unmasked
    assert L.orc_dist64_masked(p(x), p(y), p(z), p(z), 32) == 0         # zero masks
    a = z.copy(); a[0] = 1                                              # one differing bit, one mask covers it: 1/2 = 0
    m = z.copy(); m[0] = 1
    assert L.orc_dist64_masked(p(a), p(z), p(m), p(z), 32) == 0
    assert L.orc_dist64_masked(p(a), p(z), p(m), p(m), 32) == 1
    a[1] = 1; m[1] = 1                                                  # total 3 -> floor(3/2) = 1 (ONE division of the total)
    m2 = z.copy(); m2[0] = 1
    assert L.orc_dist64_masked(p(a), p(z), p(m), p(m2), 32) == 1


def test_box_blur_constant_and_known(oracle):
    b = 2
    img = np.full((20 + 2 * b, 30 + 2 * b), 77, np.uint8)
    roi = img[b:, b:]
    oracle.lib().orc_box5_inplace(roi.ctypes.data_as(C.c_void_p), 30, 20, img.strides[0])
    assert (img == 77).all()
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (24, 34)).astype(np.uint8)
    buf = src.copy()
    oracle.lib().orc_box5_inplace(buf[b:, b:].ctypes.data_as(C.c_void_p), 30, 20, buf.strides[0])
    s = sum(src[b + dy:b + dy + 20, b + dx:b + dx + 30].astype(np.int32) for dy in range(-2, 3) for dx in range(-2, 3))
    assert (buf[b:b + 20, b:b + 30] == ((s + 12) // 25)).all()        # Appendix A.4: (sum+12)/25
    assert (buf[:b] == src[:b]).all() and (buf[:, :b] == src[:, :b]).all()  # frame untouched


def test_fast_atan2(oracle):
    f = oracle.lib().orc_fastAtan2
    assert f(0.0, 1.0) == 0.0
    for (y, x), deg in {(1, 0): 90, (0, -1): 180, (-1, 0): 270, (1, 1): 45, (-1, -1): 225, (1, -1): 135}.items():
        assert abs(f(float(y), float(x)) - deg) < 0.3
    # exact bit patterns frozen from the oracle (float32, no FMA contraction)
    got = np.array([f(1.0, 1.0), f(3.0, -7.0), f(-2.5, 0.25), f(1e-3, 123.0)], np.float32)
    exp = np.load(__file__.replace("test_oracle_kat.py", "golden/fastatan2_kat.npy"))
    assert (got.view(np.uint32) == exp.view(np.uint32)).all()


def _arc_image(center, dark_from, dark_len, lo, hi=None):
    """7x7 patch: centre value `center`, circle pixels k in [dark_from, dark_from+dark_len) set to lo, rest = hi."""
    circle = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
              (-3, 1), (-2, 2), (-1, 3)]
    img = np.full((7, 7), center if hi is None else hi, np.uint8)
    img[3, 3] = center
    for k in range(dark_from, dark_from + dark_len):
        x, y = circle[k % 16]
        img[3 + y, 3 + x] = lo
    return img


def test_fast_score_and_detection(oracle):
    L = oracle.lib()
    kps = np.zeros(16, oracle.KP_DTYPE)
    # 9 contiguous circle pixels darker by 60 -> corner at threshold 20, score = min-arc-contrast - 1 = 59
    img = _arc_image(100, 3, 9, 40)
    n = L.orc_fast9_16(oracle.ptr(img), 7, 7, 7, None, 0, 20, oracle.ptr(kps), 16)
    assert n == 1 and (kps[0]["x"], kps[0]["y"], kps[0]["response"]) == (3.0, 3.0, 59.0)
    assert kps[0]["size"] == 7.0 and kps[0]["angle"] == -1.0 and kps[0]["class_id"] == -1
    assert L.orc_fast_score(oracle.ptr(img[3:, 3:]), 7, 20) == 59
    # only 8 contiguous -> not a corner
    img8 = _arc_image(100, 3, 8, 40)
    assert L.orc_fast9_16(oracle.ptr(img8), 7, 7, 7, None, 0, 20, oracle.ptr(kps), 16) == 0
    # brighter arc, contrast exactly threshold+1 -> corner with score = threshold
    imgb = _arc_image(100, 10, 9, 121)
    assert L.orc_fast9_16(oracle.ptr(imgb), 7, 7, 7, None, 0, 20, oracle.ptr(kps), 16) == 1 and kps[0]["response"] == 20.0
    # contrast == threshold is NOT a corner (strict)
    imgc = _arc_image(100, 10, 9, 120)
    assert L.orc_fast9_16(oracle.ptr(imgc), 7, 7, 7, None, 0, 20, oracle.ptr(kps), 16) == 0
    # mask == 0 at the keypoint removes it
    mask = np.full((7, 7), 255, np.uint8); mask[3, 3] = 0
    assert L.orc_fast9_16(oracle.ptr(img), 7, 7, 7, oracle.ptr(mask), 7, 20, oracle.ptr(kps), 16) == 0


def test_fast_closed_form_matches_opencv_loop(oracle):
    """score == max(t, max_arcs min(d), max_arcs min(-d)) - 1 and corner <=> that max > t (used by the HIP kernel)."""
    rng = np.random.default_rng(5)
    circle = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
              (-3, 1), (-2, 2), (-1, 3)]
    kps = np.zeros(4, oracle.KP_DTYPE)
    ncorner = 0
    for it in range(3000):
        img = rng.integers(0, 256, (7, 7)).astype(np.uint8)
        if it % 2:
            img = (img // 64 * 64).astype(np.uint8) + (it % 7)
        t = int(rng.integers(0, 60))
        v = int(img[3, 3])
        d = np.array([v - int(img[3 + y, 3 + x]) for x, y in circle])
        dd = np.concatenate([d, d])
        A = max(dd[k:k + 9].min() for k in range(16))
        B = max((-dd[k:k + 9]).min() for k in range(16))
        closed = max(t, A, B) - 1
        assert oracle.lib().orc_fast_score(oracle.ptr(img[3:, 3:]), 7, t) == closed
        n = oracle.lib().orc_fast9_16(oracle.ptr(img), 7, 7, 7, None, 0, t, oracle.ptr(kps), 4)
        # a corner whose score is 0 (t = 0, contrast 1) never survives the strict NMS against zero neighbours
        assert n == int(max(A, B) > t and closed > 0)
        if n:
            ncorner += 1
            assert kps[0]["response"] == float(closed)
    assert ncorner > 20


def test_resize_linear_properties(oracle):
    L = oracle.lib()
    src = np.full((400, 628), 131, np.uint8)
    dst = np.zeros((333, 524), np.uint8)
    L.orc_resize_linear(oracle.ptr(src), 628, 400, 628, oracle.ptr(dst), 524, 333, 524)
    assert (dst == 131).all()                                          # constant in -> constant out
    # hand-computed fixed-point value (Appendix A.1): 4x2 -> 2x1, scale 2: fx=0.5 -> coeffs 1024/1024
    s = np.array([[10, 20, 30, 40], [50, 60, 70, 80]], np.uint8)
    d = np.zeros((1, 2), np.uint8)
    L.orc_resize_linear(oracle.ptr(s), 4, 2, 4, oracle.ptr(d), 2, 1, 2)
    # column 0: T0 = 10*1024+20*1024 = 30720, T1 = 50*1024+60*1024 = 112640; b = 1024/1024
    exp0 = (((1024 * (30720 >> 4)) >> 16) + ((1024 * (112640 >> 4)) >> 16) + 2) >> 2
    assert d[0, 0] == exp0 == 35
    # identity size -> identity
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (9, 11)).astype(np.uint8)
    b = np.zeros_like(a)
    L.orc_resize_linear(oracle.ptr(a), 11, 9, 11, oracle.ptr(b), 11, 9, 11)
    assert (a == b).all()


def test_resize_nearest_and_border(oracle):
    L = oracle.lib()
    a = np.arange(6 * 8, dtype=np.uint8).reshape(6, 8)
    b = np.zeros((5, 7), np.uint8)
    L.orc_resize_nearest(oracle.ptr(a), 8, 6, 8, oracle.ptr(b), 7, 5, 7)
    xs = np.minimum(np.floor(np.arange(7) * (1.0 / (7 / 8))).astype(int), 7)
    ys = np.minimum(np.floor(np.arange(5) * (1.0 / (5 / 6))).astype(int), 5)
    assert (b == a[ys][:, xs]).all()
    buf = np.zeros((6 + 4, 8 + 4), np.uint8)
    buf[2:-2, 2:-2] = a
    L.orc_border_reflect101(oracle.ptr(buf), 8, 6, 12, 2)
    assert (buf == np.pad(a, 2, mode="reflect")).all()                 # numpy 'reflect' == BORDER_REFLECT_101


def test_camera_model(oracle, synth):
    cams = synth.lafida_cameras()
    for cam in cams:
        oc = oracle.make_ocam(cam)
        x, y, z, u, v = (C.c_double() for _ in range(5))
        rng = np.random.default_rng(0)
        for _ in range(200):
            ang = rng.uniform(0, 2 * np.pi)
            r = rng.uniform(1, cam["v0"] - 5)
            uu, vv = cam["u0"] + r * np.cos(ang), cam["v0"] + r * np.sin(ang)
            oracle.lib().orc_img2world(C.byref(oc), uu, vv, C.byref(x), C.byref(y), C.byref(z))
            assert abs(x.value ** 2 + y.value ** 2 + z.value ** 2 - 1) < 1e-12
            oracle.lib().orc_world2img(C.byref(oc), x.value, y.value, z.value, C.byref(u), C.byref(v))
            assert abs(u.value - uu) < 0.05 and abs(v.value - vv) < 0.05
        # principal point maps to the optical axis (z = -p[0]/|p[0]| = +1 since a0 < 0)
        oracle.lib().orc_img2world(C.byref(oc), cam["u0"], cam["v0"], C.byref(x), C.byref(y), C.byref(z))
        assert abs(x.value) < 1e-12 and abs(y.value) < 1e-12 and abs(z.value - 1) < 1e-12


def test_mirror_mask(oracle, synth):
    cam = synth.lafida_cameras()[0]
    m = oracle.mirror_mask(oracle.make_ocam(cam))
    assert m.shape == (480, 754)
    assert (m == synth.mirror_mask(cam)).all()
    # radius v0+22 around (col=u0,row=v0): cam_model_omni.cpp:187-212 (names swapped in the reference)
    yy, xx = np.mgrid[0:480, 0:754]
    r = np.hypot(yy - cam["v0"], xx - cam["u0"])
    assert (m[r < cam["v0"] + 21.9] == 255).all() and (m[r > cam["v0"] + 22.1] == 0).all()


def test_octtree_small(oracle):
    kp = np.zeros(5, oracle.KP_DTYPE)
    # 5 points, N=3 in a 100x50 area: nIni = cvRound(100/50) = 2 root nodes
    kp["x"] = [10, 20, 60, 70, 90]
    kp["y"] = [10, 40, 10, 40, 45]
    kp["response"] = [5, 9, 7, 7, 1]
    out = np.zeros(16, oracle.KP_DTYPE)
    n = oracle.lib().orc_distribute_octtree(oracle.ptr(kp), 5, 0, 100, 0, 50, 3, oracle.ptr(out), 16)
    # roots [0,50) has 2 pts, [50,100) has 3 -> first pass splits both: 4..5 nodes >= 3 -> stop; one best point per node
    assert n >= 3
    got = sorted((float(a), float(b)) for a, b in zip(out["x"][:n], out["y"][:n]))
    assert set(got) <= set(zip(kp["x"].tolist(), kp["y"].tolist()))
    # N larger than the point count: every point survives (all nodes end with one point)
    n = oracle.lib().orc_distribute_octtree(oracle.ptr(kp), 5, 0, 100, 0, 50, 50, oracle.ptr(out), 16)
    assert n == 5


def test_extract_end_to_end_shapes(oracle, synth):
    cam = synth.lafida_cameras()[0]
    oc = oracle.make_ocam(cam)
    img = synth.synth_image(0, 0, cam)
    mask = oracle.mirror_mask(oc)
    ex = oracle.Extractor()
    kps, d, dm = ex(img, mask, oc)
    cand0 = ex.candidates(0)
    assert len(cand0) > 217          # the oct-tree's largest-first branch (:771) is exercised
    sel = [len(ex.selected(l)) for l in range(8)]
    assert all(0 < s <= n + 2 for s, n in zip(sel, [217, 181, 151, 126, 105, 87, 73, 60]))
    assert len(kps) == sum(sel) and d.shape == (len(kps), 32) and (dm == 0).all()
    assert (kps["octave"][:sel[0]] == 0).all() and kps["size"][0] == 32.0
    assert kps["octave"][-1] == 7 and kps["size"][-1] == float(int(32 * float(np.float32(1.2)) ** 7))
    lvl0 = kps[:sel[0]]
    assert lvl0["x"].min() >= 25 and lvl0["x"].max() < 754 - 25 and lvl0["y"].min() >= 25 and lvl0["y"].max() < 480 - 25
    assert (mask[lvl0["y"].astype(int), lvl0["x"].astype(int)] == 255).all()
    # modes: dBRIEF changes descriptors but not keypoints; mdBRIEF adds masks, same descriptors as dBRIEF
    k1, d1, m1 = oracle.Extractor(do_dBrief=1)(img, mask, oc)
    k2, d2, m2 = oracle.Extractor(do_dBrief=1, learnMasks=1)(img, mask, oc)
    assert (k1 == kps).all() and (k2 == kps).all()
    assert (d1 != d).any() and (m1 == 0).all() and m2.any()
    # mdBRIEF main descriptor uses angle/RHOf instead of angle*DEG2RADf (Appendix B.2): identical up to rare rounding flips
    assert (np.unpackbits(d1 ^ d2).sum() / d1.size / 8) < 1e-3


def test_matchers_small(oracle):
    rng = np.random.default_rng(11)
    n = 200
    d2 = rng.integers(0, 256, (n, 32)).astype(np.uint8)
    perm = rng.permutation(n)
    d1 = d2[perm].copy()
    flip = rng.integers(0, 256, (n, 32)).astype(np.uint8) & rng.integers(0, 256, (n, 32)).astype(np.uint8) & \
        rng.integers(0, 256, (n, 32)).astype(np.uint8) & rng.integers(0, 256, (n, 32)).astype(np.uint8)
    d1 ^= flip                                                          # ~16 bit flips
    ones = np.full((n, 32), 255, np.uint8)
    v = np.ones(n, np.uint8)
    nm, m12 = oracle.search_kf_kf(d1, ones, v, d2, ones, v, False, 0.9)
    assert nm == n and (m12 == perm).all()
    nm2, mF = oracle.search_kf_f(d1, ones, v, d2, ones, False, 0.9)
    assert nm2 == n and (mF[perm] == np.arange(n)).all()
    # greedy order dependence: two identical queries -> the first takes the train, the second must take another or fail
    d1b = np.vstack([d1[:1], d1[:1]])
    nm3, m3 = oracle.search_kf_kf(d1b, ones[:2], v[:2], d2, ones, v, False, 0.9)
    assert m3[0] == perm[0] and m3[1] != perm[0]
    # invalid entries are skipped
    v1 = v.copy(); v1[::2] = 0
    nm4, m4 = oracle.search_kf_kf(d1, ones, v1, d2, ones, v, False, 0.9)
    assert (m4[::2] == -1).all() and (m4[1::2] == perm[1::2]).all()

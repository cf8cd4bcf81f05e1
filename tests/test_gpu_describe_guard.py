"""-m gpu: the two-pass dBRIEF / mdBRIEF descriptor (csrc/mcs_describe.hip).

The fast pass evaluates the omni model of the 2*8*descSize pattern points (x3 for mdBRIEF) with a cheaper arithmetic and sums the pattern mean as a
wave tree; it may only be used for a keypoint when NO coordinate lies within the guard band of a cvRound tie, everything else goes through the
reference's exact arithmetic (rotateAndDistortPattern, src/mdBRIEFextractorOct.cpp:250-283).  Checked here:
  * the arithmetic difference between the two forms, measured on the device on 2^27 pseudo-random pattern points per camera, stays far below the bound
    the host assumes (mcs_describe_fast_bound), which in turn is below half the guard band;
  * default mode == exact-only mode == the oracle, bit for bit, for every descriptor size and both distorted modes;
  * widening the band (almost every keypoint falls back / about half of them fall back) changes nothing in the output — the fallback path is exercised
    and the split itself is invisible;
  * a camera whose bound exceeds the band is served by the exact pass alone."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_common
    return gpu_common


def _run(G, ex, imgs, masks, cams):
    return ex.extract_host(imgs, masks, [G.mcs.make_ocam(c) for c in cams])


def _same(G, a, b):
    assert len(a) == len(b)
    for (k1, d1, m1, r1), (k2, d2, m2, r2) in zip(a, b):
        assert G.first_diff(k1, k2) is None and G.first_diff(d1, d2) is None and G.first_diff(m1, m2) is None and G.first_diff(r1, r2) is None


def test_fast_arithmetic_stays_below_the_assumed_bound(G):
    lib, ctx = G.mcs.lib(), G.ctx()
    cams = G.cams3() + [G.synth.scaled_camera(G.cams3()[1], 1280, 800)]
    flipped = dict(G.cams3()[0])                                                        # the mirror camera (p0 > 0, invP(-theta)): the other sign of p0 in the G(s) table
    flipped["p"] = [-v for v in flipped["p"]]; flipped["invP"] = [v * (-1) ** i for i, v in enumerate(flipped["invP"])]
    short = dict(G.cams3()[2]); short["invP"] = short["invP"][:6]                       # a low-degree backward polynomial
    cams += [flipped, short]
    for ci, cam in enumerate(cams):
        oc = G.mcs.make_ocam(cam)
        for ds in (16, 32, 64):
            bound = C.c_double()
            G.mcs.check(lib.mcs_describe_fast_bound(C.byref(oc), ds, C.byref(bound)))
            assert 0 < bound.value <= 0.5 * 2.0 ** -24, (ci, ds, bound.value)      # the default band holds twice the bound
        worst = 0.0
        for seed in (1, 2):
            got = C.c_double(-1.0)
            G.mcs.check(lib.mcs_selftest_describe_fast(ctx.h, C.byref(oc), seed, 1 << 26, C.byref(got)))
            worst = max(worst, got.value)
        print("camera %d: max |fast - exact| over 2^27 points = %.3e px, assumed bound %.3e" % (ci, worst, bound.value))
        assert 0 <= worst < 0.25 * bound.value, (ci, worst, bound.value)


@pytest.mark.parametrize("do_db,masks_on,ds", [(1, 1, 32), (1, 1, 16), (1, 1, 64), (1, 0, 32), (1, 0, 64)])
def test_fast_pass_equals_exact_pass_and_oracle(G, do_db, masks_on, ds):
    cams = G.cams3()
    imgs, masks, cl = [], [], []
    for f in range(4):
        i, m, c = G.frame_inputs(f)
        imgs += i; masks += m; cl += c
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=len(imgs), do_dBrief=do_db, learnMasks=masks_on, descSize=ds)
    fast = _run(G, ex, imgs, masks, cl)
    n_fb, eps = ex.describe_stats()
    nk = sum(len(r[0]) for r in fast)
    assert eps == 2.0 ** -24 and nk > 10000
    assert n_fb < 0.01 * nk, (n_fb, nk)          # the exact pass is the exception
    ex.set_describe(exact_only=True)
    exact = _run(G, ex, imgs, masks, cl)
    assert ex.describe_stats()[0] == n_fb         # exact-only mode does not go through the list
    _same(G, fast, exact)
    for i in (0, 7):
        _, kps, d, dm, rays = G.oracle_extract(imgs[i], masks[i], cl[i], do_dBrief=do_db, learnMasks=masks_on, descSize=ds)
        gk, gd, gm, gr = fast[i]
        assert G.first_diff(gk, kps) is None and G.first_diff(gd, d) is None and G.first_diff(gm, dm) is None and G.first_diff(gr, rays) is None
    ex.close()


@pytest.mark.parametrize("eps,lo,hi", [(0.05, 0.98, 1.0), (1e-4, 0.2, 0.8), (2e-6, 0.002, 0.05)])
def test_widened_guard_band_forces_the_fallback_and_changes_nothing(G, eps, lo, hi):
    imgs, masks, cl = [], [], []
    for f in range(2):
        i, m, c = G.frame_inputs(f)
        imgs += i; masks += m; cl += c
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=len(imgs), do_dBrief=1, learnMasks=1)
    ex.set_describe(exact_only=True)
    exact = _run(G, ex, imgs, masks, cl)
    nk = sum(len(r[0]) for r in exact)
    ex.set_describe(exact_only=False, guard_eps=eps)
    before = ex.describe_stats()[0]
    got = _run(G, ex, imgs, masks, cl)
    n_fb = ex.describe_stats()[0] - before
    assert lo * nk <= n_fb <= hi * nk, (eps, n_fb, nk)
    _same(G, got, exact)
    ex.close()


def test_camera_beyond_the_band_runs_exact_only(G):
    """a backward polynomial with huge alternating coefficients: the assumed bound exceeds the band, every keypoint takes the exact pass"""
    cam = dict(G.cams3()[0])
    cam["invP"] = list(cam["invP"])
    cam["invP"][10] += 4.0e8
    cam["invP"][11] -= 4.0e8 / (np.pi / 2) * 0.999
    oc = G.mcs.make_ocam(cam)
    bound = C.c_double()
    G.mcs.check(G.mcs.lib().mcs_describe_fast_bound(C.byref(oc), 32, C.byref(bound)))
    assert bound.value > 2.0 ** -24
    imgs, masks, _ = G.frame_inputs(0)
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=1, do_dBrief=1, learnMasks=1)
    got = ex.extract_host(imgs[:1], masks[:1], [oc])
    n_fb, _ = ex.describe_stats()
    assert n_fb == len(got[0][0]) > 500
    ex.set_describe(exact_only=True)
    exact = ex.extract_host(imgs[:1], masks[:1], [oc])
    _same(G, got, exact)
    ex.close()


def test_set_describe_argument_checks(G):
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=1, do_dBrief=1, learnMasks=1)
    assert G.mcs.lib().mcs_extractor_set_describe(ex.h, 0, 0.6) == G.mcs._capi.MCS_ERR_INVALID
    assert G.mcs.lib().mcs_extractor_set_describe(ex.h, 0, -1.0) == G.mcs._capi.MCS_ERR_INVALID
    ex.close()


def test_tie_distance_is_measured_for_the_exact_arithmetic(G):
    """mcs_extractor_tie_stats: the smallest | |frac(v)| - 1/2 | over the cvRound arguments of the exact arithmetic.  Checked against numpy on the ORB rotation
    (reference src/mdBRIEFextractorOct.cpp:285-301), on all three modes' real images it must stay far above the ~1e-13 px where device and host libm could
    round differently, and the fast pass must not feed it (its coordinates are guarded instead)."""
    import oracle_lib as O
    cams = G.cams3()
    img, mask = G.synth.synth_image(2, 0, cams[0]), G.synth.mirror_mask(cams[0])
    # ORB: every coordinate is measured.  The same minimum from numpy (cos / sin of glibc: the margin is 1e-10, their difference to ocml 1e-16)
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=1, nfeatures=300)
    assert ex.tie_stats() == float("inf")
    kps = ex.extract_host([img], [mask], [G.mcs.make_ocam(cams[0])])[0][0]
    got = ex.tie_stats()
    xy = np.zeros(2 * 16 * 32, np.int32)
    assert O.lib().orc_pattern(32, O.ptr(xy)) == 512
    pat = xy.astype(np.float64).reshape(-1, 2)
    ang = (kps["angle"].astype(np.float32) * np.float32(np.float32(np.pi) / np.float32(180.0))).astype(np.float64)
    c, s = np.cos(ang)[:, None], np.sin(ang)[:, None]
    x, y = pat[None, :, 0] * c - pat[None, :, 1] * s, pat[None, :, 0] * s + pat[None, :, 1] * c
    frac = np.maximum(np.abs(x - np.rint(x)), np.abs(y - np.rint(y)))
    want = 0.5 - frac.max()
    assert abs(got - want) < 1e-12 and got > 1e-10, (got, want)
    assert ex.tie_stats(reset=True) == got and ex.tie_stats() == float("inf")
    ex.close()
    # mdBRIEF, default mode: only the keypoints the guard band hands to the exact pass are measured (few or none); exact-only mode: all of them
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=1, nfeatures=500, do_dBrief=1, learnMasks=1)
    ex.extract_host([img], [mask], [G.mcs.make_ocam(cams[0])])
    n_exact, eps = ex.describe_stats()
    t_fast = ex.tie_stats(reset=True)
    assert (t_fast == float("inf")) == (n_exact == 0)
    ex.set_describe(exact_only=True)
    ex.extract_host([img], [mask], [G.mcs.make_ocam(cams[0])])
    t_all = ex.tie_stats()
    assert 1e-10 < t_all < 1e-3 and t_all <= t_fast   # ~1.5 million coordinates: the closest lies ~1e-7 from a tie
    ex.close()

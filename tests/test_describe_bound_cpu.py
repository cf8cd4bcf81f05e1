"""CPU (no GPU): the host-side table and error bound of the guarded fast descriptor arithmetic (mcs_describe_fast_table / mcs_describe_fast_bound,
csrc/mcs_capi.hip).  The fast pass replaces the reference's WorldToImg (src/cam_model_omni.cpp:146-161) by u, v = affine(x G(s), y G(s)),
G(s) = rho(atan(p0 / sqrt(s))) / sqrt(s) read from a per-camera table indexed by the bit pattern of s = x^2 + y^2.  The bound decides which cameras the
fast pass may serve, so it is checked here three ways, independently of the C++ that builds it:
  * the table's rows against G itself (numpy long double, atan included) at 40 000 random s: the error never exceeds the tail bound the library claims;
  * the claimed magnitudes (max |rho|, max sqrt(s) |s G'|, the Lipschitz constant of (x G, y G)) against finite differences of G;
  * the majorant series of the truncated tail and the rounding terms re-derived in numpy, row by row: the library's total must agree."""
import ctypes as C
import importlib
from math import comb

import numpy as np

mcs = importlib.import_module("multicol-slam_amd")
synth = importlib.import_module("multicol-slam_amd.synth")
LD = np.longdouble
JMAX = 80


def _cams():
    base = synth.lafida_cameras()
    flipped = dict(base[0])
    flipped["p"] = [-v for v in flipped["p"]]
    flipped["invP"] = [v * (-1) ** i for i, v in enumerate(flipped["invP"])]
    short = dict(base[2])
    short["invP"] = short["invP"][:6]
    return base + [synth.scaled_camera(base[1], 1280, 800), flipped, short]


def _table(cam):
    oc = mcs.make_ocam(cam)
    rows, rl, e0, bpo = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    info = (C.c_double * 6)()
    mcs.check(mcs.lib().mcs_describe_fast_table(C.byref(oc), None, C.byref(rows), C.byref(rl), C.byref(e0), C.byref(bpo), info))
    tab = np.zeros((rows.value, rl.value))
    mcs.check(mcs.lib().mcs_describe_fast_table(C.byref(oc), tab.ctypes.data_as(C.c_void_p), None, None, None, None, None))
    return tab, e0.value, bpo.value, dict(zip(("tailU", "rhoB", "dB", "lip", "seen", "inB"), list(info)))


def _G(cam, s):
    s = np.asarray(s, dtype=LD)
    n = np.sqrt(s)
    th = np.arctan(LD(cam["p"][0]) / n)
    r = np.zeros_like(s)
    for c in reversed(cam["invP"]):
        r = r * th + LD(c)
    return r / n


def _lib_bound(cam, ds):
    oc = mcs.make_ocam(cam)
    b = C.c_double()
    mcs.check(mcs.lib().mcs_describe_fast_bound(C.byref(oc), ds, C.byref(b)))
    return b.value


def test_table_rows_reproduce_G_within_the_claimed_tail():
    rng = np.random.default_rng(11)
    for cam in _cams():
        tab, e0, bpo, info = _table(cam)
        m = int(np.log2(bpo))
        nrows = tab.shape[0]
        noct = nrows // bpo
        DEG = tab.shape[1] - 1
        assert DEG >= 4 and nrows % bpo == 0
        s = np.exp2(rng.uniform(e0, e0 + noct, 40000))
        s = np.concatenate([s, np.exp2(np.arange(e0, e0 + noct, dtype=float)), np.nextafter(np.exp2(np.arange(e0 + 1, e0 + noct + 1, dtype=float)), 0)])
        bits = s.view(np.uint64)
        hi = (bits >> np.uint64(32)).astype(np.int64)
        idx = (hi >> (20 - m)) - ((1023 + e0) << m)                      # the kernel's row: exponent and top mantissa bits
        assert idx.min() == 0 and idx.max() == nrows - 1
        frac_bits = (bits & np.uint64((1 << (52 - m)) - 1)) | np.uint64(0x3FF << 52)
        tau = frac_bits.view(np.float64).astype(LD) - LD(1.0 + 1.0 / (2 << m))   # 1 + low mantissa fraction - (1 + half a bin): exact
        assert np.abs(tau).max() <= 1.0 / (2 << m)
        val = np.zeros(len(s), dtype=LD)
        for j in range(DEG, -1, -1):
            val = val * tau + tab[idx, j].astype(LD)
        err = float((np.abs(val - _G(cam, s)) * np.sqrt(s.astype(LD))).max())
        u = 2.0 ** -53
        assert np.isfinite(info["tailU"]) and err <= info["tailU"] + 2 * u * info["rhoB"], (err, info)
        assert info["seen"] <= info["tailU"] + 64 * u * info["rhoB"]
        assert info["tailU"] < 2e-8


def test_claimed_magnitudes_hold_against_finite_differences():
    for cam in _cams()[:4]:
        tab, e0, bpo, info = _table(cam)
        noct = tab.shape[0] // bpo
        s = np.exp2(np.linspace(e0, e0 + noct, 200001)[:-1]).astype(LD)
        n = np.sqrt(s)
        G = _G(cam, s)
        assert float((np.abs(G) * n).max()) <= info["rhoB"]
        h = LD(2.0) ** -20
        sG = (_G(cam, s * (1 + h)) - _G(cam, s * (1 - h))) / (2 * h)   # s G'(s)
        assert float((np.abs(sG) * n).max()) <= info["dB"] * (1 + 1e-6)
        drho = G + 2 * sG                                                # d(n G(n^2)) / dn
        assert float(np.maximum(np.abs(G), np.abs(drho)).max()) <= info["lip"] * (1 + 1e-6)
        assert float((np.maximum(np.abs(G), np.abs(drho)) * (n + 44)).max()) <= info["inB"] * (1 + 1e-6)


def _majorant_tail(cam, e0, nrows, bpo, DEG):
    """per-row truncated tail through the documented majorant series (numpy, independent of the C++): returns max over rows of sqrt(s) * tail"""
    P = np.array(cam["invP"], dtype=LD)
    nP = len(P)
    p0 = LD(cam["p"][0])
    m = int(np.log2(bpo))
    bj = np.ones(JMAX + 1, dtype=LD)
    for j in range(1, JMAX + 1):
        bj[j] = bj[j - 1] * (-(LD(2 * j - 1)) / LD(2 * j))
    abj = np.abs(bj).astype(float)
    combs = np.array([[comb(i - 1, k - 1) if 1 <= k <= i else 0 for k in range(nP)] for i in range(JMAX + 1)], dtype=float)
    ii = np.arange(JMAX + 1)[:, None]
    kk = np.arange(nP)[None, :]
    worst = 0.0
    for r in range(nrows):
        e, k = e0 + r // bpo, r % bpo
        kappa = LD(1) + (LD(k) + LD(0.5)) / LD(bpo)
        c = kappa * LD(2.0) ** e
        zc = p0 / np.sqrt(c)
        thc = np.arctan(zc)
        pk = np.array([sum(comb(j, q) * P[j] * thc ** (j - q) for j in range(q, nP)) for q in range(nP)], dtype=LD)
        apk = np.abs(pk).astype(float)
        q_ = float(abs(zc) / np.sqrt(1 + zc * zc))
        a, b = q_ / 2, 1 + q_ / 2
        eps = float(LD(2.0) ** -(m + 1) / kappa)
        with np.errstate(over="ignore", invalid="ignore"):
            terms = np.where((kk >= 1) & (kk <= ii), apk[None, :] * np.power(a, kk) * np.power(b, np.maximum(ii - kk, 0)) * combs, 0.0)
        Pi = terms.sum(1)                                              # P_0 = 0
        W = (apk[0] * abj + np.convolve(Pi, abj)[:JMAX + 1]) / float(np.sqrt(c))
        tail = float((W[DEG + 1:] * eps ** np.arange(DEG + 1, JMAX + 1)).sum())
        tail += (apk[0] + apk[1:].sum()) / float(np.sqrt(c)) * 3 * (3 * eps) ** (JMAX + 1) / (1 - 3 * eps)
        worst = max(worst, float(np.sqrt(c * (1 + eps))) * tail)
    return worst


def _bound(cam, npoints, info, tailU):
    u, hp = 2.0 ** -53, np.pi / 2
    a = [abs(v) for v in cam["invP"]]
    S = sum(v * hp ** i for i, v in enumerate(a))
    Sp = sum(i * v * hp ** (i - 1) for i, v in enumerate(a) if i)
    aff = 1 + abs(cam["c"]) + abs(cam["d"]) + abs(cam["e"])
    pp = 8 * u * (abs(cam["u0"]) + abs(cam["v0"]))
    # reference 3 roundings, fast 2, both coordinates, each relative to |ptx ax| + |pty ay| + |ukx| <= n + 44; inB = max (|G| + 2 |s G'|) (n + 44)
    inputs = aff * info["inB"] * (2 * (2 + 3) * u * 1.01)
    fast = aff * (tailU + 16 * u * info["rhoB"] + 2.01 * u * info["dB"])
    ref = aff * (12 * u * Sp + 96 * u * S) + pp
    nb = npoints // 128
    # mean: npoints - 1 sequential adds + a division (reference), 2 NB - 1 adds per lane + 6 shuffle levels + a product (fast); fixed-point subtraction 2 x 2^-33
    return 2 * (inputs + fast + ref) + (npoints + 2 * nb + 7) * u * 20480.0 + 2 * u * 8192.0 + 2.0 ** -32 * 1.001


def test_bound_matches_an_independent_derivation_and_fits_the_default_band():
    for cam in _cams()[:4]:
        tab, e0, bpo, info = _table(cam)
        tailU = 1.01 * _majorant_tail(cam, e0, tab.shape[0], bpo, tab.shape[1] - 1)
        assert abs(tailU - info["tailU"]) <= 0.02 * info["tailU"], (tailU, info["tailU"])
        for ds in (16, 32, 64):
            got, want = _lib_bound(cam, ds), _bound(cam, 2 * 8 * ds, info, tailU)
            assert abs(got - want) <= 0.02 * want, (got, want)
            assert 0 < got <= 0.5 * 2.0 ** -24


def test_cameras_the_fast_pass_must_not_serve_get_an_infinite_or_large_bound():
    cam = dict(synth.lafida_cameras()[0])
    bad = dict(cam)
    bad["p"] = [0.0] + list(cam["p"][1:])
    assert not np.isfinite(_lib_bound(bad, 32))
    bad = dict(cam)
    bad["invP"] = [v * 1e9 for v in cam["invP"]]
    assert _lib_bound(bad, 32) > 0.5 * 2.0 ** -24          # huge coefficients: beyond the default band, exact pass only
    bad = dict(cam)
    bad["invP"] = list(cam["invP"][:-1]) + [float("nan")]
    assert not np.isfinite(_lib_bound(bad, 32))


def test_device_form_of_the_table_and_its_float_tail_term():
    """mcs_describe_fast_table_packed: the rows as the DEVICE reads them.  Default build: the host's rows of doubles, unchanged, and a zero float term.  A build with
    MCS_G_PACKED=1 (48-byte rows: g0 .. g3 doubles, g4 .. g6 floats, kept as an A/B switch — it measured slower, profiles/NOTES.md round 6): the doubles must be the
    table's, the floats its coefficients rounded to nearest, the row polynomial with the tail evaluated in float arithmetic must stay within tailU + f32 term of G, and
    the term must match its independent derivation (6 relative roundings of 2^-24 on sum |g_j| |tau|^j, j = 4 .. 6, times sqrt(s))."""
    rng = np.random.default_rng(5)
    for cam in _cams()[:4]:
        tab, e0, bpo, info = _table(cam)
        oc = mcs.make_ocam(cam)
        rb, f32 = C.c_int(), C.c_double()
        mcs.check(mcs.lib().mcs_describe_fast_table_packed(C.byref(oc), None, C.byref(rb), C.byref(f32)))
        raw = np.zeros(tab.shape[0] * rb.value, np.uint8)
        mcs.check(mcs.lib().mcs_describe_fast_table_packed(C.byref(oc), raw.ctypes.data_as(C.c_void_p), None, None))
        if rb.value == 8 * tab.shape[1]:
            assert np.array_equal(raw.view(np.float64).reshape(tab.shape), tab) and f32.value == 0.0
            continue
        assert rb.value == 48 and tab.shape[1] == 7
        rows = raw.reshape(-1, 48)
        d4 = rows[:, :32].copy().view(np.float64)
        f4 = rows[:, 32:].copy().view(np.float32)
        assert np.array_equal(d4, tab[:, :4]) and np.array_equal(f4[:, :3], tab[:, 4:].astype(np.float32)) and not f4[:, 3].any()
        m = int(np.log2(bpo))
        noct = tab.shape[0] // bpo
        s = np.exp2(rng.uniform(e0, e0 + noct, 40000))
        bits = s.view(np.uint64)
        idx = ((bits >> np.uint64(32)).astype(np.int64) >> (20 - m)) - ((1023 + e0) << m)
        frac_bits = (bits & np.uint64((1 << (52 - m)) - 1)) | np.uint64(0x3FF << 52)
        tau = frac_bits.view(np.float64) - (1.0 + 1.0 / (2 << m))
        tf = tau.astype(np.float32)
        # two float FMAs: each product + sum formed in double (exact to well below a float ulp), rounded to float once
        t1 = (f4[idx, 2].astype(np.float64) * tf + f4[idx, 1]).astype(np.float32)
        tl = (t1.astype(np.float64) * tf + f4[idx, 0]).astype(np.float32)
        val = tl.astype(LD)
        for j in (3, 2, 1, 0):
            val = val * tau.astype(LD) + d4[idx, j].astype(LD)
        err = float((np.abs(val - _G(cam, s)) * np.sqrt(s.astype(LD))).max())
        u = 2.0 ** -53
        assert err <= info["tailU"] + f32.value + 8 * u * info["rhoB"], (err, info["tailU"], f32.value)
        # the term itself, row by row
        tm = 2.0 ** -(m + 1)
        t4 = sum(np.abs(tab[:, j]) * tm ** j for j in (4, 5, 6))
        r = np.arange(tab.shape[0])
        kappa = 1.0 + (r % bpo + 0.5) / bpo
        sq = np.sqrt(kappa * np.exp2(e0 + r // bpo) * (1 + tm / kappa))
        want = float((sq * (6 * 1.01 * 2.0 ** -24 * t4 + 1e-37)).max()) * 1.01
        assert abs(want - f32.value) <= 0.02 * want, (want, f32.value)
        assert f32.value < 2e-9

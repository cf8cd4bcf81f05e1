"""-m gpu: Hamming top-K kernel vs numpy brute force / the oracle's distance functions.  Bit-exact (integers)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_common
    return gpu_common


def _popcnt(a):
    return np.unpackbits(a, axis=-1).sum(-1).astype(np.int32)


def _brute(qd, td, qm=None, tm=None):
    x = qd[:, None, :] ^ td[None, :, :]
    if qm is None:
        return _popcnt(x)
    return (_popcnt(x & qm[:, None, :]) + _popcnt(x & tm[None, :, :])) // 2


def _topk_ref(D, K, elig=None):
    nq, nt = D.shape
    key = D.astype(np.int64) * (1 << 20) + np.arange(nt)[None, :]
    if elig is not None:
        key = np.where(elig, key, np.int64(1) << 40)
    if nt < K:
        key = np.concatenate([key, np.full((nq, K - nt), np.int64(1) << 40)], 1)
    order = np.argsort(key, axis=1, kind="stable")[:, :K]
    kk = np.take_along_axis(key, order, 1)
    dist = np.where(kk >= (np.int64(1) << 40), 0x7FFFFFFF, kk >> 20).astype(np.int32)
    idx = np.where(kk >= (np.int64(1) << 40), -1, order).astype(np.int32)
    return dist, idx


def test_known_answers(G):
    c = G.ctx()
    z = np.zeros(32, np.uint8)
    o = np.full(32, 255, np.uint8)
    assert c.descriptor_distance(z, z) == 0 and c.descriptor_distance(z, o) == 256
    a = z.copy(); a[0] = 1
    m = z.copy(); m[0] = 1
    assert c.descriptor_distance(a, z, m, z) == 0        # one differing bit, one mask covers it: 1/2 = 0
    assert c.descriptor_distance(a, z, m, m) == 1
    rng = np.random.default_rng(0)
    for _ in range(20):
        x, y, mx, my = (rng.integers(0, 256, 32).astype(np.uint8) for _ in range(4))
        assert c.descriptor_distance(x, y) == G.O.lib().orc_dist64(G.O.ptr(x), G.O.ptr(y), 32)
        assert c.descriptor_distance(x, y, mx, my) == G.O.lib().orc_dist64_masked(G.O.ptr(x), G.O.ptr(y), G.O.ptr(mx), G.O.ptr(my), 32)


@pytest.mark.parametrize("dim,K,masked,nq,nt", [(32, 8, False, 300, 1000), (32, 8, True, 777, 3001), (16, 2, True, 64, 257),
                                                (64, 16, False, 513, 700), (32, 32, True, 100, 20), (32, 1, False, 5, 4000)])
def test_topk_matches_bruteforce(G, dim, K, masked, nq, nt):
    rng = np.random.default_rng(dim + K + nq)
    td = rng.integers(0, 256, (nt, dim)).astype(np.uint8)
    qd = td[rng.integers(0, nt, nq)] ^ (rng.integers(0, 256, (nq, dim)) & rng.integers(0, 256, (nq, dim)) & rng.integers(0, 256, (nq, dim))).astype(np.uint8)
    qm = rng.integers(0, 256, (nq, dim)).astype(np.uint8) if masked else None
    tm = rng.integers(0, 256, (nt, dim)).astype(np.uint8) if masked else None
    thr = dim * 2
    dist, idx, cnt = G.ctx().match_topk(qd, td, K, thr, qm=qm, tm=tm)
    D = _brute(qd, td, qm, tm)
    ed, ei = _topk_ref(D, K)
    assert G.first_diff(dist, ed) is None and G.first_diff(idx, ei) is None
    assert (cnt == (D <= thr).sum(1)).all()


def test_topk_valid_and_groups(G):
    rng = np.random.default_rng(9)
    nq, nt, dim, K = 400, 1500, 32, 8
    td = rng.integers(0, 256, (nt, dim)).astype(np.uint8)
    qd = rng.integers(0, 256, (nq, dim)).astype(np.uint8)
    qv = (rng.random(nq) < 0.7).astype(np.uint8)
    tv = (rng.random(nt) < 0.6).astype(np.uint8)
    qg = rng.integers(0, 3, nq).astype(np.int32)
    tg = rng.integers(0, 3, nt).astype(np.int32)
    dist, idx, cnt = G.ctx().match_topk(qd, td, K, 100, qvalid=qv, tvalid=tv, qgroup=qg, tgroup=tg)
    D = _brute(qd, td)
    elig = (tv[None, :] != 0) & (qg[:, None] == tg[None, :]) & (qv[:, None] != 0)
    ed, ei = _topk_ref(D, K, elig)
    assert G.first_diff(dist, ed) is None and G.first_diff(idx, ei) is None
    assert (cnt == ((D <= 100) & elig).sum(1)).all()


def test_empty_sets(G):
    d, i, c = G.ctx().match_topk(np.zeros((0, 32), np.uint8), np.zeros((10, 32), np.uint8), 4, 10)
    assert d.shape == (0, 4)
    d, i, c = G.ctx().match_topk(np.zeros((3, 32), np.uint8), np.zeros((0, 32), np.uint8), 4, 10)
    assert (i == -1).all() and (d == 0x7FFFFFFF).all() and (c == 0).all()


def test_shared_reciprocal_division_is_bit_identical():
    """k_describe divides x, y and -z by the same norm through one refined reciprocal; the library's self-test runs that form and the
    plain a / d on 8 M pseudo-random operand pairs on the device: no result may differ in any bit."""
    import ctypes as C
    import gpu_common as G
    for seed in (1, 2, 3, 4):
        bad = C.c_int32(-1)
        G.mcs.check(G.mcs.lib().mcs_selftest_shared_reciprocal(G.ctx().h, seed, 2_000_000, C.byref(bad)))
        assert bad.value == 0


def test_searches_with_empty_sides(G):
    """the three brute-force searches with no query rows / no train rows (the reference's loops simply do not run)"""
    import ctypes as C
    import importlib
    cap = importlib.import_module("multicol-slam_amd._capi")
    lib, ctx = G.mcs.lib(), G.ctx()
    rng = np.random.default_rng(0)
    P = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    d = rng.integers(0, 256, (5, 32), dtype=np.uint8)
    m = np.full((5, 32), 255, np.uint8)
    ok, cam = np.ones(5, np.uint8), np.zeros(5, np.int32)
    rays = np.tile(np.array([0.0, 0.0, 1.0]), (5, 1))
    E = np.zeros(9)
    empty = np.zeros((0, 32), np.uint8)
    for nq, nt in ((0, 5), (5, 0), (0, 0)):
        q = cap.DescSet(P(d if nq else empty), P(m if nq else empty), P(ok), P(cam), nq, 32)
        t = cap.DescSet(P(d if nt else empty), P(m if nt else empty), P(ok), P(cam), nt, 32)
        for mode in range(3):
            out = np.full(8, 7, np.int32); nm = np.full(1, 7, np.int32); fb = np.zeros(1, np.int32)
            if mode == 0:
                rc = lib.mcs_search_kf_kf(ctx.h, 1, C.byref(q), 0, C.byref(t), 0, 32, 0.9, 32, cap.MEM_HOST, P(out), P(nm), P(fb))
                n_out = nq
            elif mode == 1:
                rc = lib.mcs_search_kf_f(ctx.h, 1, C.byref(q), 0, C.byref(t), 0, 32, 0.9, 32, cap.MEM_HOST, P(out), P(nm), P(fb))
                n_out = nt
            else:
                rc = lib.mcs_search_triangulation(ctx.h, 1, C.byref(q), 0, C.byref(t), 0, P(rays), P(rays), P(E), 1, 32, 32, cap.MEM_HOST, P(out), P(nm), P(fb))
                n_out = nq
            cap.check(rc)
            assert nm[0] == 0 and (out[:n_out] == -1).all(), (nq, nt, mode)

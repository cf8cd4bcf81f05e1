"""-m gpu: the greedy resolution on inputs built to make it SEQUENTIAL — clusters of near-duplicate descriptors on both sides, so that almost every query's nearest rows
are taken by lower queries and its outcome depends on a long chain of earlier outcomes; K small enough that lists run out and exact rescans happen.  The fixpoint
form (k_greedy_jacobi, few set pairs) needs as many sweeps as the longest chain, the chunked form (k_greedy_spec) as many rounds as there are conflicts; both must
return the sequential loop's result (the oracle: src/cORBmatcher.cpp:885-966 and :179-323)."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def G():
    import gpu_common
    return gpu_common


def clustered(rng, n, nclusters, flips):
    """n 32-byte descriptors in `nclusters` clusters: a centre with `flips` random bits flipped per member"""
    centres = rng.integers(0, 256, (nclusters, 32), dtype=np.uint8)
    d = centres[rng.integers(0, nclusters, n)].copy()
    for i in range(n):
        for b in rng.integers(0, 256, flips):
            d[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return d


@pytest.mark.parametrize("K", [32, 4, 1])
@pytest.mark.parametrize("nq,nt,ncl,flips", [(700, 650, 6, 3), (300, 900, 2, 2), (1500, 1400, 40, 6)])
def test_chains_of_dependent_queries(G, K, nq, nt, ncl, flips):
    cap = importlib.import_module("multicol-slam_amd._capi")
    rng = np.random.default_rng(nq + 7 * K)
    shared = rng.integers(0, 256, (ncl, 32), dtype=np.uint8)
    def make(n):
        d = shared[rng.integers(0, ncl, n)].copy()
        for i in range(n):
            for b in rng.integers(0, 256, flips):
                d[i, b >> 3] ^= np.uint8(1 << (b & 7))
        return d
    dq, dt = make(nq), make(nt)
    vq, vt = (rng.random(nq) < 0.9).astype(np.uint8), (rng.random(nt) < 0.9).astype(np.uint8)
    ones_q, ones_t = np.full_like(dq, 255), np.full_like(dt, 255)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    lib, ctx = G.mcs.lib(), G.ctx()
    # SearchByBoW(KF,KF): ratio test — with near-duplicates most queries are rejected, the accepted ones take rows away from the rest
    for ratio in (0.9, 1.0):
        q = cap.DescSet(P(dq), None, P(vq), None, nq, 32)
        t = cap.DescSet(P(dt), None, P(vt), None, nt, 32)
        m12 = np.full(nq, -7, np.int32); nm = np.zeros(1, np.int32); fb = np.zeros(1, np.int32)
        cap.check(lib.mcs_search_kf_kf(ctx.h, 1, C.byref(q), 0, C.byref(t), 0, 32, ratio, K, cap.MEM_HOST, P(m12), P(nm), P(fb)))
        en, e12 = G.O.search_kf_kf(dq, ones_q, vq, dt, ones_t, vt, False, ratio)
        assert int(nm[0]) == en and np.array_equal(m12, e12), (K, ratio, int(nm[0]), en)
    # SearchByBoW(KF,F): no ratio test against the list's tail — every query takes its nearest free row: the longest chains
    q = cap.DescSet(P(dq), None, P(vq), None, nq, 32)
    t = cap.DescSet(P(dt), None, None, None, nt, 32)
    out = np.full(nt, -7, np.int32); nm = np.zeros(1, np.int32); fb = np.zeros(1, np.int32)
    cap.check(lib.mcs_search_kf_f(ctx.h, 1, C.byref(q), 0, C.byref(t), 0, 32, 0.9, K, cap.MEM_HOST, P(out), P(nm), P(fb)))
    en, eo = G.O.search_kf_f(dq, ones_q, vq, dt, ones_t, False, 0.9)
    assert int(nm[0]) == en and np.array_equal(out, eo), (K, int(nm[0]), en)
    if K == 1:
        assert int(fb[0]) > 0, "K = 1 on clustered data must force exact rescans"
    if os.environ.get("MCS_EXPECT_JACOBI_FALLBACK"):   # (set by test_fixpoint_budget_exhausted_...: the in-order pass reports every query)
        assert int(fb[0]) == nq, (int(fb[0]), nq)


def test_both_forms_of_the_greedy_pass_agree_on_the_chains():
    """the same searches in a fresh process with MCS_GREEDY_JACOBI=0 (the chunked form for one pair too)"""
    e = dict(os.environ, MCS_GREEDY_JACOBI="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_greedy_chains.py"), "-m", "gpu", "-q", "-x", "-k", "chains_of_dependent"],
                       env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]


def test_fixpoint_budget_exhausted_falls_back_to_the_in_order_pass(sweeps="1"):
    """MCS_JACOBI_MAX_SWEEPS = 1: the fixpoint loop is given up before it can converge (the first sweep always changes outcomes), and the set is resolved in order by exact rescans
    (k_greedy_jacobi's tail) — same outcomes as the sequential loop, the fallback counter says every query went that way (a loop that ends on its guard
    must not hand out whatever it holds)"""
    e = dict(os.environ, MCS_JACOBI_MAX_SWEEPS=sweeps, MCS_EXPECT_JACOBI_FALLBACK="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_greedy_chains.py"), "-m", "gpu", "-q", "-x", "-k", "chains_of_dependent"],
                       env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]

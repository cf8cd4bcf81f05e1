"""-m gpu: the database-sweep call shapes of the brute-force searches (BASELINE configs[2] / [4]) against the oracle, bit for bit.

What bench.py's db / rig workloads call and round 1 never compared with anything:
  * mcs_search_kf_f with nsets = 32 stored keyframes and ONE shared frame (pitchF_rows = 0), host and device memory kind, K in {1, 8, 32};
  * mcs_search_kf_f_sweep: several frames x 32 keyframes in one call (the relocalisation loop of src/cTracking.cpp:1125-1221 for a batch of frames);
  * mcs_search_triangulation with the current keyframe shared by 32 neighbour pairs (pitch1_rows = 0; cLocalMapping::CreateNewMapPoints,
    src/cLocalMapping.cpp:223-270), with one essential-matrix block for all pairs and with one block per pair (mcs_search_triangulation_sweep).
Every (keyframe, frame) pair is compared with its own oracle call: O.search_kf_f / O.search_triangulation (src/cORBmatcher.cpp:179-323 without the
vocabulary restriction, :968-1155).  Integer results: tolerance 0.  Rows beyond a set's size and rows flagged invalid are padding, as in the bench layout."""
import ctypes as C
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NKF, DIM = 32, 32


@pytest.fixture(scope="module")
def G():
    import gpu_common
    return gpu_common


@pytest.fixture(scope="module")
def cap():
    return importlib.import_module("multicol-slam_amd._capi")


def P(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def noisy_copies(rng, src, n, flip_bits):
    """n rows, each a copy of a random row of src with ~flip_bits random bits flipped; some source rows are used many times (contention for the greedy)"""
    pick = rng.integers(0, max(len(src) // 3, 1), n) if rng.random() < 0.5 else rng.integers(0, len(src), n)
    noise = np.packbits(rng.random((n, src.shape[1] * 8)) < flip_bits / (src.shape[1] * 8.0), axis=1)
    return src[pick] ^ noise, pick


def make_db(seed, nframes, nF, nK, pitchF, pitchK, masked):
    """frames [nframes][pitchF] and keyframes [NKF][pitchK] (descriptors, masks, valid); keyframe rows are noisy copies of rows of the frames"""
    rng = np.random.default_rng(seed)
    fd = rng.integers(0, 256, (nframes, pitchF, DIM), dtype=np.uint8)
    fm = np.packbits(rng.random((nframes, pitchF, DIM * 8)) < 0.9, axis=2) if masked else None
    fv = np.zeros((nframes, pitchF), np.uint8)
    fv[:, :nF] = (rng.random((nframes, nF)) < 0.9)
    kd = np.zeros((NKF, pitchK, DIM), np.uint8)
    km = np.packbits(rng.random((NKF, pitchK, DIM * 8)) < 0.9, axis=2) if masked else None
    kv = np.zeros((NKF, pitchK), np.uint8)
    for k in range(NKF):
        kd[k, :nK], _ = noisy_copies(rng, fd[k % nframes, :nF], nK, 10 + 2 * (k % 5))
        kv[k, :nK] = rng.random(nK) < 0.7          # "has a good map point"
    return fd, fm, fv, kd, km, kv


def oracle_kf_f(G, kd, km, kv, nK, fd, fm, fv, nF, masked, ratio):
    """one (keyframe, frame) pair through the oracle; invalid frame rows are padding: the oracle sees the frame without them"""
    keep = np.flatnonzero(fv[:nF])
    ones = np.full((max(nK, 1), DIM), 255, np.uint8)
    n, m = G.O.search_kf_f(np.ascontiguousarray(kd[:nK]), np.ascontiguousarray(km[:nK]) if masked else ones[:nK], np.ascontiguousarray(kv[:nK]),
                           np.ascontiguousarray(fd[keep]), np.ascontiguousarray(fm[keep]) if masked else np.full((len(keep), DIM), 255, np.uint8), masked, ratio)
    full = np.full(nF, -1, np.int32)
    full[keep] = m
    return n, full


@pytest.mark.parametrize("K", [1, 8, 32])
@pytest.mark.parametrize("masked,ratio", [(True, 0.9), (False, 0.75)])
def test_kf_f_32_keyframes_shared_frame_host_and_device(G, cap, K, masked, ratio):
    nF, nK, pitchK = 901, 777, 800
    fd, fm, fv, kd, km, kv = make_db(100 + K + int(masked), 1, nF, nK, nF, pitchK, masked)
    exp = [oracle_kf_f(G, kd[k], km[k] if masked else None, kv[k], nK, fd[0], fm[0] if masked else None, fv[0], nF, masked, ratio) for k in range(NKF)]
    exp_n = np.array([e[0] for e in exp], np.int32)
    exp_m = np.stack([e[1] for e in exp])
    assert exp_n.sum() > 2000
    lib, ctx = G.mcs.lib(), G.ctx()
    # host kind
    q = cap.DescSet(P(kd), P(km), P(kv), None, nK, DIM)
    t = cap.DescSet(P(fd), P(fm), P(fv), None, nF, DIM)
    mF = np.full((NKF, nF), -7, np.int32); nm = np.full(NKF, -7, np.int32); fb = np.zeros(NKF, np.int32)
    cap.check(lib.mcs_search_kf_f(ctx.h, NKF, C.byref(q), pitchK, C.byref(t), 0, DIM, ratio, K, cap.MEM_HOST, P(mF), P(nm), P(fb)))
    assert G.first_diff(nm, exp_n) is None and G.first_diff(mF, exp_m) is None, (K, masked)
    # device kind: the same arrays resident on the GPU, outputs read back after a full synchronisation
    bufs = [G.DevBuf(a) if a is not None else None for a in (kd, km, kv, fd, fm, fv)]
    dp = lambda b: None if b is None else C.c_void_p(b.ptr.value)
    qd_ = cap.DescSet(dp(bufs[0]), dp(bufs[1]), dp(bufs[2]), None, nK, DIM)
    td_ = cap.DescSet(dp(bufs[3]), dp(bufs[4]), dp(bufs[5]), None, nF, DIM)
    o_m, o_n, o_f = G.DevBuf(np.full((NKF, nF), -7, np.int32)), G.DevBuf(np.full(NKF, -7, np.int32)), G.DevBuf(np.zeros(NKF, np.int32))
    cap.check(lib.mcs_search_kf_f(ctx.h, NKF, C.byref(qd_), pitchK, C.byref(td_), 0, DIM, ratio, K, cap.MEM_DEVICE, o_m.ptr, o_n.ptr, o_f.ptr))
    ctx.synchronize()
    assert G.first_diff(o_n.read(), exp_n) is None and G.first_diff(o_m.read(), exp_m) is None, (K, masked, "device")
    if K == 1:
        assert o_f.read().sum() > 0   # single-entry lists cannot decide the ratio test: the exact rescans ran


@pytest.mark.parametrize("K", [1, 8, 32])
def test_kf_f_sweep_frames_x_keyframes(G, cap, K):
    masked, ratio = True, 0.9
    nframes, nF, nK, pitchF, pitchK = 3, 640, 600, 700, 640
    fd, fm, fv, kd, km, kv = make_db(7 + K, nframes, nF, nK, pitchF, pitchK, masked)
    exp_n = np.zeros((nframes, NKF), np.int32)
    exp_m = np.zeros((nframes, NKF, nF), np.int32)
    for f in range(nframes):
        for k in range(NKF):
            exp_n[f, k], exp_m[f, k] = oracle_kf_f(G, kd[k], km[k], kv[k], nK, fd[f], fm[f], fv[f], nF, masked, ratio)
    assert exp_n.sum() > 3000 and (exp_n.min(axis=1) >= 0).all()
    lib, ctx = G.mcs.lib(), G.ctx()
    q = cap.DescSet(P(kd), P(km), P(kv), None, nK, DIM)
    t = cap.DescSet(P(fd), P(fm), P(fv), None, nF, DIM)
    mF = np.full((nframes, NKF, nF), -7, np.int32); nm = np.full((nframes, NKF), -7, np.int32); fb = np.zeros((nframes, NKF), np.int32)
    cap.check(lib.mcs_search_kf_f_sweep(ctx.h, NKF, C.byref(q), pitchK, nframes, C.byref(t), pitchF, DIM, ratio, K, cap.MEM_HOST, P(mF), P(nm), P(fb)))
    assert G.first_diff(nm, exp_n) is None and G.first_diff(mF, exp_m) is None
    bufs = [G.DevBuf(a) for a in (kd, km, kv, fd, fm, fv)]
    dp = lambda b: C.c_void_p(b.ptr.value)
    qd_ = cap.DescSet(dp(bufs[0]), dp(bufs[1]), dp(bufs[2]), None, nK, DIM)
    td_ = cap.DescSet(dp(bufs[3]), dp(bufs[4]), dp(bufs[5]), None, nF, DIM)
    o_m, o_n = G.DevBuf(np.full((nframes, NKF, nF), -7, np.int32)), G.DevBuf(np.full((nframes, NKF), -7, np.int32))
    cap.check(lib.mcs_search_kf_f_sweep(ctx.h, NKF, C.byref(qd_), pitchK, nframes, C.byref(td_), pitchF, DIM, ratio, K, cap.MEM_DEVICE, o_m.ptr, o_n.ptr, None))
    ctx.synchronize()
    assert G.first_diff(o_n.read(), exp_n) is None and G.first_diff(o_m.read(), exp_m) is None
    # a one-frame sweep is the shared-frame batch
    one = np.full((NKF, nF), -7, np.int32); n1 = np.zeros(NKF, np.int32)
    t1 = cap.DescSet(P(fd[1]), P(fm[1]), P(fv[1]), None, nF, DIM)
    cap.check(lib.mcs_search_kf_f_sweep(ctx.h, NKF, C.byref(q), pitchK, 1, C.byref(t1), 0, DIM, ratio, K, cap.MEM_HOST, P(one), P(n1), None))
    assert G.first_diff(one, exp_m[1]) is None


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


@pytest.mark.parametrize("K", [1, 8, 32])
@pytest.mark.parametrize("masked", [True, False])
def test_triangulation_shared_current_keyframe_32_neighbours(G, cap, K, masked):
    """the current keyframe (set 1, its rays) is shared by 32 (current, neighbour) pairs; essential matrices per pair or shared"""
    rng = np.random.default_rng(900 + K + int(masked))
    n1, n2, pitch2, NC = 700, 650, 704, 3
    d1 = rng.integers(0, 256, (n1, DIM), dtype=np.uint8)
    m1 = np.packbits(rng.random((n1, DIM * 8)) < 0.9, axis=1) if masked else None
    mp1 = (rng.random(n1) < 0.4).astype(np.uint8)            # has a map point -> not a query
    cam1 = rng.integers(0, NC, n1).astype(np.int32)
    rays1 = _unit(rng.normal(size=(n1, 3)) * [0.5, 0.5, 0.2] + [0, 0, 1.0])
    d2 = np.zeros((NKF, pitch2, DIM), np.uint8)
    m2 = np.packbits(rng.random((NKF, pitch2, DIM * 8)) < 0.9, axis=2) if masked else None
    mp2 = np.ones((NKF, pitch2), np.uint8)
    cam2 = np.zeros((NKF, pitch2), np.int32)
    rays2 = np.zeros((NKF, pitch2, 3))
    rays2[..., 2] = 1.0
    for k in range(NKF):
        d2[k, :n2], pick = noisy_copies(rng, d1, n2, 8 + k % 6)
        mp2[k, :n2] = rng.random(n2) < 0.3
        cam2[k, :n2] = np.where(rng.random(n2) < 0.85, cam1[pick], rng.integers(0, NC, n2))
        rays2[k, :n2] = _unit(rays1[pick] + rng.normal(size=(n2, 3)) * np.where(rng.random((n2, 1)) < 0.6, 0.003, 0.2))
    E = rng.normal(size=(NKF, NC, NC, 3, 3))
    for k in range(NKF):
        for c in range(NC):
            t = np.array([0.05 + 0.01 * k, 0.01 * c, 0.002 * k])
            E[k, c, c] = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    v1, v2 = (1 - mp1).astype(np.uint8), (1 - mp2).astype(np.uint8)
    ones1, ones2 = np.full((n1, DIM), 255, np.uint8), np.full((n2, DIM), 255, np.uint8)

    def oracle(k, Ek):
        return G.O.search_triangulation(d1, m1 if masked else ones1, mp1, cam1, np.ascontiguousarray(rays1), np.ascontiguousarray(d2[k, :n2]),
                                        np.ascontiguousarray(m2[k, :n2]) if masked else ones2, np.ascontiguousarray(mp2[k, :n2]),
                                        np.ascontiguousarray(cam2[k, :n2]), np.ascontiguousarray(rays2[k, :n2]), np.ascontiguousarray(Ek.reshape(NC * NC, 9)), NC, masked)

    lib, ctx = G.mcs.lib(), G.ctx()
    q = cap.DescSet(P(d1), P(m1), P(v1), P(cam1), n1, DIM)
    t = cap.DescSet(P(d2), P(m2), P(v2), P(cam2), n2, DIM)
    for per_pair in (True, False):
        exp = [oracle(k, E[k] if per_pair else E[5]) for k in range(NKF)]
        exp_n = np.array([e[0] for e in exp], np.int32); exp_m = np.stack([e[1] for e in exp])
        if per_pair:
            assert exp_n.sum() > 1500
        m12 = np.full((NKF, n1), -7, np.int32); nm = np.full(NKF, -7, np.int32); fb = np.zeros(NKF, np.int32)
        if per_pair:
            cap.check(lib.mcs_search_triangulation_sweep(ctx.h, NKF, C.byref(q), 0, C.byref(t), pitch2, P(rays1), P(rays2), P(E), NC * NC * 9, NC, DIM, K,
                                                         cap.MEM_HOST, P(m12), P(nm), P(fb)))
        else:
            Es = np.ascontiguousarray(E[5])
            cap.check(lib.mcs_search_triangulation(ctx.h, NKF, C.byref(q), 0, C.byref(t), pitch2, P(rays1), P(rays2), P(Es), NC, DIM, K, cap.MEM_HOST,
                                                   P(m12), P(nm), P(fb)))
        assert G.first_diff(nm, exp_n) is None and G.first_diff(m12, exp_m) is None, (K, masked, per_pair)
        # device kind
        arrs = (d1, m1, v1, cam1, d2, m2, v2, cam2, np.ascontiguousarray(rays1), rays2, E if per_pair else np.ascontiguousarray(E[5]))
        bufs = [G.DevBuf(a) if a is not None else None for a in arrs]
        dp = lambda b: None if b is None else C.c_void_p(b.ptr.value)
        qd_ = cap.DescSet(dp(bufs[0]), dp(bufs[1]), dp(bufs[2]), dp(bufs[3]), n1, DIM)
        td_ = cap.DescSet(dp(bufs[4]), dp(bufs[5]), dp(bufs[6]), dp(bufs[7]), n2, DIM)
        o_m, o_n = G.DevBuf(np.full((NKF, n1), -7, np.int32)), G.DevBuf(np.full(NKF, -7, np.int32))
        cap.check(lib.mcs_search_triangulation_sweep(ctx.h, NKF, C.byref(qd_), 0, C.byref(td_), pitch2, dp(bufs[8]), dp(bufs[9]), dp(bufs[10]),
                                                     NC * NC * 9 if per_pair else 0, NC, DIM, K, cap.MEM_DEVICE, o_m.ptr, o_n.ptr, None))
        ctx.synchronize()
        assert G.first_diff(o_n.read(), exp_n) is None and G.first_diff(o_m.read(), exp_m) is None, (K, masked, per_pair, "device")


def test_sweep_argument_checks(G, cap):
    lib, ctx = G.mcs.lib(), G.ctx()
    d = np.zeros((4, DIM), np.uint8)
    q = cap.DescSet(P(d), None, None, None, 4, DIM)
    out = np.zeros(64, np.int32)
    assert lib.mcs_search_kf_f_sweep(ctx.h, 0, C.byref(q), 4, 1, C.byref(q), 4, DIM, 0.9, 8, cap.MEM_HOST, P(out), P(out), None) == cap.MCS_ERR_INVALID
    assert lib.mcs_search_kf_f_sweep(ctx.h, 2, C.byref(q), 2, 1, C.byref(q), 4, DIM, 0.9, 8, cap.MEM_HOST, P(out), P(out), None) == cap.MCS_ERR_INVALID
    rays = np.zeros((4, 3)); E = np.zeros(9)
    g = cap.DescSet(P(d), None, None, P(np.zeros(4, np.int32)), 4, DIM)
    assert lib.mcs_search_triangulation_sweep(ctx.h, 2, C.byref(g), 0, C.byref(g), 0, P(rays), P(rays), P(E), 5, 1, DIM, 8, cap.MEM_HOST, P(out), P(out), None) == cap.MCS_ERR_INVALID


def test_deferred_searches_equal_in_order_searches(G, cap):
    """mcs_ctx_set_async_search: lists + greedy pass of device-memory searches run on the library's own stream; three searches on different inputs are
    issued back to back without any synchronisation, mcs_ctx_search_fence orders the caller's stream behind them.  Results = the in-order results."""
    lib = G.mcs.lib()
    ctx = G.mcs.Context(0)
    nF, nK, pitchK = 640, 600, 640
    jobs = []
    for seed in (1, 2, 3):
        fd, fm, fv, kd, km, kv = make_db(40 + seed, 1, nF, nK, nF, pitchK, True)
        bufs = [G.DevBuf(a) for a in (kd, km, kv, fd, fm, fv)]
        dp = lambda b: C.c_void_p(b.ptr.value)
        q = cap.DescSet(dp(bufs[0]), dp(bufs[1]), dp(bufs[2]), None, nK, DIM)
        t = cap.DescSet(dp(bufs[3]), dp(bufs[4]), dp(bufs[5]), None, nF, DIM)
        jobs.append((bufs, q, t))
    results = {}
    for mode in (0, 1):
        cap.check(lib.mcs_ctx_set_async_search(ctx.h, mode))
        outs = []
        for bufs, q, t in jobs:
            o_m, o_n = G.DevBuf(np.full((NKF, nF), -7, np.int32)), G.DevBuf(np.full(NKF, -7, np.int32))
            cap.check(lib.mcs_search_kf_f(ctx.h, NKF, C.byref(q), pitchK, C.byref(t), 0, DIM, 0.9, 8, cap.MEM_DEVICE, o_m.ptr, o_n.ptr, None))
            outs.append((o_m, o_n))
        for lag in (2, 1, 0):
            cap.check(lib.mcs_ctx_search_fence(ctx.h, lag))
        ctx.synchronize()
        results[mode] = [(m.read(), n.read()) for m, n in outs]
    cap.check(lib.mcs_ctx_set_async_search(ctx.h, 0))
    for (m0, n0), (m1, n1) in zip(results[0], results[1]):
        assert n0.sum() > 1000 and np.array_equal(n0, n1) and np.array_equal(m0, m1)
    assert lib.mcs_ctx_search_fence(ctx.h, 3) == cap.MCS_ERR_INVALID
    ctx.close()

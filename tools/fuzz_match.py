#!/usr/bin/env python3
"""Randomised matcher parity on the GPU box (not a test: `python tools/fuzz_match.py [seconds] [seed]`).  Random set sizes (0, 1, ragged, thousands), descriptor
sizes, K, masks, eligibility flags, ratios; descriptors drawn as clusters of near-duplicates (ties everywhere: the order is by train index, the greedy pass
runs long chains and exact rescans) or as random rows; one or several set pairs per call.
  * mcs_match_topk (Context.match_topk): distances and indices against numpy brute force;
  * mcs_search_kf_kf / mcs_search_kf_f with nsets pairs laid out at a row pitch: every pair against the oracle's sequential loops (src/cORBmatcher.cpp:885-966, :179-323).
One line per failing case with what reproduces it; exit code 1 if any failed.  The oracle is the checker here, as in tests/."""
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
mcs = G.mcs
cap = importlib.import_module("multicol-slam_amd._capi")
P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731


def popcnt(a):
    return np.unpackbits(a, axis=-1).sum(-1).astype(np.int32)


def brute(qd, td, qm=None, tm=None):
    out = np.empty((len(qd), len(td)), np.int32)
    for s in range(0, len(qd), 32):
        x = qd[s:s + 32, None, :] ^ td[None, :, :]
        out[s:s + 32] = popcnt(x) if qm is None else (popcnt(x & qm[s:s + 32, None, :]) + popcnt(x & tm[None, :, :])) // 2
    return out


def topk_ref(D, K, elig):
    nq, nt = D.shape
    big = np.int64(1) << 40
    key = D.astype(np.int64) * (1 << 20) + np.arange(nt)[None, :]
    key = np.where(elig, key, big)
    if nt < K:
        key = np.concatenate([key, np.full((nq, K - nt), big)], 1)
    order = np.argsort(key, axis=1, kind="stable")[:, :K]
    kk = np.take_along_axis(key, order, 1)
    return np.where(kk >= big, 0x7FFFFFFF, kk >> 20).astype(np.int32), np.where(kk >= big, -1, order).astype(np.int32)


def rows(rng, n, dim, style, protos):
    if style == "random" or n == 0:
        return rng.integers(0, 256, (n, dim)).astype(np.uint8)
    d = protos[rng.integers(0, len(protos), n)].copy()
    flips = int(rng.choice([0, 1, 2, 6]))
    for _ in range(flips):
        b = rng.integers(0, 8 * dim, n)
        d[np.arange(n), b >> 3] ^= (1 << (b & 7)).astype(np.uint8)
    return d


def size(rng):
    r = rng.random()
    if r < 0.08:
        return int(rng.integers(0, 3))
    if r < 0.5:
        return int(rng.integers(3, 400))
    return int(rng.integers(400, 3300))


def case_topk(rng, idx):
    dim = int(rng.choice([32, 32, 16, 64]))
    K = int(rng.choice([1, 2, 4, 8, 16, 32]))
    nq, nt = max(size(rng), 1), max(size(rng), 1)
    style = str(rng.choice(["random", "clusters"]))
    protos = rng.integers(0, 256, (int(rng.integers(1, 40)), dim)).astype(np.uint8)
    masked = rng.random() < 0.5
    qd, td = rows(rng, nq, dim, style, protos), rows(rng, nt, dim, style, protos)
    qm = rng.integers(0, 256, (nq, dim)).astype(np.uint8) if masked else None
    tm = rng.integers(0, 256, (nt, dim)).astype(np.uint8) if masked else None
    qv = (rng.random(nq) < 0.9).astype(np.uint8) if rng.random() < 0.5 else None
    tv = (rng.random(nt) < 0.8).astype(np.uint8) if rng.random() < 0.5 else None
    desc = "topk case %d: dim=%d K=%d nq=%d nt=%d %s masked=%d qv=%d tv=%d" % (idx, dim, K, nq, nt, style, masked, qv is not None, tv is not None)
    dist, ix, _ = G.ctx().match_topk(qd, td, K, -1, qm=qm, tm=tm, qvalid=qv, tvalid=tv)
    elig = np.ones((nq, nt), bool)
    if qv is not None:
        elig &= qv[:, None] != 0
    if tv is not None:
        elig &= tv[None, :] != 0
    ed, ei = topk_ref(brute(qd, td, qm, tm), K, elig)
    e = G.first_diff(dist, ed) or G.first_diff(ix, ei)
    return (desc + " -> " + e) if e else None


def case_search(rng, idx):
    dim = int(rng.choice([32, 32, 64]))
    K = int(rng.choice([1, 2, 8, 32, 32]))
    nsets = int(rng.choice([1, 1, 2, 3, 6, 12]))
    nq, nt = size(rng), size(rng)
    if nsets > 3:
        nq, nt = min(nq, 900), min(nt, 900)
    style = str(rng.choice(["random", "clusters", "clusters"]))
    masked = rng.random() < 0.5
    ratio = float(rng.choice([0.6, 0.8, 0.9, 1.0]))
    pq, pt = nq + int(rng.integers(0, 5)), nt + int(rng.integers(0, 5))   # row pitch between the sets of a call
    protos = rng.integers(0, 256, (int(rng.integers(1, 40)), dim)).astype(np.uint8)
    Q = np.zeros((nsets, max(pq, 1), dim), np.uint8); T = np.zeros((nsets, max(pt, 1), dim), np.uint8)
    QM = np.full_like(Q, 255); TM = np.full_like(T, 255)
    QV = np.zeros((nsets, max(pq, 1)), np.uint8); TV = np.zeros((nsets, max(pt, 1)), np.uint8)
    for s in range(nsets):
        Q[s, :nq], T[s, :nt] = rows(rng, nq, dim, style, protos), rows(rng, nt, dim, style, protos)
        if masked:
            QM[s, :nq], TM[s, :nt] = rng.integers(0, 256, (nq, dim)), rng.integers(0, 256, (nt, dim))
        QV[s, :nq], TV[s, :nt] = rng.random(nq) < 0.9, rng.random(nt) < 0.85
    desc = "search case %d: dim=%d K=%d nsets=%d nq=%d nt=%d pitch=%d/%d %s masked=%d ratio=%.2f" % (idx, dim, K, nsets, nq, nt, pq, pt, style, masked, ratio)
    lib, ctx = mcs.lib(), G.ctx()
    q = cap.DescSet(P(Q), P(QM) if masked else None, P(QV), None, nq, dim)
    t = cap.DescSet(P(T), P(TM) if masked else None, P(TV), None, nt, dim)
    m12 = np.full((nsets, max(nq, 1)), -7, np.int32); nm = np.zeros(nsets, np.int32); fb = np.zeros(nsets, np.int32)
    rc = lib.mcs_search_kf_kf(ctx.h, nsets, C.byref(q), Q.shape[1], C.byref(t), T.shape[1], dim, ratio, K, cap.MEM_HOST, P(m12), P(nm), P(fb))
    if rc != 0:
        return desc + " -> mcs_search_kf_kf rc %d (%s)" % (rc, lib.mcs_last_error().decode()[:120])
    for s in range(nsets):
        en, e12 = O.search_kf_kf(np.ascontiguousarray(Q[s, :nq]), np.ascontiguousarray(QM[s, :nq]), np.ascontiguousarray(QV[s, :nq]),
                                 np.ascontiguousarray(T[s, :nt]), np.ascontiguousarray(TM[s, :nt]), np.ascontiguousarray(TV[s, :nt]), masked, ratio)
        if int(nm[s]) != en or not np.array_equal(m12.reshape(-1)[s * nq:(s + 1) * nq] if nq else m12[s, :0], e12):
            return desc + " -> kf_kf set %d: %d matches, oracle %d" % (s, int(nm[s]), en)
    t2 = cap.DescSet(P(T), P(TM) if masked else None, None, None, nt, dim)
    out = np.full((nsets, max(nt, 1)), -7, np.int32)
    rc = lib.mcs_search_kf_f(ctx.h, nsets, C.byref(q), Q.shape[1], C.byref(t2), T.shape[1], dim, ratio, K, cap.MEM_HOST, P(out), P(nm), P(fb))
    if rc != 0:
        return desc + " -> mcs_search_kf_f rc %d (%s)" % (rc, lib.mcs_last_error().decode()[:120])
    for s in range(nsets):
        en, eo = O.search_kf_f(np.ascontiguousarray(Q[s, :nq]), np.ascontiguousarray(QM[s, :nq]), np.ascontiguousarray(QV[s, :nq]),
                               np.ascontiguousarray(T[s, :nt]), np.ascontiguousarray(TM[s, :nt]), masked, ratio)
        if int(nm[s]) != en or not np.array_equal(out.reshape(-1)[s * nt:(s + 1) * nt] if nt else out[s, :0], eo):
            return desc + " -> kf_f set %d: %d matches, oracle %d" % (s, int(nm[s]), en)
    return None


def case_sweep(rng, idx):
    """mcs_search_kf_f_sweep: nframes frames x nkf keyframes in one call (pair f*nkf + k), host memory; mcs_search_kf_kf_ring: every frame of a ring against the one
    before it (device memory), from a random first frame"""
    dim = 32
    K = int(rng.choice([2, 8, 32]))
    nkf, nfr = int(rng.integers(1, 7)), int(rng.integers(1, 6))
    nk, nf = min(size(rng), 1200), min(size(rng), 1200)
    style = str(rng.choice(["random", "clusters", "clusters"]))
    masked = rng.random() < 0.5
    ratio = float(rng.choice([0.7, 0.9, 1.0]))
    pk, pf = nk + int(rng.integers(0, 4)), nf + int(rng.integers(0, 4))
    protos = rng.integers(0, 256, (int(rng.integers(1, 40)), dim)).astype(np.uint8)
    KD = np.zeros((nkf, max(pk, 1), dim), np.uint8); FD = np.zeros((nfr, max(pf, 1), dim), np.uint8)
    KM = np.full_like(KD, 255); FM = np.full_like(FD, 255)
    KV = np.zeros((nkf, max(pk, 1)), np.uint8)
    for k in range(nkf):
        KD[k, :nk] = rows(rng, nk, dim, style, protos); KV[k, :nk] = rng.random(nk) < 0.9
        if masked:
            KM[k, :nk] = rng.integers(0, 256, (nk, dim))
    for f in range(nfr):
        FD[f, :nf] = rows(rng, nf, dim, style, protos)
        if masked:
            FM[f, :nf] = rng.integers(0, 256, (nf, dim))
    desc = "sweep case %d: K=%d nkf=%d nframes=%d nk=%d nf=%d pitch=%d/%d %s masked=%d ratio=%.2f" % (idx, K, nkf, nfr, nk, nf, pk, pf, style, masked, ratio)
    lib, ctx = mcs.lib(), G.ctx()
    q = cap.DescSet(P(KD), P(KM) if masked else None, P(KV), None, nk, dim)
    t = cap.DescSet(P(FD), P(FM) if masked else None, None, None, nf, dim)
    out = np.full((nfr * nkf, max(nf, 1)), -7, np.int32); nm = np.zeros(nfr * nkf, np.int32); fb = np.zeros(nfr * nkf, np.int32)
    rc = lib.mcs_search_kf_f_sweep(ctx.h, nkf, C.byref(q), KD.shape[1], nfr, C.byref(t), FD.shape[1], dim, ratio, K, cap.MEM_HOST, P(out), P(nm), P(fb))
    if rc != 0:
        return desc + " -> mcs_search_kf_f_sweep rc %d (%s)" % (rc, lib.mcs_last_error().decode()[:120])
    for f in range(nfr):
        for k in range(nkf):
            s_ = f * nkf + k
            en, eo = O.search_kf_f(np.ascontiguousarray(KD[k, :nk]), np.ascontiguousarray(KM[k, :nk]), np.ascontiguousarray(KV[k, :nk]),
                                   np.ascontiguousarray(FD[f, :nf]), np.ascontiguousarray(FM[f, :nf]), masked, ratio)
            if int(nm[s_]) != en or not np.array_equal(out.reshape(-1)[s_ * nf:(s_ + 1) * nf] if nf else out[s_, :0], eo):
                return desc + " -> pair (frame %d, keyframe %d): %d matches, oracle %d" % (f, k, int(nm[s_]), en)
    # the ring over the keyframe array (eligibility on both sides), device memory
    if nk == 0 or nkf < 2:   # (a ring has at least two frames; first + count <= nframes_total: only the predecessor wraps)
        return None
    first = int(rng.integers(0, nkf)); count = int(rng.integers(1, nkf - first + 1))
    dK, dM, dV = G.DevBuf(KD), G.DevBuf(KM), G.DevBuf(KV)
    fr = cap.DescSet(dK.ptr, dM.ptr if masked else None, dV.ptr, None, nk, dim)
    m12 = G.DevBuf(np.full((count, nk), -7, np.int32)); dn = G.DevBuf(np.zeros(count, np.int32)); dfb = G.DevBuf(np.zeros(count, np.int32))
    rc = lib.mcs_search_kf_kf_ring(ctx.h, nkf, first, count, C.byref(fr), KD.shape[1], dim, ratio, K, cap.MEM_DEVICE, m12.ptr, dn.ptr, dfb.ptr)
    if rc != 0:
        return desc + " -> mcs_search_kf_kf_ring rc %d (%s)" % (rc, lib.mcs_last_error().decode()[:120])
    G.ctx().synchronize()
    gm, gn = m12.read(), dn.read()
    for s_ in range(count):
        a, b = (first + s_) % nkf, (first + s_ - 1) % nkf
        en, e12 = O.search_kf_kf(np.ascontiguousarray(KD[a, :nk]), np.ascontiguousarray(KM[a, :nk]), np.ascontiguousarray(KV[a, :nk]),
                                 np.ascontiguousarray(KD[b, :nk]), np.ascontiguousarray(KM[b, :nk]), np.ascontiguousarray(KV[b, :nk]), masked, ratio)
        if int(gn[s_]) != en or not np.array_equal(gm[s_], e12):
            return desc + " -> ring first=%d count=%d pair %d (frames %d, %d): %d matches, oracle %d" % (first, count, s_, a, b, int(gn[s_]), en)
    return None


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def case_triangulation(rng, idx):
    """mcs_search_triangulation: queries = features WITHOUT a map point, camera groups, the epipolar test on rays and essential matrices (src/cORBmatcher.cpp:968-1155);
    nsets neighbour keyframes against one shared current keyframe (pitch1 = 0) or pairwise"""
    dim = 32
    K = int(rng.choice([2, 8, 32]))
    NC = int(rng.integers(1, 5))
    nsets = int(rng.choice([1, 1, 3, 8]))
    n1, n2 = min(max(size(rng), 1), 1500), min(max(size(rng), 1), 1500)
    masked = rng.random() < 0.5
    shared = rng.random() < 0.5
    s1 = 1 if shared else nsets
    p1, p2 = n1 + int(rng.integers(0, 4)), n2 + int(rng.integers(0, 4))
    d1 = rng.integers(0, 256, (s1, p1, dim), dtype=np.uint8)
    m1 = rng.integers(0, 256, (s1, p1, dim), dtype=np.uint8) if masked else np.full((s1, p1, dim), 255, np.uint8)
    mp1 = (rng.random((s1, p1)) < 0.4).astype(np.uint8)
    cam1 = rng.integers(0, NC, (s1, p1)).astype(np.int32)
    rays1 = _unit(rng.normal(size=(s1, p1, 3)) * [0.5, 0.5, 0.2] + [0, 0, 1.0])
    d2 = np.zeros((nsets, p2, dim), np.uint8)
    m2 = rng.integers(0, 256, (nsets, p2, dim), dtype=np.uint8) if masked else np.full((nsets, p2, dim), 255, np.uint8)
    mp2 = (rng.random((nsets, p2)) < 0.3).astype(np.uint8)
    cam2 = np.zeros((nsets, p2), np.int32)
    rays2 = np.zeros((nsets, p2, 3)); rays2[..., 2] = 1.0
    for k in range(nsets):
        a = 0 if shared else k
        pick = rng.integers(0, n1, n2)
        flip = (rng.random((n2, dim)) < 0.06) * rng.integers(0, 256, (n2, dim))
        d2[k, :n2] = d1[a, pick] ^ flip.astype(np.uint8)
        cam2[k, :n2] = np.where(rng.random(n2) < 0.85, cam1[a, pick], rng.integers(0, NC, n2))
        rays2[k, :n2] = _unit(rays1[a, pick] + rng.normal(size=(n2, 3)) * np.where(rng.random((n2, 1)) < 0.6, 0.003, 0.2))
    E = rng.normal(size=(NC, NC, 3, 3))
    for c in range(NC):
        t = np.array([0.05, 0.01 * c, 0.002])
        E[c, c] = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = np.ascontiguousarray(E)
    v1, v2 = (1 - mp1).astype(np.uint8), (1 - mp2).astype(np.uint8)
    desc = "triangulation case %d: K=%d NC=%d nsets=%d n1=%d n2=%d shared=%d masked=%d" % (idx, K, NC, nsets, n1, n2, shared, masked)
    lib, ctx = mcs.lib(), G.ctx()
    q = cap.DescSet(P(d1), P(m1) if masked else None, P(v1), P(cam1), n1, dim)
    t_ = cap.DescSet(P(d2), P(m2) if masked else None, P(v2), P(cam2), n2, dim)
    m12 = np.full((nsets, n1), -7, np.int32); nm = np.full(nsets, -7, np.int32); fb = np.zeros(nsets, np.int32)
    rc = lib.mcs_search_triangulation(ctx.h, nsets, C.byref(q), 0 if shared else p1, C.byref(t_), p2, P(rays1), P(rays2), P(E), NC, dim, K, cap.MEM_HOST, P(m12), P(nm), P(fb))
    if rc != 0:
        return desc + " -> rc %d (%s)" % (rc, lib.mcs_last_error().decode()[:120])
    for k in range(nsets):
        a = 0 if shared else k
        en, e12 = O.search_triangulation(np.ascontiguousarray(d1[a, :n1]), np.ascontiguousarray(m1[a, :n1]), np.ascontiguousarray(mp1[a, :n1]), np.ascontiguousarray(cam1[a, :n1]),
                                         np.ascontiguousarray(rays1[a, :n1]), np.ascontiguousarray(d2[k, :n2]), np.ascontiguousarray(m2[k, :n2]), np.ascontiguousarray(mp2[k, :n2]),
                                         np.ascontiguousarray(cam2[k, :n2]), np.ascontiguousarray(rays2[k, :n2]), E.reshape(NC * NC, 9), NC, masked)
        if int(nm[k]) != en or not np.array_equal(m12[k], e12):
            return desc + " -> set %d: %d matches, oracle %d, %d entries differ" % (k, int(nm[k]), en, int((m12[k] != e12).sum()))
    return None


def case_bow(rng, idx):
    """the vocabulary-restricted SearchByBoW(KF, F) (src/cORBmatcher.cpp:179-323): a keyframe feature only meets frame features of the same FeatureVector node — through
    mcs_search_kf_f with the node id as `group`, keyframe rows in (node, index) order as the reference walks them (frontend.cORBmatcher.SearchByBoW)"""
    dim = 32
    K = int(rng.choice([2, 8, 32]))
    nk, nf = min(max(size(rng), 1), 2500), min(max(size(rng), 1), 2500)
    style = str(rng.choice(["random", "clusters", "clusters"]))
    masked = rng.random() < 0.5
    ratio = float(rng.choice([0.7, 0.9, 1.0]))
    nnodes = int(rng.choice([1, 3, 20, 80]))
    protos = rng.integers(0, 256, (int(rng.integers(1, 40)), dim)).astype(np.uint8)
    dk, df = rows(rng, nk, dim, style, protos), rows(rng, nf, dim, style, protos)
    mk = rng.integers(0, 256, (nk, dim)).astype(np.uint8) if masked else np.full((nk, dim), 255, np.uint8)
    mf = rng.integers(0, 256, (nf, dim)).astype(np.uint8) if masked else np.full((nf, dim), 255, np.uint8)
    vk = (rng.random(nk) < 0.85).astype(np.uint8)
    nodek = np.where(rng.random(nk) < 0.9, rng.integers(0, nnodes, nk), -1).astype(np.int32)   # -1: a stopped word, the feature is in no node
    nodef = np.where(rng.random(nf) < 0.9, rng.integers(0, nnodes, nf), -1).astype(np.int32)
    desc = "bow case %d: K=%d nk=%d nf=%d nodes=%d %s masked=%d ratio=%.2f" % (idx, K, nk, nf, nnodes, style, masked, ratio)
    order = np.array([i for nd in sorted(set(nodek[nodek >= 0].tolist())) for i in np.nonzero(nodek == nd)[0]], np.int64)
    if len(order) == 0:
        return None
    gk = np.ascontiguousarray(nodek[order]); vf = (nodef >= 0).astype(np.uint8)
    qd, qm, qv = np.ascontiguousarray(dk[order]), np.ascontiguousarray(mk[order]), np.ascontiguousarray(vk[order])
    lib, ctx = mcs.lib(), G.ctx()
    q = cap.DescSet(P(qd), P(qm) if masked else None, P(qv), P(gk), len(order), dim)
    t = cap.DescSet(P(df), P(mf) if masked else None, P(vf), P(nodef), nf, dim)
    mF = np.full(nf, -7, np.int32); nm = np.zeros(1, np.int32); fb = np.zeros(1, np.int32)
    rc = lib.mcs_search_kf_f(ctx.h, 1, C.byref(q), 0, C.byref(t), 0, dim, ratio, K, cap.MEM_HOST, P(mF), P(nm), P(fb))
    if rc != 0:
        return desc + " -> rc %d (%s)" % (rc, lib.mcs_last_error().decode()[:120])
    got = np.where(mF >= 0, order[np.maximum(mF, 0)], -1).astype(np.int32)
    en, em = O.search_kf_f_bow(dk, mk if masked else None, vk, nodek, df, mf if masked else None, nodef, masked, ratio)
    if int(nm[0]) != en or not np.array_equal(got, em):
        return desc + " -> %d matches, oracle %d, %d entries differ" % (int(nm[0]), en, int((got != em).sum()))
    return None


def case_distinct(rng, idx):
    """cMapPoint::ComputeDistinctiveDescriptors (src/cMapPoint.cpp:294-382): per map point the observation with the least median distance to the others — a batch of
    points with 0 .. 300 observations each, noisy copies of one descriptor, exact duplicates (ties), optional masks"""
    FE = importlib.import_module("multicol-slam_amd.frontend")
    dim = int(rng.choice([32, 32, 16, 64]))
    masks = bool(rng.random() < 0.5)
    obs = []
    for n in [0, 1, 2, 3] + list(rng.integers(2, int(rng.choice([10, 40, 300])), int(rng.integers(1, 200)))):
        n = int(n)
        base = rng.integers(0, 256, dim, dtype=np.uint8)
        d = np.repeat(base[None], n, axis=0)
        d = d ^ np.packbits(rng.random((n, dim * 8)) < rng.uniform(0.0, 0.3), axis=1, bitorder="little")
        if n > 4 and rng.random() < 0.3:
            d[rng.integers(0, n)] = d[0]
        m = (rng.integers(0, 256, (n, dim), dtype=np.uint8) | rng.integers(0, 256, (n, dim), dtype=np.uint8)) if masks else None
        obs.append((d, m))
    got = FE.ComputeDistinctiveDescriptorsBatch(obs, dim, G.ctx())
    exp = np.array([O.distinctive_descriptor(d, m) for d, m in obs], np.int32)
    if not np.array_equal(got, exp):
        return "distinct case %d: dim=%d masks=%d points=%d -> %d differ, first %s" % (idx, dim, masks, len(obs), int((got != exp).sum()), np.flatnonzero(got != exp)[:5])
    return None


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    n = bad = 0
    while time.time() - t0 < budget:
        try:
            err = (case_topk, case_search, case_triangulation, case_sweep, case_bow, case_search, case_distinct)[n % 7](rng, n)
        except Exception as ex:   # an error code of the library is a finding too
            err = "case %d raised %s: %s" % (n, type(ex).__name__, str(ex)[:200])
        n += 1
        if err:
            bad += 1
            print("FAIL", err, flush=True)
            if bad >= 20:
                break
    print("fuzz_match: seed %d, %d cases, %d failures, %.0f s" % (seed, n, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

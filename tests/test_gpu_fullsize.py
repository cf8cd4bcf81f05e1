"""-m gpu: BASELINE configs[1] at its full bench size (64 three-camera 754x480 multi-frames = 192 images, mdBRIEF, N = 1000) through size-independent
properties — the oracle needs ~45 ms per image, so only a sample of the batch is compared with it directly:
  * batch invariance: the 192-image batch equals the same images extracted 48 at a time, bit for bit (checksum of every output array);
  * a sample of the batch equals the oracle;
  * the batched SearchByBoW(KF,KF) over the 63 consecutive multi-frame pairs (one call, nsets = 63) equals the per-pair calls, every train feature is
    matched at most once, and every accepted pair satisfies the reference's acceptance rule when its distances are recomputed on the host."""
import ctypes as C
import hashlib
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NF, NCAM = 64, 3


@pytest.fixture(scope="module")
def G():
    import gpu_common
    return gpu_common


@pytest.fixture(scope="module")
def batch(G):
    cams = G.cams3()
    masks = [G.synth.mirror_mask(c) for c in cams]
    imgs = [im for f in range(NF) for im in G.synth.synth_multiframe(f, cams)]
    oc = [G.mcs.make_ocam(c) for c in cams]
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=NF * NCAM, do_dBrief=1, learnMasks=1)
    full = ex.extract_host(imgs, masks * NF, oc * NF)
    parts = []
    for b0 in range(0, NF * NCAM, 48):
        parts += ex.extract_host(imgs[b0:b0 + 48], (masks * NF)[b0:b0 + 48], (oc * NF)[b0:b0 + 48])
    fallbacks = ex.describe_stats()[0]
    ex.set_describe(exact_only=True)     # every keypoint through the reference's exact descriptor arithmetic (no fast pass)
    exact = ex.extract_host(imgs, masks * NF, oc * NF)
    ex.close()
    return dict(cams=cams, masks=masks, imgs=imgs, full=full, parts=parts, exact=exact, fallbacks=fallbacks)


def digest(res):
    h = hashlib.sha256()
    for kps, d, m, rays in res:
        for a in (kps, d, m, rays):
            h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def test_batch_invariance_and_oracle_sample(G, batch):
    full, parts = batch["full"], batch["parts"]
    assert len(full) == NF * NCAM and sum(len(r[0]) for r in full) > 180000
    assert digest(full) == digest(parts)
    # ~2.9e8 pattern points: the guarded fast descriptor pass and the exact pass agree in every bit, and the exact pass is the rare exception
    assert digest(full) == digest(batch["exact"])
    assert 0 < batch["fallbacks"] < 0.005 * 2 * sum(len(r[0]) for r in full), batch["fallbacks"]
    for i in (0, 95, 191):
        cam = batch["cams"][i % NCAM]
        _, kps, d, dm, rays = G.oracle_extract(batch["imgs"][i], batch["masks"][i % NCAM], cam, do_dBrief=1, learnMasks=1)
        gk, gd, gm, gr = full[i]
        assert G.first_diff(gk, kps) is None and G.first_diff(gd, d) is None and G.first_diff(gm, dm) is None and G.first_diff(gr, rays) is None


def test_batched_search_equals_per_pair_and_obeys_the_acceptance_rule(G, batch):
    cap = importlib.import_module("multicol-slam_amd._capi")
    lib, ctx = G.mcs.lib(), G.ctx()
    full = batch["full"]
    rows = max(sum(len(full[f * NCAM + c][0]) for c in range(NCAM)) for f in range(NF))
    D = np.zeros((NF, rows, 32), np.uint8); M = np.zeros_like(D); V = np.zeros((NF, rows), np.uint8)
    for f in range(NF):
        d = np.concatenate([full[f * NCAM + c][1] for c in range(NCAM)]); m = np.concatenate([full[f * NCAM + c][2] for c in range(NCAM)])
        D[f, :len(d)], M[f, :len(d)], V[f, :len(d)] = d, m, 1
    nsets, ratio = NF - 1, 0.9
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    q = cap.DescSet(P(D[1:]), P(M[1:]), P(V[1:]), None, rows, 32)     # multi-frame f+1 ...
    t = cap.DescSet(P(D[:-1]), P(M[:-1]), P(V[:-1]), None, rows, 32)  # ... against multi-frame f
    m12 = np.full((nsets, rows), -1, np.int32); nm = np.zeros(nsets, np.int32); fb = np.zeros(nsets, np.int32)
    cap.check(lib.mcs_search_kf_kf(ctx.h, nsets, C.byref(q), rows, C.byref(t), rows, 32, ratio, 32, cap.MEM_HOST, P(m12), P(nm), P(fb)))
    assert nm.sum() > 100000 and (nm == (m12 >= 0).sum(1)).all()
    for s in (0, 31, 62):   # the same pair on its own
        q1 = cap.DescSet(P(D[s + 1]), P(M[s + 1]), P(V[s + 1]), None, rows, 32)
        t1 = cap.DescSet(P(D[s]), P(M[s]), P(V[s]), None, rows, 32)
        one = np.full(rows, -1, np.int32); n1 = np.zeros(1, np.int32); f1 = np.zeros(1, np.int32)
        cap.check(lib.mcs_search_kf_kf(ctx.h, 1, C.byref(q1), 0, C.byref(t1), 0, 32, ratio, 32, cap.MEM_HOST, P(one), P(n1), P(f1)))
        assert int(n1[0]) == int(nm[s]) and np.array_equal(one, m12[s])
    pc = np.array([bin(i).count("1") for i in range(256)], np.int64)
    for s in range(nsets):
        got = m12[s][m12[s] >= 0]
        assert len(np.unique(got)) == len(got)          # a train feature is consumed by its first taker (:949)
        assert V[s][got].all() and V[s + 1][m12[s] >= 0].all()
    for s in (5, 40):   # acceptance rule of :885-966 for every accepted pair: best < TH_LOW (masks: 32) and best < ratio * second, over the trains
        qi = np.flatnonzero(m12[s] >= 0)                # that were still free — checked here against ALL valid trains, a necessary condition
        x = D[s + 1][qi][:, None, :] ^ D[s][None, :, :]
        dist = (pc[x & M[s + 1][qi][:, None, :]].sum(-1) + pc[x & M[s][None, :, :]].sum(-1)) // 2
        dist[:, V[s] == 0] = 1 << 20
        best = dist[np.arange(len(qi)), m12[s][qi]]
        assert (best < 32).all()
        dist[np.arange(len(qi)), m12[s][qi]] = 1 << 20
        # second best among ALL valid trains is <= second best among the free ones, so `best < ratio * second_free` need not hold against it;
        # what must hold: no strictly closer train existed that was still free, i.e. every strictly closer train was taken by an earlier query
        closer = dist < best[:, None]
        taken_by = np.full(rows, rows, np.int64)
        taken_by[m12[s][qi]] = qi
        r_, c_ = np.nonzero(closer)
        assert (taken_by[c_] < qi[r_]).all()

"""The array formulation of the oct-tree used by the HIP kernel == the literal list-based oracle."""
import numpy as np
import pytest

from octree_array_model import distribute


def _run_oracle(oracle, xs, ys, resp, W, H, N):
    kp = np.zeros(len(xs), oracle.KP_DTYPE)
    kp["x"], kp["y"], kp["response"] = xs, ys, resp
    kp["class_id"] = np.arange(len(xs))
    out = np.zeros(len(xs) + 8, oracle.KP_DTYPE)
    n = oracle.lib().orc_distribute_octtree(oracle.ptr(kp), len(xs), 22, 22 + W, 22, 22 + H, N, oracle.ptr(out), len(out))
    return out["class_id"][:n].tolist()


@pytest.mark.parametrize("seed", range(40))
def test_model_matches_oracle_random(oracle, seed):
    rng = np.random.default_rng(seed)
    W = int(rng.integers(40, 800))
    H = int(rng.integers(40, 500))
    if W < H // 2:
        W = H
    n = int(rng.integers(0, 1500))
    N = int(rng.integers(1, 300))
    mode = seed % 4
    if mode == 0:      # uniform
        xs = rng.integers(3, W - 3, n)
        ys = rng.integers(3, H - 3, n)
    elif mode == 1:    # clustered (deep splits, many ties)
        cx, cy = rng.integers(3, W - 3, 6), rng.integers(3, H - 3, 6)
        c = rng.integers(0, 6, n)
        xs = np.clip(cx[c] + rng.integers(-8, 9, n), 3, W - 4)
        ys = np.clip(cy[c] + rng.integers(-8, 9, n), 3, H - 4)
    elif mode == 2:    # grid-ish with duplicates in size (tie-break stress)
        xs = (rng.integers(0, max(1, (W - 6) // 8), n) * 8 + 3)
        ys = (rng.integers(0, max(1, (H - 6) // 8), n) * 8 + 3)
    else:              # one column (split produces a single non-empty child -> "size == prevSize" exit)
        xs = np.full(n, 3 + (W - 6) // 3)
        ys = rng.integers(3, H - 3, n)
    # unique positions like FAST output (a pixel is a corner at most once), keep cell/scan order irrelevant: any order works
    pts = np.unique(np.stack([ys, xs], 1), axis=0) if n else np.zeros((0, 2), int)
    rng.shuffle(pts)
    ys, xs = pts[:, 0], pts[:, 1]
    resp = rng.integers(1, 60, len(xs))
    got = distribute(xs.tolist(), ys.tolist(), resp.tolist(), 22, 22 + W, 22, 22 + H, N)
    exp = _run_oracle(oracle, xs.astype(np.float32), ys.astype(np.float32), resp.astype(np.float32), W, H, N)
    assert got == exp


def test_model_matches_oracle_on_real_candidates(oracle, synth):
    cam = synth.lafida_cameras()[1]
    oc = oracle.make_ocam(cam)
    img = synth.synth_image(2, 1, cam)
    ex = oracle.Extractor(nfeatures=1000)
    ex(img, oracle.mirror_mask(oc), oc)
    nper = [217, 181, 151, 126, 105, 87, 73, 60]
    for lvl in range(8):
        c = ex.candidates(lvl)
        w, h = ex.level_size(lvl)
        got = distribute(c["x"].astype(int).tolist(), c["y"].astype(int).tolist(), c["response"].tolist(), 22, w - 22, 22, h - 22, nper[lvl])
        sel = ex.selected(lvl)
        exp = [(float(x) - 22, float(y) - 22) for x, y in zip(sel["x"], sel["y"])]
        assert [(float(c["x"][k]), float(c["y"][k])) for k in got] == exp

"""CPU: vocabulary loader + oracle DBoW2 descent on a hand-built tree, a synthetic vocabulary and, where the reference
checkout exists, the shipped small_orb_omni_voc_9_6.yml."""
import importlib
import os

import numpy as np
import pytest

import oracle_lib as O
import vocab_synth

io = importlib.import_module("multicol-slam_amd.io")
REAL = "/root/reference/Examples/small_orb_omni_voc_9_6.yml"


def bits(n):
    d = np.zeros(32, np.uint8)
    for b in range(n):
        d[b // 8] |= 1 << (b % 8)
    return d


def test_descent_known_answer():
    # root -> {1: 0 bits, 2: 16 bits, 3: 16 bits (tie with 2 for a 16-bit... see below)}; 2 -> {4: 12 bits, 5: 20 bits}; leaves 1, 3, 4, 5
    voc = dict(L=2, node_desc=np.stack([bits(0), bits(0), bits(16), bits(16), bits(12), bits(20)]),
               child_off=np.array([0, 3, 3, 5, 5, 5, 5], np.int32), child_idx=np.array([1, 2, 3, 4, 5], np.int32))
    q = np.stack([bits(2), bits(15), bits(17), bits(8), bits(40)])
    leaf, nid = O.bow_transform(voc, q, 1)      # nid level = L - 1 = 1
    # 2 bits: child 1 (d=2); 15: children 2 and 3 tie at d=1 -> the FIRST (2), then 4 (d=3) vs 5 (d=5) -> 4; 17: 2 (tie, first) -> 5 (3 vs 5... 17 vs 12 = 5, vs 20 = 3) -> 5
    # 8 bits: d = 8, 8, 8 -> first = 1; 40: d = 40, 24, 24 -> 2, then 28 vs 20 -> 5
    assert list(leaf) == [1, 4, 5, 1, 5] and list(nid) == [1, 2, 2, 1, 2]
    leaf0, nid0 = O.bow_transform(voc, q, 2)    # level 0 -> root
    assert list(leaf0) == list(leaf) and list(nid0) == [0] * 5
    _, nid2 = O.bow_transform(voc, q, 0)        # level 2: only paths that reach depth 2 set it
    assert list(nid2) == [0, 4, 5, 0, 5]


def test_synthetic_vocabulary_round_trip(tmp_path):
    p = str(tmp_path / "voc.yml")
    nn, nw = vocab_synth.write_vocabulary(p, k=9, L=4, seed=5)
    v = io.load_vocabulary(p)
    assert (v["k"], v["L"], v["scoringType"], v["weightingType"]) == (9, 4, 0, 0)
    assert len(v["node_desc"]) == nn + 1 and v["n_words"] == nw and v["child_off"][-1] == nn
    leaves = np.flatnonzero(np.diff(v["child_off"]) == 0)
    assert np.array_equal(np.flatnonzero(v["word_id"] >= 0), leaves) and sorted(v["word_id"][leaves]) == list(range(nw))
    assert (v["parent"][v["child_idx"]] == np.repeat(np.arange(nn + 1), np.diff(v["child_off"]))).all()
    assert 0.05 < (v["weight"][leaves] == 0).mean() < 0.3 and (v["weight"][np.diff(v["child_off"]) > 0] == 0).all()
    rng = np.random.default_rng(0)
    d = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    leaf, nid = O.bow_transform(v, d, 2)
    assert np.isin(leaf, leaves).all()
    # nid is the ancestor of the leaf at level L - levelsup = 2
    anc = leaf.copy()
    depth = np.zeros(nn + 1, np.int32)
    for i in range(1, nn + 1):
        depth[i] = depth[v["parent"][i]] + 1
    for _ in range(v["L"]):
        up = depth[anc] > 2
        anc[up] = v["parent"][anc[up]]
    assert np.array_equal(nid, anc)


@pytest.mark.skipif(not os.path.exists(REAL), reason="reference checkout not present (GPU box)")
def test_shipped_vocabulary_parses():
    v = io.load_vocabulary(REAL)
    assert (v["k"], v["L"]) == (9, 6) and len(v["node_desc"]) == 8823 and v["n_words"] == 6999
    assert np.diff(v["child_off"]).max() == 9 and v["child_off"][1] == 9
    assert (v["weight"] > 0).sum() == 5676 and abs(v["weight"].max() - 2.6390573296152584) == 0
    rng = np.random.default_rng(1)
    d = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    leaf, nid = O.bow_transform(v, d, 4)
    assert (v["word_id"][leaf] >= 0).all() and (nid > 0).all()

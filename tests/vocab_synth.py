"""Deterministic synthetic DBoW2 vocabulary in the reference's own YAML layout (the real small_orb_omni_voc_9_6.yml stays in the reference
checkout, which the GPU box does not have)."""
import numpy as np


def write_vocabulary(path, k=9, L=4, seed=11, leaf_prob=0.25, zero_weight=0.15):
    rng = np.random.default_rng(seed)
    nodes = []           # (nodeId, parentId, weight, descriptor bytes)
    words = []
    frontier = [(0, 0, rng.integers(0, 256, 32, dtype=np.uint8))]
    next_id = 1
    while frontier:
        nid, depth, d = frontier.pop(0)
        if nid != 0 and (depth == L or rng.random() < leaf_prob * (depth >= 2)):
            continue
        nchild = k if depth < 2 else int(rng.integers(2, k + 1))
        for _ in range(nchild):
            flips = rng.random(256) < 0.18
            cd = d ^ np.packbits(flips, bitorder="little")
            nodes.append([next_id, nid, 0.0, cd])
            frontier.append((next_id, depth + 1, cd))
            next_id += 1
    parents = set(n[1] for n in nodes)
    for n in nodes:
        if n[0] not in parents:           # leaf -> word
            n[2] = 0.0 if rng.random() < zero_weight else float(np.log(rng.integers(2, 40)))
            words.append((len(words), n[0]))
    with open(path, "w") as f:
        f.write("%%YAML:1.0\nvocabulary:\n   k: %d\n   L: %d\n   scoringType: 0\n   weightingType: 0\n   nodes:\n" % (k, L))
        for nid, pid, w, d in nodes:
            ws = "0." if w == 0.0 else ("%.16e" % w)
            f.write("      - { nodeId:%d, parentId:%d, weight:%s,\n          descriptor:\"%s \" }\n" % (nid, pid, ws, " ".join(str(int(v)) for v in d)))
        f.write("   words:\n")
        for wid, nid in words:
            f.write("      - { wordId:%d, nodeId:%d }\n" % (wid, nid))
    return len(nodes), len(words)

"""-m gpu: the run-time switches of libmcs_hip.so select code paths the default run does not take — MCS_NO_OVERLAP=1 (everything in order on one stream, the
path per-kernel timing uses), MCS_MATCH_FILL / MCS_MATCH_BLOCKS (train-range splits + k_match_merge even for a deep batch), MCS_DESCRIBE_EXACT=1 (the
exact descriptor pass for every keypoint), MCS_MATCH_VALU=1 (the v_bcnt matcher instead of the matrix-core one), MCS_MATCH_EXPAND_MB=0 (no room for the
expanded train sets: the v_bcnt matcher again, chosen by the size check), MCS_PYR_CHAIN=1 (the whole resize chain in one launch, alone and with MCS_NO_OVERLAP), MCS_PYR_TILES=1 (the LDS tile resize kernel that serves
scale factors above 2 instead of the column-marching one), MCS_LIST_SPLIT=0 (the fast pass's fallbacks by one wave each instead of a workgroup each), MCS_GRAPHS=0 (the launches of a small batch enqueued one by one instead of replayed from a hipGraph), MCS_GREEDY_JACOBI=0 (the chunk-by-chunk greedy pass also for few set pairs, instead of the fixpoint form), MCS_OUT_KERNEL=0 (page-locked host outputs — what Extractor.extract_host uses — by the runtime's copies instead of one launch).  Each is run in a fresh process (the switches are read once) on the same inputs; every output must equal the
default run's, bit for bit."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes as C, hashlib, importlib, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import gpu_common as G
cap = importlib.import_module("multicol-slam_amd._capi")
imgs, masks, cams = [], [], []
for f in range(3):
    i, m, c = G.frame_inputs(f)
    imgs += i; masks += m; cams += c
ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=len(imgs), nfeatures=500, do_dBrief=1, learnMasks=1)
res = ex.extract_host(imgs, masks, [G.mcs.make_ocam(c) for c in cams])
h = hashlib.sha256()
for kps, d, m, rays in res:
    for a in (kps, d, m, rays):
        h.update(np.ascontiguousarray(a).tobytes())
# 8 (frame, keyframe) pairs in one call: deep enough that the default takes the no-split path
rows = max(len(r[0]) for r in res)
D = np.zeros((len(res), rows, 32), np.uint8); M = np.zeros_like(D); V = np.zeros((len(res), rows), np.uint8)
for i, (kps, d, m, rays) in enumerate(res):
    D[i, :len(d)], M[i, :len(d)], V[i, :len(d)] = d, m, 1
P = lambda a: a.ctypes.data_as(C.c_void_p)
ns = len(res) - 1
q = cap.DescSet(P(D[1:]), P(M[1:]), P(V[1:]), None, rows, 32)
t = cap.DescSet(P(D[:-1]), P(M[:-1]), P(V[:-1]), None, rows, 32)
m12 = np.full((ns, rows), -1, np.int32); nm = np.zeros(ns, np.int32)
cap.check(G.mcs.lib().mcs_search_kf_kf(G.ctx().h, ns, C.byref(q), rows, C.byref(t), rows, 32, 0.9, 32, cap.MEM_HOST, P(m12), P(nm), None))
h.update(m12.tobytes()); h.update(nm.tobytes())
print("DIGEST", h.hexdigest(), int(nm.sum()))
'''


def _run(env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][-1].split()
    return line[1], int(line[2])


def test_run_time_switches_do_not_change_any_output():
    ref, nmatch = _run({})
    assert nmatch > 1000
    for env in ({"MCS_NO_OVERLAP": "1"}, {"MCS_MATCH_FILL": "100000000", "MCS_MATCH_BLOCKS": "4096"}, {"MCS_DESCRIBE_EXACT": "1"}, {"MCS_MATCH_VALU": "1"},
                {"MCS_MATCH_VALU": "1", "MCS_MATCH_FILL": "100000000", "MCS_MATCH_BLOCKS": "4096"}, {"MCS_MATCH_EXPAND_MB": "0"}, {"MCS_PYR_CHAIN": "1"},
                {"MCS_PYR_CHAIN": "1", "MCS_NO_OVERLAP": "1"}, {"MCS_PYR_TILES": "1"}, {"MCS_LIST_SPLIT": "0"}, {"MCS_GRAPHS": "0"}, {"MCS_GREEDY_JACOBI": "0"}, {"MCS_OUT_KERNEL": "0"}):
        got, n = _run(env)
        assert (got, n) == (ref, nmatch), env


@pytest.mark.parametrize("split", ["0", "1"])
def test_both_forms_of_the_orb_descriptor_reproduce_the_oracle(split):
    """ORB descriptors come from k_orient_b<0> + k_describe_orb (round 6: the keypoint's ray and rotation by one thread, the bits by a lean wave) for batches of 8192 rows
    and more, from the one-kernel form (describe_wave<0>) below that; MCS_ORB_SPLIT = 1 / 0 forces either for every size — the end-to-end ORB parity tests (3-image
    batches) in a fresh process with each"""
    e = dict(os.environ, MCS_ORB_SPLIT=split)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_extract.py"), "-m", "gpu", "-q", "-x", "-k", "end_to_end_bit_exact or no_mask_and_odd"],
                       env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]

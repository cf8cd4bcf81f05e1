"""-m gpu: rounding ties ENFORCED on device-kind, pipelined batches (round 6: mcs_extractor_set_tie_capture / mcs_extractor_patch_ties, csrc/mcs_tiefix.hip
k_tie_capture; reference arithmetic src/mdBRIEFextractorOct.cpp:250-301).

A device-kind batch ends with a capture launch that leaves, for every listed keypoint, its slot / level / position / angle and the 81 x 81 window of samples in
page-locked memory; the caller patches the batch's rows LATER — when the extractor's pyramid buffers already hold the next batch — and before it enqueues whatever
consumes them.  The tests widen the band (0.5 px: every exact-arithmetic keypoint is the host code's; 2e-4 px: a part), run two batches back to back, scribble over
the first batch's rows and require that exactly the listed rows come back as the oracle's; then the bench's own pipelined loop (device-resident outputs, deferred
searches, patch one step late) with the band at 2e-4 must reproduce the oracle's bits AND the oracle's matches."""
import json
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


@pytest.mark.parametrize("mode", ["orb", "dbrief", "mdbrief"])
@pytest.mark.parametrize("band", [0.5, 2e-4])
def test_rows_of_an_earlier_batch_are_patched_from_the_capture(G, mode, band):
    kw = dict(orb=dict(), dbrief=dict(do_dBrief=1), mdbrief=dict(do_dBrief=1, learnMasks=1))[mode]
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=3, nfeatures=250, **kw)
    if mode != "orb":
        ex.set_describe(exact_only=True)
    ex.set_tie_band(band)
    ex.set_tie_capture(2, 1024)
    cap = ex.cap
    batches = []
    for frame in (6, 9):
        imgs, masks, cams = G.frame_inputs(frame, 3)
        bufs = dict(img=G.DevBuf(np.stack(imgs)), msk=G.DevBuf(np.stack(masks)), nkp=G.DevBuf(np.zeros(3, np.int32)), kps=G.DevBuf(np.zeros((3, cap), G.mcs._capi.KP_DTYPE)),
                    desc=G.DevBuf(np.zeros((3, cap, 32), np.uint8)), mask=G.DevBuf(np.zeros((3, cap, 32), np.uint8)))
        batches.append((imgs, masks, cams, bufs))
    for imgs, masks, cams, b in batches:   # both batches enqueued back to back: the pyramid holds the SECOND one when the first is patched
        ex.extract_device(3, b["img"].ptr.value, 754 * 480, 754, b["msk"].ptr.value, 754 * 480, 754, [G.mcs.make_ocam(c) for c in cams], b["nkp"].ptr.value,
                          b["kps"].ptr.value, b["desc"].ptr.value, b["mask"].ptr.value, 0)
    G.ctx().synchronize()
    total = 0
    for back, (imgs, masks, cams, b) in ((1, batches[0]), (0, batches[1])):
        before_d, before_m = b["desc"].read(), b["mask"].read()
        assert G.hip().hipMemset(b["desc"].ptr, 0xA5, before_d.nbytes) == 0 and G.hip().hipMemset(b["mask"].ptr, 0x5A, before_m.nbytes) == 0
        listed, fixed = ex.patch_ties(back)
        nkp = b["nkp"].read()
        assert listed == fixed and fixed > 0
        if band == 0.5:
            assert fixed == int(nkp.sum())
        else:
            assert fixed < int(nkp.sum())
        after_d, after_m = b["desc"].read(), b["mask"].read()
        patched = 0
        for i in range(3):
            _, ok, od, odm, _ = G.oracle_extract(imgs[i], masks[i], cams[i], nfeatures=250, **kw)
            assert len(ok) == nkp[i]
            for k in range(nkp[i]):
                if (after_d[i, k] != 0xA5).any() or (after_m[i, k] != 0x5A).any():
                    patched += 1
                    assert (after_d[i, k] == od[k]).all() and (after_m[i, k] == odm[k]).all(), (mode, band, back, i, k)
        assert patched == fixed
        assert ex.patch_ties(back) == (listed, 0)   # a slot is patched once
        total += fixed
    assert ex.tie_counts()[1] == total
    # a third batch reuses the first slot: that batch can no longer be patched, and says so
    imgs, masks, cams, b = batches[0]
    ex.extract_device(3, b["img"].ptr.value, 754 * 480, 754, b["msk"].ptr.value, 754 * 480, 754, [G.mcs.make_ocam(c) for c in cams], b["nkp"].ptr.value,
                      b["kps"].ptr.value, b["desc"].ptr.value, b["mask"].ptr.value, 0)
    a, r = G.mcs._capi.C.c_int(), G.mcs._capi.C.c_int()
    assert G.mcs.lib().mcs_extractor_patch_ties(ex.h, 2, a, r) == G.mcs._capi.MCS_ERR_INVALID
    assert ex.patch_ties(0)[0] > 0
    ex.close()


def test_capture_capacity_is_reported_not_truncated(G):
    imgs, masks, cams = G.frame_inputs(4, 3)
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=3, nfeatures=250)
    ex.set_tie_band(0.5)
    ex.set_tie_capture(1, 8)
    cap = ex.cap
    b = dict(img=G.DevBuf(np.stack(imgs)), msk=G.DevBuf(np.stack(masks)), nkp=G.DevBuf(np.zeros(3, np.int32)), kps=G.DevBuf(np.zeros((3, cap), G.mcs._capi.KP_DTYPE)),
             desc=G.DevBuf(np.zeros((3, cap, 32), np.uint8)), mask=G.DevBuf(np.zeros((3, cap, 32), np.uint8)))
    ex.extract_device(3, b["img"].ptr.value, 754 * 480, 754, b["msk"].ptr.value, 754 * 480, 754, None, b["nkp"].ptr.value, b["kps"].ptr.value, b["desc"].ptr.value,
                      b["mask"].ptr.value, 0)
    a, r = G.mcs._capi.C.c_int(), G.mcs._capi.C.c_int()
    assert G.mcs.lib().mcs_extractor_patch_ties(ex.h, 0, a, r) == G.mcs._capi.MCS_ERR_CAPACITY and a.value > 8 and r.value == 0
    assert ex.fix_ties() == a.value   # the synchronous form still serves the batch
    ex.set_tie_capture(0)
    assert G.mcs.lib().mcs_extractor_patch_ties(ex.h, 0, a, r) == G.mcs._capi.MCS_ERR_INVALID
    ex.close()


@pytest.mark.parametrize("args", [[], ["--exchange", "nccl1"], ["--workload", "db", "--frames", "8"]])
def test_pipelined_loop_with_a_wide_band_reproduces_the_oracle(args):
    """bench.py's loop (device-resident outputs, deferred searches, patch one step late) with the band at 2e-4 px: every fast-pass fallback of every step is
    recomputed on the host inside the loop — and descriptors, masks, counts AND match indices are the oracle's"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["MCS_BENCH_TIE_BAND"] = "2e-4"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--steps", "4", "--warmup", "2", *args], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    cfg = out["config"]
    assert out["oracle_check"] is True and cfg["oracle_checked"]["pairs"] >= 1
    assert cfg["ties_patched_in_loop"] is True and cfg["rounding_tie_band_px"] == 2e-4
    assert cfg["ties_recomputed_on_the_host_in_loop_rank0"] == cfg["ties_listed_at_patch_time_rank0"] > 4 * cfg["descriptor_exact_pass_keypoints_per_step_rank0"] > 0


def test_host_kind_batches_recompute_from_the_capture_slot(G):
    """host-kind mcs_extract_batch with a FEW listed keypoints (<= 64: the extractor's own capture slot, no whole-level download): guard band widened to 1e-6 px so that
    a dozen keypoints fall back to the exact pass, band 0.5 px so that all of those are listed — they must come back as the oracle's"""
    imgs, masks, cams = G.frame_inputs(4, 3)
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=3, do_dBrief=1, learnMasks=1)
    ex.set_describe(guard_eps=1e-6)
    ex.set_tie_band(0.5)
    out = ex.extract_host(imgs, masks, [G.mcs.make_ocam(c) for c in cams])
    n_exact = ex.describe_stats()[0]
    listed, fixed, _ = ex.tie_counts()
    assert listed == fixed == n_exact and 0 < n_exact <= 64, n_exact
    for i in range(3):
        _, ok, od, odm, _ = G.oracle_extract(imgs[i], masks[i], cams[i], do_dBrief=1, learnMasks=1)
        assert G.first_diff(out[i][1], od) is None and G.first_diff(out[i][2], odm) is None
    ex.close()

"""-m gpu: the native rig host (multicol-slam_amd/host/rig_host.cpp) — one C++ process, one host thread + one mcs_ctx per GPU, RCCL communicators from
ncclCommInitAll, the exchange issued on the context's stream with the CALLER's communicator (the library itself never links RCCL) — the analogue of the
reference's one-thread-per-camera constructor (src/cMultiFrame.cpp:128-164).  On this one-GPU box it runs with one rank; both exchange forms go through
RCCL all the same (ncclAllGather for the database sweep, a grouped ncclSend / ncclRecv to itself for the frame ring).  Its raw outputs against the oracle:
descriptors, masks, counts and keypoint records of every image, match indices of every pair (reference semantics src/cORBmatcher.cpp:885-966 and 179-323)."""
import ctypes as C
import importlib
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "multicol-slam_amd", "host")
NCAM, W, H, NFEAT = 3, 754, 480, 300


def run_host(tmp_path, frames, keyframes):
    import oracle_lib as O
    mcs = importlib.import_module("multicol-slam_amd")
    synth = importlib.import_module("multicol-slam_amd.synth")
    rig = importlib.import_module("multicol-slam_amd.rig")
    subprocess.check_call(["make", "-s", "-C", HOST])
    cams = synth.lafida_cameras()
    imgs = np.stack([synth.synth_image(f, c, cams[c]) for c in range(NCAM) for f in range(frames)])           # camera-major: x = camera * frames + frame
    masks = np.stack([synth.mirror_mask(cams[c]) for c in range(NCAM)])
    ocams = (mcs.Ocam * NCAM)(*[mcs.make_ocam(cams[c]) for c in range(NCAM)])
    p = lambda n: str(tmp_path / n)
    imgs.tofile(p("images.bin")); masks.tofile(p("masks.bin"))
    open(p("cams.bin"), "wb").write(bytes(ocams))
    open(p("cfg.txt"), "w").write("ncam %d\nwidth %d\nheight %d\nnfeatures %d\nmode 2\nframes %d\nkeyframes %d\ngpus 1\nsteps 2\nwarmup 1\ntopk 32\nimages %s\nmasks %s\ncams %s\nout %s\n"
                                   % (NCAM, W, H, NFEAT, frames, keyframes, p("images.bin"), p("masks.bin"), p("cams.bin"), p("out")))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([os.path.join(HOST, "rig_host"), p("cfg.txt")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    lay_txt = dict(ln.split() for ln in open(p("out.layout")))
    cap, vframes = int(lay_txt["cap"]), int(lay_txt["view_frames"])
    view = rig.RigLayout(NCAM, vframes, 1, cap, 32)
    rd = lambda name, dt: np.fromfile(p("out.r0." + name), dtype=dt)
    G = rd("G", np.uint8).reshape(view.images_total, view.rows_img, view.row_stride)
    kps = rd("kps", np.uint8).reshape(NCAM * frames, cap, 28)
    oracle = {}
    for c in range(NCAM):
        for f in range(frames):
            oracle[(c, f)] = O.Extractor(nfeatures=NFEAT, do_dBrief=1, learnMasks=1)(imgs[c * frames + f], masks[c], O.make_ocam(cams[c]))
    for c in range(NCAM):
        for f in range(frames):
            ok, od, om = oracle[(c, f)]
            assert np.array_equal(kps[c * frames + f, :len(ok)].reshape(-1), np.ascontiguousarray(ok).view(np.uint8).reshape(-1)), (c, f)
    return dict(info=info, rig=rig, view=view, G=G, cap=cap, oracle=oracle, rd=rd, O=O, frames=frames)


def check_frame(R, local_frame, global_frame):
    d, m, v = R["rig"].unpack_frame(R["view"], R["G"], local_frame)
    for c in range(NCAM):
        _, od, om = R["oracle"][(c, global_frame)]
        lo = c * R["cap"]
        assert int(v[lo:lo + R["cap"]].sum()) == len(od) > 100
        assert np.array_equal(d[lo:lo + len(od)], od) and np.array_equal(m[lo:lo + len(od)], om)
    return d, m, v


def test_frame_ring_through_grouped_send_recv(tmp_path):
    F = 3
    R = run_host(tmp_path, F, 0)
    assert "ncclSend" in R["info"]["exchange"] and R["info"]["n_gpus"] == 1 and R["info"]["ms_per_step"] > 0
    assert R["view"].frames_total == F + 1
    fr = [check_frame(R, j, (j - 1) % F) for j in range(F + 1)]         # local frame 0 = the predecessor of frame 0 = frame F - 1 (cyclic)
    match = R["rd"]("match", np.int32).reshape(F, R["view"].rows_frame)
    nmatch = R["rd"]("nmatch", np.int32)
    total = 0
    for s in range(F):                                                    # pair s: frame s against frame s - 1
        (d1, m1, v1), (d0, m0, v0) = fr[s + 1], fr[s]
        n, want = R["O"].search_kf_kf(d1, m1, v1, d0, m0, v0, True, 0.9)
        assert n == nmatch[s] and np.array_equal(match[s], want), s
        total += n
    assert total > 100


def test_database_sweep_through_allgather(tmp_path):
    F, D = 2, 3
    R = run_host(tmp_path, F, D)
    assert R["info"]["exchange"] == "ncclAllGather"
    fr = [check_frame(R, f, f) for f in range(F)]
    rows = R["view"].rows_frame
    db = R["rd"]("db", np.uint8).reshape(D, rows, 64)
    dbv = R["rd"]("dbvalid", np.uint8).reshape(D, rows)
    match = R["rd"]("match", np.int32).reshape(F, D, rows)
    nmatch = R["rd"]("nmatch", np.int32).reshape(F, D)
    total = 0
    for k in range(D):
        dk, mk, vk = fr[k % F]                                            # stored keyframe k = multi-frame k % F
        assert np.array_equal(db[k][vk != 0, :32], dk[vk != 0]) and np.array_equal(dbv[k], vk)
        for f in range(F):
            df, mf, vf = fr[f]
            keep = np.flatnonzero(vf)
            n, mm = R["O"].search_kf_f(np.ascontiguousarray(db[k][:, :32]), np.ascontiguousarray(db[k][:, 32:]), dbv[k], np.ascontiguousarray(df[keep]),
                                       np.ascontiguousarray(mf[keep]), True, 0.9)
            full = np.full(rows, -1, np.int32)
            full[keep] = mm
            assert n == nmatch[f, k] and np.array_equal(match[f, k], full), (f, k)
            total += n
    assert total > 500

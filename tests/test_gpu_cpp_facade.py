"""-m gpu: the C++ host facade (include/mcs/mcs_facade.hpp) compiled with g++ and run end to end; every output array vs the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_facade_end_to_end(tmp_path):
    import gpu_common as G
    O = G.O
    cams = G.cams3()
    ncam, w, h, nframes = 3, 754, 480, 2
    exe = tmp_path / "facade_driver"
    lib_dir = os.path.join(ROOT, "multicol-slam_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_driver.cpp"),
                           "-o", str(exe), "-L" + lib_dir, "-lmcs_hip", "-Wl,-rpath," + lib_dir])
    imgs = [G.synth.synth_multiframe(f, cams) for f in range(nframes)]
    masks = [G.synth.mirror_mask(c) for c in cams]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(np.array([ncam, w, h, nframes], np.int32).tobytes())
        for c in cams:
            f.write(bytes(G.mcs.make_ocam(c)))
        for fr in imgs:
            for im in fr:
                f.write(np.ascontiguousarray(im, np.uint8).tobytes())
        for m in masks:
            f.write(np.ascontiguousarray(m, np.uint8).tobytes())
    assert C.sizeof(G.mcs.make_ocam(cams[0])) == 320
    subprocess.check_call([str(exe), str(fin), str(fout)])
    buf = open(fout, "rb").read()
    off = [0]

    def take(dtype, n):
        a = np.frombuffer(buf, dtype, n, off[0]).copy()
        off[0] += a.nbytes
        return a

    frames = []
    for f in range(nframes):
        n = take(np.int32, ncam)
        tot = int(n.sum())
        keys, d, m, rays = take(O.KP_DTYPE, tot), take(np.uint8, tot * 32).reshape(tot, 32), take(np.uint8, tot * 32).reshape(tot, 32), take(np.float64, tot * 3).reshape(tot, 3)
        s = 0
        for c in range(ncam):
            _, ek, ed, em, er = G.oracle_extract(imgs[f][c], masks[c], cams[c], nfeatures=1000, do_dBrief=1, learnMasks=1)
            assert n[c] == len(ek)
            sl = slice(s, s + n[c])
            assert G.first_diff(keys[sl], ek) is None and G.first_diff(d[sl], ed) is None and G.first_diff(m[sl], em) is None
            assert G.first_diff(rays[sl], er) is None
            s += n[c]
        frames.append((n, keys, d, m))
    n0, n1 = len(frames[0][1]), len(frames[1][1])
    cam = [np.repeat(np.arange(ncam, dtype=np.int32), fr[0]) for fr in frames]
    # SearchByBoW(KF,KF)
    nb = int(take(np.int32, 1)[0])
    m12 = take(np.int32, n0)
    en, e12 = O.search_kf_kf(frames[0][2], frames[0][3], np.ones(n0, np.uint8), frames[1][2], frames[1][3], np.ones(n1, np.uint8), True, 0.8)
    assert nb == en and np.array_equal(m12, e12) and nb > 300
    # WindowSearch(F1, F2, 60, minScaleLevel 2)
    v = [O.frame_view(fr[1], fr[2], fr[3], cam[i], [w] * ncam, [h] * ncam) for i, fr in enumerate(frames)]
    nw = int(take(np.int32, 1)[0])
    m21 = take(np.int32, n1)
    en, e21 = O.window_search(v[0][0], np.ones(n0, np.uint8), v[1][0], 60, 2, -1, 0.8, 32, True)
    assert nw == en and np.array_equal(m21, e21) and nw > 100
    # SearchForInitialization(F1, F2, prev = F1 key positions, 50)
    ni = int(take(np.int32, 1)[0])
    i12 = take(np.int32, n0)
    prev = take(np.float64, 2 * n0).reshape(n0, 2)
    p0 = np.stack([frames[0][1]["x"], frames[0][1]["y"]], axis=1).astype(np.float64)
    en, e12, ep = O.search_for_initialization(v[0][0], v[1][0], p0, 50, 0.9, 32, True)
    assert ni == en and np.array_equal(i12, e12) and np.array_equal(prev, ep) and ni > 100
    assert off[0] == len(buf)

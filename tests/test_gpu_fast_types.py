"""-m gpu: FAST TYPE_7_12 / TYPE_5_8 (`extractor.fastAgastType` 1 / 0; reference src/mdBRIEFextractorOct.cpp:869-872, 912-914) on the device: the ring is a
template parameter of k_fast_cells (csrc/mcs_fast.hip), with OpenCV 3.x's wrapped quick test and 3-pixel border kept (tests/test_oracle_fast_types.py
states both).  Candidates per level and the end-to-end outputs against the oracle, bit for bit (AGAST: tests/test_gpu_agast.py)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_common
    return gpu_common


@pytest.mark.parametrize("ftype,th,mode", [(1, 8, dict(do_dBrief=1, learnMasks=1)), (1, 20, dict(do_dBrief=0, learnMasks=0)), (0, 4, dict(do_dBrief=0, learnMasks=0)),
                                          (0, 2, dict(do_dBrief=1, learnMasks=1))])
def test_small_rings_bit_exact(G, ftype, th, mode):
    imgs, masks, cams = G.frame_inputs(2)
    rng = np.random.default_rng(ftype)
    imgs = list(imgs)
    noisy = imgs[2].astype(np.int32)                      # isolated speckles: what the 8-pixel ring (all ring pixels darker / brighter) responds to
    ys, xs = rng.integers(30, 450, 4000), rng.integers(30, 720, 4000)
    noisy[ys, xs] += rng.choice([-90, 90], 4000)
    imgs[2] = np.clip(noisy, 0, 255).astype(np.uint8)
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=3, nfeatures=600, fastThreshold=th, fastAgastType=ftype, **mode)
    res = ex.extract_host(imgs, masks, [G.mcs.make_ocam(c) for c in cams])
    total = 0
    for i in range(3):
        oex, kps, d, dm, rays = G.oracle_extract(imgs[i], masks[i], cams[i], nfeatures=600, fastThreshold=th, fastAgastType=ftype, **mode)
        for l in range(8):
            x, y, s = ex.tap_candidates(i, l)
            c = oex.candidates(l)
            assert len(x) == len(c), (i, l, len(x), len(c))
            assert G.first_diff(np.stack([x, y, s], 1), np.stack([c["x"], c["y"], c["response"]], 1).astype(np.int32)) is None, (i, l)
        gk, gd, gm, gr = res[i]
        assert G.first_diff(gk, kps) is None and G.first_diff(gd, d) is None and G.first_diff(gm, dm) is None and G.first_diff(gr, rays) is None
        total += len(kps)
    assert total > 150, total
    ex.close()


def test_large_cells_instance_and_unknown_type_rejected(G):
    """a small image (cells larger than 40 px: the 60 x 60 kernel instance) with the 12-pixel ring; unknown types are refused loudly"""
    rng = np.random.default_rng(3)
    img = np.clip(rng.normal(110, 30, (200, 260)), 0, 255).astype(np.uint8)
    ex = G.mcs.Extractor(G.ctx(), 260, 200, max_batch=1, nfeatures=300, nlevels=3, fastThreshold=12, fastAgastType=1)
    kps, d, dm, _ = ex.extract_host([img], None, [G.mcs.make_ocam(G.cams3()[0])])[0]
    _, ok, od, odm, _ = G.oracle_extract(img, None, G.cams3()[0], nfeatures=300, nlevels=3, fastThreshold=12, fastAgastType=1)
    assert len(ok) > 30 and G.first_diff(kps, ok) is None and G.first_diff(d, od) is None
    ex.close()
    with pytest.raises(G.mcs.McsError):
        G.mcs.Extractor(G.ctx(), 754, 480, fastAgastType=3)            # FAST has three types (AGAST's four: tests/test_gpu_agast.py)

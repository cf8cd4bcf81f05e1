"""-m gpu: AGAST (`extractor.useAgast` 1, `extractor.fastAgastType` 0 / 1 / 2 / 3 = AGAST_5_8 / AGAST_7_12d / AGAST_7_12s / OAST_9_16; reference
src/mdBRIEFextractorOct.cpp:869-870, 912-914) on the device: the type is a template parameter of k_fast_cells (csrc/mcs_fast.hip) — the plain segment test on the
type's ring, the ring's radius as the border inside a cell view (so neighbouring cells overlap and report a corner twice, as the reference does), and cv::AGAST's
region suppression walked by one wave.  tests/test_oracle_agast.py pins the restatement.  Candidates per level and the end-to-end outputs against the oracle, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_common
    return gpu_common


@pytest.mark.parametrize("atype,th,mode", [(0, 20, dict(do_dBrief=0, learnMasks=0)), (1, 20, dict(do_dBrief=1, learnMasks=1)), (2, 20, dict(do_dBrief=1, learnMasks=1)),
                                          (3, 20, dict(do_dBrief=1, learnMasks=1)), (2, 5, dict(do_dBrief=0, learnMasks=0)), (3, 7, dict(do_dBrief=1, learnMasks=0)),
                                          (0, 3, dict(do_dBrief=1, learnMasks=1)), (1, 1, dict(do_dBrief=0, learnMasks=0))])
def test_agast_bit_exact(G, atype, th, mode):
    imgs, masks, cams = G.frame_inputs(2)
    rng = np.random.default_rng(10 + atype)
    imgs = list(imgs)
    noisy = imgs[2].astype(np.int32)                      # isolated speckles and 2 x 2 blocks: 4-connected regions of corners with equal responses
    ys, xs = rng.integers(30, 450, 4000), rng.integers(30, 720, 4000)
    amp = rng.choice([-90, 90], 4000)
    noisy[ys, xs] += amp
    noisy[ys[:1500] + 1, xs[:1500]] += amp[:1500]
    noisy[ys[:700], xs[:700] + 1] += amp[:700]
    imgs[2] = np.clip(noisy, 0, 255).astype(np.uint8)
    imgs[1] = (imgs[1] // 24 * 24).astype(np.uint8)       # plateaus: many equal differences
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=3, nfeatures=600, fastThreshold=th, fastAgastType=atype, useAgast=1, **mode)
    res = ex.extract_host(imgs, masks, [G.mcs.make_ocam(c) for c in cams])
    total = dup = 0
    for i in range(3):
        oex, kps, d, dm, rays = G.oracle_extract(imgs[i], masks[i], cams[i], nfeatures=600, fastThreshold=th, fastAgastType=atype, useAgast=1, **mode)
        for l in range(8):
            x, y, s = ex.tap_candidates(i, l)
            c = oex.candidates(l)
            assert len(x) == len(c), (i, l, len(x), len(c))
            assert G.first_diff(np.stack([x, y, s], 1), np.stack([c["x"], c["y"], c["response"]], 1).astype(np.int32)) is None, (i, l)
            dup += len(x) - len(set(zip(x.tolist(), y.tolist())))
        gk, gd, gm, gr = res[i]
        assert G.first_diff(gk, kps) is None and G.first_diff(gd, d) is None and G.first_diff(gm, dm) is None and G.first_diff(gr, rays) is None
        total += len(kps)
    assert total > 150, total
    if atype in (0, 2):
        assert dup > 0, "cells overlap by 6 - 2 * border pixels: some corners must have been reported by two cells"
    ex.close()


@pytest.mark.parametrize("atype", [0, 1, 2, 3])
def test_agast_large_cells_instance_and_parameter_checks(G, atype):
    """a small image (cells larger than 44 px: the 64 x 64 kernel instance)"""
    rng = np.random.default_rng(30 + atype)
    img = np.clip(rng.normal(110, 30, (200, 260)), 0, 255).astype(np.uint8)
    img[40:160:7, 40:220:5] = 255
    ex = G.mcs.Extractor(G.ctx(), 260, 200, max_batch=1, nfeatures=300, nlevels=3, fastThreshold=12, fastAgastType=atype, useAgast=1)
    kps, d, dm, _ = ex.extract_host([img], None, [G.mcs.make_ocam(G.cams3()[0])])[0]
    oex, ok, od, odm, _ = G.oracle_extract(img, None, G.cams3()[0], nfeatures=300, nlevels=3, fastThreshold=12, fastAgastType=atype, useAgast=1)
    for l in range(3):
        x, y, s = ex.tap_candidates(0, l)
        c = oex.candidates(l)
        assert len(x) == len(c) and G.first_diff(np.stack([x, y, s], 1), np.stack([c["x"], c["y"], c["response"]], 1).astype(np.int32)) is None, l
    assert len(ok) > 30 and G.first_diff(kps, ok) is None and G.first_diff(d, od) is None
    ex.close()
    with pytest.raises(G.mcs.McsError):
        G.mcs.Extractor(G.ctx(), 754, 480, useAgast=1, fastAgastType=4)
    with pytest.raises(G.mcs.McsError):
        G.mcs.Extractor(G.ctx(), 754, 480, useAgast=1, fastAgastType=atype, fastThreshold=0)     # "no corner" is score 0 on the device
    with pytest.raises(G.mcs.McsError):
        G.mcs.Extractor(G.ctx(), 754, 480, useAgast=0, fastAgastType=3)                          # FAST has three types

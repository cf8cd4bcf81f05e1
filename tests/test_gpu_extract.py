"""-m gpu: the HIP extraction path (through the C ABI) vs the CPU oracle, stage by stage and end to end.  Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODES = {"orb": dict(do_dBrief=0, learnMasks=0), "dbrief": dict(do_dBrief=1, learnMasks=0), "mdbrief": dict(do_dBrief=1, learnMasks=1)}


@pytest.fixture(scope="module")
def G():
    import gpu_common
    return gpu_common


@pytest.fixture(scope="module")
def run_orb(G):
    imgs, masks, cams = G.frame_inputs(0)
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=3)
    res = ex.extract_host(imgs, masks, [G.mcs.make_ocam(c) for c in cams])
    orc = [G.oracle_extract(imgs[i], masks[i], cams[i]) for i in range(3)]
    return ex, res, orc


def test_level_tables(G, run_orb):
    ex = run_orb[0]
    assert ex.level_sizes == [(754, 480), (628, 400), (524, 333), (436, 278), (364, 231), (303, 193), (253, 161), (210, 134)]
    assert ex.features_per_level == [217, 181, 151, 126, 105, 87, 73, 60]


def test_pyramid_levels(G, run_orb):
    ex, _, orc = run_orb
    for i in range(3):
        for l in range(8):
            assert G.first_diff(ex.tap_level(i, l), orc[i][0].level_image(l)) is None, "img %d level %d" % (i, l)


def test_blurred_levels(G, run_orb):
    ex, _, orc = run_orb
    for i in range(3):
        for l in range(8):
            assert G.first_diff(ex.tap_level(i, l, blurred=True), orc[i][0].level_image(l, blurred=True)) is None, "img %d level %d" % (i, l)


def test_fast_candidates(G, run_orb):
    ex, _, orc = run_orb
    for i in range(3):
        for l in range(8):
            x, y, s = ex.tap_candidates(i, l)
            c = orc[i][0].candidates(l)
            assert len(x) == len(c), "img %d level %d: %d vs %d candidates" % (i, l, len(x), len(c))
            assert G.first_diff(np.stack([x, y, s], 1), np.stack([c["x"], c["y"], c["response"]], 1).astype(np.int32)) is None, (i, l)


def test_octree_selection(G, run_orb):
    ex, _, orc = run_orb
    for i in range(3):
        for l in range(8):
            x, y, s = ex.tap_selected(i, l)
            c = orc[i][0].selected(l)
            got = np.stack([x + 22, y + 22, s], 1)
            exp = np.stack([c["x"], c["y"], c["response"]], 1).astype(np.int32)
            assert G.first_diff(got, exp) is None, "img %d level %d" % (i, l)


@pytest.mark.parametrize("mode", list(MODES))
def test_end_to_end_bit_exact(G, mode):
    imgs, masks, cams = G.frame_inputs(1)
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=3, **MODES[mode])
    res = ex.extract_host(imgs, masks, [G.mcs.make_ocam(c) for c in cams])
    for i in range(3):
        _, kps, d, dm, rays = G.oracle_extract(imgs[i], masks[i], cams[i], **MODES[mode])
        gk, gd, gm, gr = res[i]
        assert len(gk) == len(kps) and len(kps) > 900
        for f in ("x", "y", "size", "angle", "response"):
            assert G.first_diff(gk[f].view(np.uint32), kps[f].view(np.uint32)) is None, (mode, i, f)
        assert (gk["octave"] == kps["octave"]).all() and (gk["class_id"] == -1).all()
        assert G.first_diff(gd, d) is None, (mode, i, "descriptors")
        assert G.first_diff(gm, dm) is None, (mode, i, "masks")
        assert G.first_diff(gr.view(np.uint64), rays.view(np.uint64)) is None, (mode, i, "rays")
    ex.close()


def test_no_mask_and_odd_sizes(G):
    rng = np.random.default_rng(4)
    cam = G.synth.scaled_camera(G.cams3()[0], 640, 360)
    img = G.synth.synth_image(0, 0, cam)
    ex = G.mcs.Extractor(G.ctx(), 640, 360, max_batch=2, nfeatures=500, nlevels=6, fastThreshold=12)
    noise = rng.integers(0, 256, (360, 640)).astype(np.uint8)            # dense noise: many candidates, deep oct-tree
    res = ex.extract_host([img, noise], None, [G.mcs.make_ocam(cam)] * 2)
    for im, r in zip([img, noise], res):
        _, kps, d, dm, rays = G.oracle_extract(im, None, cam, nfeatures=500, nlevels=6, fastThreshold=12)
        assert G.first_diff(r[0], kps) is None
        assert G.first_diff(r[1], d) is None
    ex.close()


def test_empty_image_gives_no_keypoints(G):
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=1)
    res = ex.extract_host([np.zeros((480, 754), np.uint8)], None, None, want_rays=False)
    assert len(res[0][0]) == 0
    ex.close()


def test_batch_equals_single(G):
    imgs, masks, cams = G.frame_inputs(3)
    oc = [G.mcs.make_ocam(c) for c in cams]
    ex3 = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=8, do_dBrief=1, learnMasks=1)
    a = ex3.extract_host(imgs * 2, masks * 2, oc * 2)
    b = ex3.extract_host(imgs[:1], masks[:1], oc[:1])
    assert G.first_diff(a[0][1], b[0][1]) is None and G.first_diff(a[3][1], b[0][1]) is None
    assert G.first_diff(a[0][2], a[3][2]) is None
    ex3.close()

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


def test_small_levels_with_cells_wider_than_40_px(G):
    """400x300, 8 levels: level 6 is 134x100, one cell row of 56 px — the FAST launch of levels 2.. takes the 60x60 LDS instance (four waves per cell)
    instead of the 40x40 one the usual level sizes get (csrc/mcs_fast.hip)"""
    cam = G.synth.scaled_camera(G.cams3()[1], 400, 300)
    img, mask = G.synth.synth_image(3, 1, cam), G.synth.mirror_mask(cam)
    ex = G.mcs.Extractor(G.ctx(), 400, 300, max_batch=1, nfeatures=600, nlevels=8, do_dBrief=1, learnMasks=1)
    (gk, gd, gm, gr), = ex.extract_host([img], [mask], [G.mcs.make_ocam(cam)])
    _, kps, d, dm, rays = G.oracle_extract(img, mask, cam, nfeatures=600, nlevels=8, do_dBrief=1, learnMasks=1)
    assert len(kps) > 100
    assert G.first_diff(gk, kps) is None and G.first_diff(gd, d) is None and G.first_diff(gm, dm) is None and G.first_diff(gr, rays) is None
    ex.close()


@pytest.mark.parametrize("w,h,sf,nl", [(275, 605, 1.2, 2), (345, 532, 1.5, 6), (376, 704, 1.2, 4)])
def test_portrait_levels_without_an_octree_root(G, w, h, sf, nl):
    """levels more than twice as tall as wide have nIni = round(width / height) = 0 oct-tree roots (reference :641-661 divides by it; defined only for a level
    without candidates: nothing from that level — the oracle's reading, and the library's since round 6; before, such extractors were refused).  275x605: every level;
    the others: the top levels only, the rest of the pyramid extracts as usual"""
    cam = G.synth.scaled_camera(G.cams3()[0], w, h)
    rng = np.random.default_rng(w)
    imgs = [G.synth.synth_image(0, 0, cam), rng.integers(0, 256, (h, w)).astype(np.uint8)]
    kw = dict(nfeatures=500, scaleFactor=sf, nlevels=nl, fastThreshold=12, do_dBrief=1, learnMasks=1)
    ex = G.mcs.Extractor(G.ctx(), w, h, max_batch=2, **kw)
    res = ex.extract_host(imgs, None, [G.mcs.make_ocam(cam)] * 2)
    total = 0
    for im, r in zip(imgs, res):
        oex = G.O.Extractor(**kw)
        oex.cap = max(oex.cap, ex.cap)
        kps, d, dm = oex(im, None, G.O.make_ocam(cam))
        assert G.first_diff(r[0], kps) is None and G.first_diff(r[1], d) is None and G.first_diff(r[2], dm) is None, (w, h)
        total += len(kps)
    assert (total == 0) == (w == 275), total
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


@pytest.mark.parametrize("desc_size,mode", [(16, "mdbrief"), (64, "mdbrief"), (64, "orb"), (16, "dbrief")])
def test_descriptor_sizes(G, desc_size, mode):
    """extractor.descSize 16 / 64 (reference: 'Extractor: 32 -> ORB , (16/32/64) -> dBRIEF and mdBRIEF')."""
    cam = G.cams3()[1]
    img, mask = G.synth.synth_image(7, 1, cam), G.synth.mirror_mask(cam)
    kw = dict(MODES[mode], descSize=desc_size, nfeatures=600)
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=1, **kw)
    gk, gd, gm, gr = ex.extract_host([img], [mask], [G.mcs.make_ocam(cam)])[0]
    _, kps, d, dm, rays = G.oracle_extract(img, mask, cam, **kw)
    assert gd.shape[1] == desc_size and G.first_diff(gk, kps) is None
    assert G.first_diff(gd, d) is None and G.first_diff(gm, dm) is None
    ex.close()


def test_big_rig_sizes_config4(G):
    """BASELINE configs[3]/[4] geometry: 1280x800, 2000 features per camera, scaled Lafida calibration."""
    cam = G.synth.scaled_camera(G.cams3()[0], 1280, 800)
    img, mask = G.synth.synth_image(1, 0, cam), G.synth.mirror_mask(cam)
    kw = dict(MODES["mdbrief"], nfeatures=2000)
    ex = G.mcs.Extractor(G.ctx(), 1280, 800, max_batch=2, **kw)
    res = ex.extract_host([img, img[::-1].copy()], [mask, mask[::-1].copy()], [G.mcs.make_ocam(cam)] * 2)
    for im, mk, r in [(img, mask, res[0]), (img[::-1].copy(), mask[::-1].copy(), res[1])]:
        _, kps, d, dm, rays = G.oracle_extract(im, mk, cam, **kw)
        assert len(kps) > 1900
        assert G.first_diff(r[0], kps) is None and G.first_diff(r[1], d) is None and G.first_diff(r[2], dm) is None
        assert G.first_diff(r[3].view(np.uint64), rays.view(np.uint64)) is None
    ex.close()


def test_other_pyramid_parameters(G):
    """scaleFactor 1.1 / 12 levels / FAST threshold 5 / 2N features (the init extractor of src/cTracking.cpp:152-158 uses th 5, 2N)."""
    cam = G.cams3()[2]
    img, mask = G.synth.synth_image(9, 2, cam), G.synth.mirror_mask(cam)
    for kw in (dict(nfeatures=2000, fastThreshold=5), dict(nfeatures=800, scaleFactor=1.1, nlevels=12), dict(nfeatures=300, scaleFactor=1.44, nlevels=4)):
        ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=1, **kw)
        gk, gd, gm, gr = ex.extract_host([img], [mask], [G.mcs.make_ocam(cam)])[0]
        _, kps, d, dm, rays = G.oracle_extract(img, mask, cam, **kw)
        assert G.first_diff(gk, kps) is None and G.first_diff(gd, d) is None, kw
        ex.close()


def test_unsupported_parameters_fail_loudly(G):
    with pytest.raises(G.mcs.McsError):
        G.mcs.Extractor(G.ctx(), 754, 480, useAgast=1, fastAgastType=4)   # AGAST: 0 .. 3 (tests/test_gpu_agast.py)
    with pytest.raises(G.mcs.McsError):
        G.mcs.Extractor(G.ctx(), 754, 480, fastAgastType=3)      # FAST: 0 / 1 / 2 = TYPE_5_8 / 7_12 / 9_16 (tests/test_gpu_fast_types.py)
    with pytest.raises(G.mcs.McsError):
        G.mcs.Extractor(G.ctx(), 754, 480, descSize=24)
    with pytest.raises(G.mcs.McsError):
        G.mcs.Extractor(G.ctx(), 120, 90)            # too small: a level has no 30-px FAST cell
    with pytest.raises(G.mcs.McsError):
        G.mcs.Extractor(G.ctx(), 754, 480, nfeatures=12000)      # level 0 would get 2604 + 3 oct-tree nodes: beyond the largest kernel instance (2048)


def test_host_images_with_padded_rows_and_gaps(G):
    """host-kind input is taken in the caller's layout (any row stride, any distance between images): the staged copy must cover exactly that span"""
    import ctypes as C
    imgs, masks, cams = G.frame_inputs(5)
    oc = (G.mcs.Ocam * 3)(*[G.mcs.make_ocam(c) for c in cams])
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=3, do_dBrief=1, learnMasks=1)
    tight = ex.extract_host(imgs, masks, list(oc))
    stride, gap = 754 + 14, 333
    pitch = 480 * stride + gap
    def pad(arrs, fill):
        buf = np.full(3 * pitch, fill, np.uint8)
        for i, a in enumerate(arrs):
            buf[i * pitch:i * pitch + 480 * stride].reshape(480, stride)[:, :754] = a
        return buf
    pi, pm = pad(imgs, 201), pad(masks, 77)
    pi, pm = pi[:2 * pitch + 479 * stride + 754].copy(), pm[:2 * pitch + 479 * stride + 754].copy()   # nothing readable behind the last pixel
    cap = ex.cap
    nkp = np.zeros(3, np.int32); kps = np.zeros((3, cap), G.mcs.KP_DTYPE); d = np.zeros((3, cap, 32), np.uint8); dm = np.zeros_like(d); rays = np.zeros((3, cap, 3))
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    G.mcs.check(G.mcs.lib().mcs_extract_batch(ex.h, 3, P(pi), pitch, stride, P(pm), pitch, stride, oc, G.mcs.MEM_HOST, P(nkp), P(kps), P(d), P(dm), P(rays)))
    for i in range(3):
        k = int(nkp[i])
        assert k == len(tight[i][0]) and G.first_diff(kps[i, :k], tight[i][0]) is None
        assert G.first_diff(d[i, :k], tight[i][1]) is None and G.first_diff(dm[i, :k], tight[i][2]) is None and G.first_diff(rays[i, :k], tight[i][3]) is None
    ex.close()


@pytest.mark.parametrize("case", [
    dict(scaleFactor=1.1, nlevels=8, nfeatures=600, descSize=32, do_dBrief=1, learnMasks=1),
    dict(scaleFactor=1.5, nlevels=5, nfeatures=500, descSize=32, do_dBrief=1, learnMasks=1),
    dict(scaleFactor=1.1, nlevels=12, nfeatures=1500, descSize=32, do_dBrief=0, learnMasks=0),
    dict(scaleFactor=1.2, nlevels=8, nfeatures=700, descSize=16, do_dBrief=1, learnMasks=1),
    dict(scaleFactor=1.2, nlevels=8, nfeatures=700, descSize=64, do_dBrief=1, learnMasks=1),
    dict(scaleFactor=1.2, nlevels=8, nfeatures=300, descSize=64, do_dBrief=1, learnMasks=0),
    dict(scaleFactor=1.3, nlevels=3, nfeatures=2000, descSize=32, do_dBrief=1, learnMasks=1, fastThreshold=9),
    # the reference sets no limit on nFeatures (src/mdBRIEFextractorOct.cpp:167-179): 5000 puts 1085 on level 0 (the 2048-node oct-tree instance), and the
    # low threshold gives every level more candidates than its quota, so the largest-first phase runs with > 1024 nodes
    dict(scaleFactor=1.2, nlevels=8, nfeatures=5000, descSize=32, do_dBrief=1, learnMasks=1, fastThreshold=7),
    dict(scaleFactor=1.2, nlevels=8, nfeatures=9000, descSize=32, do_dBrief=0, learnMasks=0, fastThreshold=5),
])
def test_parameter_space_matches_oracle(G, case):
    """the cases of tests/test_oracle_vs_ref.py::test_oracle_equals_reference_code_over_the_parameter_space (there: oracle == the reference's own code),
    here: HIP == oracle, on the Lafida sensor size and on the 1280x800 rig"""
    cams = G.cams3()
    big = G.synth.scaled_camera(cams[1], 1280, 800)
    for f, cam in ((3, cams[2]), (1, big)):
        img, mask = G.synth.synth_image(f, 1, cam), G.synth.mirror_mask(cam)
        ex = G.mcs.Extractor(G.ctx(), cam["width"], cam["height"], max_batch=1, **case)
        gk, gd, gm, gr = ex.extract_host([img], [mask], [G.mcs.make_ocam(cam)])[0]
        _, kps, d, dm, rays = G.oracle_extract(img, mask, cam, **case)
        assert len(kps) > 100 and G.first_diff(gk, kps) is None and G.first_diff(gd, d) is None and G.first_diff(gm, dm) is None, (case, cam["width"])
        assert G.first_diff(gr, rays) is None
        ex.close()

"""-m gpu: an extractor that meets more than 64 distinct camera models (the per-camera G(s) tables of the fast descriptor pass are cached per model,
csrc/mcs_capi.hip).  The 65th model arrives in the SAME batch as an image of a cached model: the table uploaded for every image must be its own camera's
(a cache trimmed in the middle of a batch would hand the fast pass another camera's G(s) — wrong descriptors without any fallback)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_more_than_64_camera_models_across_batches():
    import gpu_common as G
    cams = G.cams3()
    img0, img1 = G.synth.synth_image(0, 0, cams[0]), G.synth.synth_image(0, 1, cams[1])
    m0, m1 = G.synth.mirror_mask(cams[0]), G.synth.mirror_mask(cams[1])
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=2, nfeatures=300, do_dBrief=1, learnMasks=1)

    def variant(k):   # a distinct model per k: the principal point moves by a hundredth of a pixel
        c = dict(cams[1])
        c["u0"] = cams[1]["u0"] + 0.01 * k
        return c
    first = ex.extract_host([img0], [m0], [G.mcs.make_ocam(cams[0])])[0]          # model 1 of the cache
    for k in range(1, 64):                                                        # models 2 .. 64
        ex.extract_host([img1], [m1], [G.mcs.make_ocam(variant(k))])
    # model 1 (cached) and model 65 (new) in one batch
    newcam = variant(64)
    res = ex.extract_host([img0, img1], [m0, m1], [G.mcs.make_ocam(cams[0]), G.mcs.make_ocam(newcam)])
    for got, exp in zip(res[0], first):
        assert np.array_equal(got, exp)
    for (img, mask, cam), got in zip(((img0, m0, cams[0]), (img1, m1, newcam)), res):
        _, ok, od, om, orays = G.oracle_extract(img, mask, cam, nfeatures=300, do_dBrief=1, learnMasks=1)
        assert len(got[0]) == len(ok) > 100
        assert G.first_diff(got[1], od) is None and G.first_diff(got[2], om) is None
    # and the models keep being served by the fast pass afterwards
    again = ex.extract_host([img1, img0], [m1, m0], [G.mcs.make_ocam(variant(3)), G.mcs.make_ocam(cams[0])])
    for got, exp in zip(again[1], first):
        assert np.array_equal(got, exp)

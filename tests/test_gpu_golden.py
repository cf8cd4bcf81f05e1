"""-m gpu: the device against the REFERENCE's own outputs, without the oracle in between: tests/golden/ref_extract.npz holds what the reference's extractor code
(src/mdBRIEFextractorOct.cpp compiled unmodified, tools/gen_golden_ref.py) produced for eleven small cases — the three descriptor modes, descriptor sizes,
pyramid shapes, the small FAST rings and the four AGAST types.  Keypoints (every field, float bits), descriptors and masks must be those bytes."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def G():
    import gpu_common
    return gpu_common


def test_device_reproduces_reference_golden_vectors(G, synth):
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_extract.npz"))
    # (the generator's case tables, restated here: importing tools/gen_golden_ref.py would load the reference library, which the GPU box does not have)
    cases = [
        ("orb_376x240", 0, 0, 376, 240, 300, 1.2, 8, 20, 0, 0, 32, 2, 0), ("dbrief_376x240", 1, 1, 376, 240, 300, 1.2, 8, 20, 1, 0, 32, 2, 0),
        ("mdbrief_376x240", 2, 2, 376, 240, 300, 1.2, 8, 20, 1, 1, 32, 2, 0), ("mdbrief16_fast12_333x251", 3, 0, 333, 251, 250, 1.2, 6, 12, 1, 1, 16, 2, 0),
        ("mdbrief64_scale11_400x300", 4, 1, 400, 300, 400, 1.1, 10, 20, 1, 1, 64, 2, 0),
        ("fast7_12_376x240", 0, 1, 376, 240, 300, 1.2, 8, 8, 1, 1, 32, 1, 0), ("fast5_8_376x240", 1, 2, 376, 240, 300, 1.2, 8, 4, 0, 0, 32, 0, 0),
        ("agast5_8_376x240", 2, 0, 376, 240, 300, 1.2, 8, 20, 1, 1, 32, 0, 1), ("agast7_12d_376x240", 3, 1, 376, 240, 300, 1.2, 8, 20, 0, 0, 32, 1, 1),
        ("agast7_12s_376x240", 4, 2, 376, 240, 300, 1.2, 8, 20, 1, 1, 32, 2, 1), ("oast9_16_333x251", 5, 0, 333, 251, 250, 1.2, 6, 12, 1, 0, 32, 3, 1),
    ]
    assert sorted(n for n in g.files if n.endswith("_kps")) == sorted(c[0] + "_kps" for c in cases)
    for name, frame, ci, w, h, nf, sf, nl, th, db, lm, ds, ft, ag in cases:
        cam = synth.scaled_camera(synth.lafida_cameras()[ci], w, h)
        img = synth.synth_image(frame, ci, cam)
        mask = np.ascontiguousarray(synth.mirror_mask(cam))
        ex = G.mcs.Extractor(G.ctx(), w, h, max_batch=1, nfeatures=nf, scaleFactor=sf, nlevels=nl, fastThreshold=th, do_dBrief=db, learnMasks=lm, descSize=ds,
                             fastAgastType=ft, useAgast=ag)
        kps, d, dm, _ = ex.extract_host([img], [mask], [G.mcs.make_ocam(cam)])[0]
        gk, gd, gm = g[name + "_kps"], g[name + "_desc"], g[name + "_mask"]
        assert len(kps) == len(gk), (name, len(kps), len(gk))
        for f in gk.dtype.names:
            assert np.array_equal(np.asarray(kps[f]).view(np.uint32) if gk[f].dtype == np.float32 else kps[f],
                                  gk[f].view(np.uint32) if gk[f].dtype == np.float32 else gk[f]), (name, f)
        assert np.array_equal(d, gd) and np.array_equal(dm, gm), name
        ex.close()

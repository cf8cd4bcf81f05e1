"""-m gpu: the reference's Lafida example driven from its own file formats (calibration YAMLs, settings YAML, image list):
LoadMCS -> make_extractors -> cMultiFrame, with the shipped settings (ORB mode, 400 features, FAST 20) and the initialisation extractor
(800 features, FAST 5) of src/cTracking.cpp:152-158, vs the oracle.  Bit-exact."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_lafida_example_flow(tmp_path):
    import gpu_common as G
    import test_io_formats as T
    io = importlib.import_module("multicol-slam_amd.io")
    FE = importlib.import_module("multicol-slam_amd.frontend")
    cams = G.cams3()
    T.write_lafida_dir(str(tmp_path), cams, T.CAYLEY)
    (tmp_path / "Slam_Settings_indoor1.yaml").write_text(T.SETTINGS)
    rig = io.LoadMCS(str(tmp_path))
    ex, ini = io.make_extractors(str(tmp_path / "Slam_Settings_indoor1.yaml"), rig.GetNrCams(), ctx=G.ctx())
    # the image set arrives through the example's loader: list file -> PGM files
    imgs = G.synth.synth_multiframe(3, cams)
    lines = []
    for c, im in enumerate(imgs):
        (tmp_path / ("cam%d_1.pgm" % c)).write_bytes(b"P5\n%d %d\n255\n" % (im.shape[1], im.shape[0]) + np.ascontiguousarray(im).tobytes())
    (tmp_path / "images_and_timestamps.txt").write_text("0.00 skipped0 skipped1 skipped2\n0.04 cam0_1.pgm cam1_1.pgm cam2_1.pgm\n")
    names, stamps = io.LoadImagesAndTimestamps(2, 3, str(tmp_path))
    imgSet = [io.read_pgm(names[c][0]) for c in range(3)]
    assert stamps == [0.04] and all(np.array_equal(a, b) for a, b in zip(imgSet, imgs))
    for extractors, nfeat, fastTh in ((ex, 400, 20), (ini, 800, 5)):
        F = FE.cMultiFrame(imgSet, stamps[0], extractors, None, rig, 0)
        assert not F.masksLearned and F.descDimension == 32 and F.mnScaleLevels == 8
        s = 0
        for c in range(3):
            _, ek, ed, _, er = G.oracle_extract(imgs[c], G.synth.mirror_mask(cams[c]), cams[c], nfeatures=nfeat, fastThreshold=fastTh, do_dBrief=0, learnMasks=0)
            n = F.N[c]
            assert n == len(ek) and n > 0.8 * nfeat
            assert G.first_diff(F.mvKeys[s:s + n], ek) is None and G.first_diff(F.mDescriptors[c], ed) is None
            assert G.first_diff(F.mvKeysRays[s:s + n], er) is None
            s += n
    # the rig's poses come from the Cayley file: projecting a feature's own bearing ray lands on the feature again
    i = 5
    c = int(F.keypoint_to_cam[i])
    Xw = (rig.MtMc[c] @ np.append(F.mvKeysRays[i] * 2.5, 1.0))[:3]
    uv = rig.WorldToCamHom_fast(c, Xw)
    assert abs(uv[0] - F.mvKeys[i]["x"]) < 0.05 and abs(uv[1] - F.mvKeys[i]["y"]) < 0.05

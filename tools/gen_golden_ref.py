#!/usr/bin/env python3
"""Golden vectors from the REFERENCE's own extractor code (oracle/_ref/libmcs_ref.so = /root/reference/src/mdBRIEFextractorOct.cpp +
cam_model_omni.cpp compiled unmodified against oracle/cvshim, see oracle/Makefile `ref`).  Run in the container that has /root/reference:

    make -C oracle ref && python tools/gen_golden_ref.py

Writes tests/golden/ref_extract.npz: for a few small synthetic images and all three descriptor modes the keypoints, descriptors and masks the
reference code produced.  tests/test_oracle_vs_ref.py checks the oracle against them everywhere (the GPU box has no reference checkout)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_compare as R

CASES = [  # (name, frame, cam, width, height, nfeatures, scale, nlevels, fastTh, do_dBrief, learnMasks, descSize, use mask)
    ("orb_376x240", 0, 0, 376, 240, 300, 1.2, 8, 20, 0, 0, 32, 1),
    ("dbrief_376x240", 1, 1, 376, 240, 300, 1.2, 8, 20, 1, 0, 32, 1),
    ("mdbrief_376x240", 2, 2, 376, 240, 300, 1.2, 8, 20, 1, 1, 32, 1),
    ("mdbrief16_fast12_333x251", 3, 0, 333, 251, 250, 1.2, 6, 12, 1, 1, 16, 1),
    ("mdbrief64_scale11_400x300", 4, 1, 400, 300, 400, 1.1, 10, 20, 1, 1, 64, 1),
]
# detector options (extractor.useAgast / extractor.fastAgastType, reference src/mdBRIEFextractorOct.cpp:869-872, 912-917): the same tuple + (fastAgastType, useAgast).
# The detectors themselves are OpenCV's (restated in oracle/, forwarded by oracle/cvshim); what the reference's code contributes to these vectors is the cell loop
# with its overlapping views, the oct-tree over corners reported twice, and everything downstream.
CASES_DET = [
    ("fast7_12_376x240", 0, 1, 376, 240, 300, 1.2, 8, 8, 1, 1, 32, 1, 1, 0),
    ("fast5_8_376x240", 1, 2, 376, 240, 300, 1.2, 8, 4, 0, 0, 32, 1, 0, 0),
    ("agast5_8_376x240", 2, 0, 376, 240, 300, 1.2, 8, 20, 1, 1, 32, 1, 0, 1),
    ("agast7_12d_376x240", 3, 1, 376, 240, 300, 1.2, 8, 20, 0, 0, 32, 1, 1, 1),
    ("agast7_12s_376x240", 4, 2, 376, 240, 300, 1.2, 8, 20, 1, 1, 32, 1, 2, 1),
    ("oast9_16_333x251", 5, 0, 333, 251, 250, 1.2, 6, 12, 1, 0, 32, 1, 3, 1),
]


def case_inputs(frame, cam_idx, w, h, use_mask):
    cam = R.synth.scaled_camera(R.synth.lafida_cameras()[cam_idx], w, h)
    img = R.synth.synth_image(frame, cam_idx, cam)
    mask = np.ascontiguousarray(R.synth.mirror_mask(cam)) if use_mask else None
    return cam, img, mask


if __name__ == "__main__":
    out = {}
    for name, frame, ci, w, h, nf, sf, nl, th, db, lm, ds, um in CASES:
        cam, img, mask = case_inputs(frame, ci, w, h, um)
        k, d, m = R.run_ref(img, mask, cam, nfeatures=nf, scaleFactor=sf, nlevels=nl, fastThreshold=th, do_dBrief=db, learnMasks=lm, descSize=ds)
        out[name + "_kps"], out[name + "_desc"], out[name + "_mask"] = k, d, m
        print(name, len(k), "keypoints")
    for name, frame, ci, w, h, nf, sf, nl, th, db, lm, ds, um, ft, ag in CASES_DET:
        cam, img, mask = case_inputs(frame, ci, w, h, um)
        k, d, m = R.run_ref(img, mask, cam, nfeatures=nf, scaleFactor=sf, nlevels=nl, fastThreshold=th, do_dBrief=db, learnMasks=lm, descSize=ds, fastAgastType=ft, useAgast=ag)
        out[name + "_kps"], out[name + "_desc"], out[name + "_mask"] = k, d, m
        print(name, len(k), "keypoints")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_extract.npz"), **out)

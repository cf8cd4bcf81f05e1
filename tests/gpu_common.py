"""Helpers shared by the -m gpu parity tests: run the HIP path (through the C ABI) and the CPU oracle on the same inputs."""
import importlib

import numpy as np

import oracle_lib as O

mcs = importlib.import_module("multicol-slam_amd")
synth = importlib.import_module("multicol-slam_amd.synth")

_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        _ctx = mcs.Context(0)
    return _ctx


def cams3():
    return synth.lafida_cameras()


def frame_inputs(frame, ncam=3, cams=None):
    cams = cams or cams3()
    imgs = [synth.synth_image(frame, c, cams[c % len(cams)]) for c in range(ncam)]
    masks = [synth.mirror_mask(cams[c % len(cams)]) for c in range(ncam)]
    return imgs, masks, [cams[c % len(cams)] for c in range(ncam)]


def oracle_extract(img, mask, cam, **kw):
    ex = O.Extractor(**kw)
    oc = O.make_ocam(cam)
    kps, d, dm = ex(img, mask, oc)
    rays = np.zeros((len(kps), 3))
    if len(kps):
        O.lib().orc_rays(oc, O.ptr(kps), len(kps), O.ptr(rays))
    return ex, kps, d, dm, rays


def first_diff(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return "shape %s vs %s" % (a.shape, b.shape)
    bad = np.argwhere(a != b)
    if len(bad) == 0:
        return None
    i = tuple(bad[0])
    return "%d mismatches, first at %s: %s vs %s" % (len(bad), i, a[i], b[i])

"""CPU: the oracle's OpenCV primitive restatements against INDEPENDENT statements of what the primitives compute.
No OpenCV exists in this environment, so these primitives cannot be pinned bit for bit; what can be checked is that the
fixed-point / closed-form code agrees with the plain mathematical definition — sampling positions, border handling, corner criterion, score,
suppression, angle convention — to within the rounding the fixed-point forms are allowed:
  resize INTER_LINEAR   vs float bilinear interpolation at half-pixel centres (torch, align_corners=False): at most 1 grey level apart
  resize INTER_NEAREST  vs floor(x * scale) sampling: identical
  boxFilter 5x5         vs round(sum / 25) over a reflect-101 frame: identical (sum / 25 never ends in .5)
  FAST 9/16             vs the segment-test definition evaluated by brute force (corner set, score = largest threshold that still gives a
                        corner, strict 8-neighbour suppression, row-major order): identical
  fastAtan2             vs atan2 in degrees: within OpenCV's documented 0.3 degrees, in [0, 360)"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle_lib as O

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def rand_image(h, w, seed, smooth=True):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w)).astype(np.float64)
    if smooth:   # blocks of similar grey levels with sharp edges: corners, not noise
        img = np.kron(rng.integers(0, 256, ((h + 7) // 8, (w + 7) // 8)), np.ones((8, 8)))[:h, :w] + rng.integers(-6, 7, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("sw,sh,dw,dh", [(754, 480, 628, 400), (628, 400, 524, 333), (253, 161, 210, 134), (97, 61, 81, 51)])
def test_resize_linear_is_bilinear_at_half_pixel_centres(sw, sh, dw, dh):
    src = rand_image(sh, sw, sw + dh)
    dst = np.zeros((dh, dw), np.uint8)
    O.lib().orc_resize_linear(O.ptr(src), sw, sh, sw, O.ptr(dst), dw, dh, dw)
    ref = F.interpolate(torch.from_numpy(src.astype(np.float64))[None, None], size=(dh, dw), mode="bilinear", align_corners=False)[0, 0].numpy()
    assert np.abs(dst.astype(np.float64) - ref).max() <= 1.0 + 1e-9
    assert np.abs(dst.astype(np.float64) - ref).mean() < 0.3


@pytest.mark.parametrize("sw,sh,dw,dh", [(754, 480, 628, 400), (303, 193, 253, 161)])
def test_resize_nearest_samples_floor_positions(sw, sh, dw, dh):
    src = rand_image(sh, sw, 3, smooth=False)
    dst = np.zeros((dh, dw), np.uint8)
    O.lib().orc_resize_nearest(O.ptr(src), sw, sh, sw, O.ptr(dst), dw, dh, dw)
    xs = np.minimum(np.floor(np.arange(dw) * (sw / dw)).astype(int), sw - 1)
    ys = np.minimum(np.floor(np.arange(dh) * (sh / dh)).astype(int), sh - 1)
    assert np.array_equal(dst, src[ys][:, xs])


def test_box_filter_is_the_rounded_mean_over_a_reflect101_frame():
    h, w, b = 57, 83, 25
    img = rand_image(h, w, 11)
    buf = np.zeros((h + 2 * b, w + 2 * b), np.uint8)
    buf[b:b + h, b:b + w] = img
    O.lib().orc_border_reflect101(O.ptr(buf), w, h, w + 2 * b, b)
    assert np.array_equal(buf, np.pad(img, b, mode="reflect"))            # numpy's "reflect" is BORDER_REFLECT_101
    work = buf.copy()
    roi = work[b:, b:]                                                   # view starting at the ROI origin, same stride
    O.lib().orc_box5_inplace(C.c_void_p(work.ctypes.data + b * work.strides[0] + b), w, h, w + 2 * b)
    pad = buf.astype(np.int64)
    acc = sum(pad[b + dy:b + dy + h, b + dx:b + dx + w] for dy in range(-2, 3) for dx in range(-2, 3))
    assert not np.any((2 * acc) % 50 == 25)                              # sum / 25 never ends in .5 (25 is odd), so every rounding mode agrees
    assert np.array_equal(roi[:h, :w], np.rint(acc / 25.0).astype(np.uint8))


def fast_by_definition(img, t):
    """corner <=> 9 contiguous circle pixels all brighter than v + t or all darker than v - t; score = the largest threshold that still gives a
    corner; kept <=> score strictly above the scores of all 8 neighbours; rows and columns 3 .. size-4, row-major."""
    h, w = img.shape
    v = img[3:h - 3, 3:w - 3].astype(np.int64)
    ring = np.stack([img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int64) for dx, dy in CIRCLE])   # (16, h-6, w-6), offsets are (x, y)
    d = v[None] - ring
    arcs_dark = np.stack([np.min(np.stack([d[(k + i) % 16] for i in range(9)]), axis=0) for k in range(16)]).max(axis=0)    # all d > t' <=> t' < this
    arcs_bright = np.stack([np.min(np.stack([-d[(k + i) % 16] for i in range(9)]), axis=0) for k in range(16)]).max(axis=0)
    best = np.maximum(arcs_dark, arcs_bright)          # corner at threshold t'  <=>  best > t'
    score = np.where(best > t, best - 1, 0)            # largest t' with best > t'
    sp = np.pad(score, 1)
    nb = np.max(np.stack([sp[1 + dy:1 + dy + score.shape[0], 1 + dx:1 + dx + score.shape[1]] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0)]), axis=0)
    keep = (score > 0) & (score > nb)
    ys, xs = np.nonzero(keep)
    return [(int(x) + 3, int(y) + 3, int(score[y, x])) for y, x in zip(ys, xs)]


@pytest.mark.parametrize("seed,t", [(1, 20), (2, 7), (3, 40)])
def test_fast_9_16_matches_the_segment_test_definition(seed, t):
    img = rand_image(66, 90, seed)
    out = np.zeros(4096, O.KP_DTYPE)
    n = O.lib().orc_fast9_16(O.ptr(img), img.shape[1], img.shape[0], img.shape[1], None, 0, t, O.ptr(out), len(out))
    got = [(int(k["x"]), int(k["y"]), int(k["response"])) for k in out[:n]]
    exp = fast_by_definition(img, t)
    assert got == exp and len(exp) > 10
    assert all(k["size"] == 7.0 and k["angle"] == -1.0 and k["octave"] == 0 and k["class_id"] == -1 for k in out[:n])


def test_fast_atan2_is_atan2_in_degrees():
    rng = np.random.default_rng(5)
    y = np.concatenate([rng.normal(0, 1e4, 4000), [0, 0, 1, -1, 5, -5]]).astype(np.float32)
    x = np.concatenate([rng.normal(0, 1e4, 4000), [1, -1, 0, 0, 5, -5]]).astype(np.float32)
    got = np.array([O.lib().orc_fastAtan2(float(a), float(b)) for a, b in zip(y, x)])
    ref = np.degrees(np.arctan2(y.astype(np.float64), x.astype(np.float64))) % 360.0
    diff = np.abs((got - ref + 180.0) % 360.0 - 180.0)
    assert diff.max() < 0.3 and (got >= 0).all() and (got < 360.0 + 1e-3).all()

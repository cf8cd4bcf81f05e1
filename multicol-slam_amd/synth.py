"""Deterministic synthetic fisheye inputs.  No dataset ships with the reference
(Lafida is an external download, reference README.md:242-243) and there is no network here.

Image (frame f, camera c): canvas W x H u8, background 96; `nshapes` random rotated rectangles
(centre uniform, side 8..80 px, rotation uniform, gray uniform 0..255) in painter's order, seeded by the
camera; the whole scene shifted by (3,1) px per frame so consecutive frames match; per-pixel uniform
noise +-3 seeded by 1000*f+c; one 3x3 box pass; everything outside the mirror circle
(centre (u0,v0), radius v0+22, the reference's CreateMirrorMask level 0) set to 0.
"""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lafida_cameras():
    with open(os.path.join(_HERE, "calib", "lafida.json")) as f:
        return json.load(f)["cameras"]


def scaled_camera(cam, width, height):
    """Scale a Lafida calibration to another sensor size: u0,v0 and invP by s, p[i] by s^(1-i)."""
    s = width / cam["width"]
    out = dict(cam)
    out["width"], out["height"] = width, height
    out["u0"], out["v0"] = cam["u0"] * s, cam["v0"] * s
    out["invP"] = [v * s for v in cam["invP"]]
    out["p"] = [v * s ** (1 - i) for i, v in enumerate(cam["p"])]
    return out


def mirror_mask(cam):
    """CreateMirrorMask level 0 (reference src/cam_model_omni.cpp:163-220), float32 arithmetic like the reference."""
    h, w = cam["height"], cam["width"]
    u0 = np.float32(cam["v0"])  # sic: the reference swaps the names
    v0 = np.float32(cam["u0"])
    i = np.arange(h, dtype=np.float32)[:, None]
    j = np.arange(w, dtype=np.float32)[None, :]
    a = ((i - u0).astype(np.float64) ** 2).astype(np.float32) + ((j - v0).astype(np.float64) ** 2).astype(np.float32)
    ans = np.sqrt(a.astype(np.float32))
    return np.where(ans < (u0 + np.float32(22.0)), 255, 0).astype(np.uint8)


SCENE_LEN = 8   # frames of one scene in a long stream (stream_image): after that many (3,1)-px shifts a new scene is seeded


def _scene(cam_idx, width, height, nshapes, scene=0):
    rng = np.random.Generator(np.random.PCG64(7919 + cam_idx + 104729 * scene))
    cx = rng.uniform(0, width, nshapes)
    cy = rng.uniform(0, height, nshapes)
    sx = rng.uniform(8, 80, nshapes)
    sy = rng.uniform(8, 80, nshapes)
    rot = rng.uniform(0, np.pi, nshapes)
    gray = rng.integers(0, 256, nshapes)
    return cx, cy, sx, sy, rot, gray


def _paint(shapes, w, h, dx, dy):
    """the shapes in painter's order on a w x h canvas whose pixel (x, y) shows scene point (x - dx, y - dy)"""
    cx, cy, sx, sy, rot, gray = shapes
    img = np.full((h, w), 96, np.int32)
    for k in range(len(cx)):
        x0, y0 = cx[k] + dx, cy[k] + dy
        r = 0.5 * np.hypot(sx[k], sy[k]) + 1
        xa, xb = int(max(0, np.floor(x0 - r))), int(min(w, np.ceil(x0 + r)))
        ya, yb = int(max(0, np.floor(y0 - r))), int(min(h, np.ceil(y0 + r)))
        if xa >= xb or ya >= yb:
            continue
        yy, xx = np.mgrid[ya:yb, xa:xb]
        c, s = np.cos(rot[k]), np.sin(rot[k])
        u = (xx - x0) * c + (yy - y0) * s
        v = -(xx - x0) * s + (yy - y0) * c
        inside = (np.abs(u) <= sx[k] / 2) & (np.abs(v) <= sy[k] / 2)
        img[ya:yb, xa:xb][inside] = gray[k]
    return img


_CANVAS = {}


def synth_image(frame, cam_idx, cam, nshapes=None, scene=0):
    w, h = cam["width"], cam["height"]
    if nshapes is None:
        nshapes = int(round(600 * (w * h) / (754.0 * 480.0)))
    if 0 <= frame < SCENE_LEN:
        # the SCENE_LEN frames of a scene are integer shifts of one another: paint the scene once on a canvas with a margin and crop (same pixels as
        # painting every frame, the shapes' inside tests only see integer-shifted coordinates)
        key = (cam_idx, w, h, nshapes, scene)
        if key not in _CANVAS:
            if len(_CANVAS) > 64:
                _CANVAS.clear()
            mx, my = 3 * (SCENE_LEN - 1), SCENE_LEN - 1
            _CANVAS[key] = _paint(_scene(cam_idx, w, h, nshapes, scene), w + mx, h + my, mx, my)
        mx, my = 3 * (SCENE_LEN - 1 - frame), SCENE_LEN - 1 - frame
        img = _CANVAS[key][my:my + h, mx:mx + w]
    else:
        img = _paint(_scene(cam_idx, w, h, nshapes, scene), w, h, 3 * frame, 1 * frame)
    rng = np.random.Generator(np.random.PCG64(1000 * frame + cam_idx + 1000003 * scene))
    img = np.clip(img + rng.integers(-3, 4, img.shape), 0, 255)
    pad = np.pad(img, 1, mode="edge")
    acc = sum(pad[1 + a:1 + a + h, 1 + b:1 + b + w] for a in (-1, 0, 1) for b in (-1, 0, 1))
    img = ((acc + 4) // 9).astype(np.uint8)
    img[mirror_mask(cam) == 0] = 0
    return img


def stream_image(f, cam_idx, cam, pool=64):
    """frame f of a long synthetic stream: `pool` distinct multi-frames, a new scene every SCENE_LEN frames (so the content never drifts out of the
    image), the stream repeating after `pool` frames"""
    f %= pool
    return synth_image(f % SCENE_LEN, cam_idx, cam, scene=f // SCENE_LEN)


def synth_multiframe(frame, cams):
    return [synth_image(frame, c, cam) for c, cam in enumerate(cams)]

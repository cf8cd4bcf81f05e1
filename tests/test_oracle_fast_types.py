"""CPU: the oracle's FAST for the two small rings — FastFeatureDetector TYPE_7_12 and TYPE_5_8, selected by `extractor.fastAgastType` 1 / 0
(reference src/mdBRIEFextractorOct.cpp:869-872, 912-914; read at src/cTracking.cpp:129-131).  OpenCV is not vendored in the reference tree, so these are
restatements of OpenCV 3.x's FAST_t<patternSize> / cornerScore<patternSize> (oracle/mcs_oracle.cpp, "parity unpinned" against a real OpenCV like the
other image primitives).  What is checked here:
  * known answers worked out by hand, including the two places where OpenCV's code is NOT the "N contiguous pixels of the ring" definition its type
    names suggest: the quick rejection test reads entries 0|8 ... 7|15 of a 25-entry offset table that has wrapped around for the small rings — TYPE_5_8
    then needs all 8 ring pixels darker (or all brighter), TYPE_7_12 needs the pairs (0,8) (2,10) (4,0) (6,2) (1,9) (3,11) (5,1) (7,3) — and every ring
    keeps the 3-pixel image border of the 16-pixel ring;
  * the C++ restatement against an independent, literal Python transcription of the same published algorithm (offset table with its wrap-around,
    threshold table, run counting over N = patternSize + K + 1 entries, cornerScore's pairwise min / max loops, 3-row non-max suppression, mask filter),
    corner for corner, on random and structured images."""
import numpy as np
import pytest

OFFS = {
    16: [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)],
    12: [(0, 2), (1, 2), (2, 1), (2, 0), (2, -1), (1, -2), (0, -2), (-1, -2), (-2, -1), (-2, 0), (-2, 1), (-1, 2)],
    8: [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1)],
}
PSIZE = {0: 8, 1: 12, 2: 16}


def py_fast(img, ftype, threshold, mask=None):
    """literal transcription of FAST_t<patternSize> with non-max suppression + KeyPointsFilter::runByPixelsMask; returns [(x, y, score)] in emission order"""
    P = PSIZE[ftype]
    K, N = P // 2, P + P // 2 + 1
    h, w = img.shape
    ring = OFFS[P]
    pixel = [ring[k] if k < P else None for k in range(25)]
    for k in range(P, 25):
        pixel[k] = pixel[k - P]
    im = img.astype(np.int32)

    def px(y, x, k):
        dx, dy = pixel[k]
        return int(im[y + dy, x + dx])

    def tab(diff):   # threshold_tab[x - v + 255]
        return 1 if diff < -threshold else (2 if diff > threshold else 0)

    def score(y, x):
        v = int(im[y, x])
        d = [v - px(y, x, k) for k in range(K * 3 + 1)]
        R = K   # pairwise loops: a spans d[k+1 .. k+K], extended by d[k] and d[k+K+1]
        a0 = threshold
        for k in range(0, P, 2):
            a = min(d[k + 1:k + R + 1])
            if a <= a0:
                continue
            a0 = max(a0, min(a, d[k]))
            a0 = max(a0, min(a, d[k + R + 1]))
        b0 = -a0
        for k in range(0, P, 2):
            b = max(d[k + 1:k + R + 1])
            if b >= b0:
                continue
            b0 = min(b0, max(b, d[k]))
            b0 = min(b0, max(b, d[k + R + 1]))
        return -b0 - 1

    sc = np.zeros((h, w), np.int32)
    corner = np.zeros((h, w), bool)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = int(im[y, x])
            d = 3
            for a, b in ((0, 8), (2, 10), (4, 12), (6, 14), (1, 9), (3, 11), (5, 13), (7, 15)):
                d &= tab(px(y, x, a) - v) | tab(px(y, x, b) - v)
            found = False
            for bit, sign in ((1, -1), (2, 1)):
                if not (d & bit):
                    continue
                count = 0
                for k in range(N):
                    xk = px(y, x, k)
                    hit = xk < v - threshold if sign < 0 else xk > v + threshold
                    if hit:
                        count += 1
                        if count > K:
                            found = True
                            break
                    else:
                        count = 0
            if found:
                corner[y, x] = True
                sc[y, x] = score(y, x) & 0xFF
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if not corner[y, x]:
                continue
            s = sc[y, x]
            nb = [sc[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0)]
            if all(s > n for n in nb):
                if mask is not None and mask[int(np.float32(y) + np.float32(0.5)), int(np.float32(x) + np.float32(0.5))] == 0:
                    continue
                out.append((x, y, int(s)))
    return out


def orc_fast(oracle, img, ftype, threshold, mask=None):
    L = oracle.lib()
    img = np.ascontiguousarray(img)
    kps = np.zeros(img.size + 1, oracle.KP_DTYPE)
    m = None if mask is None else np.ascontiguousarray(mask)
    n = L.orc_fast_type(ftype, oracle.ptr(img), img.shape[1], img.shape[0], img.strides[0], None if m is None else oracle.ptr(m), 0 if m is None else m.strides[0],
                        threshold, oracle.ptr(kps), len(kps))
    assert 0 <= n <= len(kps)
    return [(int(k["x"]), int(k["y"]), int(k["response"])) for k in kps[:n]]


def ring_image(ftype, values, centre=100, size=9):
    """a size x size image of `centre` with the ring around the middle pixel set to `values` (ring order of OpenCV's offset table)"""
    img = np.full((size, size), centre, np.uint8)
    c = size // 2
    for (dx, dy), v in zip(OFFS[PSIZE[ftype]], values):
        img[c + dy, c + dx] = v
    return img


def test_type_5_8_needs_the_whole_ring(oracle):
    t = 20
    assert orc_fast(oracle, ring_image(0, [60] * 8), 0, t) == [(4, 4, 39)]           # all 8 darker by 40: score = 40 - 1
    assert orc_fast(oracle, ring_image(0, [150] * 8), 0, t) == [(4, 4, 49)]          # all 8 brighter by 50
    seven = ring_image(0, [60] * 7 + [100])                                           # 7 contiguous darker pixels: a 5-of-8 segment, yet no corner
    assert orc_fast(oracle, seven, 0, t) == []
    assert orc_fast(oracle, ring_image(0, [60, 70, 60, 60, 75, 60, 60, 60]), 0, t) == [(4, 4, 29)]   # best 5-arc avoids the 75 but must contain the 70: min d = 30
    full = np.ascontiguousarray(ring_image(0, [60] * 8))
    assert oracle.lib().orc_fast_score_type(0, full.ctypes.data + 4 * 9 + 4, 9, t) == 39


def test_type_7_12_pretest_pairs_and_run(oracle):
    t = 20
    assert orc_fast(oracle, ring_image(1, [50] * 12), 1, t) == [(4, 4, 49)]
    run7 = [50] * 7 + [100] * 5                                                       # ring entries 0..6 darker: a 7-run, and every pretest pair has a dark member
    assert orc_fast(oracle, ring_image(1, run7), 1, t) == [(4, 4, 49)]               # pairs (0,8) (2,10) (4,0) (6,2) (1,9) (3,11) (5,1) (7,3): 0,2,4/0,6/2,1,3,5/1,3 dark
    shifted = [100] * 2 + [50] * 7 + [100] * 3                                        # entries 2..8: the same 7-run two steps on — pair (1, 9) has no dark member
    assert orc_fast(oracle, ring_image(1, shifted), 1, t) == []
    assert orc_fast(oracle, ring_image(1, [50] * 6 + [100] * 6), 1, t) == []         # only 6 contiguous
    # the 3-pixel border is kept although the ring has radius 2: a corner 2 pixels from the edge is not even examined
    img = np.full((9, 9), 100, np.uint8)
    for dx, dy in OFFS[12]:
        img[2 + dy, 2 + dx] = 50
    assert orc_fast(oracle, img, 1, t) == []


@pytest.mark.parametrize("ftype", [0, 1, 2])
def test_restatement_equals_literal_python_transcription(oracle, synth, ftype):
    rng = np.random.default_rng(40 + ftype)
    cams = synth.lafida_cameras()
    scene = synth.synth_image(1, 0, cams[0])[180:228, 300:364]                      # real bench content (edges, noise)
    blobs = np.full((40, 44), 90, np.uint8)
    for _ in range(60):                                                              # isolated dark / bright dots and small squares: what the small rings respond to
        y, x, r = rng.integers(3, 37), rng.integers(3, 41), rng.integers(0, 2)
        blobs[y:y + 1 + r, x:x + 1 + r] = rng.choice([10, 40, 160, 230])
    noise = rng.integers(0, 256, (36, 36)).astype(np.uint8)
    soft = np.clip(rng.normal(120, 12, (40, 40)), 0, 255).astype(np.uint8)
    total = 0
    for img, th in ((scene, 20), (scene, 5), (blobs, 20), (blobs, 7), (noise, 30), (noise, 3), (soft, 6), (soft, 2)):
        mask = (rng.random(img.shape) < 0.8).astype(np.uint8) * 255
        for m in (None, mask):
            want = py_fast(img, ftype, th, m)
            got = orc_fast(oracle, img, ftype, th, m)
            assert got == want, (ftype, th, m is not None, len(got), len(want))
            total += len(want)
    assert total > 50


def test_extractor_accepts_the_small_rings(oracle, synth):
    cams = synth.lafida_cameras()
    img, mask = synth.synth_image(0, 0, cams[0]), synth.mirror_mask(cams[0])
    n16 = len(oracle.Extractor(nfeatures=500)(img, mask, oracle.make_ocam(cams[0]))[0])
    n12 = len(oracle.Extractor(nfeatures=500, fastAgastType=1, fastThreshold=8)(img, mask, oracle.make_ocam(cams[0]))[0])
    n8 = len(oracle.Extractor(nfeatures=500, fastAgastType=0, fastThreshold=4)(img, mask, oracle.make_ocam(cams[0]))[0])
    assert n16 > 400 and n12 > 20 and n8 > 20
    p = oracle.make_params(fastAgastType=3)                                          # FAST has three types (AGAST's four: tests/test_oracle_agast.py)
    assert not oracle.lib().orc_extractor_create(__import__("ctypes").byref(p))

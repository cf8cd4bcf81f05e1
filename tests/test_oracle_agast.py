"""CPU: the oracle's AGAST — AgastFeatureDetector AGAST_5_8 / AGAST_7_12d / AGAST_7_12s / OAST_9_16, selected by `extractor.useAgast` 1 and
`extractor.fastAgastType` 0 / 1 / 2 / 3 (reference src/mdBRIEFextractorOct.cpp:869-870, 912-914; read at src/cTracking.cpp:129-131).  OpenCV is not vendored in
the reference tree, so oracle/mcs_oracle.cpp restates OpenCV 3.x's agast.cpp / agast_score.cpp ("parity unpinned" against a real OpenCV, like the other image
primitives).  What is checked here:
  * known answers worked out by hand: the ring of every type (radius, order, N of P), the strict comparisons, the border (= the ring's radius, unlike FAST's
    constant 3), the bisection score, and the region non-maximum suppression where it differs from FAST's 3x3 test (two 4-adjacent corners never both survive;
    diagonal neighbours do; of equal responses the LATER corner in raster order takes the region; a region joined first from above and then from the left);
  * the C++ restatement against an independent Python statement of the same published algorithm (run lengths on the doubled ring instead of arc loops, the
    bisection literally, the suppression loop literally), corner for corner on random and structured images;
  * the closed form the device uses for the score (max over arcs of min(v - I) or of min(I - v), minus 1) against the bisection, pixel by pixel;
  * a property the suppression must have whatever its bookkeeping: exactly one survivor per 4-connected region of corners, carrying the region's largest
    response."""
import numpy as np
import pytest

RINGS = {   # makeAgastOffsets: (dx, dy) in ring order; N contiguous of P; border
    0: ([(-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1)], 5, 1),
    1: ([(-3, 0), (-2, 1), (-1, 2), (0, 3), (1, 2), (2, 1), (3, 0), (2, -1), (1, -2), (0, -3), (-1, -2), (-2, -1)], 7, 3),
    2: ([(-2, 0), (-2, 1), (-1, 2), (0, 2), (1, 2), (2, 1), (2, 0), (2, -1), (1, -2), (0, -2), (-1, -2), (-2, -1)], 7, 2),
    3: ([(-3, 0), (-3, 1), (-2, 2), (-1, 3), (0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1)], 9, 3),
}


def longest_run(flags):
    """longest circular run of True"""
    n = len(flags)
    if all(flags):
        return n
    best = cur = 0
    for f in list(flags) + list(flags):
        cur = cur + 1 if f else 0
        best = max(best, cur)
    return min(best, n)


def py_is_corner(im, y, x, atype, b):
    ring, N, _ = RINGS[atype]
    v = int(im[y, x])
    vals = [int(im[y + dy, x + dx]) for dx, dy in ring]
    return longest_run([i > v + b for i in vals]) >= N or longest_run([i < v - b for i in vals]) >= N


def py_score(im, y, x, atype, threshold):
    bmin, bmax = threshold, 255
    b_test = (bmax + bmin) // 2
    while True:
        if py_is_corner(im, y, x, atype, b_test):
            bmin = b_test
        else:
            bmax = b_test
        if bmin == bmax - 1 or bmin == bmax:
            return bmin
        b_test = (bmin + bmax) // 2


def py_agast(img, atype, threshold, mask=None, nms=True):
    """AGAST(img, kps, threshold, true, type) + KeyPointsFilter::runByPixelsMask; [(x, y, response)] in emission order"""
    _, _, B = RINGS[atype]
    h, w = img.shape
    im = img.astype(np.int32)
    kp = [(x, y, py_score(im, y, x, atype, threshold)) for y in range(B, h - B) for x in range(B, w - B) if py_is_corner(im, y, x, atype, threshold)]
    if not nms:
        return kp
    n = len(kp)
    flags = [-1] * n
    last_row = next_last_row = 0
    last_ind = next_last_ind = 0
    for cur in range(n):
        cx, cy, cr = kp[cur]
        if last_row + 1 < cy:
            last_row, last_ind = next_last_row, next_last_ind
        if next_last_row != cy:
            next_last_row, next_last_ind = cy, cur
        if last_row + 1 == cy:
            while kp[last_ind][0] < cx and kp[last_ind][1] == last_row:
                last_ind += 1
            if kp[last_ind][0] == cx and last_ind != cur:
                wv = last_ind
                while flags[wv] != -1:
                    wv = flags[wv]
                if cr < kp[wv][2]:
                    flags[cur] = wv
                else:
                    flags[wv] = cur
        t = cur - 1
        if cur != 0 and kp[t][1] == cy and kp[t][0] + 1 == cx:
            above = flags[cur]
            while flags[t] != -1:
                t = flags[t]
            if above == -1:
                if t != cur:
                    if cr < kp[t][2]:
                        flags[cur] = t
                    else:
                        flags[t] = cur
            elif t != above:
                if kp[above][2] < kp[t][2]:
                    flags[above] = t
                    flags[cur] = t
                else:
                    flags[t] = above
                    flags[cur] = above
    out = []
    for i in range(n):
        if flags[i] != -1:
            continue
        x, y, r = kp[i]
        if mask is not None and mask[int(np.float32(y) + np.float32(0.5)), int(np.float32(x) + np.float32(0.5))] == 0:
            continue
        out.append((x, y, r))
    return out


def orc_agast(oracle, img, atype, threshold, mask=None):
    L = oracle.lib()
    img = np.ascontiguousarray(img)
    kps = np.zeros(img.size + 1, oracle.KP_DTYPE)
    m = None if mask is None else np.ascontiguousarray(mask)
    n = L.orc_agast_type(atype, oracle.ptr(img), img.shape[1], img.shape[0], img.strides[0], None if m is None else oracle.ptr(m), 0 if m is None else m.strides[0],
                         threshold, oracle.ptr(kps), len(kps))
    assert 0 <= n <= len(kps)
    assert all(k["size"] == 7.0 and k["angle"] == -1.0 and k["octave"] == 0 and k["class_id"] == -1 for k in kps[:n])
    return [(int(k["x"]), int(k["y"]), int(k["response"])) for k in kps[:n]]


def orc_corners(oracle, img, atype, threshold):
    """the corners before the suppression, {(x, y): response}"""
    L = oracle.lib()
    img = np.ascontiguousarray(img)
    kps = np.zeros(img.size + 1, oracle.KP_DTYPE)
    n = L.orc_agast_corners(atype, oracle.ptr(img), img.shape[1], img.shape[0], img.strides[0], threshold, oracle.ptr(kps), len(kps))
    assert 0 <= n <= len(kps)
    out = [(int(k["x"]), int(k["y"]), int(k["response"])) for k in kps[:n]]
    assert out == sorted(out, key=lambda c: (c[1], c[0]))                          # raster order
    return {(x, y): r for x, y, r in out}


def ring_image(atype, values, centre=100, size=9, at=None):
    img = np.full((size, size), centre, np.uint8)
    cx, cy = at if at else (size // 2, size // 2)
    for (dx, dy), v in zip(RINGS[atype][0], values):
        img[cy + dy, cx + dx] = v
    return img


@pytest.mark.parametrize("atype", [0, 1, 2, 3])
def test_segment_lengths_strictness_and_score(oracle, atype):
    ring, N, B = RINGS[atype]
    P, t = len(ring), 20
    assert (P, N, B) == ((8, 5, 1), (12, 7, 3), (12, 7, 2), (16, 9, 3))[atype]
    assert len(set(ring)) == P and max(max(abs(dx), abs(dy)) for dx, dy in ring) == B
    for a, b in zip(ring, ring[1:] + ring[:1]):                                   # a closed 8-connected curve: consecutive ring pixels touch
        assert max(abs(a[0] - b[0]), abs(a[1] - b[1])) == 1
    centre = lambda vals, c=100: orc_corners(oracle, ring_image(atype, vals, centre=c), atype, t).get((4, 4))   # the middle pixel before the suppression (its
    # ring pixels are corners of their own in these images, and with the small rings they touch it)
    for start in (0, 3, P - 2):                                                    # N contiguous darker pixels anywhere on the ring (also across its seam): a corner
        vals = [100] * P
        for j in range(N):
            vals[(start + j) % P] = 60
        assert centre(vals) == 39                                                  # darker by 40: still a corner up to b = 39
        short = list(vals)
        short[(start + N - 1) % P] = 100                                           # N - 1 contiguous: none
        assert centre(short) is None
        broken = list(vals)
        broken[(start + N // 2) % P] = 100                                         # N - 1 darker pixels with a gap: none
        assert centre(broken) is None
    assert centre([121] * P) == 20                                                 # brighter by 21 > t: corner, response = t
    assert centre([120] * P) is None                                               # brighter by exactly t: strict comparison, none
    assert centre([80] * P) is None
    assert centre([79] * P) == 20
    assert centre([60, 70] + [60] * (N - 2) + [100] * (P - N)) == 29              # the only N-arc holds the 70: min difference 30 -> response 29
    assert centre([0] * P, 255) == 254                                             # the largest difference there is: the bisection stops at 254
    full = np.ascontiguousarray(ring_image(atype, [0] * P, centre=255))
    assert oracle.lib().orc_agast_score_type(atype, full.ctypes.data + 4 * 9 + 4, 9, t) == 254
    # a single dot is the only corner of its image (any other pixel has at most one differing ring pixel), so the whole detector returns exactly it
    dot = np.full((11, 11), 100, np.uint8)
    dot[5, 5] = 30
    assert orc_agast(oracle, dot, atype, t) == [(5, 5, 69)]


@pytest.mark.parametrize("atype", [0, 1, 2, 3])
def test_border_is_the_ring_radius(oracle, atype):
    ring, N, B = RINGS[atype]
    t = 20
    img = np.full((12, 12), 100, np.uint8)
    img[B, B] = 200                                                                # a bright dot exactly B pixels from both edges: examined, every ring pixel darker
    assert orc_agast(oracle, img, atype, t) == [(B, B, 99)]
    img2 = np.full((12, 12), 100, np.uint8)
    img2[B - 1, B] = 200                                                           # one row closer to the edge: that pixel is never a centre
    assert orc_agast(oracle, img2, atype, t) == []
    img3 = np.full((12, 12), 100, np.uint8)
    img3[12 - 1 - B, 12 - 1 - B] = 200                                             # the last examined pixel
    assert orc_agast(oracle, img3, atype, t) == [(12 - 1 - B, 12 - 1 - B, 99)]
    img4 = np.full((12, 12), 100, np.uint8)
    img4[12 - 1 - B, 12 - B] = 200
    assert orc_agast(oracle, img4, atype, t) == []


def dots(shape, pts, base=100):
    """isolated single-pixel dots: with AGAST_5_8 each is a corner of response |value - base| - 1 and none of its neighbours is (one ring pixel differs)"""
    img = np.full(shape, base, np.uint8)
    for (x, y), v in pts.items():
        img[y, x] = v
    return img


def test_region_suppression_known_answers(oracle):
    t = 20
    # two dots that touch diagonally: every other pixel sees at most 2 ring pixels differing -> exactly two corners, not 4-adjacent: both survive
    # (FAST's 3x3 test would drop the weaker one)
    img = dots((9, 9), {(3, 3): 180, (4, 4): 170})
    assert py_agast(img, 0, t, nms=False) == [(3, 3, 79), (4, 4, 69)]
    assert orc_agast(oracle, img, 0, t) == [(3, 3, 79), (4, 4, 69)]
    # a 2-pixel horizontal bar, bright: each pixel has 7 darker ring pixels in a row (the other bar pixel is the 8th) -> two 4-adjacent corners, one survivor
    bar = dots((9, 9), {(3, 4): 180, (4, 4): 170})
    assert py_agast(bar, 0, t, nms=False) == [(3, 4, 79), (4, 4, 69)]
    assert orc_agast(oracle, bar, 0, t) == [(3, 4, 79)]
    bar2 = dots((9, 9), {(3, 4): 170, (4, 4): 180})
    assert orc_agast(oracle, bar2, 0, t) == [(4, 4, 79)]
    # equal responses: `response < response` is false, so the LATER corner takes the region over (left / right, and above / below)
    tie = dots((9, 9), {(3, 4): 180, (4, 4): 180})
    assert orc_agast(oracle, tie, 0, t) == [(4, 4, 79)]
    tiev = dots((9, 9), {(4, 3): 180, (4, 4): 180})
    assert py_agast(tiev, 0, t, nms=False) == [(4, 3, 79), (4, 4, 79)]
    assert orc_agast(oracle, tiev, 0, t) == [(4, 4, 79)]
    # vertical bar, the upper one stronger: the lower one is linked to it
    assert orc_agast(oracle, dots((9, 9), {(4, 3): 190, (4, 4): 180}), 0, t) == [(4, 3, 89)]


def test_region_joined_from_above_and_from_the_left(oracle):
    """an L of three bright pixels: (4,3) above (4,4), (3,4) left of (4,4).  In raster order: (4,3), (3,4), (4,4); the last one joins the region above first, then
    the one to its left — the 'maximum above' branch of the loop.  Whatever the three responses, exactly the largest survives (ties: the later in raster order
    among the tied, by the rules above)."""
    t = 10
    for va, vl, vc in ((200, 190, 180), (190, 200, 180), (180, 190, 200), (200, 200, 180), (180, 180, 180), (200, 180, 200), (170, 200, 200)):
        img = dots((9, 9), {(4, 3): va, (3, 4): vl, (4, 4): vc}, base=100)
        all_c = py_agast(img, 0, t, nms=False)
        got = orc_agast(oracle, img, 0, t)
        assert got == py_agast(img, 0, t), (va, vl, vc)
        pos = {(x, y): r for x, y, r in all_c}
        assert {(4, 3), (3, 4), (4, 4)} <= set(pos), all_c                        # the three are corners (a 5-arc of darker pixels exists for each)
        region = [(4, 3), (3, 4), (4, 4)]
        assert len([g for g in got if (g[0], g[1]) in region]) == 1
        winner = [g for g in got if (g[0], g[1]) in region][0]
        assert winner[2] == max(pos[p] for p in region)


def components(corners):
    pos = {(x, y): i for i, (x, y, _) in enumerate(corners)}
    seen, comps = set(), []
    for p in pos:
        if p in seen:
            continue
        stack, comp = [p], []
        seen.add(p)
        while stack:
            q = stack.pop()
            comp.append(q)
            for d in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                r = (q[0] + d[0], q[1] + d[1])
                if r in pos and r not in seen:
                    seen.add(r)
                    stack.append(r)
        comps.append(comp)
    return comps


@pytest.mark.parametrize("atype", [0, 1, 2, 3])
def test_restatement_equals_python_statement_and_region_property(oracle, synth, atype):
    rng = np.random.default_rng(60 + atype)
    cams = synth.lafida_cameras()
    scene = synth.synth_image(1, 0, cams[0])[180:222, 300:348]
    blobs = np.full((36, 40), 90, np.uint8)
    for _ in range(50):
        y, x, r = rng.integers(3, 33), rng.integers(3, 37), rng.integers(0, 3)
        blobs[y:y + 1 + r, x:x + 1 + r] = rng.choice([10, 40, 160, 230])
    noise = rng.integers(0, 256, (30, 30)).astype(np.uint8)
    coarse = np.repeat(np.repeat(rng.integers(0, 4, (10, 10)).astype(np.uint8) * 60 + 20, 3, 0), 3, 1)   # plateaus: many equal responses, large regions
    soft = np.clip(rng.normal(120, 12, (32, 32)), 0, 255).astype(np.uint8)
    total = regions_gt1 = 0
    for img, th in ((scene, 20), (scene, 5), (blobs, 20), (blobs, 7), (noise, 30), (noise, 3), (coarse, 20), (coarse, 1), (soft, 6), (soft, 2)):
        mask = (rng.random(img.shape) < 0.8).astype(np.uint8) * 255
        im = img.astype(np.int32)
        for m in (None, mask):
            want = py_agast(img, atype, th, m)
            got = orc_agast(oracle, img, atype, th, m)
            assert got == want, (atype, th, m is not None, len(got), len(want))
            total += len(want)
        # one survivor per 4-connected region, with the region's largest response; the closed-form score equals the bisection
        allc = py_agast(img, atype, th, nms=False)
        kept = set((x, y) for x, y, _ in orc_agast(oracle, img, atype, th))
        resp = {(x, y): r for x, y, r in allc}
        for comp in components(allc):
            k = [p for p in comp if p in kept]
            assert len(k) == 1 and resp[k[0]] == max(resp[p] for p in comp), (atype, th, comp)
            regions_gt1 += len(comp) > 1
        ring, N, _ = RINGS[atype]
        P = len(ring)
        for x, y, r in allc[::3]:
            d = [int(im[y, x]) - int(im[y + dy, x + dx]) for dx, dy in ring]
            A = max(min(d[(k + j) % P] for j in range(N)) for k in range(P))
            Bm = max(min(-d[(k + j) % P] for j in range(N)) for k in range(P))
            assert r == max(A, Bm) - 1 and max(A, Bm) > th
    assert total > 80 and regions_gt1 > 10


def test_extractor_runs_every_agast_type(oracle, synth):
    cams = synth.lafida_cameras()
    img = synth.synth_image(0, 0, cams[0])
    mask = oracle.mirror_mask(oracle.make_ocam(cams[0]))
    counts = []
    for atype, th in ((0, 20), (1, 20), (2, 20), (3, 20)):
        kps = oracle.Extractor(nfeatures=500, fastAgastType=atype, fastThreshold=th, useAgast=1)(img, mask, oracle.make_ocam(cams[0]))[0]
        counts.append(len(kps))
        assert len(kps) > 50, (atype, len(kps))
    n_fast = len(oracle.Extractor(nfeatures=500, fastAgastType=2, fastThreshold=20)(img, mask, oracle.make_ocam(cams[0]))[0])
    assert n_fast > 50
    import ctypes as C
    p = oracle.make_params(fastAgastType=4, useAgast=1)
    assert not oracle.lib().orc_extractor_create(C.byref(p))
    p = oracle.make_params(fastAgastType=3, useAgast=0)                            # FAST has no type 3
    assert not oracle.lib().orc_extractor_create(C.byref(p))

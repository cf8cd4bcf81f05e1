"""CPU known-answer tests of the oracle's grid-window matchers (oracle/mcs_oracle.cpp "next" row): hand-built frames whose
outcome follows from reading src/cORBmatcher.cpp:326-726, 1990-2118 and src/cMultiFrame.cpp:272-353."""
import numpy as np

import oracle_lib as O

W, H = 754, 480


def kps(rows):
    """rows of (x, y, octave, angle)"""
    a = np.zeros(len(rows), O.KP_DTYPE)
    for i, (x, y, o, ang) in enumerate(rows):
        a[i]["x"], a[i]["y"], a[i]["octave"], a[i]["angle"], a[i]["size"] = x, y, o, ang, 31.0
    return a


def desc(bits_set):
    """32-byte descriptor with the first `bits_set` bits set (distance between two such = |a - b|)"""
    d = np.zeros(32, np.uint8)
    for b in range(bits_set):
        d[b // 8] |= 1 << (b % 8)
    return d


def view(rows, bits, cam=None):
    k = kps(rows)
    d = np.stack([desc(b) for b in bits]) if len(bits) else np.zeros((0, 32), np.uint8)
    c = np.zeros(len(rows), np.int32) if cam is None else np.asarray(cam, np.int32)
    return O.frame_view(k, d, None, c, [W], [H])


def test_window_membership_and_ratio():
    # probe at (100,100), window 10: features at dx = 10 (kept, "> r" rejects), 10.5 (rejected), other camera/out of window rejected
    f1, k1 = view([(100, 100, 0, 0)], [0])
    f2, k2 = view([(110, 100, 3, 0), (110.5, 100, 0, 0), (100, 300, 0, 0)], [20, 1, 0])
    n, m21 = O.window_search(f1, [1], f2, 10, 0, -1, 0.8, 32, False)
    assert n == 1 and list(m21) == [0, -1, -1]          # only feature 0 is in the window; second = INT_MAX -> ratio passes; 20 <= TH_HIGH 96
    # two candidates at distances 20 and 24: 20 <= 24*0.8 = 19.2 fails
    f2, k2 = view([(101, 100, 0, 0), (102, 100, 0, 0)], [20, 24])
    n, m21 = O.window_search(f1, [1], f2, 10, 0, -1, 0.8, 32, False)
    assert n == 0 and list(m21) == [-1, -1]
    n, m21 = O.window_search(f1, [1], f2, 10, 0, -1, 0.9, 32, False)   # 20 <= 21.6
    assert n == 1 and list(m21) == [0, -1]
    # TH_HIGH = 96 without masks: distance 97 is rejected, 96 accepted
    for dist, ok in ((96, 1), (97, 0)):
        f2, k2 = view([(101, 100, 0, 0)], [dist])
        n, _ = O.window_search(f1, [1], f2, 10, 0, -1, 0.8, 32, False)
        assert n == ok
    # level bounds on F1 (minScaleLevel > 0 only) and features without a good map point
    f1b, k1b = view([(100, 100, 1, 0), (100, 100, 5, 0), (100, 100, 3, 0)], [0, 0, 0])
    f2, k2 = view([(101, 100, 0, 0), (102, 100, 0, 0), (103, 100, 0, 0)], [1, 2, 3])
    n, m21 = O.window_search(f1b, [1, 1, 0], f2, 10, 2, 4, 0.99, 32, False)
    assert n == 0                                          # level 1 < 2, level 5 > 4, third has no map point
    n, m21 = O.window_search(f1b, [1, 1, 1], f2, 10, 2, 4, 0.99, 32, False)
    assert n == 1 and list(m21) == [2, -1, -1]            # only F1 feature 2 (level 3) searches: best = feature 0 (1 <= 2*0.99)


def test_window_search_skips_taken_features_in_order():
    # two identical probes: the first takes the closest feature, the second must skip it and take the next (ratio vs the third)
    f1, k1 = view([(100, 100, 0, 0), (100, 100, 0, 0)], [0, 0])
    f2, k2 = view([(101, 100, 0, 0), (102, 100, 0, 0), (103, 100, 0, 0)], [5, 10, 40])
    n, m21 = O.window_search(f1, [1, 1], f2, 10, 0, -1, 0.6, 32, False)
    assert n == 2 and list(m21) == [0, 1, -1]              # 5 <= 10*0.6 ; then 10 <= 40*0.6
    # grid visiting order decides ties: column-major cells (x outer), so the feature further LEFT wins an equal distance
    f2, k2 = view([(109, 100, 0, 0), (91, 100, 0, 0)], [7, 7])
    n, m21 = O.window_search(f1, [1, 0], f2, 10, 0, -1, 1.0, 32, False)
    assert n == 1 and list(m21) == [-1, 0]


def test_search_for_initialization_stealing():
    # probes 0 and 1 both look at feature 0 (same level); probe 1 is closer -> steals; probe 2 is not closer -> skipped, takes nothing
    f1, k1 = view([(50, 50, 0, 0), (50, 50, 0, 0), (50, 50, 0, 0)], [10, 4, 6])
    f2, k2 = view([(52, 50, 0, 0), (300, 300, 0, 0)], [0, 0])
    prev = np.array([[52.0, 50.0]] * 3)
    n, m12, p = O.search_for_initialization(f1, f2, prev, 10, 0.9, 32, False)
    assert n == 1 and list(m12) == [-1, 0, -1]
    assert np.array_equal(p, prev)                         # matched probe: prevMatched <- F2 key position (identical here)
    # equal distance does not steal ("vMatchedDistance[i2] <= dist" skips)
    f1, k1 = view([(50, 50, 0, 0), (50, 50, 0, 0)], [4, 4])
    n, m12, _ = O.search_for_initialization(f1, f2, prev[:2], 10, 0.9, 32, False)
    assert n == 1 and list(m12) == [0, -1]
    # same-level requirement and TH_LOW = 64 (no masks): 65 rejected
    f1, k1 = view([(50, 50, 1, 0)], [0])
    n, m12, _ = O.search_for_initialization(f1, f2, prev[:1], 10, 0.9, 32, False)
    assert n == 0
    for dist, ok in ((64, 1), (65, 0)):
        f1, k1 = view([(50, 50, 0, 0)], [dist])
        n, _, _ = O.search_for_initialization(f1, f2, prev[:1], 10, 0.9, 32, False)
        assert n == ok
    # prevMatched is updated to the matched key
    f1, k1 = view([(50, 50, 0, 0)], [1])
    n, m12, p = O.search_for_initialization(f1, f2, np.array([[55.0, 47.0]]), 10, 0.9, 32, False)
    assert n == 1 and np.array_equal(p, [[52.0, 50.0]])


def test_search_by_projection_last_best_only():
    last, kl = view([(10, 10, 2, 0), (10, 10, 2, 0), (10, 10, 2, 0), (10, 10, 2, 0)], [0, 0, 0, 0])
    cur, kc = view([(200, 200, 1, 0), (203, 200, 3, 0), (204, 200, 4, 0), (206, 200, 2, 0)], [9, 5, 1, 30])
    uv = np.array([[200.0, 200.0]] * 4)
    sc = 1.2 ** np.arange(8)
    # radius = th * 1.44; levels 1..3.  probe 0 takes feature 1 (5), probe 1 feature 0 (9), probe 2 feature 3 (30), probe 3 nothing left
    n, mc, asg = O.search_by_projection_last(cur, [0, 0, 0, 0], last, [1, 1, 1, 1], [0, 0, 0, 0], uv, [1, 1, 1, 1], sc, 5.0, 32, False)
    assert n == 3 and list(mc) == [1, 0, -1, 2] and list(asg) == [1, 1, 0, 1]
    # outliers, missing map points and points outside the mirror mask do not search; taken features are skipped
    n, mc, asg = O.search_by_projection_last(cur, [0, 1, 0, 0], last, [1, 0, 1, 1], [1, 0, 0, 0], uv, [1, 1, 0, 1], sc, 5.0, 32, False)
    assert n == 1 and list(mc) == [3, -1, -1, -1]
    # radius too small for feature 3 (dx = 6 > 2*1.44)
    n, mc, _ = O.search_by_projection_last(cur, [1, 1, 0, 0], last, [1, 0, 0, 0], [0] * 4, uv, [1] * 4, sc, 2.0, 32, False)
    assert n == 0


def test_search_by_projection_frames_sets():
    # F1 features 0 and 1 observe the same map point (id 7): only the first occurrence searches; id 9 is already in F2; id 3 is bad
    f1, k1 = view([(0, 0, 0, 0)] * 4, [0, 0, 0, 0])
    f2, k2 = view([(100, 100, 0, 0), (105, 100, 0, 0), (400, 400, 0, 0)], [3, 3, 3])
    uv = np.zeros((4, 1, 2))
    uv[:, 0] = [[101, 100], [104, 100], [100, 100], [100, 100]]
    n, m21 = O.search_by_projection_frames(f1, [7, 7, 9, 3], [0, 0, 0, 1], f2, [-1, -1, 9], uv, np.ones((4, 1), np.uint8), 10, 1.0, 32, False)
    assert n == 1 and list(m21) == [0, -1, -1]             # equal distances: feature 0 visited first (cell order), 3 <= 3*1.0


def test_rotation_histogram_filter():
    # 12 matches with rotation 0 (bin 0) and one with rotation 100 deg (bin cvRound(100/30) = 3): 1 < 0.1*12 -> removed when checkOri
    rows1 = [(20 + 30 * i, 50, 0, 0.0) for i in range(13)]
    rows2 = [(20 + 30 * i, 50, 0, 0.0 if i < 12 else 260.0) for i in range(13)]   # rot = 0 - 260 + 360 = 100
    f1, k1 = view(rows1, [0] * 13)
    f2, k2 = view(rows2, [1] * 13)
    n, m21 = O.window_search(f1, [1] * 13, f2, 5, 0, -1, 0.8, 32, False, checkOri=0)
    assert n == 13
    n, m21 = O.window_search(f1, [1] * 13, f2, 5, 0, -1, 0.8, 32, False, checkOri=1)
    assert n == 12 and m21[12] == -1 and list(m21[:12]) == list(range(12))
    prev = np.array([[r[0], r[1]] for r in rows1], np.float64)
    n, m12, _ = O.search_for_initialization(f1, f2, prev, 5, 0.9, 32, False, checkOri=1)
    assert n == 12 and m12[12] == -1


def test_world_to_cam_identity_pose():
    cam = O_cam()
    pts = np.array([[0.3, -0.2, 1.0], [0.0, 0.0, 2.0], [-1.0, 0.5, -0.5], [5.0, 0.1, 0.2]])
    uv, fl = O.world_to_cam(np.eye(4)[None], [cam], None, pts, np.zeros(4, np.int32))
    oc = O.make_ocam(cam)
    import ctypes as C
    for i, p in enumerate(pts):
        u, v = C.c_double(), C.c_double()
        O.lib().orc_world2img(C.byref(oc), C.c_double(p[0]), C.c_double(p[1]), C.c_double(p[2]), C.byref(u), C.byref(v))
        assert uv[i, 0] == u.value and uv[i, 1] == v.value
        inside = 0 < round(u.value) < cam["width"] and 0 < round(v.value) < cam["height"]
        assert (fl[i] & 1) == int(inside) and ((fl[i] >> 1) & 1) == int(p[2] <= 0)
    mask = np.zeros((cam["height"], cam["width"]), np.uint8)
    _, fl2 = O.world_to_cam(np.eye(4)[None], [cam], [mask], pts, np.zeros(4, np.int32))
    assert not (fl2 & 1).any()
    # translation: MtMc_inv * (p, 1)
    M = np.eye(4)
    M[:3, 3] = [0.5, 0.0, 0.0]
    uv3, _ = O.world_to_cam(M[None], [cam], None, pts - [0.5, 0, 0], np.zeros(4, np.int32))
    assert np.allclose(uv3, uv, atol=1e-9)


def O_cam():
    import importlib
    synth = importlib.import_module("multicol-slam_amd.synth")
    return synth.lafida_cameras()[0]


def test_distinctive_descriptor_kat():
    # distances between prefix-bit descriptors are |a - b|.  rows (bits): 0, 10, 12, 30
    #   row 0: {10, 12, 30} -> sorted[3/2 = 1] = 12 ; row 1: {2, 20} -> sorted[1] = 20 ; row 2: {18} -> 18 ; row 3 is never a candidate
    d = np.stack([desc(b) for b in (0, 10, 12, 30)])
    assert O.distinctive_descriptor(d, None) == 0
    # rows 0, 40, 41, 42: row 0 -> {40,41,42}[1] = 41 ; row 1 -> {1,2}[1] = 2 ; row 2 -> {1} = 1  => row 2
    d = np.stack([desc(b) for b in (0, 40, 41, 42)])
    assert O.distinctive_descriptor(d, None) == 2
    # ties keep the first row (strict '<'); N <= 2 -> 0; empty -> -1
    d = np.stack([desc(5)] * 6)
    assert O.distinctive_descriptor(d, None) == 0
    assert O.distinctive_descriptor(d[:2], None) == 0 and O.distinctive_descriptor(d[:1], None) == 0 and O.distinctive_descriptor(d[:0], None) == -1
    # masked distance = (popcnt(x & ma) + popcnt(x & mb)) / 2: a zero mask on one row halves its distances
    d = np.stack([desc(b) for b in (0, 40, 41, 42)])
    m = np.full((4, 32), 255, np.uint8)
    m[0] = 0
    # row 0 -> {20, 20, 21}[1] = 20 ; row 1 -> {1,2}[1] = 2 ; row 2 -> {1}  => still row 2
    assert O.distinctive_descriptor(d, m) == 2

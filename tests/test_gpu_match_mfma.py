"""-m gpu: the matrix-core form of the top-K matcher (csrc/mcs_match_mfma.hip: masked Hamming totals as FP4 dot products, taken whenever no count_le output
and no camera groups are asked for) against numpy brute force.  Bit-exact: distances and indices, ties by the lower train index."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_common
    return gpu_common


def _popcnt(a):
    return np.unpackbits(a, axis=-1).sum(-1).astype(np.int32)


def _brute(qd, td, qm=None, tm=None):
    out = np.empty((len(qd), len(td)), np.int32)
    for s in range(0, len(qd), 64):
        x = qd[s:s + 64, None, :] ^ td[None, :, :]
        out[s:s + 64] = _popcnt(x) if qm is None else (_popcnt(x & qm[s:s + 64, None, :]) + _popcnt(x & tm[None, :, :])) // 2
    return out


def _topk_ref(D, K, elig=None):
    nq, nt = D.shape
    big = np.int64(1) << 40
    key = D.astype(np.int64) * (1 << 20) + np.arange(nt)[None, :]
    if elig is not None:
        key = np.where(elig, key, big)
    if nt < K:
        key = np.concatenate([key, np.full((nq, K - nt), big)], 1)
    order = np.argsort(key, axis=1, kind="stable")[:, :K]
    kk = np.take_along_axis(key, order, 1)
    return np.where(kk >= big, 0x7FFFFFFF, kk >> 20).astype(np.int32), np.where(kk >= big, -1, order).astype(np.int32)


def _check(G, qd, td, K, qm=None, tm=None, qv=None, tv=None):
    dist, idx, cnt = G.ctx().match_topk(qd, td, K, -1, qm=qm, tm=tm, qvalid=qv, tvalid=tv)
    D = _brute(qd, td, qm, tm)
    elig = None
    if qv is not None or tv is not None:
        elig = np.ones(D.shape, bool)
        if qv is not None:
            elig &= qv[:, None] != 0
        if tv is not None:
            elig &= tv[None, :] != 0
    ed, ei = _topk_ref(D, K, elig)
    assert G.first_diff(dist, ed) is None and G.first_diff(idx, ei) is None


@pytest.mark.parametrize("dim,K,masked,nq,nt", [(32, 32, True, 777, 3001), (32, 8, False, 300, 1000), (16, 2, True, 64, 257), (16, 16, False, 130, 129),
                                                (32, 32, True, 100, 20), (32, 1, False, 5, 4000), (32, 4, True, 129, 64), (32, 16, True, 1, 1)])
def test_random_rows(G, dim, K, masked, nq, nt):
    rng = np.random.default_rng(dim + K + nq)
    td = rng.integers(0, 256, (nt, dim)).astype(np.uint8)
    qd = td[rng.integers(0, nt, nq)] ^ (rng.integers(0, 256, (nq, dim)) & rng.integers(0, 256, (nq, dim)) & rng.integers(0, 256, (nq, dim))).astype(np.uint8)
    qm = rng.integers(0, 256, (nq, dim)).astype(np.uint8) if masked else None
    tm = rng.integers(0, 256, (nt, dim)).astype(np.uint8) if masked else None
    _check(G, qd, td, K, qm, tm)


def test_many_near_duplicates_and_ties(G):
    """a handful of distinct rows repeated hundreds of times: every list is full of ties (the order is by train index) and the candidate columns
    overflow into merges all the time"""
    rng = np.random.default_rng(3)
    proto = rng.integers(0, 256, (7, 32)).astype(np.uint8)
    pm = rng.integers(0, 256, (7, 32)).astype(np.uint8)
    ti = rng.integers(0, 7, 2500)
    qi = rng.integers(0, 7, 400)
    flip = (rng.random((2500, 32)) < 0.02) * rng.integers(0, 256, (2500, 32))
    td, tm = proto[ti] ^ flip.astype(np.uint8), pm[ti]
    qd, qm = proto[qi], pm[qi]
    for K in (32, 8):
        _check(G, qd, td, K, qm, tm)
        _check(G, qd, td, K)


def test_extreme_totals(G):
    """all-zero / all-one descriptors and masks: totals 0 and 2 * 8 * dim (the largest the word can carry) must neither wrap nor collide with the padding rows"""
    for dim in (16, 32):
        z, o = np.zeros((1, dim), np.uint8), np.full((1, dim), 255, np.uint8)
        td = np.concatenate([z, o, z, o, o])
        tm = np.concatenate([o, o, z, z, o])
        for qd, qm in ((z, o), (o, o), (o, z), (z, z)):
            _check(G, np.repeat(qd, 3, 0), td, 4, np.repeat(qm, 3, 0), tm)
            _check(G, np.repeat(qd, 3, 0), td, 4)


def test_valid_flags_and_ragged_sizes(G):
    rng = np.random.default_rng(11)
    for nq, nt in ((127, 191), (128, 192), (129, 193), (400, 1500)):
        td = rng.integers(0, 256, (nt, 32)).astype(np.uint8)
        tm = rng.integers(0, 256, (nt, 32)).astype(np.uint8)
        qd = rng.integers(0, 256, (nq, 32)).astype(np.uint8)
        qm = rng.integers(0, 256, (nq, 32)).astype(np.uint8)
        qv = (rng.random(nq) < 0.7).astype(np.uint8)
        tv = (rng.random(nt) < 0.5).astype(np.uint8)
        _check(G, qd, td, 8, qm, tm, qv, tv)
    tv0 = np.zeros(1500, np.uint8)   # no eligible train row at all
    _check(G, qd, td, 8, qm, tm, None, tv0)


def test_train_set_size_at_the_index_range_boundary(G):
    """the matrix-core form carries the train index in 14 fraction bits of a float: 16 384 rows are its largest set (index 16 383 must survive the round trip
    through the float key, also as a tie-breaker), 16 385 rows go to the v_bcnt kernel — same lists either way"""
    rng = np.random.default_rng(23)
    for nt in (16384, 16385):
        td = rng.integers(0, 256, (nt, 32)).astype(np.uint8)
        tm = rng.integers(0, 256, (nt, 32)).astype(np.uint8)
        td[-3:] = td[0]; tm[-3:] = tm[0]                      # the last rows tie with row 0: order by index decides
        qd = np.concatenate([td[[0, nt - 1, nt // 2]], rng.integers(0, 256, (67, 32)).astype(np.uint8)])
        qm = np.concatenate([tm[[0, nt - 1, nt // 2]], rng.integers(0, 256, (67, 32)).astype(np.uint8)])
        _check(G, qd, td, 32, qm, tm)
        _check(G, qd, td, 4)


def test_eligible_row_counts_around_the_stage_size(G):
    """the train sets' pass compacts eligible rows into 64-row stages: 0, 1, 63, 64, 65, 128 eligible rows among 300, and all of 256 / 257 (workgroup boundary of the pass)"""
    rng = np.random.default_rng(29)
    td = rng.integers(0, 256, (300, 32)).astype(np.uint8)
    tm = rng.integers(0, 256, (300, 32)).astype(np.uint8)
    qd = td[rng.integers(0, 300, 90)] ^ (rng.integers(0, 256, (90, 32)) & rng.integers(0, 256, (90, 32)) & 3).astype(np.uint8)
    qm = rng.integers(0, 256, (90, 32)).astype(np.uint8)
    for n in (0, 1, 63, 64, 65, 128):
        tv = np.zeros(300, np.uint8)
        tv[rng.permutation(300)[:n]] = 1
        _check(G, qd, td, 8, qm, tm, None, tv)
    for nt in (256, 257):
        _check(G, qd, td[:nt], 8, qm, tm[:nt])
        _check(G, qd, td[:nt], 8, qm, tm[:nt], None, np.ones(nt, np.uint8))

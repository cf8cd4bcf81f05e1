"""-m gpu: the exchange layout of the camera-sharded rig on one GPU (world 1: the slab is the whole [camera][frame] array; the collective itself is covered
by tests/test_rig_gloo.py).  bench.py's Job runs exactly these calls:
  * mcs_extract_batch_strided + mcs_rig_pack_headers write descriptor | mask rows and the per-image counts into (cap + 1)-row blocks; mcs_rig_rows_valid
    rebuilds the row flags from the headers — compared with the plain extraction of the same images;
  * a multi-frame is a BLOCK-STRUCTURED descriptor set inside that array (mcs_desc_set.block_rows / block_pitch_rows), consumed in place by
    mcs_search_kf_kf_ring (every multi-frame against its predecessor) and mcs_search_kf_f_sweep (every multi-frame against every stored keyframe) —
    each pair compared with its own oracle call, match indices bit for bit;
  * host-memory calls with block-structured sets (the staging has to cover the whole span)."""
import ctypes as C
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NCAM, FT, NFEAT = 3, 4, 300


@pytest.fixture(scope="module")
def G():
    import gpu_common
    return gpu_common


@pytest.fixture(scope="module")
def scene(G):
    rig = importlib.import_module("multicol-slam_amd.rig")
    cap_mod = importlib.import_module("multicol-slam_amd._capi")
    cams = G.cams3()
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=NCAM * FT, nfeatures=NFEAT, do_dBrief=1, learnMasks=1)
    lay = rig.RigLayout(NCAM, FT, 1, ex.cap, 32)
    slab = lay.slab(0)
    imgs = [G.synth.synth_image(f, c, cams[c]) for c, f in slab]
    masks = [G.synth.mirror_mask(cams[c]) for c, _ in slab]
    oc = [G.mcs.make_ocam(cams[c]) for c, _ in slab]
    plain = ex.extract_host(imgs, masks, oc)
    # the same images through the strided entry point, device memory
    d_img, d_msk = G.DevBuf(np.stack(imgs)), G.DevBuf(np.stack(masks))
    Gbuf = G.DevBuf(np.zeros(lay.images_total * lay.block_bytes, np.uint8))
    nkp = G.DevBuf(np.zeros(lay.L, np.int32)); kps = G.DevBuf(np.zeros((lay.L * ex.cap, 7), np.float32)); rays = G.DevBuf(np.zeros((lay.L * ex.cap, 3)))
    valid = G.DevBuf(np.zeros(lay.images_total * lay.rows_img, np.uint8)); nall = G.DevBuf(np.zeros(lay.images_total, np.int32))
    g = Gbuf.ptr.value
    ex.extract_strided(lay.L, d_img.ptr.value, 754 * 480, 754, d_msk.ptr.value, 754 * 480, 754, oc, nkp.ptr.value, kps.ptr.value, g, g + 32, rays.ptr.value,
                       lay.rows_img, lay.row_stride)
    lib, ctx = G.mcs.lib(), G.ctx()
    cap_mod.check(lib.mcs_rig_pack_headers(ctx.h, nkp.ptr, lay.L, ex.cap, Gbuf.ptr, lay.row_stride))
    cap_mod.check(lib.mcs_rig_rows_valid(ctx.h, Gbuf.ptr, lay.images_total, ex.cap, lay.row_stride, valid.ptr, nall.ptr))
    ctx.synchronize()
    ex.status()
    return dict(rig=rig, cap=cap_mod, lay=lay, slab=slab, plain=plain, Gbuf=Gbuf, valid=valid, nall=nall, kps=kps, rays=rays, ex=ex)


def test_strided_extraction_fills_the_exchange_blocks(G, scene):
    lay, plain = scene["lay"], scene["plain"]
    Garr = scene["Gbuf"].read().reshape(lay.images_total, lay.rows_img, lay.row_stride)
    val = scene["valid"].read().reshape(lay.images_total, lay.rows_img)
    nall = scene["nall"].read()
    kps = scene["kps"].read().view(np.uint8).reshape(lay.L, lay.cap, 28)
    for x, (kp, d, m, r) in enumerate(plain):
        n = len(kp)
        assert n > 200 and nall[x] == n and int(Garr[x, lay.cap, :4].view("<i4")[0]) == n
        assert np.array_equal(Garr[x, :n, :32], d) and np.array_equal(Garr[x, :n, 32:], m)
        assert val[x, :n].all() and not val[x, n:].any()
        assert np.array_equal(kps[x, :n].reshape(-1), np.ascontiguousarray(kp).view(np.uint8).reshape(-1))
    # the numpy restatement used by the gloo test builds the same blocks
    packed = scene["rig"].pack_blocks(lay, np.stack([np.pad(p[1], ((0, lay.cap - len(p[1])), (0, 0))) for p in plain]),
                                      np.stack([np.pad(p[2], ((0, lay.cap - len(p[2])), (0, 0))) for p in plain]), np.array([len(p[0]) for p in plain]))
    for x in range(lay.images_total):
        n = int(nall[x])
        assert np.array_equal(packed[x, :n], Garr[x, :n]) and np.array_equal(packed[x, lay.cap, :4], Garr[x, lay.cap, :4])


def _frame_set(scene, frame=0):
    lay, cap = scene["lay"], scene["cap"]
    doff, moff, voff, n, stride, brows, bpitch, _ = lay.frame_desc_set(frame)
    g, v = scene["Gbuf"].ptr.value, scene["valid"].ptr.value
    return cap.DescSet(g + doff, g + moff, v + voff, None, n, stride, brows, bpitch)


@pytest.mark.parametrize("K", [32, 4])
def test_ring_search_on_block_structured_frames(G, scene, K):
    lay, cap, rig = scene["lay"], scene["cap"], scene["rig"]
    lib, ctx = G.mcs.lib(), G.ctx()
    Garr = scene["Gbuf"].read().reshape(lay.images_total, lay.rows_img, lay.row_stride)
    fr = _frame_set(scene)
    for first, count in ((0, FT), (1, 2), (FT - 1, 1)):
        m12 = G.DevBuf(np.full((count, lay.rows_frame), -7, np.int32)); nm = G.DevBuf(np.full(count, -7, np.int32)); fb = G.DevBuf(np.zeros(count, np.int32))
        cap.check(lib.mcs_search_kf_kf_ring(ctx.h, FT, first, count, C.byref(fr), lay.rows_img, 32, 0.9, K, cap.MEM_DEVICE, m12.ptr, nm.ptr, fb.ptr))
        ctx.synchronize()
        got_m, got_n = m12.read(), nm.read()
        for s in range(count):
            f = first + s
            d1, m1, v1 = rig.unpack_frame(lay, Garr, f)
            d0, m0, v0 = rig.unpack_frame(lay, Garr, (f - 1) % FT)
            n, exp = G.O.search_kf_kf(d1, m1, v1, d0, m0, v0, True, 0.9)
            assert got_n[s] == n and G.first_diff(got_m[s], exp) is None, (K, first, s)
            assert n > 300


@pytest.mark.parametrize("K", [32, 1])
def test_database_sweep_on_block_structured_frames(G, scene, K):
    lay, cap, rig = scene["lay"], scene["cap"], scene["rig"]
    lib, ctx = G.mcs.lib(), G.ctx()
    Garr = scene["Gbuf"].read().reshape(lay.images_total, lay.rows_img, lay.row_stride)
    nkf = 3
    # stored keyframes: contiguous sets of ncam*cap interleaved descriptor|mask rows (as bench.py's database), here frames 2, 0, 1 with thinned-out flags
    rng = np.random.default_rng(0)
    db = np.zeros((nkf, lay.rows_frame, lay.row_stride), np.uint8); dbv = np.zeros((nkf, lay.rows_frame), np.uint8)
    for j, f in enumerate((2, 0, 1)):
        d, m, v = rig.unpack_frame(lay, Garr, f)
        db[j, :, :32], db[j, :, 32:], dbv[j] = d, m, v & (rng.random(lay.rows_frame) < 0.8)
    ddb, dv = G.DevBuf(db), G.DevBuf(dbv)
    kf = cap.DescSet(ddb.ptr.value, ddb.ptr.value + 32, dv.ptr.value, None, lay.rows_frame, lay.row_stride)
    fr = _frame_set(scene)
    mF = G.DevBuf(np.full((FT, nkf, lay.rows_frame), -7, np.int32)); nm = G.DevBuf(np.full((FT, nkf), -7, np.int32))
    cap.check(lib.mcs_search_kf_f_sweep(ctx.h, nkf, C.byref(kf), lay.rows_frame, FT, C.byref(fr), lay.rows_img, 32, 0.9, K, cap.MEM_DEVICE, mF.ptr, nm.ptr, None))
    ctx.synchronize()
    got_m, got_n = mF.read(), nm.read()
    for f in range(FT):
        df, mf, vf = rig.unpack_frame(lay, Garr, f)
        keep = np.flatnonzero(vf)
        for j in range(nkf):
            n, m = G.O.search_kf_f(np.ascontiguousarray(db[j, :, :32]), np.ascontiguousarray(db[j, :, 32:]), dbv[j], np.ascontiguousarray(df[keep]),
                                   np.ascontiguousarray(mf[keep]), True, 0.9)
            full = np.full(lay.rows_frame, -1, np.int32)
            full[keep] = m
            assert got_n[f, j] == n and G.first_diff(got_m[f, j], full) is None, (K, f, j)
    assert got_n.sum() > 2000


def test_host_memory_call_with_block_structured_sets(G, scene):
    """the same frame through host arrays: the staging copy has to span all blocks of the set"""
    lay, cap, rig = scene["lay"], scene["cap"], scene["rig"]
    lib, ctx = G.mcs.lib(), G.ctx()
    Garr = scene["Gbuf"].read()
    val = scene["valid"].read()
    rows = Garr.reshape(-1, lay.row_stride)
    P = lambda a, off=0: C.c_void_p(a.ctypes.data + off)
    doff, moff, voff, n, stride, brows, bpitch, _ = lay.frame_desc_set(1)
    q = cap.DescSet(P(Garr, doff), P(Garr, moff), P(val, voff), None, n, stride, brows, bpitch)
    doff0, moff0, voff0, *_ = lay.frame_desc_set(0)
    t = cap.DescSet(P(Garr, doff0), P(Garr, moff0), P(val, voff0), None, n, stride, brows, bpitch)
    m12 = np.full(n, -7, np.int32); nm = np.zeros(1, np.int32)
    cap.check(lib.mcs_search_kf_kf(ctx.h, 1, C.byref(q), 0, C.byref(t), 0, 32, 0.9, 8, cap.MEM_HOST, m12.ctypes.data_as(C.c_void_p), nm.ctypes.data_as(C.c_void_p), None))
    G3 = Garr.reshape(lay.images_total, lay.rows_img, lay.row_stride)
    d1, m1, v1 = rig.unpack_frame(lay, G3, 1)
    d0, m0, v0 = rig.unpack_frame(lay, G3, 0)
    en, exp = G.O.search_kf_kf(d1, m1, v1, d0, m0, v0, True, 0.9)
    assert nm[0] == en and G.first_diff(m12, exp) is None
    # argument checks
    bad = cap.DescSet(P(Garr, doff), P(Garr, moff), P(val, voff), None, n, stride, 7, bpitch)
    assert lib.mcs_search_kf_kf(ctx.h, 1, C.byref(bad), 0, C.byref(t), 0, 32, 0.9, 8, cap.MEM_HOST, m12.ctypes.data_as(C.c_void_p), nm.ctypes.data_as(C.c_void_p), None) == cap.MCS_ERR_INVALID
    assert rows.shape[0] == lay.images_total * lay.rows_img


def test_strided_outputs_are_validated(G, scene):
    """mcs_extract_batch_strided rejects layouts its 8-byte row stores cannot serve: a row stride that is not a multiple of 8, misaligned row pointers,
    descriptor and mask rows that overlap for the given pitch / stride"""
    lay, ex = scene["lay"], scene["ex"]
    cams = G.cams3()
    imgs = [G.synth.synth_image(0, c, cams[c]) for c in range(2)]
    d_img = G.DevBuf(np.stack(imgs))
    oc = [G.mcs.make_ocam(cams[c]) for c in range(2)]
    buf = G.DevBuf(np.zeros(2 * (ex.cap + 1) * 128 + 64, np.uint8))
    nkp = G.DevBuf(np.zeros(2, np.int32)); kps = G.DevBuf(np.zeros((2 * ex.cap, 7), np.float32))
    g = buf.ptr.value

    def call(desc, mask, pitch_rows, stride):
        ex.extract_strided(2, d_img.ptr.value, 754 * 480, 754, None, 0, 0, oc, nkp.ptr.value, kps.ptr.value, desc, mask, None, pitch_rows, stride)
    call(g, g + 32, ex.cap + 1, 64)                       # the exchange layout itself is fine
    G.ctx().synchronize()
    for bad in ((g, g + 32, ex.cap + 1, 68),              # stride not a multiple of 8
                (g + 4, g + 36, ex.cap + 1, 64),          # misaligned rows
                (g, g + 16, ex.cap + 1, 64),              # mask inside the descriptor bytes
                (g, g + 40, ex.cap + 1, 64),              # mask row runs into the next descriptor row
                (g, g + 64 * 5, ex.cap + 1, 64)):         # a separate mask array that starts inside the descriptor array
        with pytest.raises(G.mcs.McsError):
            call(*bad)

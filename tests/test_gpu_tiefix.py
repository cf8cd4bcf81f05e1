"""-m gpu: rounding ties of the exact descriptor arithmetic are ENFORCED (csrc/mcs_tiefix.hip; reference src/mdBRIEFextractorOct.cpp:250-301).

The exact passes call ocml's cos / sin / atan where the reference calls glibc's; a keypoint whose cvRound arguments come within the band of a tie is listed by
the device and recomputed on the host with the host's libm.  A real keypoint within 1e-9 px of a tie turns up about once per 3000 batches, so the test widens
the band instead: with band = 0.5 px EVERY keypoint of the exact arithmetic is listed, i.e. every descriptor that leaves the call was computed by the host
code — and must still be the oracle's, bit for bit (the oracle is the same statements compiled against the same glibc); with band = 2e-4 a part of them (a third in ORB / dBRIEF, two thirds in mdBRIEF).
Device-kind batches are patched in place by mcs_extractor_fix_ties."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_common
    return gpu_common


def _inputs(G, ncam=3, frame=4):
    imgs, masks, cams = G.frame_inputs(frame, ncam)
    return imgs, masks, cams


@pytest.mark.parametrize("mode", ["orb", "dbrief", "mdbrief"])
@pytest.mark.parametrize("band", [0.5, 2e-4])
def test_host_recomputation_equals_the_oracle(G, mode, band):
    kw = dict(orb=dict(), dbrief=dict(do_dBrief=1), mdbrief=dict(do_dBrief=1, learnMasks=1))[mode]
    imgs, masks, cams = _inputs(G)
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=3, nfeatures=300, **kw)
    if mode != "orb":
        ex.set_describe(exact_only=True)   # every keypoint through the exact arithmetic, so every keypoint can be listed
    ex.set_tie_band(band)
    out = ex.extract_host(imgs, masks, [G.mcs.make_ocam(c) for c in cams])
    listed, fixed, b = ex.tie_counts()
    nk = sum(len(o[0]) for o in out)
    assert b == band and listed == fixed and fixed > 0
    if band == 0.5:
        assert fixed == nk            # all of them: the outputs below are the host code's
    else:
        assert 0 < fixed < nk
    for i in range(3):
        _, ok, od, odm, _ = G.oracle_extract(imgs[i], masks[i], cams[i], nfeatures=300, **kw)
        k, d, dm, _ = out[i]
        assert G.first_diff(k, ok) is None and G.first_diff(d, od) is None and G.first_diff(dm, odm) is None, (mode, band, i)
    # the default band lists nothing on these images (closest approach ~1e-7 px), and a negative band switches the list off
    ex.set_tie_band(0.0)
    out2 = ex.extract_host(imgs, masks, [G.mcs.make_ocam(c) for c in cams])
    assert ex.tie_counts()[:2] == (listed, fixed)
    assert ex.tie_counts()[2] == (1e-12 if mode == "orb" else 1e-9)
    for a, b_ in zip(out, out2):
        assert G.first_diff(a[1], b_[1]) is None and G.first_diff(a[2], b_[2]) is None
    ex.set_tie_band(-1.0)
    ex.extract_host(imgs, masks, [G.mcs.make_ocam(c) for c in cams])
    assert ex.tie_counts()[:2] == (listed, fixed)
    assert G.mcs.lib().mcs_extractor_set_tie_band(ex.h, 0.75) == G.mcs._capi.MCS_ERR_INVALID
    ex.close()


def test_fast_pass_keypoints_are_not_listed_and_fallbacks_are(G):
    """default mdBRIEF mode: only the keypoints the guard band hands to the exact pass can be listed"""
    imgs, masks, cams = _inputs(G)
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=3, do_dBrief=1, learnMasks=1)
    ex.set_describe(guard_eps=1e-4)   # about half of the keypoints fall back to the exact pass
    ex.set_tie_band(0.5)
    out = ex.extract_host(imgs, masks, [G.mcs.make_ocam(c) for c in cams])
    n_exact = ex.describe_stats()[0]
    listed, fixed, _ = ex.tie_counts()
    nk = sum(len(o[0]) for o in out)
    assert listed == fixed == n_exact and 0 < n_exact < nk
    for i in range(3):
        _, ok, od, odm, _ = G.oracle_extract(imgs[i], masks[i], cams[i], do_dBrief=1, learnMasks=1)
        assert G.first_diff(out[i][1], od) is None and G.first_diff(out[i][2], odm) is None
    ex.close()


def test_device_batches_are_patched_in_place(G):
    """device-kind call: the rows leave the kernels with the device libm's rounding and mcs_extractor_fix_ties replaces the listed ones.  To SEE the patch, the
    device rows are scribbled over between the batch and the fix: exactly the listed keypoints' rows come back right."""
    import ctypes as C
    imgs, masks, cams = _inputs(G, frame=6)
    ex = G.mcs.Extractor(G.ctx(), 754, 480, max_batch=3, nfeatures=250, do_dBrief=1, learnMasks=1)
    ex.set_describe(exact_only=True)
    ex.set_tie_band(2e-4)
    cap = ex.cap
    d_img, d_msk = G.DevBuf(np.stack(imgs)), G.DevBuf(np.stack(masks))
    d_nkp, d_kps = G.DevBuf(np.zeros(3, np.int32)), G.DevBuf(np.zeros((3, cap), G.mcs._capi.KP_DTYPE))
    d_desc, d_mask = G.DevBuf(np.zeros((3, cap, 32), np.uint8)), G.DevBuf(np.zeros((3, cap, 32), np.uint8))
    ocs = [G.mcs.make_ocam(c) for c in cams]
    ex.extract_device(3, d_img.ptr.value, 754 * 480, 754, d_msk.ptr.value, 754 * 480, 754, ocs, d_nkp.ptr.value, d_kps.ptr.value, d_desc.ptr.value,
                      d_mask.ptr.value, 0)
    G.ctx().synchronize()
    before_d, before_m = d_desc.read(), d_mask.read()
    assert G.hip().hipMemset(d_desc.ptr, 0xA5, before_d.nbytes) == 0 and G.hip().hipMemset(d_mask.ptr, 0x5A, before_m.nbytes) == 0
    n = ex.fix_ties()
    nkp = d_nkp.read()
    assert 0 < n < int(nkp.sum()) and ex.tie_counts()[:2] == (n, n)
    after_d, after_m = d_desc.read(), d_mask.read()
    patched = 0
    for i in range(3):
        _, ok, od, odm, _ = G.oracle_extract(imgs[i], masks[i], cams[i], nfeatures=250, do_dBrief=1, learnMasks=1)
        assert len(ok) == nkp[i]
        assert G.first_diff(before_d[i, :nkp[i]], od) is None and G.first_diff(before_m[i, :nkp[i]], odm) is None   # (on these images both libms agree anyway)
        for k in range(nkp[i]):
            if (after_d[i, k] != 0xA5).any() or (after_m[i, k] != 0x5A).any():
                patched += 1
                assert (after_d[i, k] == od[k]).all() and (after_m[i, k] == odm[k]).all()
    assert patched == n
    assert ex.fix_ties() == 0   # the list is consumed
    ex.close()

"""multicol-slam_amd — MI355X-native feature front end + brute-force Hamming matcher of MultiCol-SLAM.

Python host side over the C ABI of libmcs_hip.so (hand-written HIP for gfx950).  The class names mirror the
reference's C++ surface (mdBRIEFextractorOct / ORBextractor, cMultiFrame's extraction part, cORBmatcher); see
frontend.py.  Import with importlib.import_module("multicol-slam_amd") (the directory name carries a hyphen).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (KP_DTYPE, MASKS_RESIDENT, MEM_DEVICE, MEM_HOST, DescSet, ExtractorParams, McsError, Ocam, check, lib, make_ocam, np_ptr)

__all__ = ["Context", "Extractor", "McsError", "KP_DTYPE", "make_ocam", "ExtractorParams", "MEM_HOST", "MEM_DEVICE"]


class Context:
    """One per (process, GPU).  stream: raw hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream) or None."""

    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        check(lib().mcs_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self.h)))
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            lib().mcs_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(lib().mcs_ctx_synchronize(self.h))

    def enable_timing(self, on=True):
        check(lib().mcs_ctx_enable_timing(self.h, int(on)))

    def kernel_ms(self, name):
        ms = C.c_float()
        check(lib().mcs_ctx_kernel_ms(self.h, name.encode(), C.byref(ms)))
        return ms.value

    # ---- matcher primitives (host numpy arrays)
    def match_topk(self, qd, td, K, count_thresh, qm=None, tm=None, qvalid=None, tvalid=None, qgroup=None, tgroup=None):
        qd = np.ascontiguousarray(qd, np.uint8)
        td = np.ascontiguousarray(td, np.uint8)
        nq, dim = qd.shape
        q = DescSet(np_ptr(qd), np_ptr(qm), np_ptr(qvalid), np_ptr(qgroup), nq, dim)
        t = DescSet(np_ptr(td), np_ptr(tm), np_ptr(tvalid), np_ptr(tgroup), td.shape[0], dim)
        dist = np.zeros((nq, K), np.int32)
        idx = np.zeros((nq, K), np.int32)
        cnt = np.zeros(nq, np.int32)
        check(lib().mcs_match_topk(self.h, C.byref(q), C.byref(t), dim, K, count_thresh, MEM_HOST, np_ptr(dist), np_ptr(idx), np_ptr(cnt)))
        return dist, idx, cnt

    def descriptor_distance(self, a, b, ma=None, mb=None):
        out = C.c_int32()
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        if ma is None:
            check(lib().mcs_descriptor_distance(self.h, np_ptr(a), np_ptr(b), a.size, C.byref(out)))
        else:
            ma = np.ascontiguousarray(ma, np.uint8)
            mb = np.ascontiguousarray(mb, np.uint8)
            check(lib().mcs_descriptor_distance_masked(self.h, np_ptr(a), np_ptr(b), np_ptr(ma), np_ptr(mb), a.size, C.byref(out)))
        return out.value


class Extractor:
    """Batched device extractor (C-ABI mcs_extractor): one image size, up to max_batch images per call."""

    def __init__(self, ctx, width, height, max_batch=1, nfeatures=1000, scaleFactor=1.2, nlevels=8, fastThreshold=20,
                 do_dBrief=0, learnMasks=0, descSize=32, edgeThreshold=25, firstLevel=0, scoreType=0, patchSize=32, useAgast=0,
                 fastAgastType=2):
        self.ctx = ctx
        self.params = ExtractorParams(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, scoreType, patchSize,
                                      fastThreshold, int(useAgast), fastAgastType, int(do_dBrief), int(learnMasks), descSize)
        self.h = C.c_void_p()
        check(lib().mcs_extractor_create(ctx.h, C.byref(self.params), width, height, max_batch, C.byref(self.h)))
        cap = C.c_int32()
        check(lib().mcs_extractor_kp_capacity(self.h, C.byref(cap)))
        self.cap = cap.value
        self.width, self.height, self.max_batch, self.descSize, self.nlevels = width, height, max_batch, descSize, nlevels
        nl = C.c_int32()
        w = (C.c_int32 * 16)()
        h = (C.c_int32 * 16)()
        f = (C.c_int32 * 16)()
        check(lib().mcs_extractor_levels(self.h, C.byref(nl), w, h, f))
        self.level_sizes = [(w[i], h[i]) for i in range(nl.value)]
        self.features_per_level = [f[i] for i in range(nl.value)]

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):   # mcs_ctx_destroy releases the extractors still alive on it: after that this handle is already gone
                for p in getattr(self, "_pins", {}).values():
                    lib().mcs_host_free(self.ctx.h, p[0])
                lib().mcs_extractor_destroy(self.h)
            self._pins = {}
            self.h = None

    def _pinned(self, name, dtype, shape):
        """a page-locked array owned by this extractor (allocated once per name and size, mcs_host_alloc): host-kind calls then upload with one DMA and get
        their outputs written by one launch instead of one runtime copy per array"""
        pins = self.__dict__.setdefault("_pins", {})
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ent = pins.get(name)
        if ent is None or ent[1] < nbytes:
            if ent is not None:
                check(lib().mcs_host_free(self.ctx.h, ent[0]))
            p = C.c_void_p()
            check(lib().mcs_host_alloc(self.ctx.h, max(nbytes, 64), C.byref(p)))
            ent = pins[name] = (p, max(nbytes, 64))
        buf = (C.c_uint8 * nbytes).from_address(ent[0].value)
        return ent[0], np.frombuffer(buf, np.uint8).view(dtype).reshape(shape)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_masks(self, masks):
        """keep the (mirror) masks on the device: afterwards masks="resident" in extract_host means mask i of this set for image i (mcs_extractor_set_masks)"""
        m = np.ascontiguousarray(np.stack(masks), np.uint8)
        n, h, w = m.shape
        assert (w, h) == (self.width, self.height)
        check(lib().mcs_extractor_set_masks(self.h, n, np_ptr(m), w * h, w, MEM_HOST))

    def extract_host(self, images, masks, cams, want_rays=True):
        """images: list/array of HxW uint8; masks: same, None, or "resident" (set_masks); cams: list of Ocam or None.  Returns per-image tuples."""
        n = len(images)
        h, w = self.height, self.width
        p_img, imgs = self._pinned("img", np.uint8, (n, h, w))
        for i, im in enumerate(images):
            assert im.shape == (h, w)
            imgs[i] = im
        if isinstance(masks, str):
            assert masks == "resident"
            m = MASKS_RESIDENT
        elif masks is None:
            m = None
        else:
            m, mv = self._pinned("mask", np.uint8, (n, h, w))
            for i, mk in enumerate(masks):
                mv[i] = mk
        camarr = None
        if cams is not None:
            camarr = (Ocam * n)(*cams)
        p_n, nkp = self._pinned("nkp", np.int32, (n,))
        p_k, kps = self._pinned("kps", KP_DTYPE, (n, self.cap))
        p_d, desc = self._pinned("desc", np.uint8, (n, self.cap, self.descSize))
        p_m, dmask = self._pinned("dmask", np.uint8, (n, self.cap, self.descSize))
        p_r, rays = self._pinned("rays", np.float64, (n, self.cap, 3)) if (want_rays and cams is not None) else (None, None)
        check(lib().mcs_extract_batch(self.h, n, p_img, w * h, w, m, w * h, w, camarr, MEM_HOST, p_n, p_k, p_d, p_m, p_r))
        out = []
        for i in range(n):
            k = int(nkp[i])
            out.append((kps[i, :k].copy(), desc[i, :k].copy(), dmask[i, :k].copy(), None if rays is None else rays[i, :k].copy()))
        return out

    def extract_device(self, n, images_ptr, image_pitch, image_stride, masks_ptr, mask_pitch, mask_stride, cams, nkp_ptr, kps_ptr,
                       desc_ptr, dmask_ptr, rays_ptr):
        """Raw device pointers (ints); only enqueues on the context's stream."""
        camarr = None
        if cams is not None:
            camarr = cams if isinstance(cams, C.Array) else (Ocam * n)(*cams)
        check(lib().mcs_extract_batch(self.h, n, C.c_void_p(images_ptr), image_pitch, image_stride,
                                      C.c_void_p(masks_ptr) if masks_ptr else None, mask_pitch, mask_stride, camarr, MEM_DEVICE,
                                      C.c_void_p(nkp_ptr), C.c_void_p(kps_ptr), C.c_void_p(desc_ptr), C.c_void_p(dmask_ptr),
                                      C.c_void_p(rays_ptr) if rays_ptr else None))

    def extract_strided(self, n, images_ptr, image_pitch, image_stride, masks_ptr, mask_pitch, mask_stride, cams, nkp_ptr, kps_ptr, desc_ptr, dmask_ptr,
                        rays_ptr, out_image_pitch_rows, out_row_stride):
        """extract_device with the descriptor / mask rows written at (image * out_image_pitch_rows + k) * out_row_stride (the rig's exchange blocks)"""
        camarr = None
        if cams is not None:
            camarr = cams if isinstance(cams, C.Array) else (Ocam * n)(*cams)
        check(lib().mcs_extract_batch_strided(self.h, n, C.c_void_p(images_ptr), image_pitch, image_stride, C.c_void_p(masks_ptr) if masks_ptr else None,
                                              mask_pitch, mask_stride, camarr, C.c_void_p(nkp_ptr), C.c_void_p(kps_ptr), C.c_void_p(desc_ptr),
                                              C.c_void_p(dmask_ptr), C.c_void_p(rays_ptr) if rays_ptr else None, out_image_pitch_rows, out_row_stride))

    def status(self):
        check(lib().mcs_extractor_status(self.h))

    def set_describe(self, exact_only=False, guard_eps=0.0):
        """dBRIEF / mdBRIEF: exact_only routes every keypoint through the reference's exact arithmetic; guard_eps (px, 0 = default) is the band around
        the cvRound ties inside which the fast pass hands a keypoint to the exact pass.  The outputs are bit-identical in every setting."""
        check(lib().mcs_extractor_set_describe(self.h, int(exact_only), float(guard_eps)))

    def describe_stats(self):
        n, eps = C.c_uint64(), C.c_double()
        check(lib().mcs_extractor_describe_stats(self.h, C.byref(n), C.byref(eps)))
        return n.value, eps.value

    def tie_stats(self, reset=False):
        """smallest distance (pixels) of a cvRound argument of the exact arithmetic to a rounding tie since creation / the last reset (inf: none yet)"""
        v = C.c_double()
        check(lib().mcs_extractor_tie_stats(self.h, C.byref(v), int(reset)))
        return v.value

    def set_tie_band(self, band_px=0.0):
        """band (px) around the cvRound ties inside which an exact-arithmetic keypoint is recomputed on the host with the host's libm; 0 = default, < 0 = off"""
        check(lib().mcs_extractor_set_tie_band(self.h, float(band_px)))

    def fix_ties(self):
        """device-kind batches: recompute the last batch's listed keypoints on the host and patch the device rows; returns how many"""
        n = C.c_int()
        check(lib().mcs_extractor_fix_ties(self.h, C.byref(n)))
        return n.value

    def set_tie_capture(self, depth, max_ties=0):
        """pipelined enforcement: a ring of `depth` page-locked capture slots (one per device-kind batch in flight), `max_ties` entries each (0 = 64); depth 0 = off"""
        check(lib().mcs_extractor_set_tie_capture(self.h, int(depth), int(max_ties)))

    def patch_ties(self, back=1):
        """recompute the listed keypoints of the device-kind batch `back` calls before the latest one on the host and patch its device rows (waits for that batch
        only); returns (listed, recomputed)"""
        a, b_ = C.c_int(), C.c_int()
        check(lib().mcs_extractor_patch_ties(self.h, int(back), C.byref(a), C.byref(b_)))
        return a.value, b_.value

    def tie_counts(self):
        """(keypoints listed, keypoints recomputed on the host, band in px) since creation"""
        a, b_, band = C.c_uint64(), C.c_uint64(), C.c_double()
        check(lib().mcs_extractor_tie_counts(self.h, C.byref(a), C.byref(b_), C.byref(band)))
        return a.value, b_.value, band.value

    # ---- stage taps (parity tests)
    def tap_level(self, img, level, blurred=False):
        w, h = self.level_sizes[level]
        out = np.zeros((h, w), np.uint8)
        check(lib().mcs_extractor_tap_level(self.h, img, level, int(blurred), np_ptr(out)))
        return out

    def _tap(self, fn, img, level):
        buf = np.zeros(1 << 17, np.uint32)
        n = C.c_int32()
        check(fn(self.h, img, level, np_ptr(buf), len(buf), C.byref(n)))
        rec = buf[:n.value]
        return (rec & 0xFFF).astype(np.int32), ((rec >> 12) & 0xFFF).astype(np.int32), (rec >> 24).astype(np.int32)

    def tap_candidates(self, img, level):
        return self._tap(lib().mcs_extractor_tap_candidates, img, level)

    def tap_selected(self, img, level):
        return self._tap(lib().mcs_extractor_tap_selected, img, level)

"""Host-side mirror of the reference's class surface for the hot path (same names, argument meaning and outputs):

  mdBRIEFextractorOct / ORBextractor   include/mdBRIEFextractorOct.h:335-421, include/cORBextractor.h:61-63
  cMultiFrame (extraction part)        include/cMultiFrame.h:62-162, src/cMultiFrame.cpp:92-216,342-353
  cORBmatcher (brute-force searches)   include/cORBmatcher.h:43-133, src/cORBmatcher.cpp:46-65,179-323,885-1155
  cORBmatcher (grid-window searches)   src/cORBmatcher.cpp:67-166,326-726,1990-2118
  cMultiCamSys_ (pose, projection)     src/cam_system_omni.cpp:92-133,168-198
  DescriptorDistance64[_Masked]        src/cORBmatcher.cpp:2438-2474

Everything numeric runs in libmcs_hip.so on the GPU; this file only shapes inputs/outputs (numpy stands in for cv::Mat).
"""
import ctypes as C

import numpy as np

from . import Context, Extractor
from ._capi import KP_DTYPE, MEM_HOST, DescSet, check, lib, make_ocam, np_ptr

FRAME_GRID_ROWS, FRAME_GRID_COLS = 48, 64   # include/cMultiFrame.h

_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class cCamModelGeneral_:
    """Scaramuzza omni camera (include/cam_model_omni.h); holds the calibration and the level-0 mirror mask."""

    def __init__(self, cdeu0v0, p, invP, Iw, Ih, mirror_mask=None):
        c, d, e, u0, v0 = cdeu0v0
        self.calib = dict(c=c, d=d, e=e, u0=u0, v0=v0, p=list(p), invP=list(invP), width=int(Iw), height=int(Ih))
        self.ocam = make_ocam(self.calib)
        self._mask = mirror_mask

    @classmethod
    def from_dict(cls, cam, mirror_mask=None):
        return cls((cam["c"], cam["d"], cam["e"], cam["u0"], cam["v0"]), cam["p"], cam["invP"], cam["width"], cam["height"], mirror_mask)

    def GetWidth(self):
        return self.calib["width"]

    def GetHeight(self):
        return self.calib["height"]

    def GetMirrorMask(self, level=0):
        assert level == 0, "only level 0 is used by the extractor (src/cMultiFrame.cpp:138)"
        return self._mask


def _matx_mul(A, B):
    """cv::Matx product (s = 0; s += a(i,k) * b(k,j) in k order), so MtMc is rounded like the reference's."""
    A, B = np.asarray(A, np.float64), np.asarray(B, np.float64)
    out = np.zeros((A.shape[0], B.shape[1]))
    for i in range(A.shape[0]):
        for j in range(B.shape[1]):
            acc = np.float64(0.0)
            for k in range(A.shape[1]):
                acc = acc + A[i, k] * B[k, j]
            out[i, j] = acc
    return out


def _inv_mat(M):
    """cConverter::invMat (src/cConverter.cpp:31-44): rigid inverse, t = -(R^T) * t."""
    M = np.asarray(M, np.float64)
    Rt = M[:3, :3].T.copy()
    t = _matx_mul(-Rt, M[:3, 3:4])[:, 0]
    out = np.eye(4)
    out[:3, :3] = Rt
    out[:3, 3] = t
    return out


class cMultiCamSys_:
    """Calibrations + poses of the rig (include/cam_system_omni.h): M_t = rig pose, M_c[c] = camera c in the rig frame."""

    def __init__(self, cam_models, M_c=None, M_t=None):
        self.cams = list(cam_models)
        self.M_c = [np.asarray(m, np.float64) for m in M_c] if M_c is not None else [np.eye(4) for _ in self.cams]
        self.Set_M_t(np.eye(4) if M_t is None else M_t)

    def Set_M_t(self, M_t_):   # src/cam_system_omni.cpp:184-198
        self.M_t = np.asarray(M_t_, np.float64).copy()
        self.M_t_inv = _inv_mat(self.M_t)
        self.MtMc = [_matx_mul(self.M_t, mc) for mc in self.M_c]
        self.MtMc_inv = [_inv_mat(m) for m in self.MtMc]

    def world_to_cam(self, pts3, cam_idx, ctx=None):
        """Batched WorldToCamHom_fast + isPointInMirrorMask(u, v, 0) on the GPU -> (uv [n,2] float64, flags [n] uint8; bit0 = in mask,
        bit1 = behind the camera)."""
        pts3 = np.ascontiguousarray(pts3, np.float64).reshape(-1, 3)
        cam_idx = np.ascontiguousarray(cam_idx, np.int32)
        n = len(pts3)
        uv, flags = np.zeros((max(n, 1), 2)), np.zeros(max(n, 1), np.uint8)
        if n == 0:
            return uv[:0], flags[:0]
        nr = len(self.cams)
        M = np.ascontiguousarray(np.stack(self.MtMc_inv).reshape(nr, 16))
        ocs = (type(self.cams[0].ocam) * nr)(*[cm.ocam for cm in self.cams])
        masks = [cm.GetMirrorMask(0) for cm in self.cams]
        keep = [None if m is None else np.ascontiguousarray(m, np.uint8) for m in masks]
        mp = (C.c_void_p * nr)(*[None if m is None else m.ctypes.data for m in keep])
        ctx = ctx or default_context()
        check(lib().mcs_world_to_cam(ctx.h, np_ptr(M), ocs, nr, mp if any(m is not None for m in keep) else None, np_ptr(pts3), np_ptr(cam_idx), n,
                                     MEM_HOST, np_ptr(uv), np_ptr(flags)))
        return uv[:n], flags[:n]

    def WorldToCamHom_fast(self, c, pt3):   # src/cam_system_omni.cpp:114-133
        uv, _ = self.world_to_cam(np.asarray(pt3, np.float64)[:3].reshape(1, 3), [c])
        return uv[0]

    def GetNrCams(self):
        return len(self.cams)

    def GetCamModelObj(self, c):
        return self.cams[c]


class mdBRIEFextractorOct:
    """Same 13 constructor arguments as the reference (include/mdBRIEFextractorOct.h:339-351)."""

    HARRIS_SCORE, FAST_SCORE = 0, 1

    def __init__(self, _nfeatures=1000, _scaleFactor=1.2, _nlevels=8, _edgeThreshold=25, _firstLevel=0, _scoreType=0, _patchSize=32,
                 _fastThreshold=20, _useAgast=False, _fastAgastType=2, _do_dBrief=False, _learnMasks=False, _descSize=32, ctx=None):
        self.kw = dict(nfeatures=_nfeatures, scaleFactor=_scaleFactor, nlevels=_nlevels, edgeThreshold=_edgeThreshold, firstLevel=_firstLevel,
                       scoreType=_scoreType, patchSize=_patchSize, fastThreshold=_fastThreshold, useAgast=int(_useAgast),
                       fastAgastType=_fastAgastType, do_dBrief=int(_do_dBrief), learnMasks=int(_learnMasks), descSize=_descSize)
        self.ctx = ctx or default_context()
        self._ex = {}

    def _extractor(self, w, h, batch):
        key = (w, h)
        ex = self._ex.get(key)
        if ex is None or ex.max_batch < batch:
            if ex is not None:
                ex.close()
            ex = Extractor(self.ctx, w, h, max_batch=max(batch, 1), **self.kw)
            self._ex[key] = ex
        return ex

    def __call__(self, image, mask, camModel):
        """operator()(image, mask, keypoints, camModel, descriptors, descriptorMasks) -> (keypoints, descriptors, descriptorMasks)."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), None, None     # empty image -> silent return (:1252-1253)
        assert image.dtype == np.uint8 and image.ndim == 2   # assert(image.type() == CV_8UC1) (:1256)
        h, w = image.shape
        ex = self._extractor(w, h, 1)
        cams = None if camModel is None else [camModel.ocam]
        kps, d, dm, _ = ex.extract_host([image], None if mask is None else [mask], cams, want_rays=False)[0]
        if len(kps) == 0:
            return kps, None, None                        # _descriptors.release() (:1270-1274)
        return kps, d, dm

    def extract_rig(self, images, masks, cam_models):
        """All cameras of a multi-frame in ONE device batch (the GPU analogue of the reference's omp loop over cameras)."""
        h, w = images[0].shape
        ex = self._extractor(w, h, len(images))
        return ex.extract_host(images, masks, [c.ocam for c in cam_models], want_rays=True)

    def GetLevels(self):
        return self.kw["nlevels"]

    def GetScaleFactor(self):
        return float(np.float32(self.kw["scaleFactor"]))   # the member is the double of the FLOAT ctor argument

    def GetMasksLearned(self):
        return bool(self.kw["learnMasks"])

    def GetDescriptorSize(self):
        return self.kw["descSize"]


class ORBextractor(mdBRIEFextractorOct):
    """include/cORBextractor.h (declared but never built in the reference): the extractor in ORB mode."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, scoreType=1, fastTh=20, ctx=None):
        super().__init__(nfeatures, scaleFactor, nlevels, 25, 0, scoreType, 32, fastTh, False, 2, False, False, 32, ctx=ctx)

    def __call__(self, image, mask):
        kps, d, _ = super().__call__(image, mask, None)
        return kps, d


class cMultiFrame:
    """Extraction part of cMultiFrame::cMultiFrame (src/cMultiFrame.cpp:92-216): public fields the matcher/tracker read."""

    nNextId = 0

    def __init__(self, images, timeStamp, extractor, voc, camSystem, imgCnt=0):
        nrCams = camSystem.GetNrCams()
        self.images, self.mTimeStamp, self.camSystem, self.imgCnt = images, timeStamp, camSystem, imgCnt
        self.mpORBvocabulary, self.mBowVec, self.mFeatVec = voc, None, None
        ex0 = extractor[0] if isinstance(extractor, (list, tuple)) else extractor
        cams = [camSystem.GetCamModelObj(c) for c in range(nrCams)]
        masks = [cm.GetMirrorMask(0) for cm in cams]
        res = ex0.extract_rig(list(images), None if any(m is None for m in masks) else masks, cams)
        self.mDescriptors = [r[1] for r in res]
        self.mDescriptorMasks = [r[2] for r in res]
        self.N = [len(r[0]) for r in res]
        self.totalN = int(sum(self.N))
        self.mvKeys = np.concatenate([r[0] for r in res]) if self.totalN else np.zeros(0, KP_DTYPE)
        self.mvKeysRays = np.concatenate([r[3] for r in res]) if self.totalN else np.zeros((0, 3))
        self.keypoint_to_cam = np.concatenate([np.full(n, c, np.int32) for c, n in enumerate(self.N)]) if self.totalN else np.zeros(0, np.int32)
        self.cont_idx_to_local_cam_idx = np.concatenate([np.arange(n, dtype=np.int32) for n in self.N]) if self.totalN else np.zeros(0, np.int32)
        self.mnMinX, self.mnMinY = [0] * nrCams, [0] * nrCams
        self.mnMaxX = [cm.GetWidth() for cm in cams]
        self.mnMaxY = [cm.GetHeight() for cm in cams]
        self.mfGridElementWidthInv = [FRAME_GRID_COLS / float(self.mnMaxX[c] - self.mnMinX[c]) for c in range(nrCams)]
        self.mfGridElementHeightInv = [FRAME_GRID_ROWS / float(self.mnMaxY[c] - self.mnMinY[c]) for c in range(nrCams)]
        self.mGrids = [[[[] for _ in range(FRAME_GRID_ROWS)] for _ in range(FRAME_GRID_COLS)] for _ in range(nrCams)]
        for i in range(self.totalN):   # serial flatten + grid fill (:167-184)
            c = int(self.keypoint_to_cam[i])
            ok, gx, gy = self.PosInGrid(c, self.mvKeys[i])
            if ok:
                self.mGrids[c][gx][gy].append(i)
        self.mvbOutlier = [False] * self.totalN
        self.mvpMapPoints = [None] * self.totalN
        self.mnId = cMultiFrame.nNextId
        cMultiFrame.nNextId += 1
        self.mnScaleLevels = ex0.GetLevels()
        self.mfScaleFactor = ex0.GetScaleFactor()
        self.mvScaleFactors, self.mvLevelSigma2 = [1.0], [1.0]
        for i in range(1, self.mnScaleLevels):
            self.mvScaleFactors.append(self.mvScaleFactors[i - 1] * self.mfScaleFactor)
            self.mvLevelSigma2.append(self.mvScaleFactors[i] * self.mvScaleFactors[i])
        self.mvInvLevelSigma2 = [1 / s for s in self.mvLevelSigma2]
        self.masksLearned = ex0.GetMasksLearned()
        self.descDimension = ex0.GetDescriptorSize()

    def PosInGrid(self, cam, kp):   # src/cMultiFrame.cpp:342-353 (cvRound, not floor; bins 64 / 48 are dropped)
        posX = int(np.rint((float(kp["x"]) - self.mnMinX[cam]) * self.mfGridElementWidthInv[cam]))
        posY = int(np.rint((float(kp["y"]) - self.mnMinY[cam]) * self.mfGridElementHeightInv[cam]))
        if posX < 0 or posX >= FRAME_GRID_COLS or posY < 0 or posY >= FRAME_GRID_ROWS:
            return False, posX, posY
        return True, posX, posY

    def ComputeBoW(self):   # src/cMultiFrame.cpp:356-363 (voc = the cORBVocabulary given to the constructor)
        if not getattr(self, "mBowVec", None):
            self.mBowVec, self.mFeatVec = self.mpORBvocabulary.transform(self.all_descriptors(), 4)

    # flat (all cameras concatenated) descriptor views, the row order of mvKeys
    def all_descriptors(self):
        return np.concatenate(self.mDescriptors) if self.totalN else np.zeros((0, self.descDimension), np.uint8)

    def all_masks(self):
        return np.concatenate(self.mDescriptorMasks) if self.totalN else np.zeros((0, self.descDimension), np.uint8)


class cMultiKeyFrame:
    """Thin keyframe view: the accessors the brute-force searches use (src/cMultiKeyFrame.cpp:54,77,356-364)."""

    def __init__(self, F):
        self.camSystem = F.camSystem
        self.mDescriptors, self.mDescriptorMasks = F.mDescriptors, F.mDescriptorMasks   # shallow copies like cv::Mat
        self.mvKeys, self.mvKeysRays = F.mvKeys, F.mvKeysRays
        self.keypoint_to_cam, self.cont_idx_to_local_cam_idx = F.keypoint_to_cam, F.cont_idx_to_local_cam_idx
        self.mvpMapPoints = list(F.mvpMapPoints)
        self._d, self._m = F.all_descriptors(), F.all_masks()
        self.mBowVec, self.mFeatVec = getattr(F, "mBowVec", None), getattr(F, "mFeatVec", None)

    def GetMapPointMatches(self):
        return self.mvpMapPoints

    def GetFeatureVector(self):
        return self.mFeatVec

    def GetKeyPoints(self):
        return self.mvKeys

    def GetKeyPointsRays(self):
        return self.mvKeysRays


def _good(mp):
    return mp is not None and not (hasattr(mp, "isBad") and mp.isBad())


class cORBmatcher:
    """cORBmatcher(nnratio, checkOri, featDim, havingMasks) (src/cORBmatcher.cpp:46-65); brute-force searches only."""

    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True, featDim=32, havingMasks_=False, ctx=None, K=8):
        self.mfNNratio, self.mbCheckOrientation, self.mbFeatDim, self.havingMasks = nnratio, checkOri, featDim, havingMasks_
        if havingMasks_:
            self.TH_HIGH_, self.TH_LOW_ = int(np.floor(1.5 * featDim)), int(np.floor(featDim))
        else:
            self.TH_HIGH_, self.TH_LOW_ = 3 * featDim, 2 * featDim
        self.ctx = ctx or default_context()
        self.K = K
        self.last_fallbacks = 0

    def _rot_filter(self, variant, keys_slot, keys_partner, match, swapped, accepted=None):
        """mbCheckOrientation: rotation-consistency filter (mcs_rotation_consistency) on `match` (slot -> partner index or -1), in place.
        -> number of matches removed (0 when the matcher was built with checkOri = False)."""
        if not self.mbCheckOrientation or len(match) == 0:
            return 0
        ks, kp = np.ascontiguousarray(keys_slot), np.ascontiguousarray(keys_partner)
        if len(kp) == 0:
            return 0
        off = KP_DTYPE.fields["angle"][1]
        acc = None if accepted is None else np.ascontiguousarray(accepted, np.int32)
        rem = np.zeros(1, np.int32)
        check(lib().mcs_rotation_consistency(self.ctx.h, variant, C.c_void_p(ks.ctypes.data + off), KP_DTYPE.itemsize, C.c_void_p(kp.ctypes.data + off),
                                             KP_DTYPE.itemsize, np_ptr(acc), np_ptr(match), len(match), len(kp), int(swapped), MEM_HOST, np_ptr(rem)))
        return int(rem[0])

    def _sets(self, d1, m1, v1, g1, d2, m2, v2, g2):
        keep = [np.ascontiguousarray(d1, np.uint8), np.ascontiguousarray(d2, np.uint8)]
        mm = [None, None]
        if self.havingMasks:
            mm = [np.ascontiguousarray(m1, np.uint8), np.ascontiguousarray(m2, np.uint8)]
        q = DescSet(np_ptr(keep[0]), np_ptr(mm[0]), np_ptr(v1), np_ptr(g1), keep[0].shape[0], self.mbFeatDim)
        t = DescSet(np_ptr(keep[1]), np_ptr(mm[1]), np_ptr(v2), np_ptr(g2), keep[1].shape[0], self.mbFeatDim)
        return q, t, (keep, mm, v1, v2, g1, g2)

    def SearchByBoW(self, pKF1, other):
        """(KF,KF) -> (nmatches, vpMatches12) with vpMatches12[idx1] = map point of KF2 (src/cORBmatcher.cpp:885-966);
        (KF,F)  -> (nmatches, vpMapPointMatches) indexed by frame feature (:179-323, vocabulary restriction removed)."""
        mp1 = pKF1.GetMapPointMatches()
        v1 = np.array([_good(m) for m in mp1], np.uint8)
        fb = np.zeros(1, np.int32)
        nm = np.zeros(1, np.int32)
        if isinstance(other, cMultiKeyFrame):
            mp2 = other.GetMapPointMatches()
            v2 = np.array([_good(m) for m in mp2], np.uint8)
            q, t, keep = self._sets(pKF1._d, pKF1._m, v1, None, other._d, other._m, v2, None)
            m12 = np.full(max(len(mp1), 1), -1, np.int32)
            check(lib().mcs_search_kf_kf(self.ctx.h, 1, C.byref(q), 0, C.byref(t), 0, self.mbFeatDim, self.mfNNratio, self.K, MEM_HOST,
                                         np_ptr(m12), np_ptr(nm), np_ptr(fb)))
            self.last_fallbacks = int(fb[0])
            return int(nm[0]), [mp2[j] if j >= 0 else None for j in m12[:len(mp1)]]
        F = other
        kfv, ffv = getattr(pKF1, "mFeatVec", None), getattr(F, "mFeatVec", None)
        if kfv and ffv:
            # the reference's vocabulary-restricted search: a keyframe feature only meets frame features of the same FeatureVector node;
            # nodes ascending, the features of a node in index order = keyframe rows permuted to (node, index) order, node id as `group`
            order = np.array([i for _, lst in kfv.items() for i in lst], np.int64)
            gk = np.array([nd for nd, lst in kfv.items() for _ in lst], np.int32)
            gf = np.full(F.totalN, -1, np.int32)
            for nd, lst in ffv.items():
                gf[lst] = nd
            vf = (gf >= 0).astype(np.uint8)
            q, t, keep = self._sets(pKF1._d[order], pKF1._m[order], np.ascontiguousarray(v1[order]), gk, F.all_descriptors(), F.all_masks(), vf, gf)
            mF = np.full(max(F.totalN, 1), -1, np.int32)
            check(lib().mcs_search_kf_f(self.ctx.h, 1, C.byref(q), 0, C.byref(t), 0, self.mbFeatDim, self.mfNNratio, self.K, MEM_HOST, np_ptr(mF),
                                        np_ptr(nm), np_ptr(fb)))
            self.last_fallbacks = int(fb[0])
            mF = np.where(mF[:F.totalN] >= 0, order[np.maximum(mF[:F.totalN], 0)], -1).astype(np.int32)   # back to keyframe feature indices
            removed = self._rot_filter(0, F.mvKeys, pKF1.mvKeys, mF, True)
            return int(nm[0]) - removed, [mp1[int(i)] if i >= 0 else None for i in mF]
        q, t, keep = self._sets(pKF1._d, pKF1._m, v1, None, F.all_descriptors(), F.all_masks(), None, None)
        mF = np.full(max(F.totalN, 1), -1, np.int32)
        check(lib().mcs_search_kf_f(self.ctx.h, 1, C.byref(q), 0, C.byref(t), 0, self.mbFeatDim, self.mfNNratio, self.K, MEM_HOST, np_ptr(mF),
                                    np_ptr(nm), np_ptr(fb)))
        self.last_fallbacks = int(fb[0])
        mF = np.ascontiguousarray(mF[:F.totalN])
        removed = self._rot_filter(0, F.mvKeys, pKF1.mvKeys, mF, True)
        return int(nm[0]) - removed, [mp1[i] if i >= 0 else None for i in mF]

    def SearchByProjection(self, F, vpMapPoints, th, *rest):
        """int SearchByProjection(cMultiFrame &F, const vector<cMapPoint*> &vpMapPoints, const double th) (src/cORBmatcher.cpp:67-166).
        Map points carry what isInFrustum() left on them (src/cMultiFrame.cpp:218-270): mbTrackInView[cam], mTrackProjX/Y[cam],
        mnTrackScaleLevel[cam], mTrackViewCos[cam], plus GetDescriptor()/GetDescriptorMask() (numpy rows).  Fills F.mvpMapPoints.
        The reference's other overloads dispatch on the argument types like C++ would: (CurrentFrame, LastFrame, th) -> SearchByProjectionLast,
        (F1, F2, windowSize, vpMapPointMatches2) -> SearchByProjectionFrames."""
        if isinstance(vpMapPoints, cMultiFrame):
            if rest:
                return self.SearchByProjectionFrames(F, vpMapPoints, th, rest[0])
            return self.SearchByProjectionLast(F, vpMapPoints, th)
        from ._capi import FrameView, ProjectionSet
        nr = F.camSystem.GetNrCams()
        owner, px, py, vc, lv, pc, dd, mm = [], [], [], [], [], [], [], []
        for pMP in vpMapPoints:                       # the reference's visiting order: map point, then camera
            if pMP.isBad():
                continue
            for cam in range(nr):
                if not pMP.mbTrackInView[cam]:
                    continue
                owner.append(pMP); px.append(pMP.mTrackProjX[cam]); py.append(pMP.mTrackProjY[cam]); vc.append(pMP.mTrackViewCos[cam])
                lv.append(pMP.mnTrackScaleLevel[cam]); pc.append(cam); dd.append(pMP.GetDescriptor())
                if self.havingMasks:
                    mm.append(pMP.GetDescriptorMask())
        n = len(owner)
        if n == 0:
            return 0
        px, py, vc = (np.ascontiguousarray(v, np.float64) for v in (px, py, vc))
        lv, pc = np.ascontiguousarray(lv, np.int32), np.ascontiguousarray(pc, np.int32)
        dd = np.ascontiguousarray(np.stack(dd), np.uint8)
        mm = np.ascontiguousarray(np.stack(mm), np.uint8) if self.havingMasks else None
        keys = np.ascontiguousarray(F.mvKeys)
        fd = np.ascontiguousarray(F.all_descriptors(), np.uint8)
        fm = np.ascontiguousarray(F.all_masks(), np.uint8) if self.havingMasks else None
        fc = np.ascontiguousarray(F.keypoint_to_cam, np.int32)
        assigned = np.array([m is not None for m in F.mvpMapPoints], np.uint8)
        w, h = np.ascontiguousarray(F.mnMaxX, np.int32), np.ascontiguousarray(F.mnMaxY, np.int32)
        sc = np.ascontiguousarray(F.mvScaleFactors, np.float64)
        mp = ProjectionSet(np_ptr(px), np_ptr(py), np_ptr(vc), np_ptr(lv), np_ptr(pc), np_ptr(dd), np_ptr(mm), n, self.mbFeatDim)
        fv = FrameView(np_ptr(keys), np_ptr(fd), np_ptr(fm), np_ptr(fc), np_ptr(assigned), F.totalN, self.mbFeatDim, nr, np_ptr(w), np_ptr(h), np_ptr(sc),
                       len(sc))
        match = np.full(n, -1, np.int32)
        nm = np.zeros(1, np.int32)
        check(lib().mcs_search_by_projection(self.ctx.h, C.byref(mp), C.byref(fv), float(th), self.mfNNratio, self.mbFeatDim, MEM_HOST, np_ptr(match),
                                             np_ptr(nm)))
        for p, j in enumerate(match):
            if j >= 0:
                F.mvpMapPoints[int(j)] = owner[p]
        return int(nm[0])

    # ------------------------------------------------------------------ grid-window searches other than (F, mapPoints)
    def _frame_view(self, F, assigned):
        from ._capi import FrameView
        keys = np.ascontiguousarray(F.mvKeys)
        fd = np.ascontiguousarray(F.all_descriptors(), np.uint8)
        fm = np.ascontiguousarray(F.all_masks(), np.uint8) if self.havingMasks else None
        fc = np.ascontiguousarray(F.keypoint_to_cam, np.int32)
        w, h = np.ascontiguousarray(F.mnMaxX, np.int32), np.ascontiguousarray(F.mnMaxY, np.int32)
        sc = np.ascontiguousarray(F.mvScaleFactors, np.float64)
        fv = FrameView(np_ptr(keys), np_ptr(fd), np_ptr(fm), np_ptr(fc), np_ptr(assigned), F.totalN, self.mbFeatDim, F.camSystem.GetNrCams(), np_ptr(w),
                       np_ptr(h), np_ptr(sc), len(sc))
        return fv, (keys, fd, fm, fc, w, h, sc, assigned)

    def _window_match(self, rule, x, y, r, lo, hi, cam, rows, F1, F2, assigned):
        """probes (window centre / radius / level range / camera / descriptor row of F1) against frame F2 -> (match per probe, nmatches)"""
        from ._capi import WindowProbes
        n = len(rows)
        if n == 0:
            return np.zeros(0, np.int32), 0
        x, y, r = (np.ascontiguousarray(v, np.float64) for v in (x, y, r))
        lo, hi, cam = (np.ascontiguousarray(v, np.int32) for v in (lo, hi, cam))
        rows = np.asarray(rows, np.int64)
        dd = np.ascontiguousarray(F1.all_descriptors()[rows], np.uint8)
        mm = np.ascontiguousarray(F1.all_masks()[rows], np.uint8) if self.havingMasks else None
        self.last_accepted = np.full(n, -1, np.int32)
        pr = WindowProbes(np_ptr(x), np_ptr(y), np_ptr(r), np_ptr(lo), np_ptr(hi), np_ptr(cam), np_ptr(dd), np_ptr(mm), n, self.mbFeatDim,
                          np_ptr(self.last_accepted) if rule == 3 else None)
        fv, keep = self._frame_view(F2, assigned)
        match = np.full(n, -1, np.int32)
        nm = np.zeros(1, np.int32)
        check(lib().mcs_window_match(self.ctx.h, C.byref(pr), C.byref(fv), rule, self.mfNNratio, self.mbFeatDim, MEM_HOST, np_ptr(match), np_ptr(nm)))
        return match, int(nm[0])

    def BestInWindows(self, x, y, r, lo, hi, cam, desc, mask, F, max_dist, skip_taken=False, assigned=None):
        """The search loop of Fuse (src/cORBmatcher.cpp:1265-1719), SearchBySim3 (:1721-1988), SearchForTriangulationBetweenCameras (:1158-1263),
        SearchByProjection(pKF, Scw, ...) (:2265-2392) [skip_taken False] and of the relocalisation SearchByProjection(CurrentFrame, pKF,
        sAlreadyFound, th, ORBdist) (:2120-2263) [skip_taken True, `assigned` updated in place]: per probe (window centre x/y, radius r, level
        range lo..hi, camera, descriptor row) the closest feature of frame F inside the window -> (match [-1 if > max_dist], dist, nmatches)."""
        from ._capi import WindowProbes
        n = len(x)
        if n == 0:
            return np.zeros(0, np.int32), np.zeros(0, np.int32), 0
        x, y, r = (np.ascontiguousarray(v, np.float64) for v in (x, y, r))
        lo, hi, cam = (np.ascontiguousarray(v, np.int32) for v in (lo, hi, cam))
        dd = np.ascontiguousarray(desc, np.uint8)
        mm = np.ascontiguousarray(mask, np.uint8) if self.havingMasks else None
        pr = WindowProbes(np_ptr(x), np_ptr(y), np_ptr(r), np_ptr(lo), np_ptr(hi), np_ptr(cam), np_ptr(dd), np_ptr(mm), n, self.mbFeatDim)
        if skip_taken and assigned is None:
            assigned = np.array([m is not None for m in F.mvpMapPoints] + [0] * (F.totalN == 0), np.uint8)
        fv, keep = self._frame_view(F, assigned if skip_taken else None)
        match, dist, nm = np.full(n, -1, np.int32), np.zeros(n, np.int32), np.zeros(1, np.int32)
        check(lib().mcs_window_best(self.ctx.h, C.byref(pr), C.byref(fv), int(max_dist), int(bool(skip_taken)), self.mbFeatDim, MEM_HOST, np_ptr(match),
                                    np_ptr(dist), np_ptr(nm)))
        return match, dist, int(nm[0])

    def WindowSearch(self, F1, F2, windowSize, minScaleLevel=0, maxScaleLevel=2**31 - 1):
        """-> (nmatches, vpMapPointMatches2) (src/cORBmatcher.cpp:326-473)."""
        from ._capi import WINDOW_RATIO
        rows = [i for i, mp in enumerate(F1.mvpMapPoints) if _good(mp) and not (minScaleLevel > 0 and F1.mvKeys[i]["octave"] < minScaleLevel)
                and not (maxScaleLevel < 2**31 - 1 and F1.mvKeys[i]["octave"] > maxScaleLevel)]
        k = F1.mvKeys[rows]
        n = len(rows)
        assigned = np.zeros(max(F2.totalN, 1), np.uint8)
        match, nm = self._window_match(WINDOW_RATIO, k["x"].astype(np.float64), k["y"].astype(np.float64), np.full(n, float(windowSize)), np.full(n, -1),
                                       np.full(n, -1), F1.keypoint_to_cam[rows], rows, F1, F2, assigned)
        self.last_matches21 = np.full(F2.totalN, -1, np.int32)
        for p, j in enumerate(match):
            if j >= 0:
                self.last_matches21[int(j)] = rows[p]
        nm -= self._rot_filter(1, F2.mvKeys, F1.mvKeys, self.last_matches21, True)
        out = [F1.mvpMapPoints[int(i)] if i >= 0 else None for i in self.last_matches21]
        return nm, out

    def SearchByProjectionFrames(self, F1, F2, windowSize, vpMapPointMatches2):
        """int SearchByProjection(F1, F2, windowSize, vpMapPointMatches2) (src/cORBmatcher.cpp:476-577); the list is filled in place."""
        from ._capi import WINDOW_RATIO
        vpMapPointMatches2[:] = list(F2.mvpMapPoints)
        found = set(id(m) for m in vpMapPointMatches2 if m is not None)
        nr = F1.camSystem.GetNrCams()
        seen, sel = set(), []
        for i1, mp in enumerate(F1.mvpMapPoints):
            if mp is None or (hasattr(mp, "isBad") and mp.isBad()) or id(mp) in found or id(mp) in seen:
                continue
            seen.add(id(mp))
            sel.append(i1)
        if not sel:
            return 0
        pts = np.repeat(np.stack([np.asarray(F1.mvpMapPoints[i].GetWorldPos(), np.float64)[:3] for i in sel]), nr, axis=0)
        cams = np.tile(np.arange(nr, dtype=np.int32), len(sel))
        uv, fl = F2.camSystem.world_to_cam(pts, cams, self.ctx)
        ok = (fl & 1) != 0
        rows = np.repeat(np.asarray(sel), nr)[ok]
        lv = F1.mvKeys["octave"][rows]
        n = len(rows)
        assigned = np.array([m is not None for m in vpMapPointMatches2] + [0] * (F2.totalN == 0), np.uint8)
        match, nm = self._window_match(WINDOW_RATIO, uv[ok, 0], uv[ok, 1], np.full(n, float(windowSize)), lv, lv, cams[ok], rows, F1, F2, assigned)
        for p, j in enumerate(match):
            if j >= 0:
                vpMapPointMatches2[int(j)] = F1.mvpMapPoints[int(rows[p])]
        return nm

    def SearchByProjectionLast(self, CurrentFrame, LastFrame, th):
        """int SearchByProjection(CurrentFrame, LastFrame, th) (src/cORBmatcher.cpp:1990-2118); fills CurrentFrame.mvpMapPoints."""
        from ._capi import WINDOW_BEST
        sel = [i for i, mp in enumerate(LastFrame.mvpMapPoints) if _good(mp) and not LastFrame.mvbOutlier[i]]
        if not sel:
            return 0
        pts = np.stack([np.asarray(LastFrame.mvpMapPoints[i].GetWorldPos(), np.float64)[:3] for i in sel])
        cams = np.ascontiguousarray(LastFrame.keypoint_to_cam[sel], np.int32)
        uv, fl = CurrentFrame.camSystem.world_to_cam(pts, cams, self.ctx)
        ok = (fl & 1) != 0
        rows = np.asarray(sel)[ok]
        octv = LastFrame.mvKeys["octave"][rows]
        radius = float(th) * np.asarray(CurrentFrame.mvScaleFactors, np.float64)[octv]
        assigned = np.array([m is not None for m in CurrentFrame.mvpMapPoints] + [0] * (CurrentFrame.totalN == 0), np.uint8)
        match, nm = self._window_match(WINDOW_BEST, uv[ok, 0], uv[ok, 1], radius, octv - 1, octv + 1, cams[ok], rows, LastFrame, CurrentFrame, assigned)
        mcur = np.full(CurrentFrame.totalN, -1, np.int32)
        for p, j in enumerate(match):
            if j >= 0:
                mcur[int(j)] = rows[p]
        nm -= self._rot_filter(0, CurrentFrame.mvKeys, LastFrame.mvKeys, mcur, True)
        for j, i in enumerate(mcur):
            if i >= 0:
                CurrentFrame.mvpMapPoints[j] = LastFrame.mvpMapPoints[int(i)]
        return nm

    def SearchForInitialization(self, F1, F2, vbPrevMatched, windowSize=10):
        """-> (nmatches, vnMatches12); vbPrevMatched ([n1,2] float64) is updated in place (src/cORBmatcher.cpp:579-726)."""
        from ._capi import WINDOW_INITIALIZE
        n = F1.totalN
        rows = np.arange(n)
        lv = F1.mvKeys["octave"].astype(np.int32)
        match, nm = self._window_match(WINDOW_INITIALIZE, vbPrevMatched[:, 0], vbPrevMatched[:, 1], np.full(n, float(windowSize)), lv, lv,
                                       F1.keypoint_to_cam, rows, F1, F2, None)
        nm -= self._rot_filter(2, F1.mvKeys, F2.mvKeys, match, False, self.last_accepted)
        for i1, j in enumerate(match):
            if j >= 0:
                vbPrevMatched[i1, 0], vbPrevMatched[i1, 1] = float(F2.mvKeys[int(j)]["x"]), float(F2.mvKeys[int(j)]["y"])
        return nm, match

    def SearchForTriangulationRaw(self, pKF1, pKF2, Es):
        """-> (nmatches, vMatchedKeys1, vMatchedKeysRays1, vMatchedKeys2, vMatchedKeysRays2, vMatchedPairs) (:968-1155).
        Es: [nrCams][nrCams] 3x3 essential matrices (the reference precomputes them from the rig poses, :990-1003)."""
        nr = pKF1.camSystem.GetNrCams()
        E = np.ascontiguousarray(np.asarray(Es, np.float64).reshape(nr * nr, 9))
        v1 = np.array([m is None for m in pKF1.GetMapPointMatches()], np.uint8)    # "if (pMP1) continue"
        v2 = np.array([m is None for m in pKF2.GetMapPointMatches()], np.uint8)
        g1 = np.ascontiguousarray(pKF1.keypoint_to_cam, np.int32)
        g2 = np.ascontiguousarray(pKF2.keypoint_to_cam, np.int32)
        q, t, keep = self._sets(pKF1._d, pKF1._m, v1, g1, pKF2._d, pKF2._m, v2, g2)
        r1 = np.ascontiguousarray(pKF1.mvKeysRays, np.float64)
        r2 = np.ascontiguousarray(pKF2.mvKeysRays, np.float64)
        m12 = np.full(max(len(v1), 1), -1, np.int32)
        nm = np.zeros(1, np.int32)
        fb = np.zeros(1, np.int32)
        check(lib().mcs_search_triangulation(self.ctx.h, 1, C.byref(q), 0, C.byref(t), 0, np_ptr(r1), np_ptr(r2), np_ptr(E), nr, self.mbFeatDim,
                                             max(self.K, 16), MEM_HOST, np_ptr(m12), np_ptr(nm), np_ptr(fb)))
        self.last_fallbacks = int(fb[0])
        m12 = np.ascontiguousarray(m12[:len(v1)])
        nm[0] -= self._rot_filter(3, pKF1.mvKeys, pKF2.mvKeys, m12, False)
        pairs = [(i, int(j)) for i, j in enumerate(m12) if j >= 0]
        i1 = [p[0] for p in pairs]
        i2 = [p[1] for p in pairs]
        return int(nm[0]), pKF1.mvKeys[i1], pKF1.mvKeysRays[i1], pKF2.mvKeys[i2], pKF2.mvKeysRays[i2], pairs


def DescriptorDistance64(descr_i, descr_j, dim=32, ctx=None):
    return (ctx or default_context()).descriptor_distance(np.frombuffer(descr_i, np.uint8)[:dim], np.frombuffer(descr_j, np.uint8)[:dim])


def DescriptorDistance64Masked(descr_i, descr_j, mask_i, mask_j, dim=32, ctx=None):
    f = lambda a: np.frombuffer(a, np.uint8)[:dim]
    return (ctx or default_context()).descriptor_distance(f(descr_i), f(descr_j), f(mask_i), f(mask_j))


def ComputeDistinctiveDescriptorsBatch(observed, dim=32, ctx=None):
    """cMapPoint::ComputeDistinctiveDescriptors (src/cMapPoint.cpp:294-382) for many map points in one device call.
    observed: one (descriptors [N_k, dim] uint8, masks [N_k, dim] uint8 or None) pair per map point, rows in the order the reference's loop
    collects them.  -> int32 array, the chosen row per map point (-1 where N_k == 0: the reference keeps the old descriptor)."""
    n = len(observed)
    if n == 0:
        return np.zeros(0, np.int32)
    having = observed[0][1] is not None
    counts = [len(d) for d, _ in observed]
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum(counts)
    rows = int(off[-1])
    dd = np.zeros((max(rows, 1), dim), np.uint8)
    mm = np.zeros((max(rows, 1), dim), np.uint8) if having else None
    for k, (d, m) in enumerate(observed):
        if counts[k]:
            dd[off[k]:off[k + 1]] = d
            if having:
                mm[off[k]:off[k + 1]] = m
    best = np.full(n, -1, np.int32)
    ctx = ctx or default_context()
    check(lib().mcs_distinctive_descriptors(ctx.h, np_ptr(dd), np_ptr(mm), dim, dim, np_ptr(off), n, MEM_HOST, np_ptr(best)))
    return best


class cMapPoint:
    """The descriptor side of cMapPoint (include/cMapPoint.h): observations -> representative descriptor (+ mask)."""

    def __init__(self, Pos=None, ctx=None):
        self.mWorldPos = None if Pos is None else np.asarray(Pos, np.float64)
        self.mObservations = []          # [(keyframe, [feature indices])] in insertion order (the reference's std::map orders by POINTER)
        self.mDescriptor = self.mDescriptorMask = None
        self.mbBad = False
        self.ctx = ctx

    def isBad(self):
        return self.mbBad

    def GetWorldPos(self):
        return self.mWorldPos

    def AddObservation(self, pKF, idx):
        for kf, lst in self.mObservations:
            if kf is pKF:
                lst.append(int(idx))
                return
        self.mObservations.append((pKF, [int(idx)]))

    def _observed(self, havingMasks):
        d, m = [], []
        for kf, lst in self.mObservations:
            if hasattr(kf, "isBad") and kf.isBad():
                continue
            for l in lst:
                d.append(kf._d[l])
                if havingMasks:
                    m.append(kf._m[l])
        dim = d[0].shape[0] if d else 32
        return (np.stack(d) if d else np.zeros((0, dim), np.uint8)), ((np.stack(m) if m else np.zeros((0, dim), np.uint8)) if havingMasks else None)

    def ComputeDistinctiveDescriptors(self, havingMasks):
        if self.mbBad or not self.mObservations:
            return
        d, m = self._observed(havingMasks)
        if len(d) == 0:
            return
        b = int(ComputeDistinctiveDescriptorsBatch([(d, m)], d.shape[1], self.ctx)[0])
        self.mDescriptor = d[b].copy()
        if havingMasks:
            self.mDescriptorMask = m[b].copy()

    def GetDescriptor(self):
        return self.mDescriptor

    def GetDescriptorMask(self):
        return self.mDescriptorMask


class cORBVocabulary:
    """ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (include/cORBVocabulary.h) as far as the front end uses it:
    transform(features, levelsup) -> (BowVector, FeatureVector) (ThirdParty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1205).  The tree descent of
    every descriptor runs on the GPU (mcs_bow_transform); the two std::map's are rebuilt here with the reference's arithmetic order."""

    def __init__(self, voc, ctx=None):
        """voc: the dict of multicol-slam_amd.io.load_vocabulary (or a path to the vocabulary YAML)."""
        if isinstance(voc, str):
            from . import io as _io
            voc = _io.load_vocabulary(voc)
        self.voc = voc
        self.ctx = ctx or default_context()
        self.m_k, self.m_L = voc["k"], voc["L"]
        if voc["scoringType"] != 0 or voc["weightingType"] != 0:
            raise NotImplementedError("only TF_IDF weighting with L1 scoring (the shipped small_orb_omni_voc_9_6.yml) is mirrored")
        self.h = C.c_void_p()
        nd = np.ascontiguousarray(voc["node_desc"], np.uint8)
        co, ci = np.ascontiguousarray(voc["child_off"], np.int32), np.ascontiguousarray(voc["child_idx"], np.int32)
        check(lib().mcs_vocabulary_create(self.ctx.h, len(nd), np_ptr(nd), np_ptr(co), np_ptr(ci), self.m_L, C.byref(self.h)))

    def __del__(self):
        try:
            if self.h:
                lib().mcs_vocabulary_destroy(self.h)
        except Exception:
            pass

    def descend(self, descriptors, levelsup):
        """-> (leaf node, node at level L - levelsup) per descriptor row"""
        d = np.ascontiguousarray(descriptors, np.uint8)
        n = len(d)
        leaf, nid = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        if n:
            check(lib().mcs_bow_transform(self.h, np_ptr(d), n, d.shape[1], int(levelsup), MEM_HOST, np_ptr(leaf), np_ptr(nid)))
        return leaf[:n], nid[:n]

    def transform(self, descriptors, levelsup=4):
        """-> (BowVector {word id: value}, FeatureVector {node id: [feature indices]}), both in ascending key order like std::map."""
        leaf, nid = self.descend(descriptors, levelsup)
        words, weights = self.voc["word_id"][leaf], self.voc["weight"][leaf]
        bow, fv = {}, {}
        for i in range(len(leaf)):            # TF_IDF branch: addWeight accumulates in feature order (:1147-1163)
            w = float(weights[i])
            if w > 0:
                wid = int(words[i])
                bow[wid] = bow.get(wid, 0.0) + w
                fv.setdefault(int(nid[i]), []).append(i)
        bow = dict(sorted(bow.items()))
        norm = 0.0                            # L1 scoring: mustNormalize -> BowVector::normalize(L1) in key order (BowVector.cpp)
        for v in bow.values():
            norm += abs(v)
        if norm > 0.0:
            bow = {k: v / norm for k, v in bow.items()}
        return bow, dict(sorted(fv.items()))

"""On-disk formats of the reference's Lafida example, so its calibration / settings / image lists drive this front end unchanged
(SURVEY §8f row 4):

  read_settings            the flat `key: value` subset of OpenCV FileStorage YAML the reference's files use (`%YAML:1.0`, `#` comments)
  LoadMCS                  cSystem::LoadMCS (src/cSystem.cpp:125-180): MultiCamSys_Calibration.yaml + InteriorOrientationFisheye<c>.yaml
  cayley2rot / cayley2hom  include/misc.h:132-160, 211-224
  make_extractors          the extractor construction of cTracking::cTracking (src/cTracking.cpp:109-158)
  LoadImagesAndTimestamps  Examples/Lafida/mult_col_slam_lafida.cpp:167-198
  read_pgm                 binary PGM (P5) reader (the image decoder itself — cv::imread — is outside the path; any uint8 array works)

Host-side plumbing only; nothing numeric beyond the 3x3 Cayley formula.
"""
import os

import numpy as np

from . import frontend as FE
from . import synth


def read_settings(path):
    """-> {key: int | float | str}.  Values keep the reference's meaning: cv::FileNode converts on read, so ints stay ints here and
    callers cast like the reference does ((int)node, float node)."""
    out = {}
    with open(path, "r", encoding="latin-1") as f:
        for raw in f:
            line = raw.split("#", 1)[0].strip()
            if not line or line.startswith("%") or line == "---":
                continue
            if ":" not in line:
                continue
            key, val = line.split(":", 1)
            key, val = key.strip(), val.strip().strip('"')
            if not key or val == "":
                continue
            try:
                out[key] = int(val)
            except ValueError:
                try:
                    out[key] = float(val)
                except ValueError:
                    out[key] = val
    return out


def cayley2rot(c):   # include/misc.h:132-160
    c1, c2, c3 = (float(v) for v in c)
    c1s, c2s, c3s = c1 * c1, c2 * c2, c3 * c3
    scale = 1.0 + c1s + c2s + c3s
    R = np.array([[1 + c1s - c2s - c3s, 2 * (c1 * c2 - c3), 2 * (c1 * c3 + c2)],
                  [2 * (c1 * c2 + c3), 1 - c1s + c2s - c3s, 2 * (c2 * c3 - c1)],
                  [2 * (c1 * c3 - c2), 2 * (c2 * c3 + c1), 1 - c1s - c2s + c3s]], np.float64)
    return (1 / scale) * R


def cayley2hom(c6):   # include/misc.h:211-224
    M = np.eye(4)
    M[:3, :3] = cayley2rot(c6[:3])
    M[:3, 3] = [float(v) for v in c6[3:6]]
    return M


def load_camera(path):
    """One InteriorOrientationFisheye<c>.yaml -> cCamModelGeneral_ (+ level-0 mirror mask if Camera.mirrorMask == 1, else all ones)."""
    s = read_settings(path)
    nrpol, nrinvpol = int(s["Camera.nrpol"]), int(s["Camera.nrinvpol"])
    poly = [0.0] * max(5, nrpol)          # cv::Mat::zeros(5,1) / zeros(12,1) in the reference: shorter polynomials are zero-padded
    for i in range(nrpol):
        poly[i] = float(s["Camera.a%d" % i])
    invpoly = [0.0] * max(12, nrinvpol)
    for i in range(nrinvpol):
        invpoly[i] = float(s["Camera.pol%d" % i])
    Iw, Ih = int(s["Camera.Iw"]), int(s["Camera.Ih"])
    cal = dict(c=float(s["Camera.c"]), d=float(s["Camera.d"]), e=float(s["Camera.e"]), u0=float(s["Camera.u0"]), v0=float(s["Camera.v0"]), p=poly,
               invP=invpoly, width=Iw, height=Ih)
    mask = synth.mirror_mask(cal) if int(s.get("Camera.mirrorMask", 0)) == 1 else np.ones((Ih, Iw), np.uint8)
    return FE.cCamModelGeneral_.from_dict(cal, mask)


def LoadMCS(path2calibrations):
    """cSystem::LoadMCS -> cMultiCamSys_ with M_t = identity."""
    s = read_settings(os.path.join(path2calibrations, "MultiCamSys_Calibration.yaml"))
    nrCams = int(s["CameraSystem.nrCams"])
    M_c, models = [], []
    for c in range(nrCams):
        M_c.append(cayley2hom([float(s["CameraSystem.cam%d_%d" % (c + 1, p)]) for p in range(1, 7)]))
        models.append(load_camera(os.path.join(path2calibrations, "InteriorOrientationFisheye%d.yaml" % c)))
    return FE.cMultiCamSys_(models, M_c, np.eye(4))


def make_extractors(settings, nrCams=1, ctx=None):
    """-> (extractors, init_extractors) as cTracking builds them (src/cTracking.cpp:109-158): the tracking extractor with the settings'
    nFeatures / fastTh, the initialisation extractor with 2*nFeatures and FAST threshold 5.  One shared object serves every camera (the
    reference's per-camera instances are identical)."""
    s = read_settings(settings) if isinstance(settings, str) else settings
    nF, sf, nl = int(s["extractor.nFeatures"]), float(np.float32(s["extractor.scaleFactor"])), int(s["extractor.nLevels"])
    fastTh, score = int(s["extractor.fastTh"]), int(s["extractor.nScoreType"])
    usemd, masks = bool(int(s["extractor.usemdBRIEF"])), bool(int(s["extractor.masks"]))
    agast, ftype, dsz = bool(int(s["extractor.useAgast"])), int(s["extractor.fastAgastType"]), int(s["extractor.descSize"])
    if score not in (0, 1) or dsz not in (16, 32, 64):
        raise ValueError("extractor.nScoreType must be 0/1 and extractor.descSize 16/32/64 (asserts in src/cTracking.cpp:116,133)")
    ex = FE.mdBRIEFextractorOct(nF, sf, nl, 25, 0, score, 32, fastTh, agast, ftype, usemd, masks, dsz, ctx=ctx)
    ini = FE.mdBRIEFextractorOct(2 * nF, sf, nl, 25, 0, score, 32, 5, agast, ftype, usemd, masks, dsz, ctx=ctx)
    return [ex] * nrCams, [ini] * nrCams


def LoadImagesAndTimestamps(startFrame, endFrame, path2imgs):
    """-> (vstrImageFilenames[3][n], vTimestamps[n]); line numbers are 1-based, [startFrame, endFrame) like the reference's loop."""
    names, stamps = [[], [], []], []
    with open(os.path.join(path2imgs, "images_and_timestamps.txt"), "r") as f:
        for cnt, line in enumerate(f, 1):
            if startFrame <= cnt < endFrame:
                parts = line.split()
                if len(parts) < 4:
                    break
                try:
                    t = float(parts[0])
                except ValueError:
                    break
                stamps.append(t)
                for c in range(3):
                    names[c].append(path2imgs + "/" + parts[1 + c])
    return names, stamps


def read_pgm(path):
    """binary PGM (P5, maxval <= 255) -> uint8 [h, w]"""
    with open(path, "rb") as f:
        data = f.read()
    tok, pos = [], 0
    while len(tok) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tok.append(data[pos:end])
        pos = end
    if tok[0] != b"P5" or int(tok[3]) > 255:
        raise ValueError("only 8-bit binary PGM (P5) is supported")
    w, h = int(tok[1]), int(tok[2])
    return np.frombuffer(data, np.uint8, w * h, pos + 1).reshape(h, w).copy()


def load_vocabulary(path):
    """DBoW2 vocabulary in OpenCV-YAML form (TemplatedVocabulary::load, ThirdParty/DBoW2/DBoW2/TemplatedVocabulary.h:1573-1622) -> the flat
    arrays mcs_vocabulary_create takes plus the per-node word id / weight tables:
      k, L, scoringType, weightingType, node_desc [N+1, 32] u8 (row 0 = root, zeros), parent [N+1], child_off [N+2], child_idx [N] (children in
      file order), word_id [N+1] (-1 for inner nodes), weight [N+1] float64, n_words."""
    import re
    txt = open(path, "r", encoding="latin-1").read()
    head = txt[:txt.index("nodes:")]
    hdr = {k: int(re.search(r"\b%s:\s*(-?\d+)" % k, head).group(1)) for k in ("k", "L", "scoringType", "weightingType")}
    nodes = re.findall(r"nodeId:\s*(\d+),\s*parentId:\s*(\d+),\s*weight:\s*([^,\s]+),\s*descriptor:\s*\"([^\"]*)\"", txt)
    words = re.findall(r"wordId:\s*(\d+),\s*nodeId:\s*(\d+)", txt)
    n = len(nodes)
    if n == 0:
        raise ValueError("no nodes in %s" % path)
    node_desc = np.zeros((n + 1, 32), np.uint8)
    parent = np.zeros(n + 1, np.int32)
    weight = np.zeros(n + 1, np.float64)
    children = [[] for _ in range(n + 1)]
    for nid, pid, w, d in nodes:
        nid, pid = int(nid), int(pid)
        vals = d.split()
        if len(vals) != 32:
            raise ValueError("FORB descriptors are 32 bytes (node %d has %d)" % (nid, len(vals)))
        node_desc[nid] = np.array(vals, np.int64).astype(np.uint8)
        parent[nid], weight[nid] = pid, float(w)
        children[pid].append(nid)
    child_off = np.zeros(n + 2, np.int32)
    child_off[1:] = np.cumsum([len(c) for c in children])
    child_idx = np.array([c for lst in children for c in lst], np.int32)
    word_id = np.full(n + 1, -1, np.int32)
    for wid, nid in words:
        word_id[int(nid)] = int(wid)
    out = dict(hdr)
    out.update(node_desc=node_desc, parent=parent, child_off=child_off, child_idx=child_idx, word_id=word_id, weight=weight, n_words=len(words))
    return out

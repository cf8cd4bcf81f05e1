"""WindowSearch / SearchForInitialization / SearchByProjection(Cur, Last) of one multi-frame, a few calls each, for a rocprofv3 --kernel-trace --stats run."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mcs = importlib.import_module("multicol-slam_amd")
synth = importlib.import_module("multicol-slam_amd.synth")
FE = importlib.import_module("multicol-slam_amd.frontend")


class MP:
    def __init__(self, i):
        self.i = i

    def isBad(self):
        return False


ctx = mcs.Context(0)
cams = synth.lafida_cameras()
rig = FE.cMultiCamSys_([FE.cCamModelGeneral_.from_dict(c, synth.mirror_mask(c)) for c in cams])
ex = FE.mdBRIEFextractorOct(1000, 1.2, 8, 25, 0, 0, 32, 20, False, 2, True, True, 32, ctx=ctx)
F = [FE.cMultiFrame(synth.synth_multiframe(f, cams), 0.04 * f, [ex] * 3, None, rig, f) for f in range(2)]
rng = np.random.default_rng(5)
m = FE.cORBmatcher(0.8, False, 32, True, ctx=ctx)
F[0].mvpMapPoints = [MP(i) if rng.random() < 0.7 else None for i in range(F[0].totalN)]
F[1].mvpMapPoints = [None] * F[1].totalN
w = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for i in range(5):
    m.WindowSearch(F[0], F[1], w, 0, 2**31 - 1)
ctx.close()

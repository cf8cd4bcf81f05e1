#!/usr/bin/env python3
"""Time the native C++ rig host (multicol-slam_amd/host/rig_host) on the bench's own workloads: python tools/rig_host_bench.py [frames keyframes steps] ...
Writes the synthetic inputs to /tmp, runs the host on every visible GPU it is told to use (gpus 1 here), prints its JSON line."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mcs = importlib.import_module("multicol-slam_amd")
synth = importlib.import_module("multicol-slam_amd.synth")
HOST = os.path.join(ROOT, "multicol-slam_amd", "host")
subprocess.check_call(["make", "-s", "-C", HOST])
cams = synth.lafida_cameras()
args = [int(a) for a in sys.argv[1:]] or [64, 0, 20, 16, 32, 10]
for frames, keyframes, steps in zip(args[0::3], args[1::3], args[2::3]):
    d = "/tmp/righost_%d_%d" % (frames, keyframes)
    os.makedirs(d, exist_ok=True)
    np.stack([synth.stream_image(f, c, cams[c], 64) for c in range(3) for f in range(frames)]).tofile(d + "/images.bin")
    np.stack([synth.mirror_mask(cams[c]) for c in range(3)]).tofile(d + "/masks.bin")
    open(d + "/cams.bin", "wb").write(bytes((mcs.Ocam * 3)(*[mcs.make_ocam(cams[c]) for c in range(3)])))
    open(d + "/cfg.txt", "w").write("ncam 3\nwidth 754\nheight 480\nnfeatures 1000\nmode 2\nframes %d\nkeyframes %d\ngpus 1\nsteps %d\nwarmup 3\ntopk 32\nimages %s/images.bin\nmasks %s/masks.bin\n"
                                    "cams %s/cams.bin\nout %s/out\n" % (frames, keyframes, steps, d, d, d, d))
    r = subprocess.run([HOST + "/rig_host", d + "/cfg.txt"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.returncode == 0 else "FAILED: " + r.stderr[-500:])
    nkp = np.fromfile(d + "/out.r0.nkp", np.int32)
    print("   features per step %d" % nkp.sum())

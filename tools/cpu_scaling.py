#!/usr/bin/env python3
"""Oracle (CPU baseline) thread scaling on this box: extraction wall time of N images for several thread counts."""
import ctypes as C, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
synth = importlib.import_module("multicol-slam_amd.synth")
cams = synth.lafida_cameras()
pool = [synth.synth_multiframe(f, cams) for f in range(4)]
mk = [np.ascontiguousarray(synth.mirror_mask(c)) for c in cams]
L = O.lib()
L.orc_extract_match_many.restype = C.c_long
L.orc_extract_match_many.argtypes = [C.POINTER(O.Params), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
for mode in ((1, 1), (0, 0)):
    prm = O.make_params(do_dBrief=mode[0], learnMasks=mode[1])
    for threads in [1, 8, 32, 64, 128, 256]:
        if threads > (os.cpu_count() or 1): continue
        nf = max(4, threads)          # one multi-frame (3 images) per thread
        flat = [np.ascontiguousarray(pool[f % 4][c]) for f in range(nf) for c in range(3)]
        iptr = (C.c_void_p * len(flat))(*[a.ctypes.data for a in flat])
        mptr = (C.c_void_p * len(flat))(*[mk[i % 3].ctypes.data for i in range(len(flat))])
        ocs = (O.Ocam * len(flat))(*[O.make_ocam(cams[i % 3]) for i in range(len(flat))])
        nm = (C.c_int * nf)(); secs = (C.c_double * 2)()
        for _ in range(2):
            tot = L.orc_extract_match_many(C.byref(prm), nf, 3, iptr, 754, 480, 754, mptr, ocs, threads, 0.9, nm, secs)
        print("mode", mode, "threads", threads, "images", nf * 3, "extract %.3fs match %.3fs" % (secs[0], secs[1]),
              "-> %.1f ms/image/thread" % (secs[0] * 1e3 * threads / (nf * 3)), "Mfeat/s %.3f" % (tot / nf * (nf - 1) / (secs[0] + secs[1]) / 1e6), flush=True)

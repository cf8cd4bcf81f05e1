#!/usr/bin/env python3
"""A/B: the default workload as S independent sub-streams of F/S multi-frames on S HIP streams / library contexts (phases of one fill the
latency-bound stretches of the other).  Usage: python tools/exp_substreams.py [S ...]"""
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = bench.parse(["--no-cpu-baseline", "--no-secondary"])
e = bench.setup()
torch = e.torch
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    envs, jobs = [], []
    for i in range(S):
        ei = copy.copy(e)
        if i:
            ei.stream = torch.cuda.Stream(device=e.dev)
            ei.ctx = e.mcs.Context(e.local, ei.stream.cuda_stream)
        a = copy.copy(args)
        a.frames = 64 // S
        with torch.cuda.stream(ei.stream):
            jobs.append(bench.Job(ei, bench.Spec(a, 1)))
        envs.append(ei)

    def step():
        for j in jobs:
            j.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    feats = sum(j.local_features() for j in jobs)
    print("substreams %d: %.3f ms per 64 multi-frames, %.2f Mfeatures/s" % (S, el / n * 1e3, feats * n / el / 1e6))
    for j in jobs:
        j.close()

#!/usr/bin/env python3
"""bench.py — MultiCol-SLAM feature front end + brute-force matcher on MI355X.

A "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
F three-camera 754x480 multi-frames -> mdBRIEF extraction (pyramid, FAST, oct-tree, orientation, blur, descriptors+masks,
rays) -> SearchByBoW(KF,KF) brute force (masked Hamming, ratio 0.9) of every multi-frame against the previous multi-frame
of the stream (BASELINE.json configs[1]).  value = features extracted AND matched per second, whole job.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); the stream of multi-frames is sharded across ranks
(independent units -> weak scaling, no data-path collective); barrier + max-over-ranks timing.
torch is plumbing only (device memory, stream, process group); all compute is libmcs_hip.so through its C ABI.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=32, help="multi-frames per step and GPU (3 images each)")
    ap.add_argument("--mode", default="mdbrief", choices=["orb", "dbrief", "mdbrief"])
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--topk", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=48, help="multi-frames in the bounded CPU-baseline sample")
    ap.add_argument("--check", action="store_true", help="verify one multi-frame of the timed output against the oracle")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    mcs = importlib.import_module("multicol-slam_amd")
    synth = importlib.import_module("multicol-slam_amd.synth")
    lib = mcs.lib()

    W, H, NCAM = 754, 480, 3
    F = args.frames
    nimg = F * NCAM
    modes = {"orb": (0, 0), "dbrief": (1, 0), "mdbrief": (1, 1)}
    do_db, masks_on = modes[args.mode]
    cams = synth.lafida_cameras()

    # ---- synthetic stream shard of this rank: POOL distinct multi-frames, tiled to F (generation is untimed)
    POOL = min(F, 8)
    base = rank * 1000
    pool = [synth.synth_multiframe(base + f, cams) for f in range(POOL)]
    imgs_np = np.stack([pool[f % POOL][c] for f in range(F) for c in range(NCAM)])
    masks_np = np.stack([synth.mirror_mask(cams[c]) for _ in range(F) for c in range(NCAM)])
    stream = torch.cuda.current_stream(dev)
    ctx = mcs.Context(local, stream.cuda_stream)
    ex = mcs.Extractor(ctx, W, H, max_batch=nimg, nfeatures=args.nfeatures, do_dBrief=do_db, learnMasks=masks_on)
    cap, ds = ex.cap, 32
    d_imgs = torch.from_numpy(imgs_np).to(dev)
    d_masks = torch.from_numpy(masks_np).to(dev)
    camarr = (mcs.Ocam * nimg)(*[mcs.make_ocam(cams[i % NCAM]) for i in range(nimg)])

    # device outputs; descriptor-side buffers carry one extra multi-frame slot (slot 0 = last multi-frame of the previous step)
    rows_f = NCAM * cap
    d_nkp = torch.zeros((F + 1) * NCAM, dtype=torch.int32, device=dev)
    d_kps = torch.zeros(((F + 1) * rows_f, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros(((F + 1) * rows_f, ds), dtype=torch.uint8, device=dev)
    d_dmask = torch.zeros(((F + 1) * rows_f, ds), dtype=torch.uint8, device=dev)
    d_rays = torch.zeros(((F + 1) * rows_f, 3), dtype=torch.float64, device=dev)
    d_valid = torch.zeros((F + 1) * rows_f, dtype=torch.uint8, device=dev)
    d_match = torch.full((F * rows_f,), -1, dtype=torch.int32, device=dev)
    d_nmatch = torch.zeros(F, dtype=torch.int32, device=dev)
    d_fb = torch.zeros(F, dtype=torch.int32, device=dev)

    def p(t, row_off=0):
        return t.data_ptr() + row_off * (t.stride(0) * t.element_size() if t.dim() > 1 else t.element_size())

    q = mcs.DescSet(p(d_desc, rows_f), p(d_dmask, rows_f) if masks_on else None, p(d_valid, rows_f), None, rows_f, ds)
    t = mcs.DescSet(p(d_desc, 0), p(d_dmask, 0) if masks_on else None, p(d_valid, 0), None, rows_f, ds)

    def step():
        ex.extract_device(nimg, d_imgs.data_ptr(), W * H, W, d_masks.data_ptr(), W * H, W, camarr, p(d_nkp, NCAM), p(d_kps, rows_f), p(d_desc, rows_f),
                          p(d_dmask, rows_f), p(d_rays, rows_f))
        mcs.check(lib.mcs_rows_valid(ctx.h, C.c_void_p(p(d_nkp, 0)), (F + 1) * NCAM, cap, C.c_void_p(p(d_valid, 0))))
        mcs.check(lib.mcs_search_kf_kf(ctx.h, F, C.byref(q), rows_f, C.byref(t), rows_f, ds, 0.9, args.topk, mcs.MEM_DEVICE, C.c_void_p(d_match.data_ptr()),
                                       C.c_void_p(d_nmatch.data_ptr()), C.c_void_p(d_fb.data_ptr())))
        # the last multi-frame becomes the stored keyframe for the next step (slot F -> slot 0)
        d_nkp[:NCAM].copy_(d_nkp[F * NCAM:], non_blocking=True)
        d_desc[:rows_f].copy_(d_desc[F * rows_f:], non_blocking=True)
        d_dmask[:rows_f].copy_(d_dmask[F * rows_f:], non_blocking=True)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync_all()
    ex.status()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    ex.status()

    feats_step = int(d_nkp[NCAM:].sum().item())
    matches_step = int(d_nmatch.sum().item())
    fallbacks = int(d_fb.sum().item())
    tel = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    tfe = torch.tensor([feats_step], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tel, op=dist.ReduceOp.MAX)
        dist.all_reduce(tfe, op=dist.ReduceOp.SUM)
    elapsed_max = float(tel.item())
    feats_all = float(tfe.item())

    # ---- per-kernel device time (HIP events on the kernels' own stream), separate untimed passes
    kern = {}
    roof = None
    if rank == 0:
        ctx.enable_timing(True)
        acc = {}
        reps = 5
        for _ in range(reps):
            step()
            torch.cuda.synchronize(dev)
            for name in ("pyramid", "fast", "octree", "blur", "describe", "match", "greedy"):
                acc[name] = acc.get(name, 0.0) + ctx.kernel_ms(name)
        ctx.enable_timing(False)
        kern = {k: v / reps for k, v in acc.items()}
        nkp_total = feats_step
        # algorithmic bytes per launch (SURVEY.md §8d / DESIGN.md): per keypoint orientation disc 845 B + 512 B of samples per
        # pattern (1 ORB/dBRIEF, 3 mdBRIEF) + 28 B keypoint + 32 B descriptor (+ 32 B mask)
        per_kp = 845 + 512 * (3 if args.mode == "mdbrief" else 1) + 28 + 32 + (32 if args.mode == "mdbrief" else 0)
        S = [754 * 480, 628 * 400, 524 * 333, 436 * 278, 364 * 231, 303 * 193, 253 * 161, 210 * 134]
        alg = {
            "describe": per_kp * nkp_total,
            "pyramid": nimg * (sum(S) - S[-1] + sum(S) - S[0]),
            "fast": nimg * sum(S),
            "blur": nimg * 2 * sum(S),
        }
        dom = max(alg, key=lambda k: kern[k])
        ach = alg[dom] / (kern[dom] * 1e-3) / 1e9
        roof = {"kernel": "k_" + dom, "bound": "hbm", "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 5),
                "traffic": None, "alg_bytes_per_launch": int(alg[dom]), "avg_launch_ms": round(kern[dom], 4),
                "per_kernel_ms": {k: round(v, 4) for k, v in kern.items()},
                "per_kernel_alg_GBps": {k: round(alg[k] / (kern[k] * 1e-3) / 1e9, 1) for k in alg}}

    # ---- optional oracle check of one multi-frame of the timed output
    check = None
    if args.check and rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        f = 1
        ok = True
        nk = d_nkp.cpu().numpy()
        dd = d_desc.cpu().numpy()
        mm = d_dmask.cpu().numpy()
        for c in range(NCAM):
            _, od, om = O.Extractor(nfeatures=args.nfeatures, do_dBrief=do_db, learnMasks=masks_on)(imgs_np[f * NCAM + c], masks_np[c], O.make_ocam(cams[c]))
            i = (f + 1) * NCAM + c
            ok = ok and nk[i] == len(od) and (dd[i * cap:i * cap + len(od)] == od).all() and (mm[i * cap:i * cap + len(od)] == om).all()
        check = bool(ok)

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample of the same workload (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        nf = args.cpu_frames
        cpool = [pool[f % POOL] for f in range(nf)]
        flat = [np.ascontiguousarray(cpool[f][c]) for f in range(nf) for c in range(NCAM)]
        mk = [np.ascontiguousarray(synth.mirror_mask(cams[c])) for c in range(NCAM)]
        iptr = (C.c_void_p * len(flat))(*[a.ctypes.data for a in flat])
        mptr = (C.c_void_p * len(flat))(*[mk[i % NCAM].ctypes.data for i in range(len(flat))])
        ocs = (O.Ocam * len(flat))(*[O.make_ocam(cams[i % NCAM]) for i in range(len(flat))])
        prm = O.make_params(nfeatures=args.nfeatures, do_dBrief=do_db, learnMasks=masks_on)
        threads = os.cpu_count() or 1
        nmatch = (C.c_int * nf)()
        secs = (C.c_double * 2)()
        L = O.lib()
        L.orc_extract_match_many.restype = C.c_long
        L.orc_extract_match_many.argtypes = [C.POINTER(O.Params), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_double, C.c_void_p, C.c_void_p]
        tot = L.orc_extract_match_many(C.byref(prm), nf, NCAM, iptr, W, H, W, mptr, ocs, threads, 0.9, nmatch, secs)
        wall = secs[0] + secs[1]
        # the first multi-frame has no predecessor: count the features that were extracted AND matched
        per_frame = tot / nf
        cpu = {"value": round(per_frame * (nf - 1) / wall / 1e6, 4), "unit": "Mfeatures/s", "cores": threads, "kind": "port",
               "sample": "%d multi-frames (%d images) of the same synthetic stream: oracle extract (%s) %.2fs + SearchByBoW(KF,KF) vs previous frame %.2fs wall, OpenMP over images/frames"
               % (nf, nf * NCAM, args.mode, secs[0], secs[1]), "cpu_model": _cpu_model()}

    if rank == 0:
        ms = elapsed_max / args.steps * 1e3
        value = feats_all * args.steps / elapsed_max / 1e6
        out = {
            "metric": "Mfeatures/s extract+match, 3-cam 754x480 multi-frame", "value": round(value, 3), "unit": "Mfeatures/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8+i32 (images, Hamming) / f64 (omni model)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 3-camera 754x480 multi-frames, %s extract (N=%d, 8 levels, FAST 20) + cORBmatcher BF-Hamming "
                                   "SearchByBoW(KF,KF) vs the previous multi-frame, %d multi-frames (%d images) per step per GPU, inputs resident in HBM"
                                   % (args.mode, args.nfeatures, F, nimg), "multi_frames_per_step_per_gpu": F, "features_per_step": int(feats_all),
                       "matches_per_step_rank0": matches_step, "greedy_rescans_rank0": fallbacks, "parallelism": "stream-shard x%d" % world},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if check is not None:
            out["oracle_check"] = check
        if cpu:
            out["speedup_vs_cpu_all_cores"] = round(value / cpu["value"], 2)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()

"""world_size-2 (and 3) CPU tests of the N>1 path (gloo backend): camera / keyframe sharding and the descriptor all-gather."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ncam, F, cap, ds, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rig = importlib.import_module("multicol-slam_amd.rig")
    rng = np.random.default_rng(123)                      # every rank can regenerate the full truth
    full_d = rng.integers(0, 256, (F, ncam, cap, ds)).astype(np.uint8)
    full_m = rng.integers(0, 256, (F, ncam, cap, ds)).astype(np.uint8)
    full_n = rng.integers(0, cap + 1, (F, ncam)).astype(np.int32)
    mine = rig.camera_shard(ncam, rank, world)
    d = torch.from_numpy(full_d[:, mine].copy())
    m = torch.from_numpy(full_m[:, mine].copy())
    n = torch.from_numpy(full_n[:, mine].copy())
    ad, am, an = rig.allgather_rig(d, m, n, ncam, rank, world)
    ok = bool((ad.numpy() == full_d).all() and (am.numpy() == full_m).all() and (an.numpy() == full_n).all())
    t, u = rig.reduce_timing(1.0 + rank, 10.0 * (rank + 1), torch.device("cpu"), world)
    ok = ok and t == float(world) and u == 10.0 * world * (world + 1) / 2
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ncam", [(2, 3), (2, 6), (3, 8)])
def test_allgather_rig_gloo(world, ncam):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ncam, 2, 7, 32, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == list(range(world)) and all(ok for _, ok in res)


def test_shards_partition():
    rig = importlib.import_module("multicol-slam_amd.rig")
    for world in (1, 2, 4, 8):
        for n in (3, 6, 8, 256):
            cams = [rig.camera_shard(n, r, world) for r in range(world)]
            assert sorted(sum(cams, [])) == list(range(n))
            kfs = [rig.keyframe_shard(n, r, world) for r in range(world)]
            assert sorted(sum(kfs, [])) == list(range(n))
            assert max(len(c) for c in cams) == rig.cams_per_rank(n, world)


def test_every_rank_of_the_stream_bench_gets_images_with_content():
    """bench.py shards the synthetic stream over ranks; the scene drifts with the frame number, so a shard far down the stream would be empty
    images (ranks >= 1 once extracted ~0 features).  Every rank's frames must look like rank 0's."""
    import importlib
    import bench
    synth = importlib.import_module("multicol-slam_amd.synth")
    cam = synth.lafida_cameras()[0]
    ref = synth.synth_image(0, 0, cam).astype(np.float64).std()
    for rank in (1, 7):
        for f in bench.shard_frames(rank, 8)[::7]:
            assert synth.synth_image(f, 0, cam).astype(np.float64).std() > 0.9 * ref, (rank, f)
    assert bench.shard_frames(0, 8) == list(range(8)) and len(set(sum((bench.shard_frames(r, 8) for r in range(8)), []))) == 64

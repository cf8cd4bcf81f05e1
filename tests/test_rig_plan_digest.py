"""The per-rank plan digest of the multi-GPU runs (multicol-slam_amd/rig.py plan_digest / check_plan_digests; host/rig_host.cpp plan_digest), on CPU.

No multi-GPU hardware has been available to the builder, so the first N > 1 execution must either run the plan `bench.py --dry-run` verified or say which rank
diverged: every rank computes 32 bytes from ITS OWN arguments (slabs, transfer runs or keyframe shards, frame pairs) before the first exchange, the ranks all-gather
and compare them.  Here: the C++ host's restatement of the layout yields the same bytes as rig.py for every BASELINE workload and world size (the partitioning
contract is the reference's per-camera split, src/cMultiFrame.cpp:128-164, over ranks); a world-2 / world-3 gloo group agrees on equal arguments and names the odd
rank on unequal ones."""
import importlib
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rig = importlib.import_module("multicol-slam_amd.rig")
HOST = os.path.join(ROOT, "multicol-slam_amd", "host", "rig_host")

WORKLOADS = [(3, 64, 0, 1000), (3, 64, 32, 1000), (6, 4, 32, 2000), (8, 1, 256, 2000), (3, 2, 0, 400)]


def test_digest_depends_on_every_argument():
    base = (3, 4, 2, 1024, 0, 32, 32)
    d0 = rig.plan_digest(*base)
    assert len(d0) == 32 and d0 == rig.plan_digest(*base)
    for i, v in enumerate((6, 8, 4, 1000, 8, 64, 16)):
        a = list(base)
        a[i] = v
        assert rig.plan_digest(*a) != d0, i


@pytest.mark.skipif(not os.path.exists(HOST), reason="rig_host not built (__graft_entry__.build())")
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_native_host_computes_the_same_digest(tmp_path, world):
    for ncam, F, D, nfeat in WORKLOADS:
        cap = nfeat + 24
        cfg = tmp_path / "plan.cfg"
        cfg.write_text("ncam %d\nframes %d\nkeyframes %d\nnfeatures %d\ncap %d\ntopk 32\n" % (ncam, F, D, nfeat, cap))
        r = subprocess.run([HOST, str(cfg), "--plan-only", "--gpus", str(world)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        got = json.loads(r.stdout.strip().splitlines()[-1])
        assert got["world"] == world and got["plan_digest"] == rig.plan_digest(ncam, F, world, cap, D, 32, 32).hex(), (ncam, F, D, world)


def test_dry_run_prints_the_digest():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["ok"] and all(len(row["plan_digest"]) == 64 for row in rep["dry_run"])
    row = [x for x in rep["dry_run"] if x["workload"] == "stream"][0]
    assert row["plan_digest"] == rig.plan_digest(3, 64, 4, 1024, 0, 32, 32).hex()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, odd, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = 5 if rank == odd else 4   # the odd rank was started with another frame count
    dg = rig.plan_digest(3, frames, world, 1024, 0)

    def gather(b):
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        out = torch.empty(32 * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(out, t)
        return out.numpy().tobytes()
    try:
        q.put((rank, "ok", rig.check_plan_digests(dg, rank, world, gather)))
    except ValueError as ex:
        q.put((rank, "mismatch", str(ex)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,odd", [(2, -1), (3, -1), (3, 1), (2, 0)])
def test_ranks_compare_digests_over_gloo(world, odd):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, odd, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    if odd < 0:
        assert all(kind == "ok" for _, kind, _ in res) and len({d for _, _, d in res}) == 1
        assert res[0][2] == rig.plan_digest(3, 4, world, 1024, 0).hex()
    else:
        assert all(kind == "mismatch" for _, kind, _ in res)   # every rank sees it and stops
        if world > 2:
            assert all("rank(s) [%d]" % odd in msg for _, _, msg in res)

"""-m gpu: the bench's own jobs at BASELINE.json's full sizes, one step each, checked against the oracle — and the N > 1 code path on real RCCL.

bench.py runs in a subprocess (it brings torch, which must not share a process with the ctypes-loaded library of the other tests, tests/gpu_common.py).
Every run prints one JSON line; `oracle_check` is bench.check_against_oracle: two multi-frames (every camera: keypoint records, descriptors, masks,
counts) and the match indices of >= 2 (frame, keyframe) pairs — reference semantics src/cORBmatcher.cpp:179-323 (SearchByBoW(KF,F), vocabulary
restriction removed) and :885-966 (SearchByBoW(KF,KF)) — bit for bit.  A failed check also makes bench.py exit non-zero.

  configs[2]  3 cameras x 1000 features, 64 multi-frames x 32 stored keyframes (2048 set pairs of 3000 x 3000 rows on the matrix-core matcher)
  configs[3]  6 cameras 1280x800 x 2000 features, 4 multi-frames x 32 stored keyframes
  configs[4]  8 cameras 1280x800 x 2000 features x 8 stored keyframes (of the 256; 16 000 x 16 000 rows per pair)
  nccl1       the step of the N > 1 runs — separate send buffer, asynchronous all_gather_into_tensor on RCCL's stream, work.wait(), row flags, three
              buffer sets in rotation, matching one step late — at world size 1 over the nccl backend, stream and database workloads
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", *args], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, "bench.py %s failed (rc %d):\n%s\n%s" % (" ".join(args), p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, p.stdout[-2000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("name,args,min_pairs", [
    ("configs[2]", ["--workload", "db", "--steps", "1", "--warmup", "1"], 4),
    ("configs[3]", ["--workload", "rig", "--steps", "1", "--warmup", "1"], 4),
    ("configs[4]", ["--workload", "rig8", "--keyframes", "8", "--steps", "1", "--warmup", "1"], 2),
])
def test_full_size_workloads_against_the_oracle(name, args, min_pairs):
    out = run_bench(*args)
    cfg = out["config"]
    assert name in cfg["workload"]
    assert out["oracle_check"] is True
    assert cfg["oracle_checked"]["pairs"] >= min_pairs and cfg["oracle_checked"]["images"] >= 6
    assert cfg["matches_per_step_rank0"] > 0 and cfg["pair_distances_per_step"] > 1e9
    assert out["roofline"]["matcher"]["kernel"] == "k_match_mfma"
    tie = cfg["min_distance_to_a_rounding_tie_px_rank0"]   # the libm assumption as a checked invariant (None: the batch had no exact-pass keypoint)
    assert tie is None or tie > 1e-10


@pytest.mark.parametrize("args", [[], ["--workload", "db", "--frames", "8"]])
def test_exchange_path_on_rccl_at_world_size_1(args):
    out = run_bench("--exchange", "nccl1", "--steps", "4", "--warmup", "2", *args)
    cfg = out["config"]
    assert cfg["collective_backend"] == "nccl" and cfg["n_ranks"] == 1
    assert ("1 all-gather" if args else "point-to-point exchange") in cfg["parallelism"]
    assert out["oracle_check"] is True and cfg["oracle_checked"]["pairs"] >= 1
    assert cfg["matches_per_step_rank0"] > 0


def test_default_run_checks_every_leg():
    """the driver's command line (plus a short CPU sample): every leg carries oracle_check true, the world-1 RCCL leg ran"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--cpu-frames", "8"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["oracle_check"] is True and out["e2e"]["oracle_check"] is True
    assert all(s["oracle_check"] is True for s in out["secondary"]) and out["secondary"][0]["e2e"]["oracle_check"] is True
    assert out["exchange_world1"].get("oracle_check") is True, out["exchange_world1"]
    assert out["cpu_baseline"]["reference_threading"]["cores"] == 3
    # bit-identical as a per-run statement: no cvRound argument of the exact arithmetic (mdBRIEF fallbacks; every ORB coordinate of the shipped-settings leg)
    # came within 1e-10 px of a tie — ocml and glibc could only disagree within ~1e-13
    ties = [out["config"]["min_distance_to_a_rounding_tie_px_rank0"]] + [s["config"]["min_distance_to_a_rounding_tie_px_rank0"] for s in out["secondary"]]
    assert all(t is None or t > 1e-10 for t in ties) and ties[2] is not None, ties
    for k in ("roofline", "cpu_baseline"):
        assert out[k]


@pytest.mark.parametrize("n,args", [(2, []), (3, ["--workload", "db", "--frames", "4"])])
def test_gpus_n_starts_n_ranks_by_itself(n, args):
    """`python bench.py --gpus N` with no launcher around it starts N ranks (here on one shared GPU over gloo: a functional run of the N > 1 step) and rank 0
    prints ONE line with n_gpus = N; the split is the reference's per-camera one (src/cMultiFrame.cpp:128-164) over ranks"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env["MCS_BENCH_SHARE_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-secondary", *args],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == n and cfg["n_ranks"] == n and cfg["collective_backend"] == "gloo" and cfg["ranks_share_one_gpu"] is True
    assert out["oracle_check"] is True and cfg["oracle_checked"]["pairs"] >= 1
    assert cfg["exchange_bytes_received_per_rank_per_step"] > 0
    assert 0 < cfg["ms_per_step_fastest_rank"] <= cfg["ms_per_step_slowest_rank"] == out["ms_per_step"]


def test_gpus_n_without_n_gpus_is_refused():
    """more ranks asked for than GPUs visible (and no MCS_BENCH_SHARE_GPU): non-zero exit, no JSON line"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MCS_BENCH_SHARE_GPU"):
        env.pop(k, None)
    n = 9   # no node of this pool has more than 8
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "GPU(s) visible" in p.stderr

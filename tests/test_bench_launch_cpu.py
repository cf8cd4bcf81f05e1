"""bench.py's launch contract, the part that needs no GPU: `--gpus N` must mean N ranks on N GPUs or a non-zero exit — never a line that says n_gpus = 1
for a run that was asked for N (the reference's own split for N workers: /root/reference/src/cMultiFrame.cpp:128-164)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, **envkw):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MCS_BENCH_SHARE_GPU"):
        env.pop(k, None)
    env.update(envkw)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=300)


def test_more_ranks_than_gpus_is_refused_before_any_rank_starts():
    p = run(["--gpus", "9", "--steps", "1"])
    assert p.returncode != 0 and "GPU(s) visible" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_world_size_that_is_not_gpus_is_refused():
    p = run(["--gpus", "1", "--steps", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert p.returncode != 0 and "WORLD_SIZE is 2" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]

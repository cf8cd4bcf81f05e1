#!/bin/bash
# round 5, GPU call 5: the driver's bench command with the new legs (latency, CPU baseline per SURVEY 8d), bench-job tests (N > 1 path on one GPU), rig_host
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05c5; O=gpurun_out/r05c5
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; tail -3 $O/bench_default.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05c5/bench_default.json") if l.startswith("{")][-1])
print("value", d["value"], d["ms_per_step"], "check", d.get("oracle_check"))
print("latency", json.dumps(d.get("latency"))[:1800])
print("cpu", json.dumps(d.get("cpu_baseline"))[:1500])
print("e2e", d.get("e2e", {}).get("value"), d.get("e2e", {}).get("ms_per_step"))
print("roofline", json.dumps(d.get("roofline"))[:600])
PY
timeout 600 python -m pytest tests/test_gpu_bench_jobs.py tests/test_gpu_rig_host.py tests/test_gpu_rig.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.txt
multicol-slam_amd/host/rig_host /dev/null --gpus 4 ; echo "rig_host refuse rc $?"

#!/bin/bash
# usage: ab_latency.sh lib1 lib2 ...   — the native one-multi-frame-per-call latency (host/frame_latency through bench.py's latency leg) per library variant
for g in "$@"; do
  MCS_HIP_LIB=$PWD/gpurun_ab/libmcs_hip_$g.so LD_LIBRARY_PATH=$PWD/gpurun_ab/$g:$LD_LIBRARY_PATH timeout 600 python - "$g" <<'PY'
import sys, os, json
sys.argv = ["bench.py", "--no-cpu-baseline"]
sys.path.insert(0, os.getcwd())
import bench
args = bench.parse(["--no-cpu-baseline"])
e = bench.setup(args)
sp = bench.Spec(args, 1)
r = bench.run_latency(e, sp, calls=300, py_calls=40)
n = r.get("native", {})
print("%-8s" % sys.argv[0] if False else "", os.environ["MCS_HIP_LIB"].split("_")[-1], "median", r.get("median_ms"), "p99", r.get("p99_ms"), "extract", n.get("extract_ms", {}).get("median"), "match", n.get("match_ms", {}).get("median"), "match p90", n.get("match_ms", {}).get("p90"), "check", r.get("oracle_check"), "rescans_last", n.get("rescans_last"), "matches_last", n.get("matches_last"))
PY
done

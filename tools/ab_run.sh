#!/bin/bash
# usage: ab_run.sh "<bench args>" lib1 lib2 ...
args=$1; shift
for g in "$@"; do
  lib=$PWD/gpurun_ab/libmcs_hip_$g.so
  MCS_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-secondary $args > /tmp/ab.json 2>/tmp/ab.err
  python - "$g" <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
    m = d["roofline"].get("matcher", {})
    print("%-8s" % sys.argv[1], d["value"], "Mfeat/s", d["ms_per_step"], "ms/step", {k: v for k, v in d["roofline"]["per_kernel_ms"].items() if k in ("pyramid", "blur", "describe_fast", "fast", "match", "greedy")}, "Tpairs/s", m.get("Tpairs_per_s"), "check", d.get("oracle_check"), "matches", d["config"]["matches_per_step_rank0"])
except Exception as e:
    print(sys.argv[1], "failed", e, open("/tmp/ab.err").read()[-400:])
PY
done

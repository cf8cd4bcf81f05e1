#!/bin/bash
set -u
o=gpurun_out/c13; mkdir -p $o
timeout 500 python -m pytest tests -m gpu -x -q > $o/gputests.log 2>&1; echo "gputests rc=$?"; tail -3 $o/gputests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
for f in ("gpurun_out/c13/bench_default.json",):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("ms_per_step"), "e2e", d.get("e2e",{}).get("ms_per_step"), d.get("e2e",{}).get("value"), d.get("oracle_check"), d["roofline"].get("frac"), d["roofline"].get("per_kernel_ms"))
        for s in d.get("secondary",[]): print("  sec", s["config"]["workload"][:40], s["value"], s["ms_per_step"], s.get("e2e",{}).get("ms_per_step"))
        print("  x1", d.get("exchange_world1"))
        print("  cpu", d.get("cpu_baseline",{}).get("value"), d.get("speedup_vs_cpu_all_cores"))
    except Exception as ex: print(f, "ERR", ex, open(f.replace(".json",".err")).read()[-800:])
P

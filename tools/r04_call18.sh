#!/bin/bash
set -u
o=gpurun_out/c18; mkdir -p $o
timeout 300 python -m pytest tests/test_gpu_copy.py tests/test_gpu_bench_jobs.py -x -q > $o/tests.log 2>&1; echo "tests rc=$?"; tail -3 $o/tests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
for f in ("gpurun_out/c18/bench_default.json",):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("ms_per_step"), "e2e", d.get("e2e",{}).get("ms_per_step"), d.get("e2e",{}).get("value"), d.get("e2e",{}).get("upload_stream_queue_conflicts"), d.get("oracle_check"), d["roofline"].get("frac"))
        for s in d.get("secondary",[]): print("  sec", s["config"]["workload"][:40], s["value"], s["ms_per_step"], s.get("e2e",{}).get("ms_per_step"))
        print("  x1", d.get("exchange_world1",{}).get("ms_per_step"))
    except Exception as ex: print(f, "ERR", ex, open(f.replace(".json",".err")).read()[-800:])
P

#!/bin/bash
set -u
o=gpurun_out/c19; mkdir -p $o
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
for f in ("gpurun_out/c19/bench_default.json",):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("ms_per_step"), "e2e", d.get("e2e",{}).get("ms_per_step"), d.get("e2e",{}).get("value"), d.get("e2e",{}).get("upload_stream_queue_conflicts"), d.get("oracle_check"), d["roofline"].get("frac"))
        for s in d.get("secondary",[]): print("  sec", s["config"]["workload"][:40], s["value"], s["ms_per_step"], s.get("e2e",{}).get("ms_per_step"))
        print("  x1", d.get("exchange_world1",{}).get("ms_per_step"), "cpu", d["cpu_baseline"]["value"], d["speedup_vs_cpu_all_cores"])
    except Exception as ex: print(f, "ERR", ex, open(f.replace(".json",".err")).read()[-800:])
P
for q in 4 5 6 8; do
GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --exchange nccl1 > $o/x$q.json 2> $o/x$q.err
python - $q <<'P'
import json,sys
try:
    d=json.loads(open("gpurun_out/c19/x%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print("nccl1 queues", sys.argv[1], d["ms_per_step"], d["value"], d.get("oracle_check"))
except Exception as ex: print("ERR", ex, open("gpurun_out/c19/x%s.err"%sys.argv[1]).read()[-500:])
P
done

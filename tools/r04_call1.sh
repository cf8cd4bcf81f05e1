#!/bin/bash
# round-4 evidence call 1: GPU suite, the driver's bench command, the e2e copy A/B, kernel statistics + PMC passes of the default step
set -u
o=gpurun_out/c1; mkdir -p $o
timeout 500 python -m pytest tests -m gpu -x -q > $o/gputests.log 2>&1; echo "gputests rc=$?"; tail -3 $o/gputests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --e2e-sweep "runtime:runtime,16:16,8:8,32:32,64:64,16:runtime,runtime:16,4:4" > $o/e2e_sweep.json 2> $o/e2e_sweep.err; echo "sweep rc=$?"
python - <<'P'
import json
for f in ("gpurun_out/c1/bench_default.json","gpurun_out/c1/e2e_sweep.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("ms_per_step"), d.get("kernel_ms"), "e2e", d.get("e2e",{}).get("ms_per_step"), json.dumps(d.get("e2e_sweep")))
        for s in d.get("secondary",[]): print("  sec", s["config"]["workload"][:40], s["value"], s["ms_per_step"])
    except Exception as ex: print(f, "ERR", ex)
P
bash tools/profile_round.sh r04_stream

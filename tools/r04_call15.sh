#!/bin/bash
set -u
o=gpurun_out/c15; mkdir -p $o
for a in 0 1 2 3; do
MCS_E2E_AHEAD=$a MCS_E2E_DIAG=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check --e2e-sweep "runtime:2,runtime:2" > $o/r$a.json 2> $o/r$a.err
python - $a <<'P'
import json,sys
d=json.loads(open("gpurun_out/c15/r%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print("ahead", sys.argv[1], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["e2e_sweep"].items()})
P
grep "interval\|slowest" $o/r$a.err | cut -c1-300
done

#!/bin/bash
set -u
o=gpurun_out/c14; mkdir -p $o
for i in 1 2; do
MCS_E2E_DIAG=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check --e2e-sweep "runtime:2" > $o/r$i.json 2> $o/r$i.err
python - $i <<'P'
import json,sys
d=json.loads(open("gpurun_out/c14/r%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["e2e_sweep"].items()})
P
grep "e2e diag" $o/r$i.err | cut -c1-700
done

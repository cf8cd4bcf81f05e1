#!/bin/bash
set -u
o=gpurun_out/c23; mkdir -p $o
MCS_E2E_DIAG=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check --e2e-sweep "runtime:2" > $o/a.json 2> $o/a.err
grep "e2e diag: (index\|interval\|step starts" $o/a.err | cut -c1-1500
python - <<'P'
import json
d=json.loads(open("gpurun_out/c23/a.json").read().strip().splitlines()[-1]); print("sweep", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["e2e_sweep"].items()})
P

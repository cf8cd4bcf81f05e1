#!/bin/bash
set -u
o=gpurun_out/c20; mkdir -p $o
MCS_E2E_DIAG=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --e2e-sweep "runtime:2,runtime:2" > $o/a.json 2> $o/a.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/c20/a.json").read().strip().splitlines()[-1]); print("with check", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["e2e_sweep"].items()})
P
grep "e2e diag" $o/a.err | cut -c1-400
MCS_E2E_DIAG=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check --e2e-sweep "runtime:2,runtime:2" > $o/b.json 2> $o/b.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/c20/b.json").read().strip().splitlines()[-1]); print("no check", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["e2e_sweep"].items()})
P
grep "e2e diag" $o/b.err | cut -c1-400

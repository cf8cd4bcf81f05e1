#!/bin/bash
set -u
o=gpurun_out/c21; mkdir -p $o
show() { python - $1 <<'P'
import json,sys
d=json.loads(open("gpurun_out/c21/%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], "e2e", d.get("e2e",{}).get("ms_per_step"), [ (s["ms_per_step"], s.get("e2e",{}).get("ms_per_step")) for s in d.get("secondary",[])])
P
grep "e2e diag: GPU\|slowest" $o/$1.err | cut -c1-300; }
MCS_E2E_DIAG=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $o/nocpu.json 2> $o/nocpu.err; show nocpu
MCS_E2E_DIAG=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/full.json 2> $o/full.err; show full
MCS_E2E_DIAG=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --e2e-sweep "runtime:2" > $o/sweep.json 2> $o/sweep.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/c21/sweep.json").read().strip().splitlines()[-1]); print("sweep", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["e2e_sweep"].items()})
P
grep "e2e diag: GPU\|slowest" $o/sweep.err | cut -c1-300

#!/bin/bash
set -u
o=gpurun_out/c16; mkdir -p $o
for st in null plain; do
MCS_E2E_STREAMS=$st MCS_E2E_DIAG=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check --e2e-sweep "runtime:2,runtime:2,runtime:2,runtime:2" > $o/r$st.json 2> $o/r$st.err
python - $st <<'P'
import json,sys
d=json.loads(open("gpurun_out/c16/r%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print("streams", sys.argv[1], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["e2e_sweep"].items()})
P
grep "interval" $o/r$st.err | cut -c1-300
done

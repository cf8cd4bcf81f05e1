#!/bin/bash
set -u
o=gpurun_out/c24; mkdir -p $o
for m in 0 1 2; do for i in 1 2; do
MCS_E2E_MARK=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check --e2e-sweep "runtime:2,runtime:2" > $o/m${m}_$i.json 2> $o/m${m}_$i.err
python - $m $i <<'P'
import json,sys
d=json.loads(open("gpurun_out/c24/m%s_%s.json"%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1]); print("mark", sys.argv[1], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["e2e_sweep"].items()})
P
done; done

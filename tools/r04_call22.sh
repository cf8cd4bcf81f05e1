#!/bin/bash
set -u
o=gpurun_out/c22; mkdir -p $o
show() { python - $1 <<'P'
import json,sys
d=json.loads(open("gpurun_out/c22/%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], "e2e", d.get("e2e",{}).get("ms_per_step"), [ (s["ms_per_step"], s.get("e2e",{}).get("ms_per_step")) for s in d.get("secondary",[])], "x1", d.get("exchange_world1",{}).get("ms_per_step"))
P
}
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $o/r$i.json 2> $o/r$i.err; show r$i; done

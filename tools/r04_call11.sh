#!/bin/bash
set -u
o=gpurun_out/c11; mkdir -p $o
run() { tag=$1; cfg=$2; shift 2; env "$@" timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-check --e2e-sweep "$cfg" > $o/$tag.json 2> $o/$tag.err
python - "$tag" <<'P'
import json,sys
try:
    d=json.loads(open("gpurun_out/c11/%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print("%-18s"%sys.argv[1], d["ms_per_step"], {k:(v["ms_per_step"],v["d2h_GBps_alone"]) for k,v in d["e2e_sweep"].items()})
except Exception as ex: print(sys.argv[1],"ERR", ex, open("gpurun_out/c11/%s.err"%sys.argv[1]).read()[-600:])
P
}
for wg in 1 2 4 8 16; do for pace in 0 4 16 64; do run d_${wg}_$pace off:$wg MCS_E2E_STREAMS=plain MCS_COPY_PACE=$pace; done; done

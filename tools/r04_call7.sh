#!/bin/bash
set -u
o=gpurun_out/c7; mkdir -p $o
run() { tag=$1; cfg=$2; shift 2; env "$@" timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-check --e2e-sweep "$cfg" > $o/$tag.json 2> $o/$tag.err
python - "$tag" <<'P'
import json,sys
try:
    d=json.loads(open("gpurun_out/c7/%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print("%-14s"%sys.argv[1], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["e2e_sweep"].items()})
except Exception as ex: print(sys.argv[1],"ERR", ex, open("gpurun_out/c7/%s.err"%sys.argv[1]).read()[-400:])
P
}
# first leg of a process = the "good" pairing of plain streams; one configuration per process
for c in off:off runtime:off off:32 runtime:32 off:runtime; do run plain_$c $c MCS_E2E_STREAMS=plain; done
for c in off:off runtime:off off:32 runtime:32; do run q8_$c $c MCS_E2E_STREAMS=plain GPU_MAX_HW_QUEUES=8; done

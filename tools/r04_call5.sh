#!/bin/bash
set -u
o=gpurun_out/c5; mkdir -p $o
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-check --e2e-sweep "runtime:32,runtime:64,runtime:32,runtime:runtime" > $o/$tag.json 2> $o/$tag.err
python - "$tag" <<'P'
import json,sys
try:
    d=json.loads(open("gpurun_out/c5/%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print("%-14s"%sys.argv[1], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["e2e_sweep"].items()})
except Exception as ex: print(sys.argv[1],"ERR", ex, open("gpurun_out/c5/%s.err"%sys.argv[1]).read()[-400:])
P
}
run plain_ni2 MCS_E2E_STREAMS=plain MCS_E2E_IMAGE_BUFFERS=2
run plain_ni3 MCS_E2E_STREAMS=plain MCS_E2E_IMAGE_BUFFERS=3
run plain_ni3b MCS_E2E_STREAMS=plain MCS_E2E_IMAGE_BUFFERS=3
run p0_1 MCS_E2E_STREAMS=prio:0:1 MCS_E2E_IMAGE_BUFFERS=3
run pm1_0 MCS_E2E_STREAMS=prio:-1:0 MCS_E2E_IMAGE_BUFFERS=3
run p0_m1 MCS_E2E_STREAMS=prio:0:-1 MCS_E2E_IMAGE_BUFFERS=3
run pm1_m1 MCS_E2E_STREAMS=prio:-1:-1 MCS_E2E_IMAGE_BUFFERS=3
run ctx_ni3 MCS_E2E_STREAMS=ctx MCS_E2E_IMAGE_BUFFERS=3
run q8_plain_ni3 MCS_E2E_STREAMS=plain MCS_E2E_IMAGE_BUFFERS=3 GPU_MAX_HW_QUEUES=8
run q8_p0_1 MCS_E2E_STREAMS=prio:0:1 MCS_E2E_IMAGE_BUFFERS=3 GPU_MAX_HW_QUEUES=8

#!/bin/bash
set -u
o=gpurun_out/c8; mkdir -p $o
run() { tag=$1; cfg=$2; shift 2; env "$@" timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --e2e-sweep "$cfg" > $o/$tag.json 2> $o/$tag.err
python - "$tag" <<'P'
import json,sys
try:
    d=json.loads(open("gpurun_out/c8/%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print("%-18s"%sys.argv[1], d["ms_per_step"], {k:(v["ms_per_step"],v["oracle_check"]) for k,v in d["e2e_sweep"].items()})
except Exception as ex: print(sys.argv[1],"ERR", ex, open("gpurun_out/c8/%s.err"%sys.argv[1]).read()[-600:])
P
}
run res_rt32 runtime:32 MCS_E2E_STREAMS=plain
run res_rt16 runtime:16 MCS_E2E_STREAMS=plain
run res_rt64 runtime:64 MCS_E2E_STREAMS=plain
run res_rt8 runtime:8 MCS_E2E_STREAMS=plain
run res_rtrt runtime:runtime MCS_E2E_STREAMS=plain
run res_multi "runtime:32,runtime:32,runtime:32,runtime:16" MCS_E2E_STREAMS=plain
run res_16_32 16:32 MCS_E2E_STREAMS=plain
run old_rt32 runtime:32 MCS_E2E_STREAMS=plain MCS_E2E_OUT=stream
run res_ctxin runtime:32 MCS_E2E_STREAMS=ctx
run q5_res_rt32 runtime:32 MCS_E2E_STREAMS=plain GPU_MAX_HW_QUEUES=5
run q6_res_rt32 runtime:32 MCS_E2E_STREAMS=plain GPU_MAX_HW_QUEUES=6
run q5_offoff off:off MCS_E2E_STREAMS=plain GPU_MAX_HW_QUEUES=5

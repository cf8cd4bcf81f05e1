#!/bin/bash
set -u
o=gpurun_out/c12; mkdir -p $o
run() { tag=$1; cfg=$2; shift 2; env "$@" timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-check --e2e-sweep "$cfg" > $o/$tag.json 2> $o/$tag.err
python - "$tag" <<'P'
import json,sys
try:
    d=json.loads(open("gpurun_out/c12/%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print("%-18s"%sys.argv[1], d["ms_per_step"], {k:(v["ms_per_step"],v["h2d_GBps_alone"],v["d2h_GBps_alone"]) for k,v in d["e2e_sweep"].items()})
except Exception as ex: print(sys.argv[1],"ERR", ex, open("gpurun_out/c12/%s.err"%sys.argv[1]).read()[-600:])
P
}
run rt_2 runtime:2 MCS_E2E_STREAMS=plain
run rt_2b runtime:2 MCS_E2E_STREAMS=plain MCS_E2E_IMAGE_BUFFERS=3
run rt_off runtime:off MCS_E2E_STREAMS=plain
for wg in 2 4 8 16 32; do run h_${wg} $wg:off MCS_E2E_STREAMS=plain; done
for wg in 4 8 16; do run h_${wg}_2 $wg:2 MCS_E2E_STREAMS=plain; done
run rt_2_old runtime:2 MCS_E2E_STREAMS=plain MCS_E2E_OUT=stream
run rt_2_multi "runtime:2,runtime:2,runtime:2" MCS_E2E_STREAMS=plain

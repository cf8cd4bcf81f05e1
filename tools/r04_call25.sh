#!/bin/bash
set -u
o=gpurun_out/c25; mkdir -p $o
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --exchange nccl1 > $o/$tag.json 2> $o/$tag.err
python - $tag <<'P'
import json,sys
try:
    d=json.loads(open("gpurun_out/c25/%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print("%-16s"%sys.argv[1], d["ms_per_step"], d["value"], d.get("oracle_check"))
except Exception as ex: print("ERR", sys.argv[1], ex, open("gpurun_out/c25/%s.err"%sys.argv[1]).read()[-500:])
P
}
run base
run ch1 NCCL_MAX_NCHANNELS=1
run ch2 NCCL_MAX_NCHANNELS=2
run ch4 NCCL_MAX_NCHANNELS=4
run q8 GPU_MAX_HW_QUEUES=8
run q8ch1 GPU_MAX_HW_QUEUES=8 NCCL_MAX_NCHANNELS=1
run q8ch2 GPU_MAX_HW_QUEUES=8 NCCL_MAX_NCHANNELS=2


#!/bin/bash
# A/B of bench.py's host-buffer leg, on the GPU box from the repo root: tools/ab_e2e.sh <tag> "<h2d:d2h>[,<h2d:d2h>...]" [ENV=value ...]
#   h2d / d2h: "runtime" (hipMemcpyAsync), "off" (no copy: the leg's event structure alone; use with --no-check), or a workgroup count for mcs_copy_narrow
#   ENV: MCS_E2E_STREAMS=probed|plain|null|prio:<in>:<out>, MCS_E2E_OUT=result|stream, MCS_E2E_IMAGE_BUFFERS, MCS_E2E_AHEAD, MCS_COPY_PACE, GPU_MAX_HW_QUEUES, MCS_E2E_DIAG=1
# One configuration per process where the stream pairing matters (plain streams pair differently from one leg to the next).
# The round-4 matrix: for w in 1 2 4 8 16 32; do tools/ab_e2e.sh d$w off:$w MCS_E2E_STREAMS=plain; done      (copy-out workgroups)
#                     tools/ab_e2e.sh h runtime:off; tools/ab_e2e.sh k8 8:off                                    (copy-in: SDMA against a reading kernel)
#                     tools/ab_e2e.sh pair "runtime:2,runtime:2,runtime:2,runtime:2" MCS_E2E_STREAMS=plain MCS_E2E_DIAG=1   (1.6 / 3.0 ms alternating with the queue pairing)
tag=$1; cfg=$2; shift 2
mkdir -p gpurun_out/ab_e2e
env "$@" timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-check --e2e-sweep "$cfg" > gpurun_out/ab_e2e/$tag.json 2> gpurun_out/ab_e2e/$tag.err
python - "$tag" <<'P'
import json, sys
try:
    d = json.loads(open("gpurun_out/ab_e2e/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print("%-16s device-resident %.4f ms |" % (sys.argv[1], d["ms_per_step"]), {k: (v["ms_per_step"], v["h2d_GBps_alone"], v["d2h_GBps_alone"]) for k, v in d["e2e_sweep"].items()})
except Exception as ex:
    print(sys.argv[1], "ERR", ex, open("gpurun_out/ab_e2e/%s.err" % sys.argv[1]).read()[-500:])
P
grep "e2e diag" gpurun_out/ab_e2e/$tag.err | cut -c1-400

#!/bin/bash
# library variants (tools/ab_describe.sh build) on the default step and on configs[2]: tools/ab_bench2.sh variant...   (unknown name = the tree's library)
cd "$GRAFT_REPO_ROOT"
for g in "$@"; do
  lib=$PWD/gpurun_ab/libmcs_hip_$g.so; [ -f $lib ] || lib=$PWD/multicol-slam_amd/libmcs_hip.so
  for w in stream db; do
    MCS_HIP_LIB=$lib timeout 300 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['per_kernel_ms']
print('%-8s %-6s step %.3f ms  match %.3f greedy %.3f  check %s' % ('$g', '$w', d['ms_per_step'], k['match'], k['greedy'], d['oracle_check']))"
  done
done

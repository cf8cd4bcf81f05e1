#!/bin/bash
# A/B of k_describe build-time variants: `build name "-DFLAG=.. -DFLAG2=.."` compiles a libmcs_hip variant into gpurun_ab/ (hipcc cross-compiles here),
# `run name...` benches each variant on the GPU box.
mode=$1; shift
cd "$(dirname "$0")/.."
if [ "$mode" = build ]; then
  name=$1; flags=$2
  d=/tmp/mcs_ab_$name; mkdir -p $d/multicol-slam_amd/csrc $d/include gpurun_ab
  cp multicol-slam_amd/csrc/*.hip multicol-slam_amd/csrc/*.h multicol-slam_amd/csrc/*.inc multicol-slam_amd/csrc/Makefile $d/multicol-slam_amd/csrc/
  cp -r include/* $d/include/
  make -s -j8 -C $d/multicol-slam_amd/csrc FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math $flags" 2>&1 | grep -i "error"
  cp $d/multicol-slam_amd/libmcs_hip.so gpurun_ab/libmcs_hip_$name.so
else
  for g in "$@"; do
    lib=$PWD/gpurun_ab/libmcs_hip_$g.so; [ -f $lib ] || lib=$PWD/multicol-slam_amd/libmcs_hip.so   # an unknown name benches the tree's own library (with whatever environment the caller set)
    MCS_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 30 > /tmp/ab.json 2>/tmp/ab.err
    python - "$g" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/ab.json"))
    print("%-12s" % sys.argv[1], d["value"], "Mfeat/s", d["ms_per_step"], "ms/step", d["roofline"]["per_kernel_ms"], "matches", d["config"]["matches_per_step_rank0"])
except Exception as e:
    print(sys.argv[1], "failed", e, open("/tmp/ab.err").read()[-300:])
PY
  done
fi

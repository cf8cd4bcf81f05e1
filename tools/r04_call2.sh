#!/bin/bash
# round-4 call 2: describe-walk variants (A/B), e2e copy mechanisms continued, e2e timeline
set -u
o=gpurun_out/c2; mkdir -p $o
timeout 300 python -m pytest tests/test_gpu_extract.py tests/test_gpu_describe_guard.py tests/test_gpu_many_cameras.py tests/test_gpu_fullsize.py -x -q > $o/tests_tree.log 2>&1; echo "tests(tree) rc=$?"; tail -3 $o/tests_tree.log
for v in dma8t; do
  MCS_HIP_LIB=$PWD/gpurun_ab/libmcs_hip_$v.so timeout 300 python -m pytest tests/test_gpu_extract.py tests/test_gpu_describe_guard.py -x -q > $o/tests_$v.log 2>&1; echo "tests($v) rc=$?"; tail -2 $o/tests_$v.log
done
bash tools/ab_describe.sh run tree nodma8 dma8t nodma16 tree nodma8 2>&1 | tee $o/ab_describe.txt
bash tools/ab_kstats.sh "describe|octree|fast_cells" tree nodma8 dma8t nodma16 2>&1 | tee $o/ab_kstats.txt
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --e2e-sweep "runtime:16,runtime:8,runtime:32,runtime:4,runtime:64,runtime:16" > $o/e2e_sweep.json 2> $o/e2e_sweep.err; echo "sweep rc=$?"
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --e2e-sweep "runtime:16,16:16,runtime:runtime" > $o/e2e_sweep_q8.json 2> $o/e2e_sweep_q8.err; echo "sweep q8 rc=$?"
python - <<'P'
import json
for f in ("gpurun_out/c2/e2e_sweep.json","gpurun_out/c2/e2e_sweep_q8.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("ms_per_step"))
        for k,v in d["e2e_sweep"].items(): print("   ", k, v)
    except Exception as ex: print(f, "ERR", ex)
P
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/kt_e2e
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kt_e2e -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-check --e2e-sweep runtime:16 > /tmp/kt_e2e.json 2> /tmp/kt_e2e.err
for f in $(find /tmp/kt_e2e -name "*.csv"); do cp $f $o/e2e_$(basename $f | sed 's/^[0-9]*_//'); done
ls -la $o

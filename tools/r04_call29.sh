#!/bin/bash
set -u
o=gpurun_out/c29; mkdir -p $o
timeout 300 python -m pytest tests/test_gpu_prefetch.py tests/test_gpu_extract.py -x -q > $o/tests.log 2>&1; echo "tests rc=$?"; tail -5 $o/tests.log
for pf in 1 0 1 0; do
MCS_BENCH_PREFETCH=$pf timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-secondary > $o/p$pf.json 2> $o/p$pf.err
python - $pf <<'P'
import json,sys
try:
    d=json.loads(open("gpurun_out/c29/p%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print("prefetch", sys.argv[1], d["ms_per_step"], d["value"], d.get("oracle_check"))
except Exception as ex: print("ERR", ex, open("gpurun_out/c29/p%s.err"%sys.argv[1]).read()[-600:])
P
done

#!/bin/bash
# round-4 call 3: describe instruction diet + DMA through inline asm; e2e with the context's copy streams
set -u
o=gpurun_out/c3; mkdir -p $o
timeout 300 python -m pytest tests/test_gpu_extract.py tests/test_gpu_describe_guard.py tests/test_gpu_many_cameras.py tests/test_gpu_fullsize.py tests/test_gpu_copy.py -x -q > $o/tests_tree.log 2>&1; echo "tests(tree) rc=$?"; tail -3 $o/tests_tree.log
bash tools/ab_describe.sh run tree clamp nodma8 tree clamp 2>&1 | tee $o/ab_describe.txt
bash tools/ab_kstats.sh "describe|octree|fast_cells" tree clamp nodma8 2>&1 | tee $o/ab_kstats.txt
for st in ctx plain; do
MCS_E2E_STREAMS=$st timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --e2e-sweep "runtime:32,runtime:16,runtime:64,runtime:32,runtime:runtime,16:32" > $o/e2e_sweep_$st.json 2> $o/e2e_sweep_$st.err; echo "sweep $st rc=$?"
done
python - <<'P'
import json
for f in ("gpurun_out/c3/e2e_sweep_ctx.json","gpurun_out/c3/e2e_sweep_plain.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("ms_per_step"))
        for k,v in d["e2e_sweep"].items(): print("   ", k, v)
    except Exception as ex: print(f, "ERR", ex, open(f.replace(".json",".err")).read()[-600:])
P

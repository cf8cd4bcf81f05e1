cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_fullsize.py tests/test_gpu_rig.py tests/test_gpu_describe_guard.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -6 > $O/t.log
timeout 600 python bench.py --steps 30 --warmup 5 --check --no-cpu-baseline > $O/b_main.json 2> $O/b_main.err
MCS_NO_OVERLAP=1 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/b_noov.json 2> $O/b_noov.err
cat $O/t.log; python - <<'PY'
import json
for f in ("b_main","b_noov"):
    try:
        d=json.load(open("gpurun_out/r2e/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["per_kernel_ms"], d.get("oracle_check"), d.get("e2e",{}).get("ms_per_step"))
    except Exception as ex: print(f, "failed", ex, open("gpurun_out/r2e/%s.err"%f).read()[-800:])
PY

cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
timeout 600 python bench.py --steps 30 --warmup 5 --check --no-cpu-baseline > $O/b_main.json 2> $O/b_main.err
timeout 600 python bench.py --steps 20 --warmup 5 --frames 128 --no-cpu-baseline --no-secondary > $O/b_f128.json 2> $O/b_f128.err
timeout 600 python bench.py --steps 30 --warmup 5 --frames 32 --no-cpu-baseline --no-secondary > $O/b_f32.json 2> $O/b_f32.err
timeout 600 python bench.py --steps 30 --warmup 5 --topk 16 --no-cpu-baseline --no-secondary > $O/b_k16.json 2> $O/b_k16.err
python - <<'PY'
import json
for f in ("b_main","b_f128","b_f32","b_k16"):
    try:
        d=json.load(open("gpurun_out/r2f/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["per_kernel_ms"], d.get("oracle_check"), d.get("e2e"))
    except Exception as ex: print(f, "failed", ex, open("gpurun_out/r2f/%s.err"%f).read()[-800:])
PY

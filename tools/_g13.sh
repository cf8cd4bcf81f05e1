cd $GRAFT_REPO_ROOT
O=gpurun_out/r2m; mkdir -p $O
MCS_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --frames 8 --check > $O/b_n2.json 2> $O/b_n2.err
MCS_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --workload rig --frames 2 > $O/b_n2rig.json 2> $O/b_n2rig.err
grep -o '"oracle_check": [a-z]*' $O/b_n2.json; grep -o '"value": [0-9.]*' $O/b_n2.json $O/b_n2rig.json | head -3; tail -n 3 $O/b_n2.err
bash tools/profile_round.sh r02_stream > $O/prof_stream.log 2>&1
bash tools/profile_round.sh r02_db --workload db > $O/prof_db.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 5 --check > $O/b_main.json 2> $O/b_main.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2m/b_main.json")); print(d["value"], d["ms_per_step"], d["roofline"]["per_kernel_ms"], d.get("oracle_check"), d["e2e"]["value"], d["e2e"]["ms_per_step"], [ (s["value"], s["ms_per_step"]) for s in d["secondary"]], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY

cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rig.py tests/test_gpu_dbsweep.py -x -q -m gpu 2>&1 | tail -15 > $O/t_rig.log
timeout 600 python bench.py --steps 20 --warmup 5 --check > $O/b_main.json 2> $O/b_main.err
timeout 300 python bench.py --workload db --steps 5 --warmup 2 --no-cpu-baseline > $O/b_db.json 2> $O/b_db.err
timeout 300 python bench.py --workload rig --steps 10 --warmup 2 > $O/b_rig.json 2> $O/b_rig.err
timeout 300 python bench.py --workload rig8 --steps 3 --warmup 1 > $O/b_rig8.json 2> $O/b_rig8.err
MCS_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --frames 8 --check > $O/b_n2.json 2> $O/b_n2.err
MCS_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --workload rig --frames 2 > $O/b_n2rig.json 2> $O/b_n2rig.err
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/t_all.log
tail -n 4 $O/t_rig.log $O/t_all.log; for f in b_main b_db b_rig b_rig8 b_n2 b_n2rig; do echo == $f; tail -c 600 $O/$f.err; head -c 1500 $O/$f.json; echo; done

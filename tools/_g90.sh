cd /root/repo
bash tools/profile_round.sh r02_stream > gpurun_out/prof_stream.log 2>&1
bash tools/profile_round.sh r02_db --workload db > gpurun_out/prof_db.log 2>&1
python bench.py --steps 50 --warmup 5 --check > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --workload db --check > gpurun_out/bench_db.json 2> gpurun_out/bench_db.err
python bench.py --workload rig --check > gpurun_out/bench_rig.json 2> gpurun_out/bench_rig.err
python bench.py --workload rig8 --check --steps 5 --warmup 1 > gpurun_out/bench_rig8.json 2> gpurun_out/bench_rig8.err

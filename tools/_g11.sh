cd $GRAFT_REPO_ROOT
O=gpurun_out/r2k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_fullsize.py tests/test_gpu_describe_guard.py tests/test_gpu_rig.py -x -q -m gpu 2>&1 | tail -8 > $O/t.log
cat $O/t.log
bash tools/ab_describe.sh run main > $O/ab.log 2>&1
MCS_OCTREE_SPLIT=0 bash tools/ab_describe.sh run nosplit >> $O/ab.log 2>&1
bash tools/ab_describe.sh run main >> $O/ab.log 2>&1
cat $O/ab.log

#!/bin/bash
# round 5, GPU call 1: the distinct-point descriptor pass — full suite, the general sampler path forced (MCS_PATCH_R=12 build), A/B of step and kernel times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05c1; O=gpurun_out/r05c1
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -3 $O/pytest.txt
MCS_HIP_LIB=$PWD/gpurun_ab/libmcs_hip_patch12.so timeout 300 python -m pytest tests/test_gpu_extract.py tests/test_gpu_describe_guard.py -m gpu -x -q > $O/pytest_patch12.txt 2>&1; echo "rc $?" >> $O/pytest_patch12.txt; tail -3 $O/pytest_patch12.txt
MCS_HIP_LIB=$PWD/gpurun_ab/libmcs_hip_w20.so timeout 300 python -m pytest tests/test_gpu_extract.py tests/test_gpu_describe_guard.py -m gpu -x -q > $O/pytest_w20.txt 2>&1; echo "rc $?" >> $O/pytest_w20.txt; tail -3 $O/pytest_w20.txt
tools/ab_describe.sh run base tree f8 w20 base tree > $O/ab_run.txt 2>&1; cat $O/ab_run.txt
tools/ab_kstats.sh "describe|fast_cells|octree|orient" base tree f8 w20 > $O/ab_kstats.txt 2>&1; cat $O/ab_kstats.txt

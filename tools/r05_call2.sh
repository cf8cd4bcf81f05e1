#!/bin/bash
# round 5, GPU call 2: tie enforcement tests, the one-block (v2) and hand-pipelined (tree) distinct-point forms against round 4's kernel (base), SQ counters
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05c2; O=gpurun_out/r05c2
timeout 300 python -m pytest tests/test_gpu_tiefix.py tests/test_gpu_describe_guard.py tests/test_gpu_extract.py tests/test_gpu_fullsize.py tests/test_gpu_many_cameras.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -15 $O/pytest.txt
tools/ab_describe.sh run base tree v2 v2f8 tree base > $O/ab_run.txt 2>&1; cat $O/ab_run.txt
tools/ab_kstats.sh "describe|orient" base tree v2 v2f8 > $O/ab_kstats.txt 2>&1; cat $O/ab_kstats.txt
tools/ab_pmc.sh "describe_fast" base tree v2f8 > $O/ab_pmc.txt 2>&1; cat $O/ab_pmc.txt

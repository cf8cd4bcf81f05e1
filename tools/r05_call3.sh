#!/bin/bash
# round 5, GPU call 3: tie enforcement + frame binding tests; packed vals; A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05c3; O=gpurun_out/r05c3
timeout 600 python -m pytest tests/test_gpu_tiefix.py tests/test_gpu_dropin.py tests/test_gpu_describe_guard.py tests/test_gpu_extract.py tests/test_gpu_copy.py -m gpu -x -q -s > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; grep -v "Feature Extraction\|^$" $O/pytest.txt | tail -25
tools/ab_describe.sh run tree p1 f2 tree > $O/ab_run.txt 2>&1; cat $O/ab_run.txt
tools/ab_kstats.sh "describe_fast|orient" tree p1 f2 > $O/ab_kstats.txt 2>&1; cat $O/ab_kstats.txt

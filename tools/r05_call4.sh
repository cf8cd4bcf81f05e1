#!/bin/bash
# round 5, GPU call 4: the whole suite on the round's library (distinct-point descriptor pass, tie enforcement, frame binding), a short bench, kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05c4; O=gpurun_out/r05c4
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; grep -v "Feature Extraction\|^$" $O/pytest.txt | tail -12
timeout 300 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q -s -k constructor 2>&1 | grep "cMultiFrame constructor\|passed\|failed" | tee $O/frame_ms.txt
tools/ab_describe.sh run tree tree > $O/ab_run.txt 2>&1; cat $O/ab_run.txt
tools/ab_kstats.sh "describe|orient|fast_cells|octree|blur|resize|match|greedy|expand" tree > $O/ab_kstats.txt 2>&1; cat $O/ab_kstats.txt

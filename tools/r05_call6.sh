#!/bin/bash
# round 5, GPU call 6: FAST pass 1 on sixteen pixels per thread against four (seg4): parity tests, step and kernel times, VALU instruction counts
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05c6; O=gpurun_out/r05c6
timeout 600 python -m pytest tests/test_gpu_extract.py tests/test_gpu_fast_types.py tests/test_gpu_fullsize.py tests/test_gpu_env_paths.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.txt
tools/ab_describe.sh run seg4 tree seg4 tree > $O/ab_run.txt 2>&1; cat $O/ab_run.txt
tools/ab_kstats.sh "fast_cells|octree" seg4 tree > $O/ab_kstats.txt 2>&1; cat $O/ab_kstats.txt
MCS_NO_OVERLAP=1 tools/ab_kstats.sh "fast_cells|octree|describe_fast|blur|resize|match_mfma|greedy" seg4 tree > $O/ab_kstats_noov.txt 2>&1; cat $O/ab_kstats_noov.txt
tools/ab_pmc.sh "fast_cells" seg4 tree > $O/ab_pmc.txt 2>&1; cat $O/ab_pmc.txt

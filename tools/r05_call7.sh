#!/bin/bash
# round 5, GPU call 7: FAST with one wave per cell (bs64) against two
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05c7; O=gpurun_out/r05c7
MCS_HIP_LIB=$PWD/gpurun_ab/libmcs_hip_bs64.so timeout 600 python -m pytest tests/test_gpu_extract.py tests/test_gpu_fast_types.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.txt
tools/ab_describe.sh run bs64 tree bs64 tree > $O/ab_run.txt 2>&1; cat $O/ab_run.txt
MCS_NO_OVERLAP=1 tools/ab_kstats.sh "fast_cells" bs64 tree > $O/ab_kstats_noov.txt 2>&1; cat $O/ab_kstats_noov.txt
tools/ab_pmc.sh "fast_cells" bs64 > $O/ab_pmc.txt 2>&1; cat $O/ab_pmc.txt

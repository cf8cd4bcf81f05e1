#!/bin/bash
set -u
for pf in 1 0; do export MCS_BENCH_PREFETCH=$pf; bash tools/ab_kstats.sh "describe_fast|octree|fast_cells|resize|blur|match_mfma" tree$pf; done
bash tools/ktrace.sh pf1 2>&1 | tail -1

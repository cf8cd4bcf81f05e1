#!/bin/bash
# Evidence run for profiles/rNN (on the GPU box, from the repo root): kernel statistics of one bench workload under rocprofv3, then separate
# --pmc passes (never combined with sys/hip/hsa traces).  Usage: tools/profile_round.sh <tag> [bench.py arguments ...]   -> gpurun_out/prof_<tag>/...
set -u
tag=${1:-x}; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-check "$@" > $out/bench_under_rocprof.json 2> $out/stats.err
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES"; do
  n=$(echo $set | cut -c1-14 | tr " " _)
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc_$n -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-check "$@" > $out/pmc_$n.log 2>&1
  python tools/pmc_traffic.py $(find $out/pmc_$n -name "*counter_collection.csv") >> $out/pmc_summary.txt
done
rm -rf $out/stats $out/pmc_*/   # keep the summaries only (gpurun_out is capped)
ls -la $out

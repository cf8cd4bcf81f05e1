#!/bin/bash
# kernel statistics of one bench run under rocprofv3 (on the GPU box, from the repo root): tools/kstats.sh <tag> [bench args]  -> gpurun_out/ks_<tag>.csv
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/ks_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-check "$@" > gpurun_out/ks_$tag.json 2> gpurun_out/ks_$tag.err
cp $(find /tmp/ks_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/ks_$tag.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/ks_$tag.csv")))
for r in rows[:14]:
    print("%-90s calls %5s avg %9.1f us  total %8.2f ms  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY

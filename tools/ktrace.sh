#!/bin/bash
# per-dispatch timeline of a short bench run (rocprofv3 --kernel-trace): tools/ktrace.sh <tag> [bench args] -> gpurun_out/ktrace_<tag>.csv
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/kt_$tag
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-check "$@" > /tmp/kt_$tag.json 2> /tmp/kt_$tag.err
cp $(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1) gpurun_out/ktrace_$tag.csv
wc -l gpurun_out/ktrace_$tag.csv

#!/bin/bash
# per-dispatch timeline (kernels + memory copies) of the e2e leg: tools/e2e_trace.sh <tag> <h2d:d2h> [env assignments...] -> gpurun_out/e2etrace_<tag>_{kernel,memory_copy}_trace.csv
tag=$1; cfg=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/kt_$tag
env "$@" timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kt_$tag -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-check --e2e-sweep $cfg > /tmp/kt_$tag.json 2> /tmp/kt_$tag.err
for f in $(find /tmp/kt_$tag -name "*trace.csv"); do cp $f gpurun_out/e2etrace_${tag}_$(basename $f | sed 's/^[0-9]*_//'); done
python - "$tag" <<'P'
import json,sys
try:
    d=json.loads(open("/tmp/kt_%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["e2e_sweep"])
except Exception as ex: print("ERR", ex)
P

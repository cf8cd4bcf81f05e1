#!/bin/bash
# VALU / LDS instruction counts and the duration of the kernels matching a regex, for library variants: tools/ab_valu.sh "<regex>" variant...
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for g in "$@"; do
  lib=$PWD/gpurun_ab/libmcs_hip_$g.so; [ -f $lib ] || lib=$PWD/multicol-slam_amd/libmcs_hip.so
  rm -rf /tmp/abv_$g
  MCS_NO_OVERLAP=1 MCS_HIP_LIB=$lib timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d /tmp/abv_$g -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-check $AB_ARGS > /tmp/abv_$g.json 2> /tmp/abv_$g.err
  python - "$g" "$pat" <<'PY'
import collections, csv, glob, re, sys
g, pat = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/abv_%s/**/*counter_collection.csv" % g, recursive=True)
if not f:
    print(g, "no counters", open("/tmp/abv_%s.err" % g).read()[-300:]); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = re.sub(r"^void mcs::|^mcs::|\(.*", "", r["Kernel_Name"])[:40]
    if re.search(pat, k):
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, v in agg.items():
    steps = 3   # warmup 1 + steps 2 (+ the serialized timing passes): normalise by dispatch count of the kernel per step instead
    print("%-6s %-28s dispatches %3d " % (g, k, n[k]), " ".join("%s=%.4g" % (c.replace("SQ_", ""), x) for c, x in v.items()))
PY
done

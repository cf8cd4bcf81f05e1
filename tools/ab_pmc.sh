#!/bin/bash
# SQ counters per dispatch of the kernels matching a regex, for library variants (tools/ab_describe.sh): tools/ab_pmc.sh "<regex>" variant...
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for g in "$@"; do
  lib=$PWD/gpurun_ab/libmcs_hip_$g.so; [ -f $lib ] || lib=$PWD/multicol-slam_amd/libmcs_hip.so
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
    rm -rf /tmp/abp_$g
    MCS_HIP_LIB=$lib timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/abp_$g -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-check $AB_ARGS > /tmp/abp_$g.json 2> /tmp/abp_$g.err
    python - "$g" "$pat" <<'PY'
import collections, csv, glob, re, sys
g, pat = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/abp_%s/**/*counter_collection.csv" % g, recursive=True)
if not f:
    print(g, "no counters", open("/tmp/abp_%s.err" % g).read()[-300:]); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f[0])):
    k = re.sub(r"^void mcs::|^mcs::|\(.*", "", r["Kernel_Name"])[:40]
    if re.search(pat, k):
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, v in agg.items():
    print("%-8s %-28s" % (g, k), " ".join("%s=%.4g" % (c.replace("SQ_", ""), x / cnt[k][c]) for c, x in v.items()))
PY
  done
done

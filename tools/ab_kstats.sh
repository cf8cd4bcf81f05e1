#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of library variants built by tools/ab_describe.sh: tools/ab_kstats.sh "<kernel name regex>" variant...
# (an unknown variant name = the tree's own library); extra bench arguments through AB_ARGS
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for g in "$@"; do
  lib=$PWD/gpurun_ab/libmcs_hip_$g.so; [ -f $lib ] || lib=$PWD/multicol-slam_amd/libmcs_hip.so
  rm -rf /tmp/abk_$g
  MCS_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$g -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-check $AB_ARGS > /tmp/abk_$g.json 2> /tmp/abk_$g.err
  python - "$g" "$pat" <<'PY'
import csv, glob, json, re, sys
g, pat = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open("/tmp/abk_%s.json" % g) if l.startswith("{")][-1])
    f = glob.glob("/tmp/abk_%s/**/*kernel_stats.csv" % g, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if re.search(pat, r["Name"])]
    print("%-10s step %.3f ms |" % (g, d["ms_per_step"]), " | ".join("%s %.1f us x%s" % (re.sub(r"^void mcs::|\(.*", "", r["Name"])[:28], float(r["AverageNs"]) / 1e3, r["Calls"]) for r in rows))
except Exception as e:
    print(g, "failed", e, open("/tmp/abk_%s.err" % g).read()[-400:])
PY
done

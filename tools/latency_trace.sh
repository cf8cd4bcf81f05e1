#!/bin/bash
# Per-dispatch timeline of the latency leg's native host (one 3-camera multi-frame per call): bench.py keeps the program's input directory (MCS_KEEP_LATENCY_DIR),
# then frame_latency runs 40 calls under rocprofv3 --kernel-trace --memory-copy-trace.  -> gpurun_out/latency_trace.txt (the last call's kernels and copies, us)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/latdir; MCS_KEEP_LATENCY_DIR=/tmp/latdir timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
sed -i 's/^calls .*/calls 40/' /tmp/latdir/cfg.txt
rm -rf /tmp/lattr; timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/lattr -- multicol-slam_amd/host/frame_latency /tmp/latdir/cfg.txt > /tmp/lat.json 2>/tmp/lat.err
python - <<'P' > gpurun_out/latency_trace.txt
import csv, glob
rows = []
for f in glob.glob("/tmp/lattr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob("/tmp/lattr/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
# the last call: from the last host-to-device image copy before the last k_greedy
idx = [i for i, r in enumerate(rows) if "k_resize_cols" in r[2]]
start = idx[-7]
while start > 0 and rows[start - 1][2].startswith("COPY") : start -= 1
t0 = rows[start][0]
for s, e, n in rows[start:]:
    print("%9.1f %9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
P
cat /tmp/lat.json | tail -1

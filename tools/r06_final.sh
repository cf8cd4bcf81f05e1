#!/bin/bash
# round-6 evidence set (run on the GPU box from the repo root): GPU suite, the driver's bench command (latency leg, CPU baseline per SURVEY 8d), the other workloads,
# kernel statistics + PMC passes of four workloads, the serialized statistics, the per-dispatch timeline, the counter calibration on the path's own access patterns
# (tools/pmc_patterns), the native hosts (rig_host, frame_latency), the frame binding's constructor time -> gpurun_out/r06/ (copied to profiles/r06/)
set -u
o=gpurun_out/r06; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/gputests.log 2>&1; echo "gputests rc=$?"; grep -v "Feature Extraction" $o/gputests.log | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> $o/bench_default.err | grep -v "Feature Extraction" > $o/bench_default.json; echo "bench default rc=${PIPESTATUS[0]}"
for w in db rig rig8; do timeout 300 python bench.py --workload $w --steps 10 --warmup 2 > $o/bench_$w.json 2> $o/bench_$w.err; echo "bench $w rc=$?"; done
timeout 120 python bench.py --dry-run > $o/bench_dry_run.json 2>&1; echo "dry run rc=$?"
bash tools/profile_round.sh stream > /dev/null 2>&1
bash tools/profile_round.sh db --workload db > /dev/null 2>&1
bash tools/profile_round.sh db16 --workload db --frames 16 > /dev/null 2>&1
bash tools/profile_round.sh orb --mode orb --nfeatures 400 > /dev/null 2>&1
for w in stream db db16 orb; do for f in kernel_stats.csv pmc_summary.txt bench_under_rocprof.json; do cp gpurun_out/prof_$w/$f $o/$(echo $f | sed "s/\./_$w./"); done; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/noov; MCS_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/noov -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-check > /tmp/noov.json 2>/tmp/noov.err
cp $(find /tmp/noov -name "*kernel_stats.csv" | head -1) $o/kernel_stats_stream_nooverlap.csv
bash tools/ktrace.sh stream > /dev/null 2>&1; cp gpurun_out/ktrace_stream.csv $o/
bash tools/pmc_calibrate.sh > /dev/null 2>&1; cp gpurun_out/pmc_calibration.txt $o/
python tools/rig_host_bench.py 64 0 40 16 32 10 > $o/rig_host.txt 2>&1
timeout 200 python tools/agast_time.py 2>/dev/null | tail -1 > $o/agast_time.json
bash tools/latency_trace.sh > /dev/null 2>&1; cp gpurun_out/latency_trace.txt $o/
timeout 300 python -m pytest tests/test_gpu_dropin.py -m gpu -q -s -k constructor 2>&1 | grep "cMultiFrame constructor\|passed\|failed" > $o/frame_binding.txt
python - <<'P'
import json
for w in ("default","db","rig","rig8"):
    try:
        d=json.loads(open("gpurun_out/r06/bench_%s.json"%w).read().strip().splitlines()[-1])
        print(w, d["value"], d["ms_per_step"], d.get("oracle_check"), "e2e", d.get("e2e",{}).get("ms_per_step"), "frac", d["roofline"].get("frac"), "lat", (d.get("latency") or {}).get("median_ms"))
    except Exception as ex: print(w, "ERR", ex)
P
cat $o/frame_binding.txt
ls $o | wc -l; du -sh $o

#!/bin/bash
# tools/pmc_calibrate.sh -> gpurun_out/pmc_calibration.txt: FETCH_SIZE / WRITE_SIZE per dispatch of a copy of known size (see tools/pmc_calibrate.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_calibration.txt; : > $out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/cal_$c -- python tools/pmc_calibrate.py >> $out 2>/tmp/cal_$c.err
  python tools/pmc_traffic.py $(find /tmp/cal_$c -name "*counter_collection.csv") >> $out
done
cat $out

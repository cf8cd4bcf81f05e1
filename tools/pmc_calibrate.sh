#!/bin/bash
# tools/pmc_calibrate.sh -> gpurun_out/pmc_calibration.txt: FETCH_SIZE / WRITE_SIZE per dispatch of a copy of known size (see tools/pmc_calibrate.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_calibration.txt; : > $out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/cal_$c -- python tools/pmc_calibrate.py >> $out 2>/tmp/cal_$c.err
  python tools/pmc_traffic.py $(find /tmp/cal_$c -name "*counter_collection.csv") >> $out
done
# the path's own access widths (tools/pmc_patterns.hip): 16 / 8 / 4 bytes per lane streams, 48-byte and 36-byte row segments at random places
if [ -x tools/pmc_patterns ]; then
  for c in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" TCC_EA0_RDREQ_128B_sum; do
    n=$(echo $c | cut -c1-20 | tr " " _); rm -rf /tmp/calp_$n
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/calp_$n -- tools/pmc_patterns >> $out 2>/tmp/calp_$n.err
    python tools/pmc_traffic.py $(find /tmp/calp_$n -name "*counter_collection.csv") >> $out
  done
fi
cat $out

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --list-avail 2>/dev/null | grep -i -E "TCC_EA|TCC_REQ|TCC_READ|TCC_MISS|TCC_HIT|FETCH_SIZE|WRITE_SIZE|MALL|TCC_BUBBLE|RDREQ" | head -60
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RD_UNCACHED_32B_sum"; do
  rm -rf /tmp/cl; timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/cl -- tools/pmc_patterns > /tmp/cl.out 2>/tmp/cl.err; python tools/pmc_traffic.py $(find /tmp/cl -name "*counter_collection.csv") 2>&1 | tail -6; tail -2 /tmp/cl.err | cut -c1-300
done

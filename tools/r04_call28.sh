#!/bin/bash
for q in 8 4; do echo "queues $q"; GPU_MAX_HW_QUEUES=$q python tools/rig_host_bench.py 64 0 40 2>&1 | tail -2 | cut -c150-520; done

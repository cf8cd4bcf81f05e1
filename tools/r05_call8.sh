#!/bin/bash
# round 5, GPU call 8: FETCH_SIZE calibration on the path's own access patterns
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_calibrate.sh 2>&1 | tail -30

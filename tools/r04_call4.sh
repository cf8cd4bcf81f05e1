#!/bin/bash
set -u
mkdir -p gpurun_out
bash tools/e2e_trace.sh ctx runtime:32 MCS_E2E_STREAMS=ctx
bash tools/e2e_trace.sh plain1 runtime:32 MCS_E2E_STREAMS=plain
bash tools/e2e_trace.sh plain2 runtime:runtime MCS_E2E_STREAMS=plain

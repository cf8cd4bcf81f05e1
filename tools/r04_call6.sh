#!/bin/bash
set -u
bash tools/e2e_trace.sh q8 runtime:32 MCS_E2E_STREAMS=plain MCS_E2E_IMAGE_BUFFERS=3 GPU_MAX_HW_QUEUES=8
python tools/e2e_timeline.py gpurun_out/e2etrace_q8 3

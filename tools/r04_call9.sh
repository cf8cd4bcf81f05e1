#!/bin/bash
set -u
bash tools/e2e_trace.sh res runtime:32 MCS_E2E_STREAMS=plain
bash tools/e2e_trace.sh h2donly runtime:off MCS_E2E_STREAMS=plain
bash tools/e2e_trace.sh d2honly off:32 MCS_E2E_STREAMS=plain
bash tools/e2e_trace.sh none off:off MCS_E2E_STREAMS=plain

#!/bin/bash
# A/B of the matcher's block-count target (MCS_MATCH_BLOCKS) on the default bench: prints target, Mfeat/s, ms/step, match ms
for b in "$@"; do
  MCS_MATCH_BLOCKS=$b timeout 200 python bench.py --no-cpu-baseline > /tmp/ab.json 2>/dev/null
  python - "$b" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["per_kernel_ms"]["match"], d["roofline"]["per_kernel_ms"]["greedy"])
PY
done

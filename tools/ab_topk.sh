#!/bin/bash
# A/B of the top-K list depth feeding the greedy resolution: prints K, Mfeat/s, ms/step, match ms, greedy ms, rescans
for k in "$@"; do
  timeout 200 python bench.py --no-cpu-baseline --topk $k > /tmp/ab.json 2>/dev/null
  python - "$k" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["per_kernel_ms"]["match"], d["roofline"]["per_kernel_ms"]["greedy"], d["config"]["greedy_rescans_rank0"], d["config"]["matches_per_step_rank0"])
PY
done

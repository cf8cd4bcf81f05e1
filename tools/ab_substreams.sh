#!/bin/bash
# A/B of independent sub-streams per GPU (bench.py --substreams): prints S, Mfeat/s, ms/step
for s in "$@"; do
  timeout 200 python bench.py --no-cpu-baseline --substreams $s > /tmp/ab.json 2>/dev/null
  python - "$s" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], d["value"], d["ms_per_step"])
PY
done

#!/bin/bash
args=$1; shift
for envs in "$@"; do
  env $envs timeout 600 python bench.py --no-cpu-baseline --no-secondary $args > /tmp/ab.json 2>/tmp/ab.err
  python - "$envs" <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
    k = d["roofline"]["per_kernel_ms"]
    print("%-24s" % sys.argv[1], d["value"], d["ms_per_step"], {x: k[x] for x in ("pyramid", "fast", "octree", "blur", "describe_fast", "match")}, d.get("oracle_check"))
except Exception as e:
    print(sys.argv[1], "failed", e, open("/tmp/ab.err").read()[-400:])
PY
done

#!/bin/bash
# usage: ab_env.sh "<bench args>" "ENV=.. ENV2=.." ...   (each env set benches the tree's own library)
args=$1; shift
for envs in "$@"; do
  env $envs timeout 600 python bench.py --no-cpu-baseline --no-secondary $args > /tmp/ab.json 2>/tmp/ab.err
  python - "$envs" <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
    print("%-28s" % sys.argv[1], d["value"], "Mfeat/s", d["ms_per_step"], "ms/step", "check", d.get("oracle_check"), "matches", d["config"]["matches_per_step_rank0"])
except Exception as e:
    print(sys.argv[1], "failed", e, open("/tmp/ab.err").read()[-600:])
PY
done

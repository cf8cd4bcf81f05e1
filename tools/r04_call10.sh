#!/bin/bash
set -u
o=gpurun_out/c10; mkdir -p $o
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-secondary > $o/$tag.json 2> $o/$tag.err
python - "$tag" <<'P'
import json,sys
try:
    d=json.loads(open("gpurun_out/c10/%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print("%-18s"%sys.argv[1], d["ms_per_step"], d["value"], d.get("oracle_check"))
except Exception as ex: print(sys.argv[1],"ERR", ex, open("gpurun_out/c10/%s.err"%sys.argv[1]).read()[-600:])
P
}
run base
run m55 MCS_MATCH_CU_MASK=55555555
run m33 MCS_MATCH_CU_MASK=33333333
run m77 MCS_MATCH_CU_MASK=77777777
run m11 MCS_MATCH_CU_MASK=11111111
run m0f MCS_MATCH_CU_MASK=0f0f0f0f
run mffff0000 MCS_MATCH_CU_MASK=ffff0000
run m55g55 MCS_MATCH_CU_MASK=55555555 MCS_GREEDY_CU_MASK=55555555
run m55gaa MCS_MATCH_CU_MASK=55555555 MCS_GREEDY_CU_MASK=aaaaaaaa
run base2

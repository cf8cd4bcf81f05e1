#!/bin/bash
# functional run of the N > 1 code path on a ONE-GPU box: N ranks share cuda:0, collectives over gloo (bench.py MCS_BENCH_SHARE_GPU=1); prints value / oracle_check per run
for spec in "2 " "4 " "3 --workload db --frames 4" "8 --frames 8" "2 --workload rig --frames 2"; do
  set -- $spec; n=$1; shift
  MCS_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 3 --warmup 2 --no-cpu-baseline --no-secondary "$@" > /tmp/sh_$n.json 2> /tmp/sh_$n.err
  python - "$n" "$*" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("/tmp/sh_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
    print("ranks", sys.argv[1], sys.argv[2], "| value", d["value"], "oracle_check", d.get("oracle_check"), "|", d["config"]["parallelism"][:150])
except Exception as e:
    print("ranks", sys.argv[1], "FAILED", e, open("/tmp/sh_%s.err" % sys.argv[1]).read()[-600:])
PY
done

#!/bin/bash
# A/B of the top-K list depth feeding the greedy resolution (default step and configs[2]): prints K, workload, ms/step, match ms, greedy ms, rescans, matches
for k in "$@"; do
  for w in stream db; do
    timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 --workload $w --topk $k 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('K=$k', '$w', d['ms_per_step'], d['roofline']['per_kernel_ms']['match'], d['roofline']['per_kernel_ms']['greedy'], d['config']['greedy_rescans_rank0'], d['config']['matches_per_step_rank0'], d['oracle_check'])"
  done
done

#!/usr/bin/env python3
"""Print the per-kernel ms of a bench.py JSON line read from stdin (A/B experiments)."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]["per_kernel_ms"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "value", d["value"], "ms/step", d["ms_per_step"], "rescans", d["config"]["greedy_rescans_rank0"], r)

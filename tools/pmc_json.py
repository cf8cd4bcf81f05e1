#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the counter summaries of tools/profile_round.sh: per workload (bench.py's config.tag) and kernel the average FETCH_SIZE /
WRITE_SIZE (KiB per dispatch) and SQ_INSTS_VALU (wave instructions per dispatch).  bench.py reports (fetch + write) * 1024 as roofline.traffic and the
instruction count in roofline.valu_issue when its workload tag matches.
Usage: pmc_json.py out.json  tag=path/to/pmc_summary.txt:path/to/bench_under_rocprof.json ..."""
import ast
import json
import re
import sys

NAMES = {"k_describe_fast": "describe", "k_match_partial": "match", "k_match_mfma": "match", "k_fast_cells": "fast", "k_blur": "blur", "k_resize_level": "pyramid", "k_octree": "octree",
         "k_greedy_spec": "greedy", "k_orient_a": "orient_a", "k_orient_b": "orient_b", "k_describe_list": "describe_list", "k_describe": "describe_exact", "k_expand_train": "match_expand", "k_resize_chain": "pyramid", "k_resize_cols": "pyramid", "k_copy_narrow": "copy"}
# FETCH_SIZE = TCC_EA0_RDREQ x 64 B, and EVERY read request the L2 sends to the memory side of this chip is a 128-byte request (round 5, tools/pmc_patterns.hip under
# rocprofv3 -> profiles/r06/pmc_calibration.txt: TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ for 16 / 8 / 4-byte-per-lane streams AND for scattered 48-byte / 36-byte row
# segments; the streams read exactly 0.500 of their bytes, the segment patterns exactly (line touches) x 64 B).  So the bytes that really cross the L2's memory side
# are 2 x FETCH_SIZE for every kernel — round 4 doubled only the 16-byte-per-lane readers and so reported the keypoint stage at 0.81x its algorithmic bytes; it is
# ~1.6x.  WRITE_SIZE reads 1.00 of a copy's bytes.  What is counted is the L2's miss traffic: lines served by the 256 MiB Infinity Cache are included.
FETCH_FACTOR = 2.0
out = {"_comment": "Memory-side traffic of the L2 and VALU instructions per launch from separate rocprofv3 --pmc passes (tools/profile_round.sh; KiB per dispatch, wave instructions "
                   "per dispatch).  fetch_kib = 2 x FETCH_SIZE for EVERY kernel: the counter tallies each 128-byte read request as 64 bytes, and all requests are 128-byte "
                   "ones (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ on streams and on scattered 36 / 48-byte segments alike: profiles/r06/pmc_calibration.txt); fetch_kib_raw keeps "
                   "the counter's own reading.  WRITE_SIZE is used as read (1.00 on a copy of known size).  Infinity-Cache hits are part of the count (it is the L2's miss "
                   "traffic, an upper bound of the HBM traffic).  Kernels launched several times per step (the 7 k_resize_cols launches) are summed per step.", "workloads": {}}
for arg in sys.argv[2:]:
    label, rest = arg.split("=", 1)
    summary, benchjson = rest.split(":")
    tag = json.loads(open(benchjson).read().strip().splitlines()[-1])["config"]["tag"]
    kern = {}
    rows = []
    for line in open(summary):
        m = re.match(r"\S+ (?:void )?mcs::(\w+)[^{]*(\{.*\}) dispatches (\d+)", line)
        if m and NAMES.get(m.group(1)):
            rows.append((NAMES[m.group(1)], ast.literal_eval(m.group(2)), int(m.group(3))))
    # launches per step of a kernel = its dispatches over those of the once-per-step descriptor kernel in the same counter pass (the resize chain is 7
    # launches, FAST 3 when the extraction runs in its two overlapped chains)
    for name, vals, disp in rows:
        ref = [d for n, v, d in rows if n in ("describe", "describe_exact") and set(v) == set(vals)]
        mult = disp / ref[0] if ref else 1.0
        k = kern.setdefault(name, {})
        if "FETCH_SIZE" in vals:
            k["fetch_kib"] = round(vals["FETCH_SIZE"] * mult * FETCH_FACTOR, 1)
            k["fetch_kib_raw"] = round(vals["FETCH_SIZE"] * mult, 1)
        if "WRITE_SIZE" in vals:
            k["write_kib"] = round(vals["WRITE_SIZE"] * mult, 1)
        if "SQ_INSTS_VALU" in vals:
            k["valu_insts"] = int(vals["SQ_INSTS_VALU"] * mult)
    out["workloads"][tag] = {"label": label, "kernels": kern}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print("wrote", sys.argv[1], list(out["workloads"]))

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
# FETCH_SIZE reads HALF the bytes of a 16-byte-per-lane coalesced stream on gfx950 (the guide; calibrated here on a copy of known size, tools/pmc_calibrate.sh ->
# profiles/r04/pmc_calibration.txt: 262 158 KiB read for 524 288 KiB copied, WRITE_SIZE 524 331 KiB for the same bytes): the kernels whose reads ARE such streams
# (the matcher's global_load_lds operand stream, the copy kernel) are doubled; dword-granular readers matched a known byte count within 14 % (k_blur, round 1).
WIDE_READERS = {"match": 2.0, "copy": 2.0}
out = {"_comment": "HBM traffic and VALU instructions per launch from separate rocprofv3 --pmc passes (tools/profile_round.sh; FETCH_SIZE / WRITE_SIZE in KiB per "
                   "dispatch, SQ_INSTS_VALU in wave instructions per dispatch).  FETCH_SIZE of the kernels that read 16 bytes per lane in whole-wave streams (match: "
                   "global_load_lds of the expanded operands) is DOUBLED (gfx950 tallies their 128-byte requests at 64 bytes: profiles/r04/pmc_calibration.txt, a copy of "
                   "known size reads 0.50 of its bytes, writes 1.00); fetch_kib_raw keeps the counter's own reading.  k_describe_fast requests its patches as 48-byte row "
                   "segments (three 16-byte lanes): not calibrated, left as read.  Kernels launched several times per step (the 7 k_resize_cols launches) are summed per step.", "workloads": {}}
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
            k["fetch_kib"] = round(vals["FETCH_SIZE"] * mult * WIDE_READERS.get(name, 1.0), 1)
            if name in WIDE_READERS:
                k["fetch_kib_raw"] = round(vals["FETCH_SIZE"] * mult, 1)
        if "WRITE_SIZE" in vals:
            k["write_kib"] = round(vals["WRITE_SIZE"] * mult, 1)
        if "SQ_INSTS_VALU" in vals:
            k["valu_insts"] = int(vals["SQ_INSTS_VALU"] * mult)
    out["workloads"][tag] = {"label": label, "kernels": kern}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print("wrote", sys.argv[1], list(out["workloads"]))

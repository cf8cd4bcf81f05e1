#!/usr/bin/env python3
"""Instruction mix of one kernel in a device assembly file (hipcc --cuda-device-only -S): static counts per class, and the same for the hottest loop
bodies (label ... backward branch).  Usage: isa_mix.py file.s mangled_kernel_name_substring"""
import collections
import re
import sys

src = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^(\S*%s\S*):[^\n]*\n(.*?)\n\.Lfunc_end' % re.escape(name), src, re.S | re.M)
if not m:
    sys.exit("kernel not found")
body = m.group(2).split('\n')


def klass(k):
    if k.startswith('v_') and 'f64' in k:
        return 'v_f64'
    if k.startswith(('v_rsq', 'v_rcp', 'v_sqrt', 'v_sin', 'v_cos', 'v_exp', 'v_log')):
        return 'v_trans'
    if k.startswith('v_'):
        return 'v_32'
    if k.startswith('s_load') or k.startswith('s_buffer_load'):
        return 'smem'
    if k.startswith('s_waitcnt'):
        return 'waitcnt'
    if k.startswith('s_'):
        return 'salu'
    if k.startswith(('global_', 'flat_', 'buffer_')):
        return 'vmem'
    if k.startswith('ds_'):
        return 'lds'
    if k.startswith('scratch_'):
        return 'scratch'
    return k


ins = []
for l in body:
    t = l.strip()
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
        continue
    ins.append(t.split()[0])
print(m.group(1), "static instructions:", len(ins))
print(dict(collections.Counter(klass(k) for k in ins)))
print(collections.Counter(ins).most_common(25))

#!/usr/bin/env python3
"""Instruction mix of the basic blocks of one kernel in an -S listing: python tools/isa_loop.py listing.s <mangled-substring>"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(_Z\w*' + re.escape(key) + r'\w*):[^\n]*\n(.*?)\n\s+\.end_amdhsa_kernel', s, re.S | re.M)
print(m.group(1))
blocks, cur = [], ["entry", []]
for l in m.group(2).split('\n'):
    lm = re.match(r'^(\.LBB\d+_\d+):', l)
    if lm:
        blocks.append(cur); cur = [lm.group(1), []]
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur[1].append(t.split()[0] + (" -> " + t.split()[-1] if t.startswith(("s_cbranch", "s_branch")) else ""))
blocks.append(cur)
for name, ins in blocks:
    c = collections.Counter()
    for i in ins:
        op = i.split()[0]
        k = ("trans" if re.match(r'v_(rcp|rsq|sqrt|exp|log|sin|cos)', op) else "vop3" if re.match(r'v_(fma|div_fixup|div_fmas|div_scale|mad|bfe|perm|med3|add3|lshl_add|cndmask)', op) else
             "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_barrier", "s_cbranch", "s_branch", "s_nop")) else
             "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "ctl")
        c[k] += 1
    br = [i for i in ins if "->" in i]
    print("%-12s n=%4d %s %s" % (name, len(ins), dict(c), br))

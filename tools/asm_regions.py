#!/usr/bin/env python3
"""Developer aid: per-region instruction census of one kernel in a hipcc -S listing (scratch, AGPR moves, MFMA, memory)."""
import re, collections, sys
path, sym = sys.argv[1], sys.argv[2]
step = int(sys.argv[3]) if len(sys.argv) > 3 else 500
s = open(path).read()
i = s.index(sym + ':')
j = s.index('.amdhsa_kernel', i)
lines = s[i:j].split('\n')
labels = {}
for n, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = n
for n, l in enumerate(lines):
    m = re.search(r'(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
    if m and labels.get(m.group(2), 10**9) < n and n - labels[m.group(2)] > 2500: print('loop', labels[m.group(2)], n)
for a in range(0, len(lines), step):
    c = collections.Counter()
    for l in lines[a:a + step]:
        t = l.strip().split()
        if not t or t[0][0] in ';.': continue
        op = t[0]
        key = ('scratch_st' if op.startswith('scratch_store') else 'scratch_ld' if op.startswith('scratch_load') else 'acc_rd' if 'accvgpr_read' in op else
               'acc_wr' if 'accvgpr_write' in op or 'accvgpr_mov' in op else 'mfma' if op.startswith('v_mfma') else 'gload' if op.startswith('global_load') else
               'gstore' if op.startswith('global_store') else 'ds' if op.startswith('ds_') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'other')
        c[key] += 1
    print(a, dict(sorted(c.items())))

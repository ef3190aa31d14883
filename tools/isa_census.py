#!/usr/bin/env python3
"""Developer aid: instruction census of the main (largest) loop of one kernel in a `hipcc -S --cuda-device-only` listing.

    hipcc <build flags of accelerated-tinympc_amd/build.py> -S --cuda-device-only csrc/admm_rowlane.hip -o /tmp/rowlane.s
    python tools/isa_census.py /tmp/rowlane.s _ZN7tinympc19admm_rowlane_kernelILi12ELi4ELi30ELb1ELb0ELb0ELb0ELb0EEEvNS_9RowParamsE
"""
import re, collections, sys
s = open(sys.argv[1]).read()
sym = sys.argv[2]
i = s.index(sym + ':'); j = s.index('.amdhsa_kernel', i)
lines = s[i:j].split('\n')
labels = {}
for n, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = n
loops = []
for n, l in enumerate(lines):
    m = re.search(r'(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
    if m and labels.get(m.group(2), 10**9) < n: loops.append((labels[m.group(2)], n))
a, b = max(loops, key=lambda t: t[1] - t[0])
c = collections.Counter(); nops = 0; dpp = 0
for l in lines[a:b]:
    t = l.strip().split()
    if not t or t[0][0] in ';.' or t[0].endswith(':'): continue
    if t[0] == 's_nop': nops += int(t[1]) + 1
    if '_dpp' in t[0]: dpp += 1
    c[t[0]] += 1
tot = sum(c.values())
valu = sum(v for k, v in c.items() if k.startswith('v_'))
print(f'main loop: lines {a}..{b}: {tot} instructions, {valu} VALU of which {dpp} DPP, s_nop wait states {nops}')
for k, v in c.most_common(40): print(f'  {k:28s}{v}')

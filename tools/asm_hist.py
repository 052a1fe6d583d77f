#!/usr/bin/env python
"""Static instruction histogram per kernel of a gfx950 .s file (hipcc -S --cuda-device-only)."""
import collections
import re
import sys

s = open(sys.argv[1]).read()
parts = re.split(r'\n\t\.section\t\.text\.', s)
for part in parts[1:]:
    name = part.split(',', 1)[0]
    end = part.find('.end_amdhsa_kernel')
    body = part[:part.find('\t.section\t.rodata')] if '\t.section\t.rodata' in part else part
    ins = [l.strip().split()[0] for l in body.split('\n')
           if l.startswith('\t') and not l.strip().startswith('.') and not l.strip().startswith(';')]
    c = collections.Counter(ins)
    groups = collections.Counter()
    for k, v in c.items():
        g = 'other'
        if k.startswith('v_'): g = 'valu'
        if k.startswith('s_'): g = 'salu'
        if k.startswith('ds_'): g = 'lds'
        if k.split('_')[0] in ('global', 'buffer', 'flat', 'scratch'): g = 'vmem'
        if k in ('v_readlane_b32', 'v_writelane_b32'): g = 'sgpr_spill'
        if re.match(r'v_(sqrt|rcp|rsq|sin|cos|log|exp)_f(32|64)', k): g = 'trans'
        if k.startswith('s_cbranch') or k.startswith('s_branch'): g = 'branch'
        groups[g] += v
    print(name[:60], 'total', len(ins), dict(groups))
    if len(sys.argv) > 2:
        print('   ', c.most_common(int(sys.argv[2])))

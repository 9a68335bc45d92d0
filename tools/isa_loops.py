#!/usr/bin/env python3
"""Per-loop instruction mix of one kernel in a hipcc -S listing: python tools/isa_loops.py file.s <mangled-kernel-prefix>"""
import re
import sys
from collections import Counter

path, prefix = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().endswith(':') or (l.startswith(prefix) and ': ' in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i


def cls(op):
    if op.startswith('v_'):
        if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')):
            return 'lane'
        if '_f64' in op:
            return 'v_f64'
        return 'v_other'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith(('s_load', 's_buffer_load')):
        return 'smem'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem' if not op.startswith('scratch_') else 'scratch'
    if op.startswith('ds_'):
        return 'lds'
    return 'other'


loops = []
for i, l in enumerate(body):
    m = re.match(r'^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'^\s+s_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
for a, b in sorted(set(loops)):
    c = Counter()
    for l in body[a:b + 1]:
        m = re.match(r'^\s+([a-z_0-9]+)', l)
        if m and not l.strip().startswith(';'):
            c[cls(m.group(1))] += 1
    tot = sum(c.values())
    if tot > 80:
        print(f'loop lines {a}-{b} ({tot} instrs):', dict(c))

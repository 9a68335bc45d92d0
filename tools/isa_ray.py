#!/usr/bin/env python3
"""Instruction mix of ONE ray of the lane sweep in a hipcc -S -gline-tables-only listing: the stretch between two scan clusters
(ds_bpermute groups): python tools/isa_ray.py file.s <mangled-kernel-prefix> [source.hip]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith(pref))
body = lines[start:]
end = next(i for i, l in enumerate(body) if l.startswith('.Lfunc_end'))
body = body[:end]
bp = [i for i, l in enumerate(body) if 'ds_bpermute_b32' in l]
cl = []
for i in bp:
    if cl and i - cl[-1][-1] < 200: cl[-1].append(i)
    else: cl.append([i])
print('scan clusters:', [(c[0], len(c)) for c in cl])
src = open(sys.argv[3]).read().split('\n') if len(sys.argv) > 3 else None
for a, b in zip(cl[:-1], cl[1:]):
    c = Counter(); lane = Counter(); mov = Counter(); cur = None
    for l in body[a[0]:b[0]]:
        m = re.match(r'\s+\.loc\s+\d+\s+(\d+)', l)
        if m: cur = int(m.group(1))
        m = re.match(r'^\s+([a-z_0-9]+)', l)
        if m and not l.strip().startswith(('.', ';')):
            op = m.group(1); c[op] += 1
            if op.startswith(('v_readlane', 'v_writelane')): lane[cur] += 1
            if op.startswith('v_mov_b'): mov[cur] += 1
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    f64 = sum(v for k, v in c.items() if k.startswith('v_') and '_f64' in k)
    print(f'-- ray body {a[0]}..{b[0]}: VALU {valu} (f64 {f64}), SALU {sum(v for k, v in c.items() if k.startswith("s_"))}, '
          f'cndmask {c["v_cndmask_b32_e64"] + c["v_cndmask_b32_e32"]}, lane ops {c["v_readlane_b32"] + c["v_writelane_b32"]}, '
          f'mov {c["v_mov_b64_e32"] + c["v_mov_b32_e32"]}, waits {c["s_waitcnt"]}')
    for name, cnt in (('lane ops', lane), ('v_mov', mov)):
        print('   ', name, 'by source line:', sorted(cnt.items(), key=lambda x: -x[1])[:10])

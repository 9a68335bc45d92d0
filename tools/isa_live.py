#!/usr/bin/env python3
"""VGPR liveness of one kernel in a hipcc -S listing (compile with -gline-tables-only for source lines):
   python tools/isa_live.py file.s <mangled-kernel-prefix> [source-file-substring]
Prints the maximum number of live VGPRs, where it is reached, and for the registers live there the source line
of their definition.  Approximate def/use rules (first operand = destination, stores / LDS adds only read)."""
import re
import sys
from collections import Counter, defaultdict

path, prefix = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and (l.rstrip().endswith(':') or ': ' in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start + 1:end]

files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))

RC = 's' if (len(sys.argv) > 3 and sys.argv[3] == 'sgpr') else 'v'
REG = re.compile(r'\b%s(\d+)\b|\b%s\[(\d+):(\d+)\]' % (RC, RC))


def regs_of(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


instrs = []   # (op, defs, uses, srcline, text)
label_at = {}
cur_loc = None
for l in body:
    s = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        label_at[m.group(1)] = len(instrs)
        continue
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m:
        cur_loc = (int(m.group(1)), int(m.group(2)))
        continue
    if not s or s.startswith(';') or s.startswith('.'):
        continue
    s = s.split(';')[0].strip()
    m = re.match(r'^([a-z_0-9]+)\s*(.*)$', s)
    if not m:
        continue
    op, rest = m.group(1), m.group(2)
    ops = [t.strip() for t in rest.split(',')] if rest else []
    defs, uses = [], []
    nodst = op.startswith(('buffer_store', 'global_store', 'scratch_store', 'flat_store', 'ds_write', 'ds_add', 'ds_max', 'ds_min',
                           'v_cmpx', 'global_atomic', 'buffer_atomic', 'v_nop', 's_cmp', 's_cbranch', 's_branch', 's_waitcnt', 's_nop',
                           's_barrier', 's_endpgm', 's_setprio', 's_sleep', 's_bitcmp', 's_setreg', 's_setpc', 's_sendmsg'))
    if RC == 'v' and op.startswith('s_'):
        nodst = True
    if op.startswith('v_cmp') and op.endswith('_e32'):
        nodst = True
    if op.startswith(('ds_add_rtn', 'ds_max_rtn')):
        nodst = False
    ndst = 0 if nodst else 1
    if op.startswith('v_div_scale'):
        ndst = 2
    for i, t in enumerate(ops):
        (defs if i < ndst else uses).extend(regs_of(t))
    if op.startswith(('v_fmac', 'v_mac', 'v_writelane')) or '_dpp' in op or 'dpp' in rest:
        uses.extend(defs)
    instrs.append((op, defs, uses, cur_loc, s))

n = len(instrs)
succ = [[] for _ in range(n)]
for i, (op, d, u, loc, s) in enumerate(instrs):
    if op == 's_endpgm':
        continue
    m = re.search(r'(\.LBB\d+_\d+)', s)
    if op == 's_branch' and m and m.group(1) in label_at:
        succ[i].append(label_at[m.group(1)])
        continue
    if op.startswith('s_cbranch') and m and m.group(1) in label_at:
        succ[i].append(label_at[m.group(1)])
    if i + 1 < n:
        succ[i].append(i + 1)

live_in = [set() for _ in range(n)]
changed = True
it = 0
while changed and it < 60:
    changed = False
    it += 1
    for i in range(n - 1, -1, -1):
        out = set()
        for j in succ[i]:
            if j < n:
                out |= live_in[j]
        new = (out - set(instrs[i][1])) | set(instrs[i][2])
        if new != live_in[i]:
            live_in[i] = new
            changed = True
cnt = [len(s) for s in live_in]
mx = max(cnt)
at = cnt.index(mx)
print(f'{n} instructions, max live VGPRs {mx} at instruction {at}: {instrs[at][4]}  (source {instrs[at][3]})')
# reaching definitions (nearest preceding def in listing order: approximate)
lastdef = {}
defline = {}
for i in range(at + 1):
    for r in instrs[i][1]:
        lastdef[r] = i
by_src = Counter()
for r in sorted(live_in[at]):
    i = lastdef.get(r)
    loc = instrs[i][3] if i is not None else None
    by_src[(loc, instrs[i][0] if i is not None else '?')] += 1
filt = sys.argv[3] if len(sys.argv) > 3 else None
for (loc, op), c in sorted(by_src.items(), key=lambda kv: (kv[0][0] or (0, 0))):
    fn = files.get(loc[0], '?') if loc else '?'
    print(f'  {c:3d} regs  def {op:28s} {fn.split("/")[-1]}:{loc[1] if loc else "?"}')
# profile every 200 instructions
print('profile (instr index: live):', ' '.join(f'{i}:{cnt[i]}' for i in range(0, n, max(1, n // 60))))

#!/usr/bin/env python3
"""Register spills stored while lanes are switched off: a static check of the gfx950 listings of every kernel (CPU only).

What this looks for.  A VGPR the register allocator spills is stored to the lane's private segment ("scratch") with the CURRENT
exec mask.  ROCm 7.2's allocator may put that store at the top of the block that JOINS a lane-divergent `if`, in front of the
`s_or_b64 exec, exec, sN` that switches the other lanes back on (lanesweep_kernel, round 5: the `if (c.blk == 0)` around the
boundary-intensity fill, three lanes of 64):

    .LBB0_181:
        scratch_store_dword off, v246, off offset:24 ; 4-byte Folded Spill     <- only the lanes of the `else` side
        s_or_b64 exec, exec, s[4:5]
        ...
        scratch_load_dword v246, off, off offset:24  ; 4-byte Folded Reload    <- all lanes

The lanes that were off get whatever the slot held before -- what an earlier kernel of this or ANOTHER process left in the
queue's scratch memory: results that differ from box to box, from process to process and with what ran before (the deviation
that rounds 3 and 4 chased in the device allocations).  The dynamic check is LWHIP_SCRATCH_POISON (tests/test_padding.py);
this is the static one, run by lightweaver_amd.build on the listings the compiler writes next to the objects
(-save-temps=obj): the build FAILS when a kernel is reported.

Method.  Per kernel, the exec state is a stack of open regions carried along the control-flow graph (labels, branches,
fall-through; a block takes the state of the first edge that reaches it):
    s_and_saveexec_b64 sN, cond          opens a region (mask that closes it: sN; `s_xor_b64 sM, exec, sN` moves it to sM)
    s_or_saveexec_b64 sD, sN             closes the region of sN -- all lanes of the enclosing region are on -- and the
    s_xor_b64 exec, exec, sD             that follows opens the else side (closed by sD)
    s_or_b64 exec, exec, sN              closes the region of sN
    s_andn2_b64 / s_and_b64 exec, ...    lanes leaving a loop: a region that the next unmatched s_or_b64 exec closes
Reaching definitions per spill slot over the same graph: a store replaces the reaching stores that ran with the same lanes or
fewer, and leaves the others (a store with lanes off updates the slot for its lanes only -- legitimate after a store with all
lanes on: `if (c) x = f(x)`).  A reload is REPORTED when no store that reaches it ran with at least the reload's lanes on.

usage: python -m lightweaver_amd.spill_lint file.s [...]      exit code 1 if anything is reported
       python -m lightweaver_amd.spill_lint --build UNIT.hip [-DNAME ...]   compile a unit of csrc/ to a listing first
       SPILL_LINT_VERBOSE=1: every slot's stores and reloads with their region stacks
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # the repository
SAVE = re.compile(r'^\s*s_(and|or|andn2|xor|orn2)_saveexec_b64\s+(s\[\d+:\d+\]|vcc)')
OR_EXEC = re.compile(r'^\s*s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\]|vcc)')
NARROW = re.compile(r'^\s*s_(andn2|and|xor)_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\]|vcc)')
MOV_EXEC = re.compile(r'^\s*s_mov_b64\s+exec,\s*(\S+)')
SPILL = re.compile(r'^\s*(scratch_store_\w+)\s+off,\s*(\S+),\s*off(?:\s+offset:(\d+))?\s*;.*Spill')
RELOAD = re.compile(r'^\s*(scratch_load_\w+)\s+(\S+),\s*off,\s*off(?:\s+offset:(\d+))?\s*;.*Reload')
ACC_SPILL = re.compile(r'^\s*v_accvgpr_write_b32\s+(a\d+),\s*(v\d+)\s*;.*Spill')
ACC_RELOAD = re.compile(r'^\s*v_accvgpr_read_b32\s+(v\d+),\s*(a\d+)\s*;.*Reload')
WIDTH = {'dword': 4, 'dwordx2': 8, 'dwordx3': 12, 'dwordx4': 16, 'b32': 4, 'b64': 8, 'b96': 12, 'b128': 16}


LABEL = re.compile(r'^(\.LBB\d+_\d+):')
BRANCH = re.compile(r'^\s*s_(cbranch_\w+|branch)\s+(\.LBB\d+_\d+)')
XOR_RENAME = re.compile(r'^\s*s_xor_b64\s+(s\[\d+:\d+\]|vcc),\s*exec,\s*(s\[\d+:\d+\]|vcc)')
SAVE_MOV = re.compile(r'^\s*s_mov_b64\s+(s\[\d+:\d+\]|vcc),\s*exec\b')


def lint_kernel(name, lines, first_line, verbose=False):
    """The exec state as a stack of open regions, carried along the control-flow graph (labels, branches, fall-through): a
    block takes the state of the first edge that reaches it (depth-first from the entry); a later edge with another state is
    counted (structured code agrees at every join once the joining block's own `s_or_b64 exec` has run -- and that comes
    after the block's label, so joins of an if / else DO arrive with different states: the one kept is the first, the
    restore pops whichever)."""
    # blocks
    blocks = [('entry', [])]
    for n, l in enumerate(lines, first_line):
        m = LABEL.match(l)
        if m:
            blocks.append((m.group(1), []))
            continue
        blocks[-1][1].append((n, l))
    index = {lab: i for i, (lab, _) in enumerate(blocks)}
    entry_state = {0: ()}
    order = [0]
    nreg = [0]
    stores, loads = {}, {}
    mismatch = 0
    visited = set()
    events = {}   # block -> [('S' | 'L', slot, line, region ids)] in program order
    succs = {}    # block -> set of blocks

    def pop_to(stack, sg):
        for i in range(len(stack) - 1, -1, -1):
            if stack[i][1] == sg:
                return stack[:i], True
        return stack, False

    work = [0]
    while work:
        bi = work.pop()
        if bi in visited:
            continue
        visited.add(bi)
        stack = list(entry_state[bi])
        pending = None
        fall = True
        ev = events.setdefault(bi, [])
        succs.setdefault(bi, set())
        for n, l in blocks[bi][1]:
            m = SAVE.match(l)
            if m:
                kind, sg = m.group(1), m.group(2)
                if kind == 'or':
                    # `s_or_saveexec_b64 D, S` between the two sides of an if / else: the region whose mask is in S ends and
                    # ALL lanes of the enclosing region are on until `s_xor_b64 exec, exec, D` switches the first side's off
                    ms = re.match(r'^\s*s_or_saveexec_b64\s+(\S+),\s*(\S+)', l)
                    stack, _ = pop_to(stack, ms.group(2))
                    pending = sg
                    continue
                nreg[0] += 1
                stack = stack + [(nreg[0], sg)]
                continue
            m = XOR_RENAME.match(l)
            if m:
                # `s_and_saveexec_b64 sB, cond` + `s_xor_b64 sA, exec, sB`: the mask that closes the region moves to sA
                if stack and stack[-1][1] == m.group(2):
                    stack = stack[:-1] + [(stack[-1][0], m.group(1))]
                continue
            m = OR_EXEC.match(l)
            if m:
                stack, ok = pop_to(stack, m.group(1))
                if not ok:
                    stack, _ = pop_to(stack, None)   # (a loop's lanes come back through a register not seen saving them)
                continue
            m = SAVE_MOV.match(l)
            if m:
                pending = m.group(1)
                continue
            m = NARROW.match(l)
            if m:
                if pending is not None:
                    nreg[0] += 1
                    stack = stack + [(nreg[0], pending)]
                    pending = None
                elif not (stack and stack[-1][1] is None):
                    nreg[0] += 1
                    stack = stack + [(nreg[0], None)]
                continue
            m = MOV_EXEC.match(l)
            if m:
                if m.group(1) == '-1':
                    stack = []
                else:
                    stack, _ = pop_to(stack, m.group(1).rstrip(','))
                continue
            ids = tuple(i for i, _ in stack)
            m = SPILL.match(l)
            if m:
                w = WIDTH.get(m.group(1).split('_')[-1], 4)
                off = int(m.group(3) or 0)
                for b in range(off, off + w, 4):
                    stores.setdefault(b, []).append((n, ids))
                    ev.append(('S', b, n, ids))
                continue
            m = RELOAD.match(l)
            if m:
                w = WIDTH.get(m.group(1).split('_')[-1], 4)
                off = int(m.group(3) or 0)
                for b in range(off, off + w, 4):
                    loads.setdefault(b, []).append((n, ids))
                    ev.append(('L', b, n, ids))
                continue
            m = ACC_SPILL.match(l)
            if m:
                stores.setdefault(m.group(1), []).append((n, ids))
                ev.append(('S', m.group(1), n, ids))
                continue
            m = ACC_RELOAD.match(l)
            if m:
                loads.setdefault(m.group(2), []).append((n, ids))
                ev.append(('L', m.group(2), n, ids))
                continue
            m = BRANCH.match(l)
            if m:
                ti = index.get(m.group(2))
                if ti is not None:
                    succs[bi].add(ti)
                    if ti not in entry_state:
                        entry_state[ti] = tuple(stack)
                        work.append(ti)
                    elif tuple(i for i, _ in entry_state[ti]) != ids:
                        mismatch += 1
                if m.group(1) == 'branch':
                    fall = False
                    break
                continue
            if re.match(r'^\s*s_endpgm', l) or re.match(r'^\s*s_setpc_b64', l):
                fall = False
                break
        if fall and bi + 1 < len(blocks):
            ti = bi + 1
            succs[bi].add(ti)
            if ti not in entry_state:
                entry_state[ti] = tuple(stack)
                work.append(ti)
            elif tuple(i for i, _ in entry_state[ti]) != tuple(i for i, _ in stack):
                mismatch += 1
    # which stores reach which reload: reaching definitions per slot over the same graph (a store kills the earlier ones)
    preds = {}
    for b, ss in succs.items():
        for t in ss:
            preds.setdefault(t, set()).add(b)
    # transfer of one store: it replaces the reaching stores of the slot that ran with the same lanes or fewer (their region
    # stack extends this one's); a store with lanes off leaves the others' values in the slot for the lanes it did not write
    def apply(cur, slot, n, ids):
        old = cur.get(slot, frozenset())
        keep = frozenset(d for d in old if d[1][:len(ids)] != ids)
        cur[slot] = keep | frozenset([(n, ids)])

    IN = {b: {} for b in events}     # block -> {slot: frozenset of (line, ids)}
    OUT = {}
    todo = list(events)
    inq = set(todo)
    while todo:
        b = todo.pop()
        inq.discard(b)
        new_in = {}
        for pb in preds.get(b, ()):
            for slot, defs in OUT.get(pb, {}).items():
                cur = new_in.get(slot)
                new_in[slot] = defs if cur is None else (cur | defs)
        IN[b] = new_in
        out = dict(new_in)
        for kind, slot, n, ids in events[b]:
            if kind == 'S':
                apply(out, slot, n, ids)
        if out != OUT.get(b):
            OUT[b] = out
            for t in succs.get(b, ()):
                if t in events and t not in inq:
                    todo.append(t)
                    inq.add(t)
    # a reload is reported when NO store that reaches it ran with (at least) the reload's lanes on
    bad = []
    for b, ev in events.items():
        cur = dict(IN[b])
        for kind, slot, n, ids in ev:
            if kind == 'S':
                apply(cur, slot, n, ids)
                continue
            defs = cur.get(slot, ())
            if defs and not any(ids[:len(sst)] == sst for _, sst in defs):
                for sl, sst in defs:
                    bad.append((slot, sl, sst, n, ids))
    if verbose:
        for slot, ss in sorted(stores.items(), key=lambda kv: str(kv[0])):
            for sl, sst in ss:
                print(f'    slot {slot}: store line {sl} regions {list(sst)}; reloads ' + ', '.join(f'{ll} {list(lst)}' for ll, lst in loads.get(slot, [])))
    nsp = sum(len(v) for v in stores.values())
    if verbose:
        print(f'    ({len(blocks)} blocks, {len(visited)} reached, {mismatch} edges arrive with another state than the first)')
    return nsp, bad


def lint_file(path, quiet=False, report=None):
    """-> number of reported slots; report (a list) receives one (kernel, spill stores, reported slots) per kernel with spills."""
    txt = open(path, errors='replace').read().split('\n')
    kernels = []   # (name, first line index, last)
    cur = None
    for i, l in enumerate(txt):
        m = re.match(r'^(\w+):\s*(;.*)?$', l)
        if m and not l.startswith('.L') and i + 1 < len(txt) and cur is None and ('@function' in ''.join(txt[max(0, i - 6):i])):
            cur = (m.group(1), i)
        if cur and re.match(r'^\s*s_endpgm', l):
            pass
        if cur and l.startswith('.Lfunc_end'):
            kernels.append((cur[0], cur[1], i))
            cur = None
    total_bad = 0
    for name, a, b in kernels:
        nsp, bad = lint_kernel(name, txt[a:b], a + 1, verbose=bool(os.environ.get('SPILL_LINT_VERBOSE')))
        if nsp == 0 and not bad:
            continue
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r'\(lwhip::.*', '', dn).replace('lwhip::', '').replace('(anonymous namespace)::', '').replace('void ', '')
        slots = sorted({str(s) for s, *_ in bad}, key=lambda s: (len(s), s))
        if report is not None:
            report.append((dn, nsp, len(slots)))
        if not quiet or bad:
            print(f'{os.path.basename(path)}: {dn}: {nsp} spill stores (dwords), {len(slots)} slot(s) stored with lanes off and reloaded outside'
                  + (': ' + ' '.join(slots[:24]) if slots else ''))
        seen = set()
        for slot, sl, sst, ll, lst in bad:
            if (slot, sl) in seen:
                continue
            seen.add((slot, sl))
            if not quiet:
                print(f'    slot {slot}: store line {sl} (regions {list(sst)}) -- reload line {ll} (regions {list(lst)})')
        total_bad += len(slots)
    return total_bad


def build_listing(unit, defs):
    sys.path.insert(0, ROOT)
    from lightweaver_amd import build
    out = os.path.join(tempfile.gettempdir(), 'spill_lint_' + os.path.splitext(unit)[0] + ('_' + '_'.join(d.strip('-D') for d in defs) if defs else '') + '.s')
    cmd = [build._hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-S', '--cuda-device-only'] + defs + ['-o', out, os.path.join(build.CSRC, unit)]
    subprocess.run(cmd, check=True, cwd=build.CSRC, stderr=subprocess.DEVNULL)
    return out


if __name__ == '__main__':
    args = sys.argv[1:]
    files = []
    if args and args[0] == '--build':
        files.append(build_listing(args[1], args[2:]))
    else:
        files = args
    nbad = sum(lint_file(f) for f in files)
    print(f'{nbad} slot(s) reported')
    sys.exit(1 if nbad else 0)

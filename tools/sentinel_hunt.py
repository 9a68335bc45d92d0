#!/usr/bin/env python3
"""Which device buffer is read before anything wrote it?  (run on the GPU box: gpurun -- python3 tools/sentinel_hunt.py)

LWHIP_PAD_SENTINEL fills the fp64 state allocations of every context with a FINITE sentinel instead of zeros, so a result that
depends on never-written memory changes -- deterministically, no statistics (NaN, the pattern of LWHIP_POISON, is swallowed by
the fmin / fmax of the Steffen derivative).  One value is not enough: 6.7e299 in an opacity is "optically thick" and can be as
harmless as zero, so several values are tried (LWHIP_PAD_SENTINEL_VALUE), and small valid-looking integers in the index /
argument tables (LWHIP_PAD_SENTINEL_INT).

  stage 0   one test with LWHIP_TRACE_ALLOC: proof that the sentinel is applied, and the allocation numbers
  stage 1   the chosen tests under every (value, int) configuration           -> which tests depend on padding at all
  stage 2   the first failing test under a bisection over the allocation numbers -> which buffer it is

Output: gpurun_out/sentinel/{trace.log, cfg_*.log, table.txt}
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out', 'sentinel')
os.makedirs(OUT, exist_ok=True)
tests = sys.argv[1:] or ['tests/test_hip_parity.py', 'tests/test_split_order.py', 'tests/test_prd.py', 'tests/test_hprd.py']
CONFIGS = [('rand', {'LWHIP_PAD_SENTINEL_VALUE': 'rand'}), ('randsign', {'LWHIP_PAD_SENTINEL_VALUE': 'randsign'}), ('big', {}), ('one', {'LWHIP_PAD_SENTINEL_VALUE': '1.0'}), ('milli', {'LWHIP_PAD_SENTINEL_VALUE': '1e-3'}),
           ('neg', {'LWHIP_PAD_SENTINEL_VALUE': '-1.0'}), ('e4', {'LWHIP_PAD_SENTINEL_VALUE': '3e4'}),
           ('tiny', {'LWHIP_PAD_SENTINEL_VALUE': '1e-12'}),
           ('int1', {'LWHIP_PAD_SENTINEL_INT': '1'}), ('int2', {'LWHIP_PAD_SENTINEL_INT': '2'}), ('int7', {'LWHIP_PAD_SENTINEL_INT': '7'})]


def run(env_extra, args, log):
    env = dict(os.environ)
    env['LWHIP_DEBUG'] = '1'   # (the diagnosis knobs are read only with it)
    env.update(env_extra)
    with open(log, 'w') as f:
        p = subprocess.run([sys.executable, '-m', 'pytest', '-m', 'gpu', '-q', '--tb=line', '-p', 'no:cacheprovider'] + args,
                           cwd=ROOT, env=env, stdout=f, stderr=subprocess.STDOUT)
    return p.returncode


def failed_ids(log):
    ids = []
    for line in open(log, errors='replace'):
        m = re.match(r'^(FAILED|ERROR) (\S+)', line)
        if m:
            ids.append(m.group(2))
    return ids


tab = open(os.path.join(OUT, 'table.txt'), 'w')


def say(line):
    print(line, flush=True)
    tab.write(line + '\n')
    tab.flush()


trace = os.path.join(OUT, 'trace.log')
run({'LWHIP_PAD_SENTINEL': 'all', 'LWHIP_PAD_SENTINEL_INT': '1', 'LWHIP_TRACE_ALLOC': '1'},
    ['tests/test_hip_parity.py::test_lane_sweep_ray_split_vs_oracle', '-s'], trace)
txt = open(trace, errors='replace').read()
nmax = max([int(m) for m in re.findall(r'lwhip alloc #(\d+) ', txt)] or [0])
say('stage 0: %d allocations traced, %d with the sentinel, %d with the int sentinel, %d cleared; highest number %d'
    % (txt.count('lwhip alloc #'), txt.count(' sentinel'), txt.count(' int-sentinel'), txt.count(' cleared'), nmax))
bad_cfg = []
for name, env in CONFIGS:
    e = {'LWHIP_PAD_SENTINEL': 'all'}
    e.update(env)
    log = os.path.join(OUT, 'cfg_%s.log' % name)
    rc = run(e, tests + ['-rfE'], log)
    bad = failed_ids(log)
    tail = [ln.strip() for ln in open(log, errors='replace') if ' passed' in ln or ' failed' in ln or 'error' in ln.lower()][-1:]
    say('stage 1 [%s %s]: rc %d, failing %d  %s' % (name, env, rc, len(bad), ' | '.join(tail)))
    for b in bad:
        say('      ' + b)
    if bad:
        bad_cfg.append((name, env, bad))
for name, env, bad in bad_cfg[:2]:
    picked = bad[:1]
    say('stage 2 [%s]: bisection over the allocation numbers 0 .. %d, test: %s' % (name, nmax, picked[0]))

    def fails(lo, hi):
        e = {'LWHIP_PAD_SENTINEL': '%d-%d' % (lo, hi)}
        e.update(env)
        log = os.path.join(OUT, '%s_%03d_%03d.log' % (name, lo, hi))
        rc = run(e, picked + ['-rfE'], log)
        say('  allocations %3d .. %3d: %s' % (lo, hi, 'FAIL' if rc else 'ok'))
        return rc != 0

    def bisect(lo, hi, depth=0):
        if lo == hi:
            say('  -> allocation number %d' % lo)
            return
        mid = (lo + hi) // 2
        left = fails(lo, mid)
        right = fails(mid + 1, hi)
        if left:
            bisect(lo, mid, depth + 1)
        if right:
            bisect(mid + 1, hi, depth + 1)
        if not left and not right:
            say('  (neither half of %d .. %d fails alone: several allocations together)' % (lo, hi))

    if fails(0, nmax):
        bisect(0, nmax)

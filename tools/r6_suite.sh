#!/bin/bash
# the whole -m gpu suite, then the default bench run with a one-screen summary of its JSON line   usage: bash tools/r6_suite.sh [tag]
TAG=${1:-r6}
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -25
timeout 900 python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
python3 - <<PY
import json
try:
    d = json.loads([l for l in open('gpurun_out/bench_$TAG.json') if l.startswith('{')][-1])
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/bench_$TAG.err').read()[-3000:]); raise SystemExit(1)
r = d['roofline']
print('value', round(d['value'], 1), 'ms/step', round(d['ms_per_step'], 5), 'kernel_ms', round(r['kernel_ms'], 5), 'frac', round(r['frac'], 4),
      'parity', d.get('parity_checked'), d.get('max_rel_err'), 'cpu', d.get('cpu_baseline', {}).get('value'))
for k, v in d.get('aux', {}).items():
    if isinstance(v, dict):
        print(' aux', k, {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items() if not isinstance(vv, (dict, list, str)) or kk in ('failed',)})
    else:
        print(' aux', k, v)
PY

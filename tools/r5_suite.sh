#!/bin/bash
# the whole -m gpu suite, then the default bench run
mkdir -p gpurun_out/r5
timeout 2400 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r5/suite.log 2>&1; echo "suite rc $?"
tail -15 gpurun_out/r5/suite.log
timeout 1500 python3 bench.py > gpurun_out/r5/bench.json 2> gpurun_out/r5/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r5/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','max_rel_err','parity_checked')})
print('roofline', {k:d['roofline'][k] for k in ('frac','kernel_ms')}, d['roofline']['secondary']['frac'])
for k,v in d['aux'].items():
    if isinstance(v,dict):
        print(k, {kk:vv for kk,vv in v.items() if kk in ('ms_per_step','ms_per_subiter','ms_per_batch_iteration','column_iters_per_s','ms_per_iteration','ratio_to_headline','host_build_s','contexts_and_profiles_s','sweep','failed','max_rel_err')})
    else:
        print(k,v)
PY

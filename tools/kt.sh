#!/bin/bash
# kernel durations of a short bench run under given env: bash tools/kt.sh "ENV=1 ENV2=2"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  rm -rf /tmp/kt_out
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_out -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /tmp/kt.json 2> /tmp/kt.err
  echo "== $e: $(python3 -c "import json;d=json.load(open('/tmp/kt.json'));print('%.0f it/s %.3f ms/step'%(d['value'],d['ms_per_step']))" 2>/dev/null)"
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/kt_out/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0].replace('void lwhip::','')
    if any(k in n for k in ('raymarch','tile_p','apply','sweep_')):
        print('   %-45s %8.1f us x %s'%(n[:45], float(r['AverageNs'])/1e3, r['Calls']))
PY
done

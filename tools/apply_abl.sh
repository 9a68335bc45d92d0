#!/bin/bash
# apply_kernel's duration (rocprofv3 average) under the ablation builds AP_ABL_* of lwhip_kernels.hip (wrong results on purpose)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in ${@:-base apng apnd apnr apnone}; do
  if [ "$v" = base ]; then unset LWHIP_LIB; else export LWHIP_LIB=$R/lightweaver_amd/variants/liblwhip_$v.so; fi
  rm -rf /tmp/ap_st
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ap_st -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-aux > /tmp/ap_b.json 2>/dev/null
  python3 - << PY
import csv, glob
for r in csv.DictReader(open(glob.glob('/tmp/ap_st/*/*kernel_stats.csv')[0])):
    if 'apply_kernel' in r['Name'] or 'lanesweep' in r['Name']:
        print('$v', r['Name'].split('(')[0][-40:], 'avg %.2f us  min %.2f' % (float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done

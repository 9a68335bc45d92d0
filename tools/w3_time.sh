#!/bin/bash
# the lane sweep compiled for THREE wavefronts per SIMD (-DLS_WAVES_PER_EU=3: the compiler must fit 168 VGPRs and spills what it cannot)
# against the shipped two: bench step, kernel time, registers and scratch of the launched instance   usage: bash tools/w3_time.sh
R=$GRAFT_REPO_ROOT
bash $R/tools/variant_time.sh base w3 base w3
OUT=$R/gpurun_out/w3; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export LWHIP_LIB=$R/lightweaver_amd/variants/liblwhip_w3.so
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-aux > $OUT/bench.json 2> $OUT/err
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/t/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'lanesweep' in r['Kernel_Name']]
r = rows[-1]
print({k: r[k] for k in r if any(s in k for s in ('VGPR', 'SGPR', 'Scratch', 'LDS', 'Workgroup_Size', 'Grid_Size'))})
d = sorted(int(x['End_Timestamp']) - int(x['Start_Timestamp']) for x in rows)
print('lanesweep launches', len(rows), 'median ns', d[len(d) // 2])
PY
python3 -c "
import json; d=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][-1]); print('parity', d.get('parity_checked'), d.get('max_rel_err'))"

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python3 - << PY
import csv,glob
f=glob.glob("$R/gpurun_out/tl/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last iterations' kernels (the population solves of the aux section come after them)
rows=[r for r in rows if any(k in r['Kernel_Name'] for k in ('sweep_kernel','reduce_stage','apply_kernel'))]
sel=rows[-12:]
t0=int(sel[0]['Start_Timestamp'])
prev_end=None
for r in sel:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-prev_end)/1e3 if prev_end else 0
    print('%-40s start %8.1f us dur %7.1f us gap %6.1f us'%(r['Kernel_Name'][:40],(s-t0)/1e3,(e-s)/1e3,gap))
    prev_end=e
PY

#!/bin/bash
# kernel timeline of the sharded path on one GPU (tools/shard_probe.py under rocprofv3 --kernel-trace): what the fixed price
# of N > 1 is made of -- start / duration / gap of the last launches.  usage: bash tools/shard_timeline.sh [Nlambda]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-1280}
rm -rf $R/gpurun_out/tl_shard
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_shard -- python3 $R/tools/shard_probe.py $N > /dev/null 2>&1
python3 - << PY
import csv,glob
f=glob.glob("$R/gpurun_out/tl_shard/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=rows[-16:]
t0=int(sel[0]['Start_Timestamp'])
prev_end=None
for r in sel:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-prev_end)/1e3 if prev_end else 0
    print('%-60s start %8.1f us dur %7.1f us gap %6.1f us'%(r['Kernel_Name'].replace('lwhip::','').replace('(anonymous namespace)::','')[:60],(s-t0)/1e3,(e-s)/1e3,gap))
    prev_end=e
PY

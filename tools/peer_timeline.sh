#!/bin/bash
# kernel timeline of the peer exchange on one GPU (tools/peer_probe.py under rocprofv3 --kernel-trace): start / duration / gap of
# the last launches -- sweep end -> reduce_stage2 -> peer_publish -> apply start of a rank whose peers are on time.
# usage: bash tools/peer_timeline.sh [world] [Nlambda per shard]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl_peer
python3 $R/tools/peer_probe.py ${1:-2} ${2:-1280}
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_peer -- python3 $R/tools/peer_probe.py ${1:-2} ${2:-1280} > /dev/null 2>&1
python3 - << PY
import csv,glob
f=glob.glob("$R/gpurun_out/tl_peer/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the peer rounds are before the one-call steps of the lone shard: take a window that ends at the last peer_publish
last=max(i for i,r in enumerate(rows) if 'peer_publish' in r['Kernel_Name'])
sel=rows[last-13:last+4]
t0=int(sel[0]['Start_Timestamp'])
for r in sel:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%-46s queue %-4s start %8.1f us dur %7.1f us end %8.1f'%(r['Kernel_Name'].replace('lwhip::','').replace('(anonymous namespace)::','').replace('void ','')[:46],r.get('Queue_Id','?'),(s-t0)/1e3,(e-s)/1e3,(e-t0)/1e3))
PY

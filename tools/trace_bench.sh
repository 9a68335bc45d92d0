#!/bin/bash
# HIP API + kernel + copy trace of a short bench run: what is the device doing during the closing synchronize?
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_bench
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/out.json 2> $OUT/err.log
python3 - << PY
import csv, glob
api = list(csv.DictReader(open(glob.glob("$OUT/*/*hip_api_trace.csv")[0])))
ker = list(csv.DictReader(open(glob.glob("$OUT/*/*kernel_trace.csv")[0])))
cps = glob.glob("$OUT/*/*memory_copy_trace.csv")
cp = list(csv.DictReader(open(cps[0]))) if cps else []
syncs = [r for r in api if r['Function'] == 'hipDeviceSynchronize']
print('hipDeviceSynchronize calls', len(syncs))
for s in syncs:
    b, e = int(s['Start_Timestamp']), int(s['End_Timestamp'])
    print('sync dur %.3f ms' % ((e - b) / 1e6))
    for k in ker:
        kb, ke = int(k['Start_Timestamp']), int(k['End_Timestamp'])
        if ke > b and kb < e:
            print('   kernel', k['Kernel_Name'][:50], 'start %+.3f ms dur %.3f ms' % ((kb - b) / 1e6, (ke - kb) / 1e6))
    for c in cp:
        cb, ce = int(c['Start_Timestamp']), int(c['End_Timestamp'])
        if ce > b and cb < e:
            print('   copy', c.get('Direction'), 'start %+.3f dur %.3f ms' % ((cb - b) / 1e6, (ce - cb) / 1e6))
# slowest API calls
api.sort(key=lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp']), reverse=True)
for r in api[:12]:
    print(r['Function'], '%.3f ms' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
PY

#!/bin/bash
# usage: tools/pmc2d.sh "<counters>" tag   -- one rocprofv3 --pmc pass over the 2D iteration (tests/diag/bench2d.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $R/gpurun_out/pmc2d_$2 -- python3 $R/tests/diag/bench2d.py 256 0.5 > $R/gpurun_out/pmc2d_$2.log 2>&1
python3 - << PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/pmc2d_$2/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
first=None
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "2d" not in k and "cont_kernel" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if first is None: first=r["Counter_Name"]
    if r["Counter_Name"]==first: cnt[k]+=1
for k,v in agg.items():
    print(k[:60], cnt[k], {c: "%.4g"%(x/max(cnt[k],1)) for c,x in v.items()})
PY

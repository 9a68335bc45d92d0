#!/bin/bash
# counters of the 2D iteration's kernels: bash tools/pmc_2d.sh "COUNTERS..." ["COUNTERS..." ...]  (one pass per group)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grp in "$@"; do
  rm -rf /tmp/pmc2
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc2 -- python3 $R/tests/diag/bench2d.py > /dev/null 2> /tmp/pmc2.err
  python3 - << PY
import csv,glob,collections
f=glob.glob("/tmp/pmc2/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    for s in ("fs2d_coef","rates2d","fs2d_scan","gather2d","cont_kernel"):
        if s in k:
            agg[s][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[s][r["Counter_Name"]]+=1
for s in agg:
    print(s, {c: "%.4g"%(x/max(cnt[s][c],1)) for c,x in agg[s].items()})
PY
done

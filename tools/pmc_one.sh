#!/bin/bash
# counters of one kernel (name substring) under an env: bash tools/pmc_one.sh <kernel-substr> "ENV=.." "COUNTERS..."
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
K=$1; E=$2; shift 2
for grp in "$@"; do
  rm -rf /tmp/pmc1
  env $E rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc1 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux > /dev/null 2> /tmp/pmc1.err
  python3 - << PY
import csv,glob,collections
f=glob.glob("/tmp/pmc1/*/*counter_collection.csv")[0]
agg=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    if "$K" not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
print({c: "%.4g"%(x/max(cnt[c],1)) for c,x in agg.items()})
PY
done

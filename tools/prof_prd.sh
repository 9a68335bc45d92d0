#!/bin/bash
# kernel-level timing of the PRD sub-iterations (bench.py --aux) under rocprofv3
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_prd
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --aux > $OUT/bench.json 2> $OUT/err.log
python3 - << PY
import csv, glob
st = glob.glob("$OUT/stats/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(st)):
    print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
PY

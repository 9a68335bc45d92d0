#!/bin/bash
# kernel-level timing of the 2D iteration (tests/diag/bench2d.py) under rocprofv3
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof2d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tests/diag/bench2d.py ${1:-256} ${2:-0.5} > $OUT/out.log 2> $OUT/err.log
tail -2 $OUT/out.log
python3 - << PY
import csv, glob
st = glob.glob("$OUT/stats/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(st)):
    print(r['Name'][:80], r['Calls'], r['AverageNs'], r['Percentage'])
PY

#!/bin/bash
# Kernel-trace stats of one side measurement of bench.py: bash tools/prof_aux.sh prd|columns|2d [extra bench args]
R=$GRAFT_REPO_ROOT
W=$1; shift
OUT=$R/gpurun_out/prof_aux_$W
rm -rf $OUT/stats; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --aux-only $W "$@" > $OUT/bench.json 2> $OUT/stats.err
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv | cut -c1-220
cat $OUT/bench.json

#!/bin/bash
# which HIP runtime calls the set-up of a 512-column batch spends its time in   usage: bash tools/columns_api_trace.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/columns_api; rm -rf $OUT; mkdir -p $OUT
python3 $R/tools/create_threads_time.py 2>&1 | tail -9
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --stats --output-format csv -d $OUT/t -- python3 $R/bench.py --aux-only columns --no-cpu-baseline > $OUT/bench.json 2> $OUT/err
f=$(ls $OUT/t/*/*hip_api_stats.csv | head -1); cp $f $OUT/hip_api_stats.csv
head -16 $OUT/hip_api_stats.csv | cut -c1-150
python3 -c "
import json; d=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][-1]); print({k:v for k,v in d['C4_columns'].items() if not isinstance(v,dict)})"

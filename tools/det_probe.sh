#!/bin/bash
# kernel-level timing of the fixed-order mode against the default at 10 240 wavelengths: where its extra time goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r5/detprof; mkdir -p $O
for m in default det; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$m -o $m -- python3 $R/tools/det_time.py $m 10240 2560 > $O/$m.log 2>&1
  grep Nlambda $O/$m.log
  f=$(find $O/$m -name "*kernel_stats.csv" | head -1)
  echo "== $m: $f"; head -7 "$f" | cut -c1-220
done
echo "== det, nobody waits (turns all zero: the sums' order is the timing's)"
LWHIP_DEBUG=1 LWHIP_DET_NOWAIT=1 python3 $R/tools/det_time.py det 10240 2560 2>&1 | grep Nlambda

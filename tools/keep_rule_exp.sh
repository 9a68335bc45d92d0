#!/bin/bash
# the tail rule of the lane sweep: the automatic choice against "cut numCU / 2 chunks" (LWHIP_LANE_KEEP = chunks - 128) at several sizes
R=$GRAFT_REPO_ROOT
cd $R
for n in 6656 7680 8704 9216 10240 11264 12288 14336; do
  ch=$(( ( (n + 2) / 3 + 3 ) / 4 ))
  for k in auto $((ch - 128)) $((ch - 64)) $((ch - 192)); do
    if [ "$k" = auto ]; then unset LWHIP_LANE_KEEP; else export LWHIP_LANE_KEEP=$k; fi
    python3 bench.py --nlambda $n --steps 30 --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n chunks $ch keep $k: ms/step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done

#!/bin/bash
# one partial round of the lane sweep: the automatic number of chunks cut against ALL of them (LWHIP_LANE_ONE_CUT=99999), interleaved
R=$GRAFT_REPO_ROOT
cd $R
run() { python3 bench.py --nlambda $1 --steps 40 --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 cut ${LWHIP_LANE_ONE_CUT:-auto}: ms/step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for r in 1 2; do
  for n in 3840 4096 4352 4608 4864 5120 5376; do
    unset LWHIP_LANE_ONE_CUT; run $n; export LWHIP_LANE_ONE_CUT=99999; run $n; unset LWHIP_LANE_ONE_CUT
  done
done

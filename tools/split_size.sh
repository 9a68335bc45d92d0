#!/bin/bash
# the main sweep's ray split S at shard sizes (what a rank of a 2 / 4 / 8-GPU run sweeps)   usage: bash tools/split_size.sh
R=$GRAFT_REPO_ROOT
export LWHIP_DEBUG=1
for n in 1280 2560 3840 5120; do
  for s in 0 1 2 4; do
    LWHIP_LANE_SPLIT=$s python3 $R/bench.py --nlambda $n --steps 200 --warmup 20 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('nlambda $n split $s', 'ms/step %.4f  kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done

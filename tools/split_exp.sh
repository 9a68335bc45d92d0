#!/bin/bash
# lane sweep: a tile's rays split over S wavefronts (LWHIP_LANE_SPLIT) against the wavelength count
R=$GRAFT_REPO_ROOT
export LWHIP_SWEEP=lanes
for n in 1280 2560 5120; do
  for S in 1 2 4; do
    LWHIP_LANE_SPLIT=$S python3 $R/bench.py --nlambda $n --steps 40 --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('Nlambda=$n S=$S', 'ms/step %.4f  kernel_ms %.4f  err %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d.get('max_rel_err')))"
  done
done

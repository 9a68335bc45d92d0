#!/bin/bash
# lane sweep at 10 240 wavelengths: the launch's last chunks cut into pieces whose tiles' rays are split over LWHIP_LANE_TAIL
# wavefronts; LWHIP_LANE_KEEP chunks (of the heaviest-first dispatch order) stay whole
R=$GRAFT_REPO_ROOT
export LWHIP_SWEEP=lanes
for cfg in "0 512" "2 512" "4 512" "2 448" "4 448" "2 384" "4 384" "2 320" "4 320" "2 256" "4 256"; do
  set -- $cfg
  LWHIP_LANE_TAIL=$1 LWHIP_LANE_KEEP=$2 python3 $R/bench.py --steps 60 --warmup 5 ${NLA:+--nlambda $NLA} --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tail S=$1 keep=$2', 'ms/step %.4f  kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done

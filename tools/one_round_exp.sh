#!/bin/bash
# one partial round of the lane sweep (3 072 < wavelengths < 6 144): the number of chunks cut in two (heaviest first), LWHIP_LANE_ONE_CUT
R=$GRAFT_REPO_ROOT
cd $R
for n in 3328 3840 4352 4864 5376; do
  ch=$(( ( (n + 2) / 3 + 3 ) / 4 ))
  for k in auto 0 $(( (512 - ch) / 2 )) $(( 512 - ch - 64 )) $ch; do
    if [ "$k" = auto ]; then unset LWHIP_LANE_ONE_CUT; else export LWHIP_LANE_ONE_CUT=$k; fi
    python3 bench.py --nlambda $n --steps 30 --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n chunks $ch cut $k: ms/step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done

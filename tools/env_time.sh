#!/bin/bash
# time the default bench step under several environments of the library: bash tools/env_time.sh "A=1" "LWHIP_LANE_TAIL=0" ...
R=${GRAFT_REPO_ROOT:-.}
for e in "$@"; do
  env $e python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', 'ms/step %.4f  kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done

R=$GRAFT_REPO_ROOT
export LWHIP_SWEEP=lanes
for n in 10240 8192 6400; do for cfg in "0 512" "2 512" "2 600" "2 700" "2 800"; do
  set -- $cfg
  LWHIP_LANE_TAIL=$1 LWHIP_LANE_KEEP=$2 python3 $R/bench.py --steps 60 --warmup 5 --nlambda $n --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n tail S=$1 keep=$2', 'ms/step %.4f  kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done

R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for m in lanes march; do
  LWHIP_SWEEP=$m python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('10240 rep$rep $m', 'ms/step %.4f  kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
for n in 12288 15360 20480; do for m in lanes march; do
  LWHIP_SWEEP=$m python3 $R/bench.py --steps 40 --warmup 5 --nlambda $n --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n $m', 'ms/step %.4f  kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done

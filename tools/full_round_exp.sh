for n in 4608 5120 5632 6144; do for f in 85 200; do echo -n "$n FULL=$f: "; LWHIP_LANE_FULL=$f python3 bench.py --nlambda $n --steps 30 --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done; done

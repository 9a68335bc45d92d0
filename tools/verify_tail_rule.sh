R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
for n in 6656 7680 8704 9216 10240 11264 12288 14336 20480; do python3 bench.py --nlambda $n --steps 30 --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n: ms/step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-aux --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver form: %.1f it/s ms_per_step %.4f kernel %.4f frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"; done

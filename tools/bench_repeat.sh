R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3 4 5; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-aux --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20/5  ms_per_step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done
for i in 1 2; do python3 bench.py --gpus 1 --steps 100 --warmup 10 --no-aux --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('100/10 ms_per_step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done

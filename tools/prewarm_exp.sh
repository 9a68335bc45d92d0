#!/bin/bash
# the driver's form of the bench (20 steps, 5 warm-up) against the number of untimed clock-settle steps before them
R=$GRAFT_REPO_ROOT
cd $R
for pw in 0 60 200 500 1000; do for i in 1 2 3; do LWHIP_BENCH_PREWARM=$pw python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-aux --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prewarm $pw: ms_per_step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done; done

#!/bin/bash
# kernel times of a short bench run under rocprofv3 + the bench's own numbers: bash tools/kt.sh ["ENV=.. ENV2=.."]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_st
env $1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_st -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-aux > /tmp/kt_b.json 2>/dev/null
head -6 /tmp/kt_st/*/*kernel_stats.csv | cut -c1-160
cd $R
for i in 1 2 3; do env $1 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f sweep %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done

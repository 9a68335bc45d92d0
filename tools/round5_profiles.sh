#!/bin/bash
# Everything the round-5 profiles/ files are refreshed from (run on the GPU box): the default bench under rocprofv3 with counters,
# the aux entries (incl. the fixed-order mode and the realistic overlapping-lines grid), the default bench line in the driver's form.
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile.sh r05 > gpurun_out/profile_r05.log 2>&1
for w in prd columns 2d fixed realistic; do bash tools/prof_aux.sh $w > gpurun_out/prof_aux_$w.log 2>&1; done
python3 bench.py --steps 100 --warmup 10 --no-aux --no-cpu-baseline > gpurun_out/bench_r05_100.json 2> /dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r05_driver.json 2> gpurun_out/bench_r05_driver.err
tail -c 400 gpurun_out/bench_r05_100.json
tail -c 3000 gpurun_out/bench_r05_driver.json

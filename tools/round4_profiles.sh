#!/bin/bash
# Everything the round-4 profiles/ files are refreshed from (run on the GPU box): the default bench under rocprofv3 with counters,
# the aux entries, the size sweep, the default bench line and the driver's form of it.
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile.sh r04 > gpurun_out/profile_r04.log 2>&1
for w in prd columns 2d; do bash tools/prof_aux.sh $w > gpurun_out/prof_aux_$w.log 2>&1; done
bash tools/size_sweep.sh > gpurun_out/size_sweep_r04.txt 2>&1
python3 bench.py --steps 100 --warmup 10 --no-aux --no-cpu-baseline > gpurun_out/bench_r04_100.json 2> /dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r04_driver.json 2> gpurun_out/bench_r04_driver.err
tail -c 400 gpurun_out/bench_r04_100.json
tail -c 3000 gpurun_out/bench_r04_driver.json

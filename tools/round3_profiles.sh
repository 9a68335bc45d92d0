#!/bin/bash
# Everything the round-3 profiles/ directory is made from (run on the GPU box): the default bench under rocprofv3 with
# counters (the lane sweep, the default, at 10 240 wavelengths), the lane sweep at two shard sizes, the march forced at the full
# size, the aux entries, the size sweep and the default bench line.
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile.sh r03 > gpurun_out/profile_r03.log 2>&1
bash tools/profile.sh r03_lanes_1280 --nlambda 1280 > gpurun_out/profile_r03_lanes_1280.log 2>&1
bash tools/profile.sh r03_lanes_5120 --nlambda 5120 > gpurun_out/profile_r03_lanes_5120.log 2>&1
LWHIP_SWEEP=march bash tools/profile.sh r03_march_10240 > gpurun_out/profile_r03_march_10240.log 2>&1
for w in prd columns 2d; do bash tools/prof_aux.sh $w > gpurun_out/prof_aux_$w.log 2>&1; done
bash tools/size_sweep.sh > gpurun_out/size_sweep_r03.txt 2>&1
python3 bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err
tail -c 600 gpurun_out/bench_r03.json

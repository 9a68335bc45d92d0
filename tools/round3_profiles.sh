#!/bin/bash
# Everything the round-3 profiles/ directory is made from (run on the GPU box): the default bench under rocprofv3 with
# counters (march sweep at 10 240 wavelengths), the lane sweep at a shard size and at the full size, the aux entries.
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile.sh r03 > gpurun_out/profile_r03.log 2>&1
LWHIP_SWEEP=lanes bash tools/profile.sh r03_lanes_5120 --nlambda 5120 > gpurun_out/profile_r03_lanes_5120.log 2>&1
LWHIP_SWEEP=lanes bash tools/profile.sh r03_lanes_10240 > gpurun_out/profile_r03_lanes_10240.log 2>&1
for w in prd columns 2d; do bash tools/prof_aux.sh $w > gpurun_out/prof_aux_$w.log 2>&1; done
python3 bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err
tail -c 600 gpurun_out/bench_r03.json

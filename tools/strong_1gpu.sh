#!/bin/bash
# The N = 2 strong split of the bench problem on ONE GPU (two ranks share device 0, gloo all-reduce staged through the
# host): per-rank step time and the host time spent issuing the collective, next to the single-rank run.  Functional
# evidence for the sharded path with the real kernels; not a scaling measurement.
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('1 rank  : ms/step %.4f  value %.0f it/s  sweep %s kernel_ms %.4f' % (d['ms_per_step'], d['value'], d['config']['sweep'], d['roofline']['kernel_ms']))"
for n in 2 4; do
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --share-gpu --backend gloo --steps 100 --warmup 10 --no-cpu-baseline --no-aux --no-weak 2>/dev/null | grep '^{' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n ranks on one GPU (strong): ms/step %.4f  value %.0f it/s  per-rank wavelengths %d  sweep %s kernel_ms %.4f  allreduce_us %.1f' % (d['ms_per_step'], d['value'], d['config']['Nlambda_per_gpu'], d['config']['sweep'], d['roofline']['kernel_ms'], d.get('allreduce_us', 0)))"
done

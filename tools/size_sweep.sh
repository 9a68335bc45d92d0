#!/bin/bash
# step time of the iteration against the wavelength count, both sweeps: bash tools/size_sweep.sh
R=$GRAFT_REPO_ROOT
for n in 1280 2560 3840 5120 6144 7680 10240; do for m in lanes march; do
  LWHIP_SWEEP=$m python3 $R/bench.py --steps 40 --warmup 5 --nlambda $n --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n $m', 'ms/step %.4f  kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done

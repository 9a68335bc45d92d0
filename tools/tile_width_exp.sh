#!/bin/bash
# march: sweep-kernel time against the wavelength cap of tiles with two slots (LWHIP_TILE_LH) and with one (LWHIP_TILE_L1)
R=$GRAFT_REPO_ROOT
export LWHIP_SWEEP=march
for cfg in "12 12" "10 12" "8 12" "6 12" "8 10" "6 10" "8 11" "9 12" "7 12"; do
  set -- $cfg
  LWHIP_TILE_LH=$1 LWHIP_TILE_L1=$2 LWHIP_VERBOSE=1 python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-aux 2> /tmp/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('LH=$1 L1=$2', 'ms/step %.4f  kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  grep -o "tiles=[0-9]*" /tmp/err.txt | head -1
done

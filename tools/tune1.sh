#!/bin/bash
# quick knob sweep
for cfg in "10 512 8" "5 512 8" "5 1024 8" "2 1024 8" "10 256 8" "10 1024 8" "10 512 16" "10 512 4" "5 2048 8" "3 1024 8"; do
  set -- $cfg
  LWHIP_G=$1 LWHIP_CHUNKS=$2 LWHIP_P=$3 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('G=$1 CH=$2 P=$3', 'it/s=%.0f'%d['value'], 'kernel_ms=%.3f'%d['roofline']['kernel_ms'], 'ms/step=%.3f'%d['ms_per_step'])"
done

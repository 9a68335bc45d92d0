#!/bin/bash
# quick knob sweep: "G CHUNKS"
for cfg in "5 384" "5 512" "5 640" "5 768" "5 1024" "10 512" "4 768"; do
  set -- $cfg
  LWHIP_G=$1 LWHIP_CHUNKS=$2 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('G=$1 CH=$2', 'it/s=%.0f'%d['value'], 'kernel_ms=%.3f'%d['roofline']['kernel_ms'], 'ms/step=%.3f'%d['ms_per_step'])"
done

#!/bin/bash
export LWHIP_DEBUG=1
python3 -m pytest tests/test_hip_parity.py tests/test_timed_sizes.py tests/test_split_order.py tests/test_prd.py tests/test_plugin.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6
for r in 1 2; do
for pr in 1 0; do
  LWHIP_PAIR_RAYS=$pr python3 bench.py --no-aux --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('pair $pr', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done; done

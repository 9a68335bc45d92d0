#!/bin/bash
# quick loop for the march kernel: parity subset, phase stamps, bench timing
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -3
for D in 4 7; do
LWHIP_D=$D python tools/stamps.py 2>&1 | grep stamps | tail -1
LWHIP_D=$D python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('D=$D it/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'])"
done

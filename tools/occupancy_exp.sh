#!/bin/bash
# The occupancy experiment of VERDICT r02 #1 on the ray-column march: the same kernel held to 3 / 4 wavefronts per SIMD
# (amdgpu_waves_per_eu), single problem (10 240 wavelengths) and the 512-column batch (abundant parallelism).
R=$GRAFT_REPO_ROOT
export LWHIP_SWEEP=march
for v in base occ3 occ4; do
  if [ "$v" = base ]; then unset LWHIP_LIB; else export LWHIP_LIB=$R/lightweaver_amd/variants/liblwhip_$v.so; fi
  python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v single: ms/step %.4f  sweep kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  python3 $R/bench.py --aux-only columns 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())['C4_columns']; print('$v columns: ms/batch-iteration %.2f  column-iterations/s %.0f' % (d['ms_per_batch_iteration'], d['column_iters_per_s']))"
done

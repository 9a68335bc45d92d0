#!/bin/bash
# A/B on ONE box: in-order runs of tests/test_hip_parity.py with nothing cleared, alternating between the build whose slot adds
# are flat atomics (rounds 3 / 4) and the shipped one (ds_add_f64)
export LWHIP_DEBUG=1   # (the diagnosis knobs are read only with it)
N=${1:-40}
mkdir -p gpurun_out/flake
fa=0; fb=0
for i in $(seq 1 $N); do
  LWHIP_NO_CLEAR=1 LWHIP_LIB=$PWD/lightweaver_amd/variants/liblwhip_flatacc.so python3 -m pytest tests/test_hip_parity.py -m gpu -q -x -s --tb=short -p no:cacheprovider > gpurun_out/flake/A_$i.log 2>&1
  if [ $? -ne 0 ]; then fa=$((fa+1)); echo "A run $i FAIL $(grep -m1 -E '^E  ' gpurun_out/flake/A_$i.log | cut -c1-200)"; grep DIAG gpurun_out/flake/A_$i.log | head -8; else rm -f gpurun_out/flake/A_$i.log; fi
  LWHIP_NO_CLEAR=1 python3 -m pytest tests/test_hip_parity.py -m gpu -q -x -s --tb=short -p no:cacheprovider > gpurun_out/flake/B_$i.log 2>&1
  if [ $? -ne 0 ]; then fb=$((fb+1)); echo "B run $i FAIL $(grep -m1 -E '^E  ' gpurun_out/flake/B_$i.log | cut -c1-200)"; grep DIAG gpurun_out/flake/B_$i.log | head -8; else rm -f gpurun_out/flake/B_$i.log; fi
done
echo "A (flat atomics for the slots' adds): $fa of $N failed; B (ds_add_f64, shipped): $fb of $N failed" | tee -a gpurun_out/flake/summary.txt

#!/bin/bash
# N in-order runs of tests/test_hip_parity.py (one process each) under an environment; counts the failing runs.
#   tools/flake_probe.sh 50 noclear LWHIP_NO_CLEAR=1
#   tools/flake_probe.sh 50 noclear_sync LWHIP_NO_CLEAR=1 LWHIP_SYNC_ALLOC=1
# (round 4 saw the 1.77e-6 deviation of the PRD atom's rates in 5 of 40 such runs before device allocations were cleared)
export LWHIP_DEBUG=1   # (the diagnosis knobs are read only with it)
N=${1:-40}; TAG=${2:-default}; shift 2
mkdir -p gpurun_out/flake
fails=0
for i in $(seq 1 $N); do
  log=gpurun_out/flake/${TAG}_$i.log
  env "$@" python3 -m pytest tests/test_hip_parity.py -m gpu -q -x -s --tb=short -p no:cacheprovider > $log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "run $i: FAIL $(grep -m1 -E '^(FAILED|E  )' $log)"; grep DIAG $log | cut -c1-600 | head -24; else rm -f $log; fi
done
echo "$TAG ($*): $fails of $N runs failed" | tee -a gpurun_out/flake/summary.txt

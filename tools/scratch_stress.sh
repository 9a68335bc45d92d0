#!/bin/bash
# The whole -m gpu suite N times in seeded random orders with the scratch memory poisoned in front of every kernel launch
# (LWHIP_SCRATCH_POISON=1) and the fp64 state allocations filled with different plausible values (LWHIP_PAD_SENTINEL=all, rand).
# usage: bash tools/scratch_stress.sh [N=3] [first seed=1]  -> gpurun_out/r05_stress/{run_SEED.log,summary.txt}
export LWHIP_DEBUG=1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-3}
S0=${2:-1}
OUT=$R/gpurun_out/r05_stress
mkdir -p $OUT
export LWHIP_SCRATCH_POISON=1 LWHIP_PAD_SENTINEL=all LWHIP_PAD_SENTINEL_VALUE=rand
: > $OUT/summary.txt
for ((i = 0; i < N; ++i)); do
  seed=$((S0 + i))
  python3 -m pytest $R/tests -q -m gpu -p no:cacheprovider --lw-shuffle $seed --deselect tests/test_padding.py > $OUT/run_$seed.log 2>&1
  rc=$?
  echo "seed $seed rc $rc: $(grep -E "passed|failed|error" $OUT/run_$seed.log | tail -n 1)" >> $OUT/summary.txt
  grep -E "^FAILED|^ERROR" $OUT/run_$seed.log | head -5 >> $OUT/summary.txt
done
cat $OUT/summary.txt

#!/bin/bash
# VERDICT r03 "Next round" #1: the whole -m gpu suite N times in seeded random orders under the diagnostic builds --
#   LWHIP_POISON=1       device allocations are filled with 0xFF when made and 0xFE when given back,
#   LWHIP_CHECK_ZERO=1   the stage-1 reduce buffer is checked (on the device) to hold zeros whenever a sweep starts adding to it,
#   LWHIP_LIB=.../variants/liblwhip_poison.so  (-DLWHIP_DBG_POISON_LDS: every workgroup of every kernel starts with its
#                        whole LDS allocation holding a NaN pattern)
# usage: bash tools/parity_stress.sh [N=20] [first seed=1]  -> gpurun_out/r04_stress/{run_SEED.log,summary.txt}
export LWHIP_DEBUG=1   # (the diagnosis knobs are read only with it)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-20}
S0=${2:-1}
OUT=$R/gpurun_out/r04_stress
mkdir -p $OUT
export LWHIP_POISON=1 LWHIP_CHECK_ZERO=1
[ -f $R/lightweaver_amd/variants/liblwhip_poison.so ] && export LWHIP_LIB=$R/lightweaver_amd/variants/liblwhip_poison.so
echo "lib: ${LWHIP_LIB:-shipped}" > $OUT/summary.txt
for ((i = 0; i < N; ++i)); do
  seed=$((S0 + i))
  python3 -m pytest $R/tests -q -m gpu -p no:cacheprovider --lw-shuffle $seed -x > $OUT/run_$seed.log 2>&1
  rc=$?
  echo "seed $seed rc $rc: $(grep -E "passed|failed|error" $OUT/run_$seed.log | tail -n 1)" >> $OUT/summary.txt
done
cat $OUT/summary.txt

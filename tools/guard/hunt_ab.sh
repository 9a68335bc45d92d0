#!/bin/bash
# The A/B that names the writer of the stray host store and shows the fix (profiles/r06_stray_write.md):
#   D  guardalloc + LWHIP_DESTROY_STREAMS=1 (what rounds 1-5 did: hipStreamDestroy of a context's stream), pytest's own
#      fault handler off so that guardalloc's is the first to see the fault: address, the freed block it falls into and who
#      allocated it, native backtrace of the storing thread
#   P  guardalloc + the default (pooled streams, never destroyed): no fault expected
# usage: bash tools/guard/hunt_ab.sh [secD] [secP]
export LWHIP_DEBUG=1 LWHIP_NO_CLEAR=1
SD=${1:-300}; SP=${2:-420}
OUT=gpurun_out/guard; mkdir -p $OUT; rm -f $OUT/*.log $OUT/summary_ab.txt
gcc -O2 -fPIC -shared -o $OUT/guardalloc.so tools/guard/guardalloc.c -ldl || exit 1
phase() {
  local name=$1 budget=$2; shift 2
  local t0=$SECONDS n=0 fails=0
  while [ $((SECONDS - t0)) -lt $budget ]; do
    n=$((n+1)); log=$OUT/${name}_$n.log
    env "$@" > $log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "$name run $n: rc $rc $(grep -m1 -E '^(FAILED|E  )' $log | cut -c1-300)"
      [ $fails -le 3 ] && grep -A34 'GUARDALLOC: signal' $log | cut -c1-400; else rm -f $log; fi
  done
  echo "phase $name: $fails of $n runs failed ($((SECONDS - t0)) s)" | tee -a $OUT/summary_ab.txt
}
PYT="python3 -m pytest tests/test_hip_parity.py -m gpu -q -x -s --tb=short -p no:cacheprovider -p no:faulthandler"
phase D_destroy $SD LD_PRELOAD=$PWD/$OUT/guardalloc.so LWHIP_DESTROY_STREAMS=1 $PYT
phase P_pooled $SP LD_PRELOAD=$PWD/$OUT/guardalloc.so $PYT
echo "box: $(rocm-smi --showuniqueid 2>/dev/null | grep -m1 'Unique ID:')" | tee -a $OUT/summary_ab.txt

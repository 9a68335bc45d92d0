#!/bin/bash
# One box, a time budget per phase (seconds): who writes the stray 4 bytes into the oracle's input copy?
#   A  tests/test_hip_parity.py under tools/guard/guardalloc.so: freed heap blocks of the victim's size window are PROT_NONE for
#      good -- a CPU store through a stale pointer faults with the writer's native backtrace
#   B  the same file with LWHIP_TEST_GUARD=1: the oracle's inputs in read-only pages -- changed content with no fault = a device
#      writer; a fault = a CPU writer that aims at the array itself
#   C  control: tools/guard/control.py -- torch + the oracle only, liblwhip.so never loaded
#   P  plain in-order runs (what rounds 3-5 hunted with), to see whether this box shows the deviation at all
# usage: bash tools/guard/hunt.sh [secA] [secB] [secC] [secP]
export LWHIP_DEBUG=1 LWHIP_NO_CLEAR=1
SA=${1:-600}; SB=${2:-400}; SC=${3:-300}; SP=${4:-300}
OUT=gpurun_out/guard; mkdir -p $OUT; rm -f $OUT/*.log
gcc -O2 -fPIC -shared -o $OUT/guardalloc.so tools/guard/guardalloc.c -ldl || exit 1
phase() { # name seconds env... -- command...
  local name=$1 budget=$2; shift 2
  local t0=$SECONDS n=0 fails=0
  while [ $((SECONDS - t0)) -lt $budget ]; do
    n=$((n+1)); log=$OUT/${name}_$n.log
    env "$@" > $log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "$name run $n: rc $rc $(grep -m1 -E '^(FAILED|E  )' $log | cut -c1-300)"
      grep -E 'GUARD|DIAG' $log | cut -c1-700 | head -60; else rm -f $log; fi
  done
  echo "phase $name: $fails of $n runs failed ($((SECONDS - t0)) s)" | tee -a $OUT/summary.txt
}
PYT="python3 -m pytest tests/test_hip_parity.py -m gpu -q -x -s --tb=short -p no:cacheprovider"
phase A_whole $((SA/2)) LD_PRELOAD=$PWD/$OUT/guardalloc.so $PYT
phase A_split $((SA/2)) LD_PRELOAD=$PWD/$OUT/guardalloc.so $PYT -k "ray_split or generic_tiles_in_the_fixed"
phase B_whole $((SB/2)) LWHIP_TEST_GUARD=1 $PYT
phase B_split $((SB/2)) LWHIP_TEST_GUARD=1 $PYT -k "ray_split"
phase C_control $SC python3 tools/guard/control.py
phase P_plain $SP $PYT
echo "box: $(rocm-smi --showuniqueid 2>/dev/null | grep -m1 'Unique ID:')" | tee -a $OUT/summary.txt

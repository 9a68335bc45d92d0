#!/bin/bash
# which never-written device memory reaches the rates (round 4): tests/test_hip_parity.py N times in order with only the fp64
# allocations numbered [FROM, TO) of every context cleared; failures mean the culprit is outside the range.
# usage: bash tools/clear_bisect.sh FROM TO [N=40]
R=$GRAFT_REPO_ROOT
cd $R
export LWHIP_CLEAR_FROM=$1 LWHIP_CLEAR_TO=$2
N=${3:-40}
n=0
for i in $(seq 1 $N); do
  python -m pytest tests/test_hip_parity.py -x -q -m gpu > /tmp/fl.log 2>&1
  case "$(grep -E 'passed|failed' /tmp/fl.log | tail -1)" in *failed*) n=$((n+1)); grep -E '^FAILED|beyond tol' /tmp/fl.log | cut -c1-400;; esac
done
echo "clear [$1, $2): $n failures of $N"

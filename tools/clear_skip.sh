#!/bin/bash
# the inverse probe of clear_bisect.sh: everything cleared EXCEPT the fp64 allocations numbered [FROM, TO) -- a failure proves the
# never-written memory that reaches the rates lies in that range.  usage: bash tools/clear_skip.sh FROM TO [N=40]
R=$GRAFT_REPO_ROOT
cd $R
export LWHIP_CLEAR_SKIP_FROM=$1 LWHIP_CLEAR_SKIP_TO=$2
N=${3:-40}
n=0
for i in $(seq 1 $N); do
  python -m pytest tests/test_hip_parity.py -x -q -m gpu > /tmp/fl.log 2>&1
  case "$(grep -E 'passed|failed' /tmp/fl.log | tail -1)" in *failed*) n=$((n+1));; esac
done
echo "not cleared [$1, $2): $n failures of $N"

#!/bin/bash
# PRD sub-iteration under experiment builds / knobs   usage: bash tools/prd_ab.sh "LIB ENV=.. ENV=.." ...   (LIB = base or a variant name)
R=$GRAFT_REPO_ROOT
export LWHIP_DEBUG=1
SPECS=("$@")
for rep in 1 2; do
for spec in "${SPECS[@]}"; do
  set -- $spec
  v=$1; shift
  if [ "$v" = base ]; then unset LWHIP_LIB; else export LWHIP_LIB=$R/lightweaver_amd/variants/liblwhip_$v.so; fi
  env "$@" python3 $R/bench.py --aux-only prd --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$spec', round(d['C3_prd']['ms_per_subiter'], 5))"
done
done

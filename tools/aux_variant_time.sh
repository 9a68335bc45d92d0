#!/bin/bash
# time one side measurement of bench.py (prd | columns | 2d) under several experiment builds: bash tools/aux_variant_time.sh 2d base rb8 ...
R=$GRAFT_REPO_ROOT
A=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset LWHIP_LIB; else export LWHIP_LIB=$R/lightweaver_amd/variants/liblwhip_$v.so; fi
  python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --aux-only $A 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d=d.get('aux',d); k=list(d.keys())[0]; e=d[k]; print('$v', {q:e[q] for q in e if q.startswith('ms_') or q.endswith('_per_s')})"
done

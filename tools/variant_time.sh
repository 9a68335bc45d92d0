#!/bin/bash
# time the default bench step under several experiment builds of the library: bash tools/variant_time.sh name1 name2 ...
# (lightweaver_amd/variants/liblwhip_<name>.so, built by hand with extra -D flags; "base" = the shipped library)
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset LWHIP_LIB; else export LWHIP_LIB=$R/lightweaver_amd/variants/liblwhip_$v.so; fi
  python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms/step %.4f  kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done

#!/bin/bash
# the fixed-order mode's waits taken apart (host-side switches, LWHIP_DET_NOWAIT bits: 1 no line / mixed turns, 2 no pure turns, 4 wavefront order,
# 8 no mixed-slot turns, 16 no turns in blend tiles, 32 no turns at the line slot of one-line tiles)
for v in 0 8 16 24 32 56; do
  echo "== LWHIP_DET_NOWAIT=$v"; LWHIP_DEBUG=1 LWHIP_DET_NOWAIT=$v python3 tools/det_time.py det 10240 2>&1 | grep Nlambda
done

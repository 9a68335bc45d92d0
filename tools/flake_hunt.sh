#!/bin/bash
# One box: does the deviation of rounds 3 / 4 / 5 (Gamma / Rij of the PRD atom in tests/test_hip_parity.py) show here?  N in-order runs of the
# file with nothing cleared; the test itself prints what it finds about the ORACLE's inputs when it fails (round 5: in the failing
# runs it is the oracle's result that is off, the HIP runs agree with what the oracle gives elsewhere).   usage: bash tools/flake_hunt.sh [N]
N=${1:-40}
rm -f gpurun_out/flake/summary.txt
bash tools/flake_probe.sh $N noclear LWHIP_NO_CLEAR=1 | tail -60
echo "box: $(rocm-smi --showuniqueid 2>/dev/null | grep -m1 'Unique ID:')"

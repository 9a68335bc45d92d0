#!/bin/bash
# One box: does the 1.77e-6 deviation of rounds 3 / 4 (Gamma / Rij of the PRD atom in tests/test_hip_parity.py) show here with
# nothing cleared?  If it does, the same count under the things that would tell what it is: clearing, blocking launches,
# serialised kernels, poisoned scratch.   usage: bash tools/flake_hunt.sh [N]
N=${1:-40}
rm -f gpurun_out/flake/summary.txt
bash tools/flake_probe.sh $N noclear LWHIP_NO_CLEAR=1 | tail -4
if grep -q "noclear.*: 0 of" gpurun_out/flake/summary.txt; then echo "not on this box ($(rocm-smi --showuniqueid 2>/dev/null | grep -m1 "Unique ID:"))"; exit 0; fi
bash tools/flake_probe.sh $N cleared LWHIP_DEBUG=1 | tail -3
bash tools/flake_probe.sh $N blocking LWHIP_NO_CLEAR=1 HIP_LAUNCH_BLOCKING=1 | tail -3
bash tools/flake_probe.sh $N serial LWHIP_NO_CLEAR=1 AMD_SERIALIZE_KERNEL=3 | tail -3
bash tools/flake_probe.sh $N scratch LWHIP_NO_CLEAR=1 LWHIP_SCRATCH_POISON=1 | tail -3
cat gpurun_out/flake/summary.txt

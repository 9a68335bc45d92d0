#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin): one line per kernel."""
import re
import subprocess
import sys

name = None
d = {}
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        name = m.group(1)
        d[name] = {}
        continue
    m = re.search(r'remark:\s+([A-Za-z][\w \[\]/]*?): (\S+)', l)
    if m and name:
        d[name][m.group(1).strip()] = m.group(2)
for n, v in d.items():
    dn = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'\(.*', '', dn).replace('lwhip::', '').replace('void ', '')
    print(f"{dn:58s} VGPR {v.get('VGPRs', '?'):>4} SGPRspill {v.get('SGPRs Spill', '?'):>4} VGPRspill "
          f"{v.get('VGPRs Spill', '?'):>5} scratch {v.get('ScratchSize [bytes/lane]', '?'):>5} occ {v.get('Occupancy [waves/SIMD]', '?')}")

"""Where a generic-kind result deviates: the blended problem of tests/test_hip_parity.py under split factors and the fixed-order mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
os.environ['LWHIP_DEBUG'] = '1'
import numpy as np
import torch
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models
from lightweaver_amd import _abi as abi
from oracle.bindings import OracleContext
from test_fs2d import blended_atoms

prob = models.build_problem(models.falc82(), blended_atoms(2, 0.3), Nrays=3)
lines = np.zeros(prob.Nlambda, dtype=int); conts = np.zeros(prob.Nlambda, dtype=int)
for a in prob.atoms:
    for t in a.trans:
        (lines if t.type == abi.LINE else conts)[t.Nblue:t.Nred] += 1
q = prob.copy()
with OracleContext(q) as oc:
    q.gamma_prefill(); oc.formal_sol_gamma_matrices()
if os.environ.get('LWHIP_GEN_DIAG_MARCH'):
    os.environ['LWHIP_SWEEP'] = 'march'
    for n in (2, 3):
        pm = models.build_problem(models.falc82(), blended_atoms(n, 0.3), Nrays=3)
        with Context(pm) as ctx:
            pm.gamma_prefill(); ctx.formal_sol_gamma_matrices(); ctx.stat_equil()
    del os.environ['LWHIP_SWEEP']
for split, det in (('1', False), ('2', False), ('4', False), ('0', True)):
    if split != '0': os.environ['LWHIP_LANE_SPLIT'] = split
    else: os.environ.pop('LWHIP_LANE_SPLIT', None)
    p = prob.copy()
    with Context(p, deterministic=det) as ctx:
        p.gamma_prefill(); ctx.formal_sol_gamma_matrices()
    eJ = np.abs(p.J - q.J) / np.abs(q.J)
    la = np.unique(np.nonzero(eJ > 1e-9)[0])
    print(f'split {split} det {det}: J max rel err {eJ.max():.3e} at {len(la)} wavelengths {la[:20].tolist()} lines there {lines[la[:20]].tolist()} conts {conts[la[:20]].tolist()}')
    if len(la):
        bad = set(la.tolist())
        print('   la: ' + ' '.join(f"{l}{'*' if l in bad else ''}({lines[l]},{conts[l]})" for l in range(max(la[0] - 6, 0), min(la[-1] + 7, prob.Nlambda))))
        l0 = la[0]
        print(f'   J computed / J oracle at la {l0} by depth: ' + ' '.join(f'{v:.3g}' for v in (p.J[l0] / q.J[l0])))
        l1 = la[-1]
        print(f'   J computed / J oracle at la {l1} by depth: ' + ' '.join(f'{v:.3g}' for v in (p.J[l1] / q.J[l1])))
    for ia, (x, y) in enumerate(zip(p.atoms, q.atoms)):
        for kr, (tx, ty) in enumerate(zip(x.trans, y.trans)):
            e = max(np.max(np.abs(tx.Rij - ty.Rij) / np.abs(ty.Rij)), np.max(np.abs(tx.Rji - ty.Rji) / np.abs(ty.Rji)))
            if e > 1e-9:
                print(f'   atom {ia} trans {kr} type {tx.type} {tx.i}->{tx.j} [{tx.Nblue},{tx.Nred}) err {e:.3e}')

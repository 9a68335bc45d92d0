"""Hunt for the rare deviation of the PRD problem's Ca II rates (1.77e-6 in Rij / Gamma of atom 1, J and I fine) seen in
tests/test_hip_parity.py::test_lane_sweep_ray_split_vs_oracle[True-*]: the test's own sequence of contexts, many times in
one process, every run compared transition by transition with the oracle.  python3 tests/diag/prd_rates_flake.py [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models
from oracle.bindings import OracleContext

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
os.environ['LWHIP_SWEEP'] = 'lanes'
atmos = models.perturbed(models.falc82(), seed=7)
plain = models.build_problem(atmos, [models.H_6(0.5), models.CaII_6(0.5)], Nrays=5)
prd = models.falc_h_ca(Nrays=5, lineScale=0.4, prd=True, atmos=atmos)
q = prd.copy(); q.gamma_prefill(); oc = OracleContext(q); oc.formal_sol_gamma_matrices()
want_fs = [t.Rij.copy() for t in q.atoms[1].trans]
wantG = q.atoms[1].Gamma.copy()
oc.redistribute_prd(2, 0.0)
want = [t.Rij.copy() for t in q.atoms[1].trans]

def rel(a, b):
    d = np.abs(a - b); s = np.abs(b).max()
    return d.max() / s if s > 0 else d.max()

bad = 0
for rep in range(reps):
    for prob, nprd in ((plain, 0), (plain, 0), (plain, 0), (prd, 2), (prd, 2), (prd, 2)):
        os.environ['LWHIP_LANE_SPLIT'] = str((rep % 3 == 0) * 3 + 1 if False else (1, 2, 4)[rep % 3])
        p = prob.copy()
        with Context(p) as ctx:
            p.gamma_prefill()
            ctx.formal_sol_gamma_matrices()
            if nprd:
                rij_fs = [t.Rij.copy() for t in p.atoms[1].trans]
                G_fs = p.atoms[1].Gamma.copy()
                ctx.redistribute_prd(nprd, 0.0)
        if nprd:
            errs_fs = [rel(a, b) for a, b in zip(rij_fs, want_fs)]
            errs = [rel(t.Rij, w) for t, w in zip(p.atoms[1].trans, want)]
            eG = rel(G_fs, wantG)
            if max(errs) > 1e-9 or max(errs_fs) > 1e-9 or eG > 1e-9:
                bad += 1
                k = int(np.argmax(errs_fs)) if max(errs_fs) > 1e-9 else int(np.argmax(errs))
                t = p.atoms[1].trans[k]
                d = np.abs(rij_fs[k] - want_fs[k]) / np.abs(want_fs[k]).max()
                print(f'rep {rep} split {os.environ["LWHIP_LANE_SPLIT"]}: after formal solution Rij err {["%.2e" % e for e in errs_fs]} Gamma {eG:.2e}; '
                      f'after PRD {["%.2e" % e for e in errs]}; worst transition {k} (type {t.type}, i {t.i}, j {t.j}, Nblue {t.Nblue}, Nred {t.Nred}) '
                      f'depths with err > 1e-9: {np.nonzero(d > 1e-9)[0].tolist()[:20]} max at {int(np.argmax(d))}', flush=True)
print(f'{bad} deviating runs of {reps * 3}')

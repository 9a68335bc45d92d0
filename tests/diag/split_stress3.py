"""As the test file runs them: the prd=False case's three contexts, then the prd=True S=2 context, compared with the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models
from oracle.bindings import OracleContext
os.environ['LWHIP_SWEEP'] = 'lanes'
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
atmos = models.perturbed(models.falc82(), seed=7)
probP = models.falc_h_ca(Nrays=5, lineScale=0.4, prd=True, atmos=atmos)
probN = models.build_problem(atmos, [models.H_6(0.5), models.CaII_6(0.5)], Nrays=5)
q = probP.copy(); q.gamma_prefill()
oc = OracleContext(q); oc.formal_sol_gamma_matrices(); oc.redistribute_prd(2, 0.0)
def run(prob, S, prd):
    os.environ['LWHIP_LANE_SPLIT'] = str(S)
    p = prob.copy()
    with Context(p) as ctx:
        p.gamma_prefill(); ctx.formal_sol_gamma_matrices()
        if prd:
            ctx.redistribute_prd(2, 0.0)
    return p
nbad = 0
for i in range(N):
    for S in (2, 2, 1):
        run(probN, S, False)
    for S in (2, 2, 1):
        p = run(probP, S, True)
        for ia, (a, b) in enumerate(zip(p.atoms, q.atoms)):
            for it, (ta, tb) in enumerate(zip(a.trans, b.trans)):
                for nm in ('Rij', 'Rji'):
                    x, y = getattr(ta, nm), getattr(tb, nm)
                    m = np.max(np.abs(y))
                    if m > 0 and (np.abs(x - y) / m).max() > 1e-9:
                        nbad += 1
                        e = np.abs(x - y) / m
                        k = int(np.argmax(e))
                        print('iter', i, 'S', S, 'atom', ia, 'trans', it, 'type', ta.type, nm, 'err', float(e.max()), 'k', k, 'n bad k', int((e > 1e-9).sum()), 'prd' if ta.rhoPrd is not None else '', flush=True)
print('done', N, 'bad', nbad)

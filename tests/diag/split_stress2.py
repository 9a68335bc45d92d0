"""The sequence of test_lane_sweep_ray_split_vs_oracle[True-2], many times, against the oracle: where does it deviate?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models
from oracle.bindings import OracleContext
os.environ['LWHIP_SWEEP'] = 'lanes'
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
atmos = models.perturbed(models.falc82(), seed=7)
prob = models.falc_h_ca(Nrays=5, lineScale=0.4, prd=True, atmos=atmos)
q = prob.copy(); q.gamma_prefill()
oc = OracleContext(q); oc.formal_sol_gamma_matrices(); oc.redistribute_prd(2, 0.0)
def run(S):
    os.environ['LWHIP_LANE_SPLIT'] = str(S)
    p = prob.copy()
    with Context(p) as ctx:
        p.gamma_prefill(); ctx.formal_sol_gamma_matrices(); ctx.redistribute_prd(2, 0.0)
    return p
nbad = 0
for i in range(N):
    for S in (2, 2, 1, 4):
        p = run(S)
        for ia, (a, b) in enumerate(zip(p.atoms, q.atoms)):
            for it, (ta, tb) in enumerate(zip(a.trans, b.trans)):
                for nm in ('Rij', 'Rji'):
                    x, y = getattr(ta, nm), getattr(tb, nm)
                    m = np.max(np.abs(y))
                    if m > 0:
                        e = np.abs(x - y) / m
                        if e.max() > 1e-9:
                            nbad += 1
                            k = int(np.argmax(e))
                            print('iter', i, 'S', S, 'atom', ia, 'trans', it, nm, 'err', float(e.max()), 'at k', k, 'got', x[k], 'want', y[k], 'prd' if ta.rhoPrd is not None else '', flush=True)
print('done', N, 'bad', nbad)

"""Run-to-run stability of Gamma / rates of the lane sweep under split factors, with and without PRD (diagnosis)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models
os.environ['LWHIP_SWEEP'] = 'lanes'
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
atmos = models.perturbed(models.falc82(), seed=7)
def run(prob, nprd, S):
    os.environ['LWHIP_LANE_SPLIT'] = str(S)
    p = prob.copy()
    with Context(p) as ctx:
        p.gamma_prefill()
        ctx.formal_sol_gamma_matrices()
        mid = [t.Rij.copy() for a in p.atoms for t in a.trans]
        if nprd:
            ctx.redistribute_prd(nprd, 0.0)
    return p, mid
def worst(a, b):
    w = 0.0
    for x, y in zip(a, b):
        m = np.max(np.abs(y))
        if m > 0:
            w = max(w, float(np.max(np.abs(x - y)) / m))
    return w
for prd in (True, False):
    prob = models.falc_h_ca(Nrays=5, lineScale=0.4, prd=prd, atmos=atmos)
    for S in (2, 1, 4):
        ref, refMid = run(prob, 2 if prd else 0, S)
        refR = [t.Rij for a in ref.atoms for t in a.trans]
        bad = []
        for i in range(N):
            q, mid = run(prob, 2 if prd else 0, S)
            e1 = worst(mid, refMid)
            e2 = worst([t.Rij for a in q.atoms for t in a.trans], refR)
            if e1 > 1e-9 or e2 > 1e-9:
                bad.append((i, e1, e2))
        print('prd', prd, 'S', S, 'runs', N, 'bad', len(bad), bad[:5], flush=True)

"""Where do two runs of the lane sweep with split rays differ?  (diagnosis)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models
os.environ['LWHIP_SWEEP'] = 'lanes'
S = sys.argv[1] if len(sys.argv) > 1 else '2'
os.environ['LWHIP_LANE_SPLIT'] = S
atmos = models.perturbed(models.falc82(), seed=7)
prob = models.falc_h_ca(Nrays=5, lineScale=0.4, prd=True, atmos=atmos)
def run(nprd):
    p = prob.copy()
    with Context(p) as ctx:
        p.gamma_prefill()
        ctx.formal_sol_gamma_matrices()
        if nprd:
            ctx.redistribute_prd(nprd, 0.0)
    return p
for nprd in (0, 1, 2):
    ref = run(nprd)
    for rep in range(4):
        q = run(nprd)
        dJ = np.argwhere(ref.J != q.J)
        dI = np.argwhere(ref.I != q.I)
        print('S', S, 'nprd', nprd, 'rep', rep, 'J differs at', len(dJ), 'points', dJ[:6].tolist(), 'max rel', float(np.max(np.abs(ref.J - q.J) / np.abs(ref.J))), 'I differs', len(dI))
print('--- across split factors, after 1 PRD sub-iteration')
res = {}
for s in ('1', '2', '4'):
    os.environ['LWHIP_LANE_SPLIT'] = s
    res[s] = run(1)
for s in ('2', '4'):
    d = np.abs(res[s].I - res['1'].I) / np.abs(res['1'].I)
    bad = np.argwhere(d > 1e-9)
    print('S', s, 'vs 1: I differs (>1e-9) at', len(bad), 'entries; wavelengths', sorted(set(bad[:, 0].tolist()))[:40], 'max', float(d.max()))
    dj = np.abs(res[s].J - res['1'].J) / np.abs(res['1'].J)
    print('   J max rel', float(dj.max()))

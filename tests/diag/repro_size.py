"""Diagnosis: the lane sweep on resampled FAL-C columns of several depths in ONE process (python repro_size.py Ns:Nrays ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch; torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models
os.environ.setdefault('LWHIP_SWEEP', 'lanes')
for spec in sys.argv[1:]:
    Ns, Nr = (int(x) for x in spec.split(':'))
    atmos = models.resample(models.falc82(), Ns)
    prob = models.build_problem(atmos, [models.H_6(0.2)], Nrays=Nr)
    p = prob.copy()
    with Context(p) as ctx:
        print(spec, 'kind', ctx.sweep_kind(), 'Nla', p.Nlambda, flush=True)
        ctx.formal_sol_gamma_matrices()
        print(spec, 'fs ok', flush=True)

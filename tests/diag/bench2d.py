"""BASELINE configs[4] in miniature: a 256 x 82 x-periodic atmosphere built from perturbed FAL-C columns, H active,
2D BESSER short characteristics.  Times formal_sol_gamma_matrices (device resident) and, with --ref, the reference
core (oracle/_ref) on the same problem."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models

Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ls = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
base = models.falc82()
cols = [models.perturbed(base, seed=1234 + j) for j in range(Nx)]
x = np.linspace(0.0, 25.0e3 * (Nx - 1), Nx)
t0 = time.perf_counter()
prob = models.build_problem_2d(cols, x, [models.H_6(ls)])   # geometry: lwhip_build_intersections
t1 = time.perf_counter()
print(f'Nx={Nx} Nz={prob.grid2d.Nz} Nspace={prob.Nspace} Nlambda={prob.Nlambda} rays={prob.Nrays}x2 '
      f'NlongChar={prob.grid2d.substepOff.size - 1}; host build {t1 - t0:.1f}s')
with Context(prob) as ctx:
    ctx.formal_sol_gamma_matrices(deviceResident=True)
    torch.cuda.synchronize()
    n = 5
    t2 = time.perf_counter()
    for _ in range(n):
        up = ctx.formal_sol_gamma_matrices(deviceResident=True)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f'device: {1e3 * (t3 - t2) / n:.2f} ms per iteration ({n / (t3 - t2):.1f} it/s), dJMax {up.dJMax:.3e}')
if '--ref' in sys.argv:   # CPU baseline leg: the reference core through the test infrastructure
    from oracle import bindings
    q = prob.copy()
    rc = bindings.RefContext(q, Nthreads=min(os.cpu_count() or 1, 64))
    q.gamma_prefill(); rc.formal_sol_gamma_matrices()
    t4 = time.perf_counter()
    q.gamma_prefill(); rc.formal_sol_gamma_matrices()
    t5 = time.perf_counter()
    print(f'reference core (scalar scheme, {min(os.cpu_count() or 1, 64)} threads): {1e3 * (t5 - t4):.0f} ms per iteration')

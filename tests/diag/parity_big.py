"""Where do the HIP path, the C oracle and the reference core (scalar / SIMD scheme) differ on the bench grid?
usage (GPU box): python tools/parity_big.py [Nlambda]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models
from oracle import bindings

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
prob = models.throughput_grid(NlambdaTarget=n, Nrays=5)
runs = {}
g = prob.copy()
with Context(g) as c:
    c.formal_sol_gamma_matrices()
runs['hip'] = g
o = prob.copy(); o.gamma_prefill()
bindings.OracleContext(o).formal_sol_gamma_matrices(Nthreads=32)
runs['oracle'] = o
if bindings.ref_available():
    r = prob.copy(); r.gamma_prefill()
    bindings.RefContext(r, Nthreads=32).formal_sol_gamma_matrices()
    runs['ref_scalar'] = r
    for isa in ('AVX512', 'AVX2FMA'):
        if bindings.simd_plugin(isa):
            s = prob.copy(); s.gamma_prefill()
            bindings.RefContext(s, scheme=bindings.simd_plugin(isa), Nthreads=32).formal_sol_gamma_matrices()
            runs['ref_' + isa] = s

def worst(a, b):
    nz = b != 0
    e = np.zeros_like(b)
    e[nz] = np.abs(a[nz] - b[nz]) / np.abs(b[nz])
    i = np.unravel_index(np.argmax(e), e.shape)
    return float(e[i]), tuple(int(x) for x in i), float(a[i]), float(b[i])

names = list(runs)
for i, x in enumerate(names):
    for y in names[i + 1:]:
        a, b = runs[x], runs[y]
        print(f'{x} vs {y}: J {worst(a.J, b.J)}  I {worst(a.I, b.I)[:2]}  '
              f'Gamma0 {worst(a.atoms[0].Gamma, b.atoms[0].Gamma)[:2]} Gamma1 {worst(a.atoms[1].Gamma, b.atoms[1].Gamma)[:2]}')
# structure of the wavelength with the worst J difference hip vs oracle
e, idx, _, _ = worst(runs['hip'].J, runs['oracle'].J)
la = idx[0]
act = [(ia, kr, t.type, t.Nblue, t.Nred) for ia, a in enumerate(prob.atoms) for kr, t in enumerate(a.trans) if t.Nblue <= la < t.Nred]
print('worst J wavelength', la, prob.wavelength[la], 'active', act)
d = np.abs(runs['hip'].J - runs['oracle'].J) / np.abs(runs['oracle'].J)
bad = np.where(d.max(axis=1) > 1e-9)[0]
print('wavelengths with J rel err > 1e-9:', len(bad), bad[:40])
print('depth profile of the error at worst la:', np.array2string(d[la], precision=2))

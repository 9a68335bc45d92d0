"""Debug: where does Gamma differ from the golden iter1 vectors? (GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from helpers import FIXTURES, load_fixture, variant_problem
from lightweaver_amd.context import Context
name = sys.argv[1] if len(sys.argv) > 1 else FIXTURES[0]
base, d = load_fixture(name)
p = variant_problem(base, d, 'iter1')
with Context(p, lib_path=os.environ.get("LWHIP_LIB")) as ctx:
    ctx.formal_sol_gamma_matrices()
for ia, a in enumerate(p.atoms):
    if a.detailed:
        continue
    G = a.Gamma
    ref = d[f'out/iter1/Gamma{ia}']
    err = np.abs(G - ref) / np.maximum(np.abs(ref), 1e-300)
    idx = np.argwhere(err > 1e-9)
    print('atom', ia, 'entries beyond 1e-9:', len(idx), 'of', err.size)
    pairs = {}
    for to, fr, k in idx:
        pairs.setdefault((int(to), int(fr)), []).append(int(k))
    for (to, fr), ks in sorted(pairs.items()):
        print('  Gamma[%d,%d]: %d depths, k in [%d, %d], max err %.3e' % (to, fr, len(ks), min(ks), max(ks), err[to, fr].max()))

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests'))
import numpy as np
from helpers import load_fixture, variant_problem, rel_err
from lightweaver_amd.context import Context
base, d = load_fixture('falc_h_ca_small')
p = variant_problem(base, d, 'iter1')
with Context(p) as ctx:
    ctx.formal_sol_gamma_matrices()
Jg = d['out/iter1/J']
err = np.abs(p.J/Jg-1)
bad = np.argwhere(err > 1e-9)
print('G env', os.environ.get('LWHIP_G'), 'max err', err.max(), 'nbad', len(bad), 'first bad', bad[:5].tolist(), 'I err', rel_err(p.I, d['out/iter1/I']))
if len(bad):
    la,k = bad[0]; print(p.J[la,k], Jg[la,k], 'bad la unique', np.unique(bad[:,0])[:20], 'bad k unique', np.unique(bad[:,1])[:20])

"""Generate lightweaver_amd/harness/data/falc82.npz -- the FAL-C 82-point model atmosphere table.

Runs ONLY in the build container (needs /root/reference).  The table (Fontenla, Avrett &
Loeser 1993, model C) is read as DATA out of the reference's lightweaver/fal.py by evaluating
the numeric array literals `cmass, temp, ne, vel, vturb, nh` found there (fal.py:8-430); no
reference code is imported, executed beyond those literals, or copied.  The committed .npz is an
input fixture: column mass [g cm^-2], T [K], ne [cm^-3], vel [km/s], vturb [km/s], nh[6,82] [cm^-3].

    python lightweaver_amd/harness/data/make_falc_table.py
"""
import ast
import os

import numpy as np

SRC = '/root/reference/lightweaver/fal.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'falc82.npz')
WANT = ['cmass', 'temp', 'ne', 'vel', 'vturb', 'nh']


def main():
    tree = ast.parse(open(SRC).read())
    env = {'np': np}
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 \
                and isinstance(node.targets[0], ast.Name) and node.targets[0].id in WANT:
            code = compile(ast.Module(body=[node], type_ignores=[]), SRC, 'exec')
            exec(code, env)
    data = {k: np.ascontiguousarray(env[k], dtype=np.float64) for k in WANT}
    assert data['temp'].shape == (82,) and data['nh'].shape == (6, 82)
    np.savez_compressed(OUT, **data)
    print('wrote', OUT, {k: v.shape for k, v in data.items()})


if __name__ == '__main__':
    main()

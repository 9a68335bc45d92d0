"""Flat (de)serialisation of a `Problem` to a dict of numpy arrays (np.savez-able).

The host arrays are the source of truth for a context at all times (as in the reference, where
pickling a Context re-creates all plugin scratch: Source/LwMiddleLayer.pyx:2977-3037), so saving a
Problem is a complete checkpoint of the hot path's state; the golden fixtures under tests/golden/
use the same format.
"""
import numpy as np

from . import _abi as abi
from .model import AtomData, Boundary, Problem, TransitionData

_T_SCALARS = ['type', 'i', 'j', 'Nblue', 'Nred', 'lambda0', 'Aji', 'Bji', 'Bij', 'dopplerWidth']
_T_ARRAYS = ['wavelength', 'alpha', 'phi', 'wphi', 'aDamp', 'rhoPrd', 'Qelast']


def problem_to_dict(prob: Problem, include_phi=True) -> dict:
    d = {}
    for k in ['height', 'temperature', 'muz', 'wmu', 'wavelength', 'bgChi', 'bgEta', 'bgSca',
              'vlosMu', 'J']:
        d[k] = getattr(prob, k)
    d['meta'] = np.array([prob.formalSolver, int(prob.storeDepthData), len(prob.atoms),
                          prob.zLowerBc.type, prob.zUpperBc.type], dtype=np.int64)
    for name, bc in (('zLowerBc', prob.zLowerBc), ('zUpperBc', prob.zUpperBc)):
        if bc.type == abi.BC_CALLABLE:
            d[f'{name}/idxs'] = bc.idxs
            d[f'{name}/bcData'] = bc.bcData
    if getattr(prob, 'grid2d', None) is not None:
        d.update(prob.grid2d.to_dict('grid2d/'))
    for ia, a in enumerate(prob.atoms):
        p = f'atom{ia}'
        d[f'{p}/meta'] = np.array([a.Nlevel, int(a.detailed), len(a.trans)], dtype=np.int64)
        d[f'{p}/name'] = np.array(a.name)
        for k in ['n', 'nStar', 'nTotal', 'vBroad']:
            d[f'{p}/{k}'] = getattr(a, k)
        if not a.detailed:
            d[f'{p}/C'] = a.C
        for kr, t in enumerate(a.trans):
            q = f'{p}/trans{kr}'
            d[f'{q}/scalars'] = np.array([float(getattr(t, s)) for s in _T_SCALARS])
            for k in _T_ARRAYS:
                v = getattr(t, k)
                if v is None or (k == 'phi' and not include_phi):
                    continue
                d[f'{q}/{k}'] = v
    return d


def problem_from_dict(d) -> Problem:
    meta = d['meta']
    atoms = []
    for ia in range(int(meta[2])):
        p = f'atom{ia}'
        Nlevel, detailed, Ntrans = (int(x) for x in d[f'{p}/meta'])
        trans = []
        for kr in range(Ntrans):
            q = f'{p}/trans{kr}'
            s = d[f'{q}/scalars']
            kw = {name: (int(round(v)) if name in ('type', 'i', 'j', 'Nblue', 'Nred') else float(v))
                  for name, v in zip(_T_SCALARS, s)}
            for k in _T_ARRAYS:
                if f'{q}/{k}' in d:
                    kw[k] = np.array(d[f'{q}/{k}'], dtype=np.float64)
            trans.append(TransitionData(**kw))
        atoms.append(AtomData(name=str(d[f'{p}/name']), Nlevel=Nlevel,
                              n=np.array(d[f'{p}/n']), nStar=np.array(d[f'{p}/nStar']),
                              nTotal=np.array(d[f'{p}/nTotal']), vBroad=np.array(d[f'{p}/vBroad']),
                              trans=trans, detailed=bool(detailed),
                              C=np.array(d[f'{p}/C']) if f'{p}/C' in d else None))
    bcs = {}
    for name, typ in (('zLowerBc', int(meta[3])), ('zUpperBc', int(meta[4]))):
        b = Boundary(type=typ)
        if typ == abi.BC_CALLABLE:
            b.idxs = np.array(d[f'{name}/idxs'], dtype=np.int32)
            b.bcData = np.array(d[f'{name}/bcData'], dtype=np.float64)
        bcs[name] = b
    grid = None
    if 'grid2d/x' in d:
        from .grid2d import Grid2d
        grid = Grid2d.from_dict(d, 'grid2d/')
    prob = Problem(grid2d=grid, height=d['height'], temperature=d['temperature'], muz=d['muz'], wmu=d['wmu'],
                   wavelength=d['wavelength'], bgChi=d['bgChi'], bgEta=d['bgEta'],
                   bgSca=d['bgSca'], atoms=atoms, vlosMu=d['vlosMu'], J=np.array(d['J']),
                   formalSolver=int(meta[0]), zLowerBc=bcs['zLowerBc'], zUpperBc=bcs['zUpperBc'],
                   storeDepthData=bool(meta[1]))
    prob.gamma_prefill()
    return prob


def save_problem(path, prob: Problem, **kw):
    np.savez_compressed(path, **problem_to_dict(prob, **kw))


def load_problem(path) -> Problem:
    with np.load(path) as z:
        return problem_from_dict({k: z[k] for k in z.files})

"""Shared test helpers: golden-fixture loading, variant construction, error metrics."""
import os

import numpy as np

from lightweaver_amd import _abi as abi
from lightweaver_amd.io import problem_from_dict
from lightweaver_amd.model import Boundary, Problem

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FIXTURES = ['falc_h_ca_small', 'falc_h_vel']

# Stated fp64 tolerances: relative, element-wise, against the reference's scalar
# scheme after one call on identical inputs; 1e-6 on the converged J.
TOL_ONE_CALL = 1e-9
TOL_CONVERGED = 1e-6


def load_fixture(name):
    with np.load(os.path.join(GOLDEN, name + '.npz')) as z:
        d = {k: z[k] for k in z.files}
    return problem_from_dict(d), d


def rebuild(p: Problem, **kw) -> Problem:
    args = dict(height=p.height, temperature=p.temperature, muz=p.muz, wmu=p.wmu,
                wavelength=p.wavelength, bgChi=p.bgChi, bgEta=p.bgEta, bgSca=p.bgSca,
                atoms=p.atoms, vlosMu=p.vlosMu, J=p.J, formalSolver=p.formalSolver,
                zLowerBc=p.zLowerBc, zUpperBc=p.zUpperBc, storeDepthData=p.storeDepthData)
    args.update(kw)
    q = Problem(**args)
    q.gamma_prefill()
    return q


def variant_problem(base: Problem, d, variant) -> Problem:
    """The input problem of a golden variant (see tests/golden/make_golden.py)."""
    p = base.copy()
    if variant == 'linear':
        p.formalSolver = abi.FS_LINEAR_1D
    elif variant == 'besser':
        p.formalSolver = abi.FS_BESSER_1D
    elif variant == 'bc':
        p.zLowerBc = Boundary(abi.BC_CALLABLE, idxs=d['in/bc/idxs'], bcData=d['in/bc/bcData'])
        p.zUpperBc = Boundary(abi.BC_THERMALISED)
    elif variant == 'prd_detailed':
        ca = p.atoms[1]
        ca.detailed = True
        for kr in range(2):
            ca.trans[kr].rhoPrd = d[f'in/prd/rho{kr}']
        p = rebuild(p, storeDepthData=False)
    p.gamma_prefill()
    return p


def rel_err(a, b):
    """max element-wise |a-b|/|b| (elements where the reference is exactly 0 must match exactly)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    nz = b != 0.0
    err = 0.0
    if nz.any():
        err = float(np.max(np.abs(a[nz] - b[nz]) / np.abs(b[nz])))
    if (~nz).any():
        err = max(err, float(np.max(np.abs(a[~nz]))))
    return err


def collect(prob, what=('J', 'I', 'Gamma', 'R', 'n')):
    out = {}
    if 'J' in what:
        out['J'] = prob.J
    if 'I' in what:
        out['I'] = prob.I
    for ia, a in enumerate(prob.atoms):
        if 'Gamma' in what and not a.detailed:
            out[f'Gamma{ia}'] = a.Gamma
        if 'n' in what:
            out[f'n{ia}'] = a.n
        if 'R' in what:
            out[f'Rij{ia}'] = np.stack([t.Rij for t in a.trans])
            out[f'Rji{ia}'] = np.stack([t.Rji for t in a.trans])
    return out


def compare_to_golden(prob, d, variant, keys=None, tol=TOL_ONE_CALL):
    got = collect(prob)
    worst = {}
    for k, v in got.items():
        gk = f'out/{variant}/{k}'
        if gk not in d or (keys is not None and k not in keys and k.rstrip('0123456789') not in keys):
            continue
        worst[k] = rel_err(v, d[gk])
    bad = {k: e for k, e in worst.items() if not e <= tol}
    assert not bad, f'{variant}: beyond tol {tol}: {bad} (all: {worst})'
    return worst


def gamma_term_scale(atom):
    """Cancellation-aware scale of every Gamma element of an active atom, from the REFERENCE side's own outputs: an
    off-diagonal element is the collisional pre-fill plus, per transition between the two levels, the integrals
    sum (Uji + Vji I) w = Rji (element (i, j)) / sum Vij I w = Rij (element (j, i)) minus preconditioning terms of at
    most that size -- a sum of terms of both signs whose magnitude is |C| + R, not |Gamma|.  The diagonal is minus the
    column sum (finalise_Gamma, SimdFullIterationTemplates.hpp:491-508).  |dGamma| <= tol * scale is the bound a sum
    formed in a different order can be held to; |dGamma| <= tol * |Gamma| is not where the terms cancel."""
    Nl = atom.Nlevel
    S = np.abs(np.asarray(atom.C, dtype=np.float64)) if atom.C is not None else np.zeros_like(atom.Gamma)
    S = S.copy().reshape(Nl, Nl, -1)
    for t in atom.trans:
        S[t.i, t.j] += np.abs(t.Rji)
        S[t.j, t.i] += np.abs(t.Rij)
    for i in range(Nl):
        S[i, i] = 0.0
    for i in range(Nl):
        S[i, i] = S[:, i].sum(axis=0)
    return S.reshape(atom.Gamma.shape)


def gamma_err_scaled(got, want_atom):
    """max |Gamma_got - Gamma_ref| / scale (gamma_term_scale of the reference atom)."""
    S = gamma_term_scale(want_atom)
    d = np.abs(np.asarray(got, dtype=np.float64) - want_atom.Gamma)
    nz = S > 0.0
    assert np.all(d[~nz] == 0.0)
    return float(np.max(d[nz] / S[nz])) if nz.any() else 0.0


class ReadOnlyInputs:
    """DIAGNOSIS (LWHIP_TEST_GUARD=1): every INPUT array of a problem moved into pages of its own and
    made read-only for the lifetime of this object.  A CPU store into one of them dies with SIGSEGV at the store (pytest's
    faulthandler and tools/guard/guardalloc.so print the stack); a device / SDMA write goes past the CPU's page protection,
    so `changed()` naming an array with no fault in between means the writer is not a CPU thread."""
    OUTPUTS = ('J', 'I', 'Rij', 'Rji', 'Gamma', 'n', 'rhoPrd', 'depthChi', 'depthEta', 'depthI')

    def __init__(self, prob):
        import ctypes
        import mmap
        self._libc = ctypes.CDLL(None, use_errno=True)
        self._maps, self._saved = [], {}
        seen = set()

        def walk(obj, path):
            if id(obj) in seen or obj is None:
                return
            seen.add(id(obj))
            if isinstance(obj, (list, tuple)):
                for i, v in enumerate(obj):
                    if isinstance(v, np.ndarray):
                        continue    # (arrays held in lists are not descriptor inputs of Problem)
                    walk(v, f'{path}[{i}]')
            elif hasattr(obj, '__dict__'):
                for k, v in list(vars(obj).items()):
                    if k == '_keepalive':
                        continue
                    if isinstance(v, np.ndarray):
                        if k in self.OUTPUTS or v.nbytes == 0 or not v.flags.c_contiguous:
                            continue
                        n = (v.nbytes + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE
                        mm = mmap.mmap(-1, n)
                        w = np.frombuffer(mm, dtype=v.dtype, count=v.size).reshape(v.shape)
                        w[...] = v
                        setattr(obj, k, w)
                        self._maps.append((f'{path}.{k}', mm, w, n))
                        self._saved[f'{path}.{k}'] = v.copy()
                    else:
                        walk(v, f'{path}.{k}')
        walk(prob, 'p')
        self._protect(mmap.PROT_READ)

    def _protect(self, prot):
        import ctypes
        for name, mm, w, n in self._maps:
            rc = self._libc.mprotect(ctypes.c_void_p(w.ctypes.data), ctypes.c_size_t(n), ctypes.c_int(prot))
            assert rc == 0, (name, ctypes.get_errno())

    def changed(self):
        """Names (+ element indices, old and new bytes) of the guarded arrays whose content is no longer what was put there."""
        out = []
        for name, mm, w, n in self._maps:
            old = self._saved[name]
            if not np.array_equal(w, old, equal_nan=True):
                idx = np.nonzero(w.ravel() != old.ravel())[0]
                out.append((name, idx[:8].tolist(), [w.ravel()[i].tobytes().hex() for i in idx[:8]],
                            [old.ravel()[i].tobytes().hex() for i in idx[:8]], hex(w.ctypes.data)))
        return out

    def release(self):
        import mmap
        self._protect(mmap.PROT_READ | mmap.PROT_WRITE)

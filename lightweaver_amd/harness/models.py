"""Harness model builders: FAL-C atmosphere, a 6-level H and a 6-level Ca II model atom, the
global wavelength grid, LTE populations, collisional rates, damping, background -> `Problem`.

Host-side input producers only (see physics.py).  The model atoms are authored here because the
reference's `rh_atoms.py` is absent from the mount; level energies, oscillator
strengths and threshold cross-sections are standard literature values (NIST ASD / Kramers),
good to the accuracy a throughput-and-parity harness needs.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from .. import _abi as abi
from ..model import AtomData, Problem, TransitionData
from . import physics as ph

_HERE = os.path.dirname(os.path.abspath(__file__))
FALC_FIXTURE = os.path.join(_HERE, 'data', 'falc82.npz')


# ------------------------------------------------------------------------------------------------
# atmosphere
# ------------------------------------------------------------------------------------------------
@dataclass
class Atmosphere1d:
    height: np.ndarray       # [m], index 0 = top
    temperature: np.ndarray  # [K]
    ne: np.ndarray           # [m^-3]
    vturb: np.ndarray        # [m/s]
    vlos: np.ndarray         # [m/s]
    nHTot: np.ndarray        # [m^-3]
    nH: np.ndarray           # [6, Nspace] hydrogen level populations [m^-3]

    @property
    def Nspace(self):
        return self.height.shape[0]


MASS_PER_H = 1.4271  # amu per hydrogen nucleus for a solar mixture


def falc82() -> Atmosphere1d:
    """FAL-C on its native 82 points (lightweaver/fal.py:8-431 via harness/data/falc82.npz, written by harness/data/make_falc_table.py)."""
    d = np.load(FALC_FIXTURE)
    cmass = d['cmass'] * 1e-3 / ph.CM_TO_M**2          # kg m^-2
    nH = d['nh'] / ph.CM_TO_M**3
    nHTot = nH.sum(axis=0)
    rho = ph.Amu * MASS_PER_H * nHTot
    height = np.zeros(82)
    for k in range(1, 82):
        height[k] = height[k - 1] - 2.0 * (cmass[k] - cmass[k - 1]) / (rho[k - 1] + rho[k])
    return Atmosphere1d(height=height, temperature=d['temp'].copy(), ne=d['ne'] / ph.CM_TO_M**3,
                        vturb=d['vturb'] * ph.KM_TO_M, vlos=d['vel'] * ph.KM_TO_M,
                        nHTot=nHTot, nH=nH)


def resample(atmos: Atmosphere1d, Nspace: int) -> Atmosphere1d:
    """Log-linear resampling in column-mass-like coordinate to Nspace points (the reference's
    benchmark resamples FAL-C to 500 points, lightweaver/benchmark.py:19-45)."""
    x = np.linspace(0.0, 1.0, atmos.Nspace)
    xn = np.linspace(0.0, 1.0, Nspace)
    li = lambda a: np.interp(xn, x, a)
    lg = lambda a: np.exp(np.interp(xn, x, np.log(a)))
    return Atmosphere1d(height=li(atmos.height), temperature=lg(atmos.temperature),
                        ne=lg(atmos.ne), vturb=li(atmos.vturb), vlos=li(atmos.vlos),
                        nHTot=lg(atmos.nHTot),
                        nH=np.stack([lg(atmos.nH[i]) for i in range(atmos.nH.shape[0])]))


def perturbed(atmos: Atmosphere1d, seed: int, dT=0.05, dv=2.0e3) -> Atmosphere1d:
    """Seeded perturbation of a column (1.5D batches)."""
    rng = np.random.default_rng(seed)
    ker = np.ones(5) / 5.0
    sm = lambda a: np.convolve(np.pad(a, 2, mode='edge'), ker, mode='valid')
    fT = 1.0 + dT * sm(rng.standard_normal(atmos.Nspace))
    v = dv * sm(rng.standard_normal(atmos.Nspace))
    return Atmosphere1d(height=atmos.height.copy(), temperature=atmos.temperature * fT,
                        ne=atmos.ne * fT, vturb=atmos.vturb.copy(), vlos=atmos.vlos + v,
                        nHTot=atmos.nHTot.copy(), nH=atmos.nH.copy())


# ------------------------------------------------------------------------------------------------
# model atoms
# ------------------------------------------------------------------------------------------------
@dataclass
class LineModel:
    i: int
    j: int
    f: float
    qCore: float
    qWing: float
    Nlambda: int
    prd: bool = False
    starkLinear: bool = False
    vdw: float = 3.0e-14      # m^3 s^-1 per neutral H


@dataclass
class ContinuumModel:
    i: int
    j: int
    alpha0: float             # threshold cross-section [m^2]
    minLambda: float          # [nm]
    Nlambda: int


@dataclass
class AtomModel:
    name: str
    mass: float               # [amu]
    abundance: float          # n / nHTot
    E_cm: List[float]         # level energies [cm^-1]
    g: List[float]
    stage: List[int]
    lines: List[LineModel]
    continua: List[ContinuumModel]
    extraCE: List[Tuple[int, int, float]] = field(default_factory=list)  # (i, j, Upsilon)
    nQuantum: Optional[List[int]] = None   # principal quantum numbers (hydrogenic Stark)

    @property
    def Nlevel(self):
        return len(self.E_cm)

    @property
    def E_J(self):
        return np.asarray(self.E_cm) * 100.0 * ph.HC

    def lambda0(self, i, j):
        return ph.HC / (self.E_J[j] - self.E_J[i]) / ph.NM_TO_M


def H_6(lineScale=1.0) -> AtomModel:
    """5 bound levels + H II.  f-values: Wiese et al. (NIST); Kramers-like thresholds."""
    n = lambda base: max(int(round(base * lineScale)) | 1, 5)
    f = {(0, 1): 0.4162, (0, 2): 0.07910, (0, 3): 0.02899, (0, 4): 0.01394,
         (1, 2): 0.6407, (1, 3): 0.1193, (1, 4): 0.04467,
         (2, 3): 0.8421, (2, 4): 0.1506, (3, 4): 1.038}
    quad = {(0, 1): (15.0, 600.0, 101), (0, 2): (10.0, 250.0, 51), (0, 3): (3.0, 100.0, 21),
            (0, 4): (3.0, 100.0, 21), (1, 2): (3.0, 250.0, 71), (1, 3): (3.0, 250.0, 41),
            (1, 4): (3.0, 250.0, 41), (2, 3): (2.0, 30.0, 21), (2, 4): (2.0, 30.0, 21),
            (3, 4): (2.0, 30.0, 21)}
    lines = [LineModel(i, j, f[(i, j)], quad[(i, j)][0], quad[(i, j)][1], n(quad[(i, j)][2]),
                       starkLinear=True, vdw=1.0e-14) for (i, j) in f]
    alpha0 = [6.152e-22, 1.379e-21, 2.149e-21, 2.923e-21, 3.699e-21]
    minL = [22.794, 91.176, 205.147, 364.705, 569.852]
    continua = [ContinuumModel(i, 5, alpha0[i], minL[i], n(20)) for i in range(5)]
    return AtomModel('H', 1.008, 1.0,
                     [0.0, 82258.211, 97491.219, 102822.766, 105290.508, 109677.617],
                     [2, 8, 18, 32, 50, 1], [0, 0, 0, 0, 0, 1], lines, continua,
                     nQuantum=[1, 2, 3, 4, 5, 0])


def CaII_6(lineScale=1.0, prd=False) -> AtomModel:
    """Ca II 4s, 3d(3/2,5/2), 4p(1/2,3/2) + Ca III.  H & K and the infrared triplet."""
    n = lambda base: max(int(round(base * lineScale)) | 1, 5)
    lines = [
        LineModel(0, 3, 0.330, 7.0, 600.0, n(101), prd=prd, vdw=2.0e-14),    # H 396.85
        LineModel(0, 4, 0.682, 7.0, 600.0, n(101), prd=prd, vdw=2.0e-14),    # K 393.37
        LineModel(1, 3, 0.0597, 3.0, 150.0, n(41), vdw=2.0e-14),             # 866.2
        LineModel(1, 4, 0.0120, 3.0, 150.0, n(41), vdw=2.0e-14),             # 849.8
        LineModel(2, 4, 0.0722, 3.0, 150.0, n(41), vdw=2.0e-14),             # 854.2
    ]
    continua = [ContinuumModel(0, 5, 2.03e-23, 50.0, n(20)),
                ContinuumModel(1, 5, 6.15e-22, 60.0, n(20)),
                ContinuumModel(2, 5, 6.15e-22, 60.0, n(20)),
                ContinuumModel(3, 5, 2.40e-22, 70.0, n(20)),
                ContinuumModel(4, 5, 2.40e-22, 70.0, n(20))]
    return AtomModel('Ca', 40.078, 10.0**(6.36 - 12.0),
                     [0.0, 13650.19, 13710.88, 25191.51, 25414.40, 95751.87],
                     [2, 4, 6, 2, 4, 1], [1, 1, 1, 1, 1, 2], lines, continua,
                     extraCE=[(0, 1, 5.0), (0, 2, 7.0), (1, 2, 20.0), (3, 4, 10.0)])


# ------------------------------------------------------------------------------------------------
# Problem construction
# ------------------------------------------------------------------------------------------------
def _collisional_rates(m: AtomModel, nStar, T, ne):
    """C[to, from, k]: effective-collision-strength de-excitation for every bound pair, Seaton
    ionisation for every continuum, upward/recombination rates by detailed balance."""
    Nl, Ns = m.Nlevel, T.shape[0]
    Cm = np.zeros((Nl, Nl, Ns))
    ne_cm3 = ne * 1e-6
    ups = {(l.i, l.j): 1.0 + 10.0 * l.f for l in m.lines}
    for (i, j, u) in m.extraCE:
        ups[(i, j)] = u
    for (i, j), u in ups.items():
        down = ne_cm3 * 8.63e-6 * u / (m.g[j] * np.sqrt(T))
        Cm[i, j] += down
        Cm[j, i] += down * nStar[j] / nStar[i]
    for c in m.continua:
        dE = m.E_J[c.j] - m.E_J[c.i]
        u = dE / (ph.KBoltzmann * T)
        up = ne_cm3 * 1.55e13 * 0.2 * (c.alpha0 * 1e4) * np.exp(-u) / u / np.sqrt(T)
        Cm[c.j, c.i] += up
        Cm[c.i, c.j] += up * nStar[c.i] / nStar[c.j]
    return Cm


def _damping(m: AtomModel, l: LineModel, Aji_all, lambda0, vBroad, atmos: Atmosphere1d):
    """Returns (aDamp, Qelast): damping parameter and the elastic-collision part of the width [s^-1]."""
    gRad = sum(A for (i, j), A in Aji_all.items() if j in (l.i, l.j))
    gEl = l.vdw * atmos.nH[0] * (atmos.temperature / 1.0e4)**0.3
    if l.starkLinear and m.nQuantum is not None:
        nl, nu = m.nQuantum[l.i], m.nQuantum[l.j]
        a1 = 0.642 if nl == 1 else 1.0
        gEl = gEl + a1 * 0.6 * (nu * nu - nl * nl) * atmos.ne**(2.0 / 3.0) * ph.CM_TO_M**2
    else:
        gEl = gEl + 1.0e-12 * atmos.ne  # quadratic-Stark-like
    return (gRad + gEl) * lambda0 * ph.NM_TO_M / (4.0 * np.pi * vBroad), gEl


def grid_size(models: List[AtomModel], lambdaReference=500.0) -> int:
    """Number of points of the global wavelength grid build_problem would make (no arrays built)."""
    own = []
    for m in models:
        for l in m.lines:
            own.append(ph.line_quadrature(m.lambda0(l.i, l.j), l.qCore, l.qWing, l.Nlambda))
        for c in m.continua:
            own.append(np.linspace(c.minLambda, m.lambda0(c.i, c.j), c.Nlambda))
    return int(np.unique(np.concatenate(own + [np.array([lambdaReference])])).size)


def build_problem(atmos: Atmosphere1d, models: List[AtomModel], Nrays=5,
                  formalSolver=abi.FS_BEZIER3_1D, detailed: Tuple[str, ...] = (),
                  storeDepthData=False, lambdaReference=500.0, computeProfiles=True) -> Problem:
    """Assemble the Context-boundary arrays.  The global grid is the sorted unique union of all
    transition grids plus lambdaReference (lightweaver/atomic_set.py:1027-1088); each
    transition's grid is the global slice [Nblue, Nred)."""
    T, ne = atmos.temperature, atmos.ne
    Ns = atmos.Nspace
    muz, wmu = ph.gauss_legendre_mu(Nrays)
    own = []
    for m in models:
        for l in m.lines:
            own.append(ph.line_quadrature(m.lambda0(l.i, l.j), l.qCore, l.qWing, l.Nlambda))
        for c in m.continua:
            own.append(np.linspace(c.minLambda, m.lambda0(c.i, c.j), c.Nlambda))
    grid = np.unique(np.sort(np.concatenate(own + [np.array([lambdaReference])])))

    atoms = []
    it = iter(own)
    for m in models:
        nTotal = m.abundance * atmos.nHTot
        if m.name == 'H':
            nTotal = atmos.nHTot.copy()
        vBroad = np.sqrt(2.0 * ph.KBoltzmann * T / (ph.Amu * m.mass) + atmos.vturb**2)
        nStar = ph.lte_populations(m.E_J, m.g, m.stage, nTotal, T, ne)
        Aji_all = {}
        for l in m.lines:
            Aji_all[(l.i, l.j)] = ph.einstein_coefficients(l.f, m.g[l.i], m.g[l.j],
                                                           m.lambda0(l.i, l.j))[0]
        trans = []
        for l in m.lines:
            g = next(it)
            lam0 = m.lambda0(l.i, l.j)
            Aji, Bji, Bij = ph.einstein_coefficients(l.f, m.g[l.i], m.g[l.j], lam0)
            Nblue = int(np.searchsorted(grid, g[0]))
            Nred = int(np.searchsorted(grid, g[-1])) + 1
            w = grid[Nblue:Nred].copy()
            aDamp, Qelast = _damping(m, l, Aji_all, lam0, vBroad, atmos)
            t = TransitionData(type=abi.LINE, i=l.i, j=l.j, Nblue=Nblue, Nred=Nred, lambda0=lam0,
                               wavelength=w, Aji=Aji, Bji=Bji, Bij=Bij,
                               dopplerWidth=ph.CLight / lam0, aDamp=aDamp, Qelast=Qelast)
            if l.prd:
                t.rhoPrd = np.ones((Nred - Nblue, Ns))
            trans.append(t)
        for c in m.continua:
            g = next(it)
            lamEdge = m.lambda0(c.i, c.j)
            Nblue = int(np.searchsorted(grid, g[0]))
            Nred = int(np.searchsorted(grid, g[-1])) + 1
            w = grid[Nblue:Nred].copy()
            trans.append(TransitionData(type=abi.CONTINUUM, i=c.i, j=c.j, Nblue=Nblue, Nred=Nred,
                                        lambda0=lamEdge, wavelength=w, dopplerWidth=1.0,
                                        alpha=c.alpha0 * (w / lamEdge)**3))
        isDetailed = m.name in detailed
        a = AtomData(name=m.name, Nlevel=m.Nlevel, n=nStar.copy(), nStar=nStar, nTotal=nTotal,
                     vBroad=vBroad, trans=trans, detailed=isDetailed)
        if not isDetailed:
            a.C = _collisional_rates(m, nStar, T, ne)
        atoms.append(a)

    chi, eta, sca, planck = ph.h_minus_background(grid, T, ne, atmos.nH[0], with_planck=True)
    vlosMu = muz[:, None] * atmos.vlos[None, :]
    prob = Problem(height=atmos.height, temperature=T, muz=muz, wmu=wmu, wavelength=grid,
                   bgChi=chi, bgEta=eta, bgSca=sca, atoms=atoms, vlosMu=vlosMu,
                   formalSolver=formalSolver, storeDepthData=storeDepthData)
    # initial J = B_nu(T) (LTE start)
    prob.J[...] = planck
    if computeProfiles:
        compute_profiles_host(prob)
    prob.gamma_prefill()
    return prob


def compute_profiles_host(prob: Problem):
    """phi, wphi of every line on the host with SciPy's wofz (the same Faddeeva package the
    reference vendors as Source/Faddeeva.cc).  Formulae: Transition::compute_phi_la and
    compute_wphi (Source/FormalScalar.cpp:28-51,106-134)."""
    from scipy.special import wofz
    sqrtPi = np.sqrt(np.pi)
    sign = np.array([-1.0, 1.0])
    for a in prob.atoms:
        for t in a.trans:
            if t.type != abi.LINE:
                continue
            vBase = (t.wavelength - t.lambda0) * ph.CLight / t.lambda0        # [Nl]
            v = (vBase[:, None, None, None]
                 + sign[None, None, :, None] * prob.vlosMu[None, :, None, :]) / a.vBroad
            H = wofz(v + 1j * t.aDamp).real
            t.phi[...] = H / (sqrtPi * a.vBroad)
            wl = t.wlambda()
            wphi = np.einsum('l,m,lmdk->k', wl, 0.5 * prob.wmu, t.phi)
            t.wphi[...] = 1.0 / wphi


def falc_h_ca(Nrays=5, lineScale=1.0, formalSolver=abi.FS_BEZIER3_1D, prd=False,
              atmos: Optional[Atmosphere1d] = None, **kw) -> Problem:
    """FAL-C, H (6 levels) + Ca II (6 levels) active: BASELINE.json configs[1] at lineScale=1
    (Nlambda ~ 1e3) and the north-star throughput grid at larger lineScale."""
    atmos = atmos if atmos is not None else falc82()
    return build_problem(atmos, [H_6(lineScale), CaII_6(lineScale, prd=prd)], Nrays=Nrays,
                         formalSolver=formalSolver, **kw)


def falc_h(Nrays=5, lineScale=1.0, formalSolver=abi.FS_BEZIER3_1D, **kw) -> Problem:
    """FAL-C with only H active (BASELINE.json configs[0], plot_SimpleLineTest-like)."""
    return build_problem(falc82(), [H_6(lineScale)], Nrays=Nrays, formalSolver=formalSolver, **kw)


def D_6(lineScale=1.0) -> AtomModel:
    """A deuterium-like twin of the hydrogen model: the same levels with the isotope shift of the energies (reduced mass:
    +2.72e-4) and 2e-5 of the abundance.  Its Lyman / Balmer lines sit 0.03 nm blueward of hydrogen's, inside their cores
    and wings: with both atoms active a wavelength of those lines carries TWO lines and the bound-free continua coupled to
    each of them -- the overlap the stock H + Ca II pair never has (throughput_grid(realistic=True), bench.py's
    aux.realistic_1d)."""
    import copy
    m = copy.deepcopy(H_6(lineScale))
    m.name = 'D'
    m.E_cm = [e * (1.0 + 2.72e-4) for e in m.E_cm]
    m.abundance = 2e-5 * m.abundance
    return m


def throughput_grid(NlambdaTarget=10240, Nrays=5, realistic=False, **kw) -> Problem:
    """FAL-C H + Ca II with the line/continuum quadratures scaled so that the global grid has
    ~NlambdaTarget points (north-star synthetic throughput grid, Ndepth = 82, Nmu = 5; N x 10 240
    for the N-GPU weak-scaling runs).  realistic=True adds the deuterium-like twin of hydrogen (D_6): pairwise
    overlapping lines with coupled continua of both atoms, three active atoms' continua in the ultraviolet."""
    prd = kw.get('prd', False)
    atoms = lambda ls: [H_6(ls), CaII_6(ls, prd=prd)] + ([D_6(ls)] if realistic else [])
    size = lambda ls: grid_size(atoms(ls))
    lo, hi = 1.0, 40.0
    while size(hi) < NlambdaTarget and hi < 1e5:
        lo, hi = hi, hi * 4.0
    best = hi
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        n = size(mid)
        best = mid
        if n < NlambdaTarget:
            lo = mid
        else:
            hi = mid
        if n == NlambdaTarget:
            break
    if realistic:
        kw.pop('prd', None)
        return build_problem(falc82(), atoms(best), Nrays=Nrays, **kw)
    return falc_h_ca(Nrays=Nrays, lineScale=best, **kw)


# ------------------------------------------------------------------------------------------------
# 2D (x-periodic) problems from columns: BASELINE configs[4] in miniature
# ------------------------------------------------------------------------------------------------
def rays_2d(Nmuz=2):
    """A small angle set for the half space mu_z > 0 with both signs of mu_x (toObs = False flips both, as the
    reference does, Source/FormalScalar2d.cpp:766-771): Gauss-Legendre in mu_z, two azimuths each, weights
    summing to 1."""
    muz, w = ph.gauss_legendre_mu(Nmuz)
    mux = np.sqrt(1.0 - muz**2) * np.cos(np.pi / 5.0)
    return (np.concatenate([mux, -mux]), np.concatenate([muz, muz]), np.concatenate([0.5 * w, 0.5 * w]))


def build_problem_2d(columns: List[Atmosphere1d], x: np.ndarray, models_: List[AtomModel], grid2d_factory=None,
                     Nmuz=2, lambdaReference=500.0) -> Problem:
    """An x-periodic 2D problem whose columns are 1D atmospheres on a common height grid.  Every per-depth array
    of the 1D problems is laid out [.., Nz, Nx] (index k * Nx + j).  `grid2d_factory(x, z, mux, muz, T[Nz, Nx])`
    returns the lightweaver_amd.grid2d.Grid2d (intersection table); by default the library's own builder
    (grid2d.build_grid2d = lwhip_build_intersections), the tests also pass the reference's through oracle/_ref."""
    Nx = len(columns)
    assert x.shape == (Nx,) and all(np.array_equal(c.height, columns[0].height) for c in columns)
    mux, muz, wmu = rays_2d(Nmuz)
    Nrays = muz.size
    cols = [build_problem(c, models_, Nrays=Nrays, lambdaReference=lambdaReference, computeProfiles=False)
            for c in columns]
    p0 = cols[0]
    Nz = p0.Nspace

    def lay(arrs):        # [Nx] arrays of shape [..., Nz] -> [..., Nz * Nx]
        a = np.stack(arrs, axis=-1)
        return np.ascontiguousarray(a.reshape(a.shape[:-2] + (Nz * Nx,)))
    T = lay([c.temperature for c in cols])
    atoms = []
    for ia, a0 in enumerate(p0.atoms):
        trans = []
        for kr, t0 in enumerate(a0.trans):
            kw = dict(type=t0.type, i=t0.i, j=t0.j, Nblue=t0.Nblue, Nred=t0.Nred, lambda0=t0.lambda0,
                      wavelength=t0.wavelength.copy(), Aji=t0.Aji, Bji=t0.Bji, Bij=t0.Bij, dopplerWidth=t0.dopplerWidth)
            if t0.type == abi.LINE:
                kw['aDamp'] = lay([c.atoms[ia].trans[kr].aDamp for c in cols])
                kw['Qelast'] = lay([c.atoms[ia].trans[kr].Qelast for c in cols])
                if t0.rhoPrd is not None:
                    kw['rhoPrd'] = np.ones((t0.Nred - t0.Nblue, Nz * Nx))
            else:
                kw['alpha'] = t0.alpha.copy()
            trans.append(TransitionData(**kw))
        atoms.append(AtomData(name=a0.name, Nlevel=a0.Nlevel, n=lay([c.atoms[ia].n for c in cols]),
                              nStar=lay([c.atoms[ia].nStar for c in cols]),
                              nTotal=lay([c.atoms[ia].nTotal for c in cols]),
                              vBroad=lay([c.atoms[ia].vBroad for c in cols]), trans=trans, detailed=a0.detailed,
                              C=lay([c.atoms[ia].C for c in cols]) if a0.C is not None else None))
    vz = lay([c.vlos for c in columns])
    vlosMu = muz[:, None] * vz[None, :]
    if grid2d_factory is None:
        from ..grid2d import build_grid2d as grid2d_factory
    grid = grid2d_factory(x, columns[0].height, mux, muz, T.reshape(Nz, Nx))
    prob = Problem(height=np.repeat(columns[0].height, Nx), temperature=T, muz=muz, wmu=wmu,
                   wavelength=p0.wavelength, bgChi=lay([c.bgChi for c in cols]), bgEta=lay([c.bgEta for c in cols]),
                   bgSca=lay([c.bgSca for c in cols]), atoms=atoms, vlosMu=vlosMu, J=lay([c.J for c in cols]),
                   grid2d=grid)
    compute_profiles_host(prob)
    prob.gamma_prefill()
    return prob

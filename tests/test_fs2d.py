"""2D short characteristics (first stage): the formal solver primitive
piecewise_besser_2d + interp_linear_2d on an x-periodic grid.

CPU: the C restatement (oracle/lw_oracle_2d.inc) against the real core compiled from /root/reference
(oracle/_ref: build_intersection_list + piecewise_besser_2d) and against the committed fixture.
GPU: lwhip_formal_solver_2d through the C ABI against the oracle.  Tolerance 1e-9 relative."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, rel_err
from lightweaver_amd import _abi as abi
from lightweaver_amd.grid2d import Grid2d
from oracle import bindings

HAVE_REF = os.path.exists(bindings.REF_LIB)
FIXTURE = os.path.join(GOLDEN, 'fs2d_small.npz')


def small_grid_inputs(Nx=14, Nz=18, seed=0):
    """A FAL-C-like stratification with horizontal structure; three rays of an il7n3-like set (positive and
    negative mu_x, shallow and steep), so that both interpolation axes and long characteristics occur."""
    rng = np.random.default_rng(seed)
    x = np.linspace(0.0, 1.3e6, Nx)
    z = np.linspace(2.1e6, -1.0e5, Nz) + 0.0          # height, decreasing with index
    z[1:-1] += 2.0e4 * rng.standard_normal(Nz - 2)    # non-uniform
    z = np.sort(z)[::-1].copy()
    muz = np.array([0.18, 0.55, 0.86])
    mux = np.sqrt(1.0 - muz**2) * np.array([0.9, -0.75, 0.4])
    T = 4500.0 + 3000.0 * np.exp(-z / 8e5)[:, None] * (1.0 + 0.1 * rng.random((Nz, Nx)))
    return x, z, mux, muz, T


def fields(grid, seed):
    rng = np.random.default_rng(seed)
    Nz, Nx = grid.Nz, grid.Nx
    chi = 3e-6 * np.exp(-grid.z / 2.5e5)[:, None] * (1.0 + 0.5 * rng.random((Nz, Nx)))
    S = 2e-8 * (1.0 + 0.8 * np.sin(np.linspace(0, 5, Nz))[:, None] * rng.random((Nz, Nx)) + rng.random((Nz, Nx)))
    return chi, S


def load_golden():
    with np.load(FIXTURE) as zf:
        d = {k: zf[k] for k in zf.files}
    return Grid2d.from_dict(d), d


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
@pytest.mark.parametrize('bc', [(abi.BC_THERMALISED, abi.BC_ZERO), (abi.BC_ZERO, abi.BC_THERMALISED)])
def test_oracle_2d_besser_matches_reference(bc):
    x, z, mux, muz, T = small_grid_inputs()
    ref = bindings.Ref2d(x, z, mux, muz, T, zLowerBc=bc[0], zUpperBc=bc[1])
    grid = ref.grid()
    assert grid.substepOff.size > 1 and set(np.unique(grid.uw['axis'])) == {0, 1, 2}
    for seed in (1, 2):
        chi, S = fields(grid, seed)
        for mu in range(grid.Nrays):
            for toObs in (0, 1):
                Ir, Pr = ref.besser(mu, toObs, 500.0, chi, S)
                Io, Po = bindings.oracle_2d_besser(grid, mu, toObs, 500.0, chi, S)
                np.testing.assert_array_equal(Io, Ir)
                np.testing.assert_array_equal(Po, Pr)


def test_oracle_2d_besser_golden():
    grid, d = load_golden()
    n = 0
    for mu in range(grid.Nrays):
        for toObs in (0, 1):
            I, P = bindings.oracle_2d_besser(grid, mu, toObs, float(d['wav']), d['chi'], d['S'])
            np.testing.assert_array_equal(I, d['I'][mu, toObs])
            np.testing.assert_array_equal(P, d['Psi'][mu, toObs])
            n += 1
    assert n == 6


def test_2d_vertical_uniform_limit():
    """Horizontally uniform chi and S: every column carries the same intensity (periodic grid)."""
    grid, d = load_golden()
    chi = np.repeat(d['chi'][:, :1], grid.Nx, axis=1)
    S = np.repeat(d['S'][:, :1], grid.Nx, axis=1)
    g2 = Grid2d(**{**grid.__dict__, 'temperature': np.repeat(grid.temperature[:, :1], grid.Nx, axis=1)})
    I, _ = bindings.oracle_2d_besser(g2, 2, 1, float(d['wav']), chi, S)
    assert rel_err(I, np.repeat(I[:, :1], grid.Nx, axis=1)) <= 1e-10


@pytest.mark.gpu
def test_hip_2d_besser_matches_oracle(gpu):
    from lightweaver_amd.grid2d import formal_solver_2d
    grid, d = load_golden()
    rays = np.arange(2 * grid.Nrays, dtype=np.int32)
    chi = np.broadcast_to(d['chi'], (rays.size,) + d['chi'].shape)
    S = np.broadcast_to(d['S'], (rays.size,) + d['S'].shape)
    I, P = formal_solver_2d(grid, float(d['wav']), rays, chi, S)
    for p, ray in enumerate(rays):
        assert rel_err(I[p], d['I'][ray >> 1, ray & 1]) <= 1e-9
        assert rel_err(P[p], d['Psi'][ray >> 1, ray & 1]) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize('Nx,Nz', [(5, 4), (70, 33), (256, 82)])
def test_hip_2d_besser_sizes(gpu, Nx, Nz):
    """Other sizes (incl. BASELINE config 5's 256 x 82) on stencil tables from the reference, where it is
    present; skipped on the GPU box, which has no reference (the committed fixture covers it there)."""
    if not HAVE_REF:
        pytest.skip('needs oracle/_ref for the intersection table')
    from lightweaver_amd.grid2d import formal_solver_2d
    x, z, mux, muz, T = small_grid_inputs(Nx, Nz, seed=3)
    ref = bindings.Ref2d(x, z, mux, muz, T)
    grid = ref.grid()
    chi, S = fields(grid, 5)
    rays = np.arange(2 * grid.Nrays, dtype=np.int32)
    I, P = formal_solver_2d(grid, 500.0, rays, np.broadcast_to(chi, (6,) + chi.shape), np.broadcast_to(S, (6,) + S.shape))
    for p, ray in enumerate(rays):
        Io, Po = bindings.oracle_2d_besser(grid, int(ray >> 1), int(ray & 1), 500.0, chi, S)
        assert rel_err(I[p], Io) <= 1e-9 and rel_err(P[p], Po) <= 1e-9


# ---- the whole iteration on a 2D problem: CPU side pinned (the device path is the next stage) --------------
def small_2d_problem():
    """10 perturbed 24-point FAL-C columns, H active, 4 rays; the intersection table is the reference's."""
    from lightweaver_amd.harness import models
    base = models.resample(models.falc82(), 24)
    Nx = 10
    cols = [models.perturbed(base, seed=50 + j) for j in range(Nx)]
    x = np.linspace(0.0, 9.0e5, Nx)
    return models.build_problem_2d(cols, x, [models.H_6(0.12)],
                                   lambda x, z, mux, muz, T: bindings.Ref2d(x, z, mux, muz, T).grid())


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
def test_oracle_2d_iteration_matches_reference():
    """formal_sol_gamma_matrices + stat_eq on a 2D problem: the oracle (intensity core + 2D solver) against the
    real core (its own build_intersection_list, piecewise_besser_2d, interp_linear_2d), bit for bit."""
    p = small_2d_problem()
    q = p.copy()
    ref, orc = bindings.RefContext(p), bindings.OracleContext(q)
    for it in range(3):
        p.gamma_prefill()
        q.gamma_prefill()
        assert ref.formal_sol_gamma_matrices() == orc.formal_sol_gamma_matrices()[:2]
        np.testing.assert_array_equal(q.J, p.J)
        np.testing.assert_array_equal(q.I, p.I)
        for a, b in zip(q.atoms, p.atoms):
            np.testing.assert_array_equal(a.Gamma, b.Gamma)
            for ta, tb in zip(a.trans, b.trans):
                np.testing.assert_array_equal(ta.Rij, tb.Rij)
        assert ref.stat_equil() == orc.stat_equil() == 0
        for a, b in zip(q.atoms, p.atoms):
            np.testing.assert_array_equal(a.n, b.n)
    assert p.I.shape == (p.Nlambda, p.Nrays, 10) and np.ptp(p.I[5, 0]) > 0     # columns differ


def test_2d_problem_round_trips_through_io():
    from lightweaver_amd import io
    from lightweaver_amd.model import Problem
    grid, d = load_golden()
    Ns = grid.Nx * grid.Nz
    prob = Problem(height=np.repeat(grid.z, grid.Nx), temperature=grid.temperature.ravel(), muz=grid.muz,
                   wmu=np.full(grid.Nrays, 1.0 / grid.Nrays), wavelength=np.array([500.0, 501.0]),
                   bgChi=np.ones((2, Ns)), bgEta=np.ones((2, Ns)), bgSca=np.zeros((2, Ns)), atoms=[], grid2d=grid)
    rt = io.problem_from_dict(io.problem_to_dict(prob))
    assert rt.grid2d is not None and rt.I.shape == (2, grid.Nrays, grid.Nx)
    assert np.array_equal(rt.grid2d.uw['fracX'], grid.uw['fracX'])


def load_2d_problem():
    """The committed 2D problem (inputs incl. the reference's intersection table) and the reference's outputs
    after two iterations (tests/golden/make_golden.py: make_2d_fixture)."""
    from helpers import load_fixture
    return load_fixture('falc2d_small')


def test_oracle_2d_iteration_golden():
    prob, gold = load_2d_problem()
    orc = bindings.OracleContext(prob)
    for it in range(2):
        prob.gamma_prefill()
        orc.formal_sol_gamma_matrices()
        assert orc.stat_equil() == 0
    np.testing.assert_array_equal(prob.J, gold['out/J'])
    np.testing.assert_array_equal(prob.I, gold['out/I'])
    np.testing.assert_array_equal(prob.atoms[0].n, gold['out/n0'])


@pytest.mark.gpu
def test_hip_2d_iteration_matches_golden(gpu):
    """formal_sol_gamma_matrices + stat_equil on the 2D problem through the device pipeline (cont -> gather2d ->
    fs2d -> rates2d -> apply), against the reference's committed outputs."""
    from lightweaver_amd.context import Context
    prob, gold = load_2d_problem()
    q = prob.copy()
    orc = bindings.OracleContext(q)
    with Context(prob) as ctx:
        for it in range(2):
            up = ctx.formal_sol_gamma_matrices()
            q.gamma_prefill()
            dJ, idx = orc.formal_sol_gamma_matrices()
            assert up.dJMax == pytest.approx(dJ, rel=1e-9)
            assert rel_err(prob.J, q.J) <= 1e-9 * (it + 1)
            assert rel_err(prob.I, q.I) <= 1e-9 * (it + 1)
            for a, b in zip(prob.atoms, q.atoms):
                # Gamma entries are sums over wavelengths and rays of terms of both signs on top of the collisional
                # pre-fill; the device adds per-wavelength ray sums where the reference adds term by term
                assert rel_err(a.Gamma, b.Gamma) <= 1e-8
                for ta, tb in zip(a.trans, b.trans):
                    assert rel_err(ta.Rij, tb.Rij) <= 1e-9 * (it + 1) and rel_err(ta.Rji, tb.Rji) <= 1e-9 * (it + 1)
            ctx.stat_equil()
            assert orc.stat_equil() == 0
    assert rel_err(prob.J, gold['out/J']) <= 1e-8
    assert rel_err(prob.atoms[0].n, gold['out/n0']) <= 1e-8


@pytest.mark.gpu
def test_hip_2d_device_resident_and_batches(gpu, monkeypatch):
    """Same result with one wavelength per batch and with everything resident in HBM."""
    from lightweaver_amd.context import Context
    prob, gold = load_2d_problem()
    ref = prob.copy()
    with Context(ref) as ctx:
        for it in range(2):
            ctx.formal_sol_gamma_matrices()
            ctx.stat_equil()
    monkeypatch.setenv('LWHIP_BATCH2D', '1')
    with Context(prob) as ctx:
        for it in range(2):
            ctx.formal_sol_gamma_matrices(deviceResident=True)
            ctx.stat_equil(deviceResident=True)
        ctx.download(abi.ALL_OUTPUTS | abi.POPS)
    assert rel_err(prob.J, ref.J) <= 1e-11 and rel_err(prob.atoms[0].n, ref.atoms[0].n) <= 1e-10


@pytest.mark.gpu
def test_hip_2d_wavelength_shards_sum_to_whole(gpu):
    """BASELINE config 5's multi-GPU split on one device: two 2D contexts owning [0, m) and [m, Nlambda); their
    reduce buffers (Gamma / rate partial sums, 8 * Ntrans * 4 * Nspace bytes -- MBs in 2D, so this is the one
    bandwidth-relevant all-reduce) are added through zero-copy torch views, as ShardedIteration does with RCCL."""
    import torch
    from lightweaver_amd.context import Context
    from lightweaver_amd.distributed import DeviceReduceBuffer
    prob, _ = load_2d_problem()
    q = prob.copy()
    with Context(q) as whole:
        upWhole = whole.formal_sol_gamma_matrices()
    m = prob.Nlambda // 2 + 1
    pa, pb = prob.copy(), prob.copy()
    with Context(pa, laStart=0, laEnd=m, worldSize=2, worldRank=0) as A, \
            Context(pb, laStart=m, laEnd=prob.Nlambda, worldSize=2, worldRank=1) as B:
        views = []
        for c in (A, B):
            c.prob.gamma_prefill()
            c.upload(abi.GAMMA)
            c.fs_partial()
            c.synchronize()
            ptr, n = c.reduce_buffer()
            views.append(torch.as_tensor(DeviceReduceBuffer(ptr, n), device='cuda'))
        total = views[0] + views[1]
        for c, v in zip((A, B), views):
            v.copy_(total)
            torch.cuda.synchronize()
            up = c.fs_finalise()
            assert up.dJMax == pytest.approx(upWhole.dJMax, rel=1e-12) and up.dJMaxIdx == upWhole.dJMaxIdx
            c.download(abi.ALL_OUTPUTS)
    for p in (pa, pb):
        assert rel_err(p.atoms[0].Gamma, q.atoms[0].Gamma) <= 1e-11
        for ta, tb in zip(p.atoms[0].trans, q.atoms[0].trans):
            assert rel_err(ta.Rij, tb.Rij) <= 1e-11 and rel_err(ta.Rji, tb.Rji) <= 1e-11
    assert rel_err(pa.J[:m], q.J[:m]) <= 1e-13 and rel_err(pb.J[m:], q.J[m:]) <= 1e-13
    assert rel_err(pa.I[:m], q.I[:m]) <= 1e-13 and rel_err(pb.I[m:], q.I[m:]) <= 1e-13


# ---- formal_sol (intensity only) on 2D problems ------------------------------------------------------------------
@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
@pytest.mark.parametrize('upOnly', [True, False])
def test_oracle_2d_formal_sol_matches_reference(upOnly):
    prob, _ = load_2d_problem()
    prob.I[:] = -1.0
    q = prob.copy()
    bindings.OracleContext(prob).formal_sol(upOnly=upOnly)
    bindings.RefContext(q).formal_sol(upOnly=upOnly)
    np.testing.assert_array_equal(prob.I, q.I)
    assert np.all(prob.I >= 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize('upOnly', [True, False])
def test_hip_2d_formal_sol_matches_oracle(gpu, upOnly):
    """lwhip_formal_sol on a 2D problem: emergent intensity of every up-going ray, J / Gamma / rates untouched."""
    from lightweaver_amd.context import Context
    prob, _ = load_2d_problem()
    q = prob.copy()
    bindings.OracleContext(q).formal_sol(upOnly=upOnly)
    J0, G0 = prob.J.copy(), prob.atoms[0].Gamma.copy()
    prob.I[:] = -1.0
    with Context(prob) as ctx:
        ctx.formal_sol(upOnly=upOnly)
        ctx.download(abi.ALL_OUTPUTS)
    assert rel_err(prob.I, q.I) <= 1e-9
    assert np.array_equal(prob.J, J0) and np.array_equal(prob.atoms[0].Gamma, G0)


def blended_atoms(ncopy, ls=0.25):
    """H plus `ncopy` Ca II model atoms with slightly scaled level energies: their lines overlap each other (2 copies:
    up to 4 lines at one wavelength, 3 copies: up to 6), which the stock H + Ca II pair never does (at most 2)."""
    import copy
    from lightweaver_amd.harness import models
    ca = models.CaII_6(ls)

    def shifted(eps, name):
        m = copy.deepcopy(ca)
        m.name, m.E_cm, m.abundance = name, [e * (1 + eps) for e in m.E_cm], 0.5 * m.abundance
        return m
    return [models.H_6(ls)] + [shifted(2e-4 * (q - ncopy // 2), f'Ca{q}') for q in range(ncopy)]


@pytest.mark.gpu
def test_hip_2d_overlapping_lines(gpu):
    """Four lines at one wavelength on a small 2D grid (the MAXL = 4 instantiations of gather2d / rates2d), three
    active atoms, mixed and pure continua of all of them; built with the library's own geometry."""
    from lightweaver_amd.context import Context
    from lightweaver_amd.harness import models
    base = models.falc82()
    cols = [models.perturbed(base, seed=300 + j) for j in range(6)]
    prob = models.build_problem_2d(cols, np.linspace(0.0, 5 * 40e3, 6), blended_atoms(2))
    q = prob.copy()
    orc = bindings.OracleContext(q)
    with Context(prob) as ctx:
        for it in range(2):
            up = ctx.formal_sol_gamma_matrices()
            q.gamma_prefill()
            dJ, _ = orc.formal_sol_gamma_matrices()
            assert up.dJMax == pytest.approx(dJ, rel=1e-9)
            assert rel_err(prob.J, q.J) <= 1e-9 * (it + 1) and rel_err(prob.I, q.I) <= 1e-9 * (it + 1)
            for a, b in zip(prob.atoms, q.atoms):
                assert rel_err(a.Gamma, b.Gamma) <= 1e-8
                for ta, tb in zip(a.trans, b.trans):
                    assert rel_err(ta.Rij, tb.Rij) <= 1e-9 * (it + 1) and rel_err(ta.Rji, tb.Rji) <= 1e-9 * (it + 1)
            ctx.stat_equil()
            assert orc.stat_equil() == 0
    for a, b in zip(prob.atoms, q.atoms):
        assert rel_err(a.n, b.n) <= 1e-8


# ---- PRD sub-iterations on 2D problems ------------------------------------------------------------------------------
def prd_problem_2d():
    from lightweaver_amd.harness import models
    base = models.falc82()
    cols = [models.perturbed(base, seed=400 + j) for j in range(5)]
    return models.build_problem_2d(cols, np.linspace(0.0, 4 * 40e3, 5), [models.H_6(0.2), models.CaII_6(0.2, prd=True)])


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
def test_oracle_2d_prd_matches_reference():
    """redistribute_prd on a 2D problem (scattering integral per point, the PRD rates pass through the 2D formal
    solver): the oracle against the core, bit for bit, two iterations with three sub-iterations each."""
    prob = prd_problem_2d()
    p, q = prob.copy(), prob.copy()
    oc, rc = bindings.OracleContext(p), bindings.RefContext(q)
    for it in range(2):
        p.gamma_prefill()
        q.gamma_prefill()
        assert oc.formal_sol_gamma_matrices() == rc.formal_sol_gamma_matrices()
        assert oc.stat_equil() == rc.stat_equil() == 0
        ra, rb = oc.redistribute_prd(3, 1e-2), rc.redistribute_prd(3, 1e-2)
        assert np.array_equal(ra['dRho'], rb['dRho']) and np.array_equal(ra['dRhoMaxIdx'], rb['dRhoMaxIdx'])
        assert np.array_equal(ra['dJPrdMax'], rb['dJPrdMax'])
    assert np.array_equal(p.J, q.J)
    for a, b in zip(p.atoms, q.atoms):
        assert np.array_equal(a.n, b.n)
        for t, u in zip(a.trans, b.trans):
            if t.rhoPrd is not None:
                assert np.array_equal(t.rhoPrd, u.rhoPrd)
            assert np.array_equal(t.Rij, u.Rij) and np.array_equal(t.Rji, u.Rji)


@pytest.mark.gpu
def test_hip_2d_prd_matches_oracle(gpu):
    """lwhip_redistribute_prd on a 2D problem: rho, J, the PRD lines' rates and the reported changes."""
    from lightweaver_amd.context import Context
    prob = prd_problem_2d()
    q = prob.copy()
    orc = bindings.OracleContext(q)
    with Context(prob) as ctx:
        for it in range(2):
            ctx.formal_sol_gamma_matrices()
            ctx.stat_equil()
            q.gamma_prefill()
            orc.formal_sol_gamma_matrices()
            assert orc.stat_equil() == 0
            up = ctx.prd_redistribute(3, 1e-2)
            ro = orc.redistribute_prd(3, 1e-2)
            assert up.NprdSubIter == ro['NprdSubIter']
            assert rel_err(up.dRho, ro['dRho']) <= 1e-5          # see tests/test_prd.py on dRho's conditioning
            assert rel_err(up.dJPrdMax, ro['dJPrdMax']) <= 1e-7
            tol = 1e-8 * (it + 1)
            assert rel_err(prob.J, q.J) <= tol
            for a, b in zip(prob.atoms, q.atoms):
                for t, u in zip(a.trans, b.trans):
                    if t.rhoPrd is not None:
                        assert rel_err(t.rhoPrd, u.rhoPrd) <= tol
                    assert rel_err(t.Rij, u.Rij) <= tol and rel_err(t.Rji, u.Rji) <= tol


# ---- fixed (CALLABLE) x boundaries (Source/FormalScalar2d.cpp:806-852, 884-898) ------------------------------------
def x_boundaries(Nrays, Nz, Nla=1, seed=0):
    from lightweaver_amd.model import Boundary
    rng = np.random.default_rng(seed)
    idxs = np.arange(2 * Nrays, dtype=np.int32).reshape(Nrays, 2)
    mk = lambda: Boundary(abi.BC_CALLABLE, idxs=idxs.copy(), bcData=rng.uniform(0.5, 2.0, (Nla, 2 * Nrays, Nz)) * 2e-8)
    return mk(), mk()


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
@pytest.mark.parametrize('bc', [(abi.BC_THERMALISED, abi.BC_ZERO), (abi.BC_ZERO, abi.BC_THERMALISED)])
def test_oracle_2d_besser_fixed_x_matches_reference(bc):
    """The column a ray enters through is prescribed, the march starts one column in, the last column has no downwind
    point (linear step; thermalised z boundary: the preceding column's gradient), no long characteristics."""
    x, z, mux, muz, T = small_grid_inputs()
    muz = np.append(muz, 1.0)            # and a vertical ray, which still traces every column
    mux = np.append(mux, 0.0)
    lo, hi = x_boundaries(muz.size, z.size)
    ref = bindings.Ref2d(x, z, mux, muz, T, zLowerBc=bc[0], zUpperBc=bc[1], xLowerBc=lo, xUpperBc=hi)
    grid = ref.grid()
    assert grid.substepOff.size == 1 and not grid.periodic
    for seed in (1, 2):
        chi, S = fields(grid, seed)
        for mu in range(grid.Nrays):
            for toObs in (0, 1):
                Ir, Pr = ref.besser(mu, toObs, 500.0, chi, S)
                Io, Po = bindings.oracle_2d_besser(grid, mu, toObs, 500.0, chi, S)
                np.testing.assert_array_equal(Io, Ir)
                np.testing.assert_array_equal(Po, Pr)
                if mux[mu] != 0.0:       # the prescribed column really is the boundary data
                    sgn = mux[mu] if toObs else -mux[mu]
                    b, jb = (lo, 0) if sgn > 0 else (hi, grid.Nx - 1)
                    np.testing.assert_array_equal(Io[:, jb], b.bcData[0, b.idxs[mu, toObs]])


def fixed_x_problem(seed=3):
    """small_2d_problem with both x boundaries CALLABLE: the incoming radiation is the Planck function of the
    boundary column's own temperature, modulated per ray (a physically sane irradiation, so that the statistical
    equilibrium stays well conditioned)."""
    from lightweaver_amd.grid2d import build_grid2d
    from lightweaver_amd.harness import physics as ph
    from lightweaver_amd.model import Boundary
    p = small_2d_problem()
    g = p.grid2d
    rng = np.random.default_rng(seed)
    idxs = np.arange(2 * g.Nrays, dtype=np.int32).reshape(g.Nrays, 2)

    def side(col):
        T = g.temperature[:, col]
        B = np.stack([ph.planck_nu(T, lam) for lam in p.wavelength])          # [Nlambda, Nz]
        return Boundary(abi.BC_CALLABLE, idxs=idxs.copy(),
                        bcData=B[:, None, :] * rng.uniform(0.6, 1.0, (1, 2 * g.Nrays, 1)))
    p.grid2d = build_grid2d(g.x, g.z, g.mux, g.muz, g.temperature, g.zLowerBc, g.zUpperBc, xLowerBc=side(0),
                            xUpperBc=side(g.Nx - 1))
    return p


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
def test_oracle_2d_iteration_fixed_x_matches_reference():
    p = fixed_x_problem()
    q = p.copy()
    ref, orc = bindings.RefContext(p), bindings.OracleContext(q)
    for it in range(2):
        p.gamma_prefill()
        q.gamma_prefill()
        assert ref.formal_sol_gamma_matrices() == orc.formal_sol_gamma_matrices()[:2]
        np.testing.assert_array_equal(q.J, p.J)
        np.testing.assert_array_equal(q.I, p.I)
        for a, b in zip(q.atoms, p.atoms):
            np.testing.assert_array_equal(a.Gamma, b.Gamma)
            for ta, tb in zip(a.trans, b.trans):
                np.testing.assert_array_equal(ta.Rij, tb.Rij)
        assert ref.stat_equil() == orc.stat_equil() == 0
        for a, b in zip(q.atoms, p.atoms):
            np.testing.assert_array_equal(a.n, b.n)


@pytest.mark.gpu
@pytest.mark.parametrize('bc', [(abi.BC_THERMALISED, abi.BC_ZERO), (abi.BC_ZERO, abi.BC_THERMALISED)])
def test_hip_2d_besser_fixed_x_matches_oracle(gpu, bc):
    from lightweaver_amd.grid2d import build_grid2d, formal_solver_2d
    x, z, mux, muz, T = small_grid_inputs(Nx=37, Nz=29)
    muz, mux = np.append(muz, 1.0), np.append(mux, 0.0)
    lo, hi = x_boundaries(muz.size, z.size)
    grid = build_grid2d(x, z, mux, muz, T, bc[0], bc[1], xLowerBc=lo, xUpperBc=hi)
    chi, S = fields(grid, 4)
    rays = np.arange(2 * grid.Nrays)
    I, Psi = formal_solver_2d(grid, 500.0, rays, np.broadcast_to(chi, (rays.size,) + chi.shape),
                              np.broadcast_to(S, (rays.size,) + S.shape))
    for r in rays:
        Io, Po = bindings.oracle_2d_besser(grid, r // 2, r % 2, 500.0, chi, S)
        assert rel_err(I[r], Io) <= 1e-9 and rel_err(Psi[r], Po) <= 1e-9, r


@pytest.mark.gpu
def test_hip_2d_iteration_fixed_x_matches_oracle(gpu):
    """The device 2D pipeline on a grid with CALLABLE x boundaries: iteration + stat_equil against the oracle (which is
    bit-identical to the core, test_oracle_2d_iteration_fixed_x_matches_reference)."""
    from lightweaver_amd.context import Context
    prob = fixed_x_problem()
    q = prob.copy()
    orc = bindings.OracleContext(q)
    with Context(prob) as ctx:
        for it in range(2):
            up = ctx.formal_sol_gamma_matrices()
            q.gamma_prefill()
            dJ, idx = orc.formal_sol_gamma_matrices()
            assert up.dJMax == pytest.approx(dJ, rel=1e-9)
            assert rel_err(prob.J, q.J) <= 1e-9 * (it + 1)
            assert rel_err(prob.I, q.I) <= 1e-9 * (it + 1)
            for a, b in zip(prob.atoms, q.atoms):
                assert rel_err(a.Gamma, b.Gamma) <= 1e-8
                for ta, tb in zip(a.trans, b.trans):
                    assert rel_err(ta.Rij, tb.Rij) <= 1e-9 * (it + 1) and rel_err(ta.Rji, tb.Rji) <= 1e-9 * (it + 1)
            ctx.stat_equil()
            assert orc.stat_equil() == 0
            assert rel_err(prob.atoms[0].n, q.atoms[0].n) <= 1e-8


# ---- CALLABLE z boundaries of a 2D grid (Source/FormalScalar2d.cpp:930-938): bcData [Nlambda, Nmu, Nx] ------------------
def callable_z_problem(seed=5):
    """small_2d_problem irradiated from above: zUpperBc CALLABLE (a fraction of the Planck function of the top row's
    temperature, per column and ray), zLowerBc THERMALISED."""
    from lightweaver_amd.harness import physics as ph
    from lightweaver_amd.model import Boundary
    p = small_2d_problem()
    g = p.grid2d
    rng = np.random.default_rng(seed)
    idxs = np.arange(2 * g.Nrays, dtype=np.int32).reshape(g.Nrays, 2)
    B = np.stack([ph.planck_nu(g.temperature[0], lam) for lam in p.wavelength])          # [Nlambda, Nx]
    p.zUpperBc = Boundary(abi.BC_CALLABLE, idxs=idxs, bcData=B[:, None, :] * rng.uniform(0.05, 0.3, (1, 2 * g.Nrays, 1)))
    g.zUpperBc = abi.BC_CALLABLE
    return p


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
def test_oracle_2d_iteration_callable_z_matches_reference():
    p = callable_z_problem()
    q = p.copy()
    ref, orc = bindings.RefContext(p), bindings.OracleContext(q)
    for it in range(2):
        p.gamma_prefill()
        q.gamma_prefill()
        assert ref.formal_sol_gamma_matrices() == orc.formal_sol_gamma_matrices()[:2]
        np.testing.assert_array_equal(q.J, p.J)
        np.testing.assert_array_equal(q.I, p.I)
        for a, b in zip(q.atoms, p.atoms):
            np.testing.assert_array_equal(a.Gamma, b.Gamma)
        assert ref.stat_equil() == orc.stat_equil() == 0
    base = small_2d_problem()               # the irradiation really enters: J differs from the ZERO-boundary run
    bo = bindings.OracleContext(base)
    base.gamma_prefill()
    bo.formal_sol_gamma_matrices()
    assert rel_err(q.J, base.J) > 1e-3


@pytest.mark.gpu
def test_hip_2d_iteration_callable_z_matches_oracle(gpu):
    """(No population update in between: with this irradiation the rate matrix of the top rows is so ill conditioned
    that the 2e-8 agreement of Gamma turns into 1e-2 in n -- for the reference's own schemes as much as for this one.)"""
    from lightweaver_amd.context import Context
    prob = callable_z_problem()
    q = prob.copy()
    orc = bindings.OracleContext(q)
    with Context(prob) as ctx:
        for it in range(2):
            up = ctx.formal_sol_gamma_matrices()
            q.gamma_prefill()
            dJ, idx = orc.formal_sol_gamma_matrices()
            assert up.dJMax == pytest.approx(dJ, rel=1e-9)
            assert rel_err(prob.J, q.J) <= 1e-9 * (it + 1) and rel_err(prob.I, q.I) <= 1e-9 * (it + 1)
            for a, b in zip(prob.atoms, q.atoms):
                # (Gamma: sums of terms of both signs, see test_hip_2d_iteration_matches_golden; the irradiated top rows
                # cancel more strongly: 1.9e-8 measured)
                assert rel_err(a.Gamma, b.Gamma) <= 5e-8

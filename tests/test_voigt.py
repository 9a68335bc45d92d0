"""Voigt / Faddeeva: the restated algorithm (oracle) against the reference's vendored package and
SciPy's wofz (the same S. G. Johnson code), and the device kernel against both."""
import numpy as np
import pytest
from scipy.special import wofz

from helpers import load_fixture, rel_err
from lightweaver_amd import _abi as abi
from oracle import bindings
from oracle.bindings import OracleContext

TOL_VOIGT = 1e-13

# (a, v) points covering every branch of Faddeeva::w for Im z >= 0 (Source/Faddeeva.cc:680-972):
# x = 0, y = 0, continued fraction (nu = 1, 2, general), Algorithm 916 sums with x < 5e-4, x < 10,
# y > 5, and the large-x / tiny-y tail.
BRANCH_POINTS = [(a, v)
                 for a in [0.0, 1e-12, 1e-10, 1e-6, 1e-3, 0.09, 0.11, 1.0, 4.9, 5.1, 6.9, 7.1, 50.0, 1e3]
                 for v in [0.0, 1e-5, 4e-4, 6e-4, 0.3, 3.0, 5.9, 6.1, 7.9, 8.1, 9.9, 10.1, 15.0, 27.9,
                           28.1, 300.0, 3999.0, 4001.0, 1e6, 2e7, -0.7, -12.0]]


def _random_points(n=20000, seed=0):
    rng = np.random.default_rng(seed)
    return list(zip(10 ** rng.uniform(-6, 1.2, n), 10 ** rng.uniform(-3, 3, n) * rng.choice([-1, 1], n)))


def test_oracle_voigt_vs_scipy_wofz():
    lib = OracleContext.lib_()
    pts = BRANCH_POINTS + _random_points()
    a = np.array([p[0] for p in pts])
    v = np.array([p[1] for p in pts])
    want = wofz(v + 1j * a).real
    got = np.array([lib.lworacle_voigt_H(x, y) for x, y in pts])
    assert rel_err(got, want) <= TOL_VOIGT


@pytest.mark.skipif(not bindings.ref_available(), reason='oracle/_ref not built')
def test_oracle_voigt_vs_reference_faddeeva():
    ref, ora = bindings.RefContext.lib_(), OracleContext.lib_()
    pts = BRANCH_POINTS + _random_points(5000, seed=1)
    got = np.array([ora.lworacle_voigt_H(a, v) for a, v in pts])
    want = np.array([ref.lwref_voigt_H(a, v) for a, v in pts])
    assert rel_err(got, want) <= TOL_VOIGT


@pytest.mark.parametrize('name', ['falc_h_ca_small', 'falc_h_vel'])
def test_oracle_profiles_vs_golden(name):
    """phi / wphi of the fixtures were produced by the reference's own Faddeeva (SciPy wofz is the
    same code; equality with Transition::compute_phi is asserted in test_oracle_vs_reference)."""
    base, _ = load_fixture(name)
    p = base.copy()
    for a in p.atoms:
        for t in a.trans:
            if t.type == abi.LINE:
                t.phi[...] = 0.0
                t.wphi[...] = 0.0
    OracleContext(p).compute_profiles()
    for a, b in zip(p.atoms, base.atoms):
        for t, u in zip(a.trans, b.trans):
            if t.type == abi.LINE:
                assert rel_err(t.phi, u.phi) <= TOL_VOIGT
                assert rel_err(t.wphi, u.wphi) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['falc_h_ca_small', 'falc_h_vel', 'falc2d_small'])
def test_device_profiles_vs_golden(gpu, name):
    from lightweaver_amd.context import Context
    base, _ = load_fixture(name)
    p = base.copy()
    for a in p.atoms:
        for t in a.trans:
            if t.type == abi.LINE:
                t.phi[...] = 0.0
                t.wphi[...] = 0.0
    with Context(p) as ctx:
        ctx.compute_profiles()
    for a, b in zip(p.atoms, base.atoms):
        for t, u in zip(a.trans, b.trans):
            if t.type == abi.LINE:
                assert rel_err(t.phi, u.phi) <= 1e-12
                assert rel_err(t.wphi, u.wphi) <= 1e-12


@pytest.mark.gpu
def test_device_profiles_on_a_wavelength_shard(gpu):
    """wphi is a sum over the line's whole grid: a shard that owns only part of a line must still
    get the full normalisation."""
    from lightweaver_amd.context import Context
    base, _ = load_fixture('falc_h_ca_small')
    p = base.copy()
    m = p.Nlambda // 2
    for a in p.atoms:
        for t in a.trans:
            if t.type == abi.LINE:
                t.phi[...] = 0.0
                t.wphi[...] = 0.0
    with Context(p, laStart=0, laEnd=m, worldSize=2, worldRank=0) as ctx:
        ctx.compute_profiles()
    for a, b in zip(p.atoms, base.atoms):
        for t, u in zip(a.trans, b.trans):
            if t.type == abi.LINE:
                assert rel_err(t.wphi, u.wphi) <= 1e-12
                hi = min(max(m - t.Nblue, 0), t.Nlambda)
                assert rel_err(t.phi[:hi], u.phi[:hi]) <= 1e-12
                assert not t.phi[hi:].any()


@pytest.mark.gpu
def test_device_profiles_2d_domain_beyond_one_workgroup(gpu):
    """A 2D domain of 40 x 82 = 3 280 points: phi per (line wavelength, ray, direction, point) and the wphi sums
    tiled over workgroups, against the harness's host profiles (scipy wofz)."""
    import numpy as np
    from lightweaver_amd.context import Context
    from lightweaver_amd.harness import models
    basecol = models.falc82()
    cols = [models.perturbed(basecol, seed=77 + j) for j in range(40)]
    base = models.build_problem_2d(cols, np.linspace(0.0, 39 * 30e3, 40), [models.H_6(0.3)])
    p = base.copy()
    for a in p.atoms:
        for t in a.trans:
            if t.type == abi.LINE:
                t.phi[...] = 0.0
                t.wphi[...] = 0.0
    with Context(p) as ctx:
        ctx.compute_profiles()
    for a, b in zip(p.atoms, base.atoms):
        for t, u in zip(a.trans, b.trans):
            if t.type == abi.LINE:
                assert rel_err(t.phi, u.phi) <= 1e-12
                assert rel_err(t.wphi, u.wphi) <= 1e-12


# ---- atmosphere updates with device-generated profiles (f1) ---------------------------------------------------------
def _adopt_atmosphere(p, fresh):
    """Overwrite, in place, every input of `p` that depends on the atmosphere with `fresh`'s (same model atoms, same
    grids): what the Python layer's update_deps() does after a temperature / velocity change."""
    for name in ('temperature', 'vlosMu', 'bgChi', 'bgEta', 'bgSca', 'J'):
        getattr(p, name)[...] = getattr(fresh, name)
    for a, b in zip(p.atoms, fresh.atoms):
        for name in ('n', 'nStar', 'nTotal', 'vBroad', 'C'):
            getattr(a, name)[...] = getattr(b, name)
        for t, u in zip(a.trans, b.trans):
            if t.type == abi.LINE:
                t.aDamp[...] = u.aDamp
                if t.Qelast is not None:
                    t.Qelast[...] = u.Qelast


def _oracle_run(q, niter):
    oc = OracleContext(q)
    for _ in range(niter):
        q.gamma_prefill()
        dJ, _ = oc.formal_sol_gamma_matrices()
    assert oc.stat_equil() == 0
    return dJ


def _assert_column_matches(p, q, tol=1e-9):
    assert rel_err(p.J, q.J) <= tol and rel_err(p.I, q.I) <= tol
    for a, b in zip(p.atoms, q.atoms):
        assert rel_err(a.Gamma, b.Gamma) <= 10 * tol
        assert rel_err(a.n, b.n) <= 1e-7
        for t, u in zip(a.trans, b.trans):
            if t.type == abi.LINE:
                assert rel_err(t.phi, u.phi) <= tol and rel_err(t.wphi, u.wphi) <= tol


@pytest.mark.gpu
@pytest.mark.parametrize('batched', [False, True])
def test_profiles_follow_atmosphere_updates(gpu, batched):
    """A column whose temperature and line-of-sight velocity change re-derives phi / wphi on the device: uploading
    LWHIP_ATMOS | LWHIP_NSTAR marks device-generated profiles stale and the next sweep -- of the context, or of the
    fused batch the column belongs to -- regenerates them first (compute_profile, Source/LwMiddleLayer.pyx:3244-3288;
    Transition::compute_phi, Source/LwTransition.hpp:155-190).  Against the oracle run of the perturbed column."""
    from lightweaver_amd.batch import ColumnBatch
    from lightweaver_amd.context import Context
    from lightweaver_amd.harness import models
    base = models.falc82()
    mk = lambda atmos, prof: models.falc_h_ca(Nrays=3, lineScale=0.3, atmos=atmos, computeProfiles=prof)
    ncol, j = (4, 2) if batched else (1, 0)
    probs = [mk(models.perturbed(base, seed=40 + c), False) for c in range(ncol)]
    newAtmos = models.perturbed(base, seed=977, dT=0.08, dv=3.5e3)
    up = abi.ATMOS | abi.NSTAR | abi.BACKGROUND | abi.COLLISIONS | abi.POPS | abi.J
    down = abi.ALL_OUTPUTS | abi.POPS | abi.PROFILES
    if batched:
        with ColumnBatch(probs) as batch:
            assert batch._batch is not None
            for _ in range(2):
                batch.formal_sol_gamma_matrices()
            _adopt_atmosphere(probs[j], mk(newAtmos, False))
            batch.contexts[j].upload(up)
            for _ in range(2):
                ups = batch.formal_sol_gamma_matrices()
            batch.stat_equil()
            for c in (j, 0):
                batch.contexts[c].download(down)
        dJ = ups[j].dJMax
        # an untouched column is unaffected by its neighbour's update
        q0 = mk(models.perturbed(base, seed=40), True)
        oc = OracleContext(q0)
        for _ in range(4):
            q0.gamma_prefill()
            oc.formal_sol_gamma_matrices()
        assert rel_err(probs[0].J, q0.J) <= 1e-8
    else:
        with Context(probs[0]) as ctx:
            ctx.compute_profiles(deviceResident=True)
            for _ in range(2):
                ctx.formal_sol_gamma_matrices(deviceResident=True)
            _adopt_atmosphere(probs[0], mk(newAtmos, False))
            ctx.upload(up)
            for _ in range(2):
                u = ctx.formal_sol_gamma_matrices(deviceResident=True)
            ctx.stat_equil(deviceResident=True)
            ctx.download(down)
        dJ = u.dJMax
    q = mk(newAtmos, True)
    dJref = _oracle_run(q, 2)
    assert dJ == pytest.approx(dJref, rel=1e-7)
    _assert_column_matches(probs[j], q)


@pytest.mark.gpu
def test_batch_profiles_in_one_launch_match_per_line_launches(gpu):
    """lwhip_batch_compute_profiles (every line of every column through the device list of the Voigt / wphi / retile
    kernels) against lwhip_compute_profiles of each column alone and against the oracle: phi AND wphi bit for bit (the same
    function of the same arguments; the order of wphi's wavelength sums is a function of the line alone, not of the
    launch: voigt_wphi_kernel)."""
    from lightweaver_amd.batch import ColumnBatch
    from lightweaver_amd.context import Context
    from lightweaver_amd.harness import models
    base = models.falc82()
    mk = lambda seed: models.falc_h_ca(Nrays=3, lineScale=0.4, atmos=models.perturbed(base, seed=seed, dv=2.0e3),
                                       computeProfiles=False)
    seeds = [11, 12, 13, 14, 15]
    batchProbs, soloProbs = [mk(s) for s in seeds], [mk(s) for s in seeds]
    with ColumnBatch(batchProbs) as batch:
        assert batch._batch is not None
        batch.compute_profiles()        # a second time: the tickets / lists are reusable
        ups = batch.formal_sol_gamma_matrices()
        for c in batch.contexts:
            c.download(abi.PROFILES | abi.ALL_OUTPUTS)
    for p, u in zip(soloProbs, ups):
        with Context(p) as ctx:
            ctx.compute_profiles(deviceResident=True)
            v = ctx.formal_sol_gamma_matrices(deviceResident=True)
            ctx.download(abi.PROFILES | abi.ALL_OUTPUTS)
        assert u.dJMax == pytest.approx(v.dJMax, rel=1e-11)
    q = mk(seeds[2])
    OracleContext(q).compute_profiles()
    for p, s in zip(batchProbs, soloProbs):
        for a, b in zip(p.atoms, s.atoms):
            for t, u in zip(a.trans, b.trans):
                if t.type == abi.LINE:
                    np.testing.assert_array_equal(t.phi, u.phi)
                    np.testing.assert_array_equal(t.wphi, u.wphi)
        assert rel_err(p.J, s.J) <= 1e-11
    for a, b in zip(batchProbs[2].atoms, q.atoms):
        for t, u in zip(a.trans, b.trans):
            if t.type == abi.LINE:
                assert rel_err(t.phi, u.phi) <= 1e-12 and rel_err(t.wphi, u.wphi) <= 1e-12

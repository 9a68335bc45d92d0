"""The wavefront-autonomous ("march") variant of the sweep (lightweaver_amd/csrc/lwhip_march.hip),
selected with LWHIP_SWEEP=3: same C ABI, same results.  It is not the default in this round (it is
not yet faster than the LDS-column sweep), so these tests keep it honest."""
import numpy as np
import pytest

from helpers import (FIXTURES, TOL_CONVERGED, TOL_ONE_CALL, compare_to_golden, load_fixture, rel_err,
                     variant_problem)
from lightweaver_amd import _abi as abi
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models
from oracle import bindings

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _march(monkeypatch):
    monkeypatch.setenv('LWHIP_SWEEP', '3')


@pytest.mark.parametrize('name', FIXTURES)
@pytest.mark.parametrize('variant', ['iter1', 'linear', 'besser', 'bc', 'prd_detailed', 'lambda'])
def test_march_variants_golden(gpu, name, variant):
    base, d = load_fixture(name)
    if f'out/{variant}/J' not in d:
        pytest.skip('variant not in this fixture')
    p = variant_problem(base, d, variant)
    with Context(p) as ctx:
        ctx.formal_sol_gamma_matrices(lambdaIterate=(variant == 'lambda'))
    compare_to_golden(p, d, variant)
    if variant == 'iter1':
        s = slice(None, None, 8)
        assert rel_err(p.depthI[s], d['out/iter1/depthI']) <= TOL_ONE_CALL
        assert rel_err(p.depthChi[s], d['out/iter1/depthChi']) <= TOL_ONE_CALL


@pytest.mark.parametrize('upOnly,key', [(True, 'fs_up'), (False, 'fs_all')])
def test_march_formal_sol_golden(gpu, upOnly, key):
    base, d = load_fixture('falc_h_ca_small')
    p = base.copy()
    with Context(p) as ctx:
        ctx.formal_sol(upOnly=upOnly)
    assert rel_err(p.I, d[f'out/{key}/I']) <= TOL_ONE_CALL


@pytest.mark.parametrize('Nspace,Nrays,D', [(3, 1, 0), (5, 2, 0), (64, 4, 0), (65, 1, 0), (200, 3, 0),
                                            (82, 5, 2), (82, 5, 3), (82, 5, 4), (82, 5, 8)])
def test_march_ragged_sizes_and_segment_lengths(gpu, monkeypatch, Nspace, Nrays, D):
    """Depth counts around the wavefront width and every points-per-lane instantiation."""
    if D:
        monkeypatch.setenv('LWHIP_D', str(D))
    atmos = models.resample(models.falc82(), Nspace)
    prob = models.build_problem(atmos, [models.H_6(0.2)], Nrays=Nrays)
    p, q = prob.copy(), prob.copy()
    with bindings.OracleContext(p) as oc:
        oc.formal_sol_gamma_matrices()
    with Context(q) as ctx:
        ctx.formal_sol_gamma_matrices()
    assert rel_err(q.J, p.J) <= TOL_ONE_CALL
    for a, b in zip(q.atoms, p.atoms):
        assert rel_err(a.Gamma, b.Gamma) <= TOL_ONE_CALL
        for ta, tb in zip(a.trans, b.trans):
            assert rel_err(ta.Rij, tb.Rij) <= TOL_ONE_CALL


def test_march_iter20_golden_device_resident(gpu):
    base, d = load_fixture('falc_h_ca_small')
    p = base.copy()
    with Context(p) as ctx:
        for it in range(20):
            ctx.formal_sol_gamma_matrices(deviceResident=True)
            if it >= 3:
                ctx.stat_equil(deviceResident=True)
        ctx.download(abi.ALL_OUTPUTS | abi.POPS)
    compare_to_golden(p, d, 'iter20', keys=('J', 'n'), tol=TOL_CONVERGED)


def test_march_prd(gpu):
    from test_prd import golden_prd_problem, run_iterations
    prob, gold = golden_prd_problem()
    p, upd = run_iterations(Context, prob)
    assert [u.NprdSubIter for u in upd] == list(gold['NprdSubIter'])
    assert rel_err(p.J, gold['J']) <= 1e-8

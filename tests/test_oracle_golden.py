"""The C oracle (oracle/lw_oracle.c) against the golden vectors produced by the real Lightweaver
core (tests/golden/make_golden.py).  Runs everywhere, no GPU, no reference needed."""
import numpy as np
import pytest

from helpers import (FIXTURES, TOL_CONVERGED, compare_to_golden, load_fixture, rel_err,
                     variant_problem)
from oracle.bindings import OracleContext

# The oracle follows the reference operation by operation; with -ffp-contract=off it reproduces
# the scalar scheme exactly on this toolchain.  Allow a few ulp for other libm builds.
TOL_ORACLE = 1e-13


@pytest.mark.parametrize('name', FIXTURES)
def test_iter1_and_stateq(name):
    base, d = load_fixture(name)
    p = variant_problem(base, d, 'iter1')
    oc = OracleContext(p)
    dJ = np.zeros(p.Nlambda)
    dJMax, idx = oc.formal_sol_gamma_matrices(dJ=dJ)
    compare_to_golden(p, d, 'iter1', tol=TOL_ORACLE)
    assert rel_err(dJ, d['out/iter1/dJ']) <= TOL_ORACLE
    assert rel_err(dJMax, d['out/iter1/dJMax']) <= TOL_ORACLE
    # reference single-thread bookkeeping quirk (SimdFullIterationTemplates.hpp:627) is restated
    assert idx == int(d['out/iter1/dJMaxIdx'])
    s = slice(None, None, 8)
    assert rel_err(p.depthChi[s], d['out/iter1/depthChi']) <= TOL_ORACLE
    assert rel_err(p.depthEta[s], d['out/iter1/depthEta']) <= TOL_ORACLE
    assert rel_err(p.depthI[s], d['out/iter1/depthI']) <= TOL_ORACLE
    assert oc.stat_equil() == 0
    compare_to_golden(p, d, 'stateq', keys=('n',), tol=TOL_ORACLE)


@pytest.mark.parametrize('name', FIXTURES)
@pytest.mark.parametrize('variant', ['linear', 'besser', 'bc', 'prd_detailed'])
def test_variants(name, variant):
    base, d = load_fixture(name)
    if f'out/{variant}/J' not in d:
        pytest.skip('variant not in this fixture')
    p = variant_problem(base, d, variant)
    OracleContext(p).formal_sol_gamma_matrices()
    compare_to_golden(p, d, variant, tol=TOL_ORACLE)


@pytest.mark.parametrize('name', FIXTURES)
def test_lambda_iterate(name):
    base, d = load_fixture(name)
    p = variant_problem(base, d, 'lambda')
    OracleContext(p).formal_sol_gamma_matrices(lambdaIterate=True)
    compare_to_golden(p, d, 'lambda', keys=('Gamma',), tol=TOL_ORACLE)


@pytest.mark.parametrize('name', FIXTURES)
@pytest.mark.parametrize('upOnly,key', [(True, 'fs_up'), (False, 'fs_all')])
def test_formal_sol(name, upOnly, key):
    base, d = load_fixture(name)
    p = base.copy()
    J0 = p.J.copy()
    OracleContext(p).formal_sol(upOnly=upOnly)
    assert rel_err(p.I, d[f'out/{key}/I']) <= TOL_ORACLE
    assert np.array_equal(p.J, J0)


@pytest.mark.parametrize('name', FIXTURES)
def test_iter20(name):
    base, d = load_fixture(name)
    p = base.copy()
    oc = OracleContext(p)
    for it in range(20):
        p.gamma_prefill()
        oc.formal_sol_gamma_matrices()
        if it >= 3:
            assert oc.stat_equil() == 0
    compare_to_golden(p, d, 'iter20', keys=('J', 'n'), tol=1e-11)


@pytest.mark.parametrize('name', FIXTURES)
def test_threaded_oracle_matches_single(name):
    """The OpenMP leg (cpu_baseline with cores > 1) only reorders the Gamma/R sums."""
    base, d = load_fixture(name)
    p = variant_problem(base, d, 'iter1')
    dJMax, idx = OracleContext(p).formal_sol_gamma_matrices(Nthreads=4)
    compare_to_golden(p, d, 'iter1', tol=1e-10)
    dJ = d['out/iter1/dJ']
    assert dJMax == dJ.max() and idx == int(np.argmax(dJ))


def test_singular_matrix_is_reported():
    base, d = load_fixture('falc_h_vel')
    p = base.copy()
    p.atoms[0].Gamma[...] = 0.0
    p.atoms[0].Gamma[2, :, 5] = 0.0
    oc = OracleContext(p)
    # an all-zero row (other than the eliminated one) -> "Singular Matrix" (LuSolve.cpp:22-23)
    from lightweaver_amd import _abi as abi
    assert oc.stat_equil() == abi.ERR_SINGULAR

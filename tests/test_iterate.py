"""iterate_ctx_se host mirror (lightweaver/iterate_ctx.py:85-208): the loop run to the reference's default
convergence criteria on the GPU (device resident) must take the same number of iterations and land on the same
J and populations as the same loop driven on the oracle (1e-6, BASELINE north star)."""
import numpy as np
import pytest

from helpers import TOL_CONVERGED, load_fixture, rel_err
from lightweaver_amd import _abi as abi
from lightweaver_amd.iterate import DefaultConvergenceCriteria, dPopsMax, dRhoMax, iterate_ctx_se
from lightweaver_amd.context import IterationUpdate
from oracle import bindings


class OracleLoopContext:
    """The Context surface iterate_ctx_se uses, computing on the CPU oracle (host arrays authoritative)."""

    def __init__(self, prob, ng=None):
        self.prob = prob
        self.oc = bindings.OracleContext(prob)
        self.ng = None
        if ng is not None:   # Ng objects exist from Context construction on (LwMiddleLayer.pyx:2900)
            self.ng = [bindings.NgHandle('oracle', *ng, sol=a.n) for a in prob.atoms if not a.detailed]

    def formal_sol_gamma_matrices(self, deviceResident=False):
        self.prob.gamma_prefill()
        dJ = np.zeros(self.prob.Nlambda)
        self.oc.formal_sol_gamma_matrices(dJ=dJ)
        return IterationUpdate(updatedJ=True, dJMax=float(dJ.max()), dJMaxIdx=int(np.argmax(dJ)))

    def stat_equil(self, deviceResident=False):
        act = [a for a in self.prob.atoms if not a.detailed]
        before = [a.n.copy() for a in act]
        assert self.oc.stat_equil() == 0
        if self.ng is not None:   # rel_diff_ng_accelerate (LwMiddleLayer.pyx:3318-3346)
            acc = [g.accelerate(a.n) for g, a in zip(self.ng, act)]
            mc = [g.max_change() for g in self.ng]
            return IterationUpdate(updatedPops=True, dPops=[m[0] for m in mc], dPopsMaxIdx=[m[1] for m in mc],
                                   ngAccelerated=any(acc))
        ch = [np.abs((a.n - b) / a.n).ravel() for a, b in zip(act, before)]
        return IterationUpdate(updatedPops=True, dPops=[float(c.max()) for c in ch],
                               dPopsMaxIdx=[int(np.argmax(c)) for c in ch])

    def prd_redistribute(self, maxIter=3, tol=1e-2, deviceResident=False):
        u = self.oc.redistribute_prd(maxIter, tol)
        n = u['NprdSubIter']
        return IterationUpdate(updatedRho=True, updatedJ=True, NprdSubIter=n, dRho=u['dRho'],
                               dRhoMaxIdx=u['dRhoMaxIdx'], dJPrdMax=u['dJPrdMax'], dJPrdMaxIdx=u['dJPrdMaxIdx'],
                               dJMax=float(u['dJPrdMax'][n - 1]) if n else 0.0)


def test_convergence_criteria_follow_the_reference():
    conv = DefaultConvergenceCriteria(None, 5e-3, 1e-3, None)
    J = IterationUpdate(updatedJ=True, dJMax=1e-3)
    pops = IterationUpdate(updatedPops=True, dPops=[5e-4, 2e-4])
    assert conv.is_converged(J, pops, None)
    assert not conv.is_converged(IterationUpdate(updatedJ=True, dJMax=1e-2), pops, None)
    assert not conv.is_converged(J, IterationUpdate(updatedPops=True, dPops=[5e-4, 2e-3]), None)
    prdU = IterationUpdate(updatedRho=True, NprdSubIter=2, dRho=np.array([[1.0, 2.0], [0.05, 0.2]]), dJMax=1e-4)
    assert dPopsMax(pops) == 5e-4 and dRhoMax(prdU) == 0.2
    assert conv.is_converged(J, pops, prdU)                       # rhoTol None: dRho not considered
    assert not DefaultConvergenceCriteria(None, 5e-3, 1e-3, 0.1).is_converged(J, pops, prdU)


def test_iterate_on_the_oracle_converges():
    base, _ = load_fixture('falc_h_vel')
    n = iterate_ctx_se(OracleLoopContext(base.copy()), NmaxIter=60, JTol=5e-2, popsTol=5e-2)
    assert 3 < n < 59


@pytest.mark.gpu
@pytest.mark.parametrize('prd', [False, True])
def test_iterate_ctx_se_device_resident_matches_oracle_loop(gpu, prd):
    from lightweaver_amd.context import Context
    if prd:
        from test_prd import golden_prd_problem
        prob, _ = golden_prd_problem()
    else:
        prob, _ = load_fixture('falc_h_ca_small')
    # plain MALI without Ng acceleration converges slowly on this start: loose tolerances keep the test short,
    # the point is that both loops stop at the same iteration on the same state
    kw = dict(NmaxIter=60, JTol=1e-1, popsTol=6e-2, prd=prd)
    q = prob.copy()
    nOracle, finO = iterate_ctx_se(OracleLoopContext(q), returnFinalConvergence=True, **kw)
    p = prob.copy()
    with Context(p) as ctx:
        nHip, finH = iterate_ctx_se(ctx, returnFinalConvergence=True, **kw)
        ctx.download(abi.ALL_OUTPUTS | abi.POPS | abi.RHOPRD)
    assert nHip == nOracle and 3 < nHip < 59
    assert finH[0].dJMax == pytest.approx(finO[0].dJMax, rel=1e-6)
    assert np.allclose(finH[1].dPops, finO[1].dPops, rtol=1e-5)
    assert rel_err(p.J, q.J) <= TOL_CONVERGED
    for a, b in zip(p.atoms, q.atoms):
        assert rel_err(a.n, b.n) <= TOL_CONVERGED


HAVE_REF = __import__('os').path.exists(bindings.REF_LIB)


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
@pytest.mark.parametrize('ng', [(0, 0, 0), (2, 3, 5), (3, 4, 2), (1, 1, 0)])
def test_oracle_ng_matches_the_reference_struct(ng):
    """A geometrically converging vector sequence pushed through both implementations: identical
    accelerate() decisions, solutions and max_change at every step."""
    rng = np.random.default_rng(7)
    target = 1.0 + rng.random(300)
    err = rng.standard_normal(300)
    x0 = target + err
    a, b = bindings.NgHandle('ref', *ng, sol=x0), bindings.NgHandle('oracle', *ng, sol=x0)
    xa, xb = x0.copy(), x0.copy()
    nacc = 0
    for it in range(30):
        for x in (xa, xb):           # the iteration being accelerated: a slowly converging linear map
            x[:] = target + 0.9 * (x - target) + 0.05 * np.roll(x - target, 1)
        ra, rb = a.accelerate(xa), b.accelerate(xb)
        assert ra == rb
        nacc += ra
        np.testing.assert_array_equal(xa, xb)
        assert a.max_change() == b.max_change()
    assert nacc == (0 if ng[0] == 0 else nacc) and (ng[0] == 0 or nacc >= 3)


def test_oracle_ng_without_constructor_only_tracks_changes():
    g = bindings.NgHandle('oracle')
    x = np.array([1.0, 2.0, 4.0])
    assert g.max_change() == (0.0, 0)
    assert not g.accelerate(x.copy())
    assert g.max_change() == (0.0, 0)
    assert not g.accelerate(np.array([1.0, 2.5, 4.0]))
    assert g.max_change() == (pytest.approx(0.2), 1)


@pytest.mark.gpu
def test_iterate_with_ng_acceleration_matches_oracle_loop(gpu):
    """NgOptions(2, 3, 5) on the device against the same loop on the oracle with the restated Ng: same
    acceleration steps, same iteration count, same converged state; and fewer iterations than plain MALI."""
    from lightweaver_amd.context import Context
    prob, _ = load_fixture('falc_h_ca_small')
    kw = dict(NmaxIter=80, JTol=5e-2, popsTol=3e-2)
    ng = (2, 3, 5)
    q = prob.copy()
    nOracle = iterate_ctx_se(OracleLoopContext(q, ng=ng), **kw)
    nPlain = iterate_ctx_se(OracleLoopContext(prob.copy()), **kw)
    p = prob.copy()
    with Context(p) as ctx:
        ctx.configure_ng(*ng)
        nHip = iterate_ctx_se(ctx, **kw)
        ctx.download(abi.ALL_OUTPUTS | abi.POPS)
    assert nHip == nOracle < nPlain
    assert rel_err(p.J, q.J) <= TOL_CONVERGED
    for a, b in zip(p.atoms, q.atoms):
        assert rel_err(a.n, b.n) <= TOL_CONVERGED

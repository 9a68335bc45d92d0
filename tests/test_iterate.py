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

    def __init__(self, prob):
        self.prob = prob
        self.oc = bindings.OracleContext(prob)

    def formal_sol_gamma_matrices(self, deviceResident=False):
        self.prob.gamma_prefill()
        dJ = np.zeros(self.prob.Nlambda)
        self.oc.formal_sol_gamma_matrices(dJ=dJ)
        return IterationUpdate(updatedJ=True, dJMax=float(dJ.max()), dJMaxIdx=int(np.argmax(dJ)))

    def stat_equil(self, deviceResident=False):
        act = [a for a in self.prob.atoms if not a.detailed]
        before = [a.n.copy() for a in act]
        assert self.oc.stat_equil() == 0
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

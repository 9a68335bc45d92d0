"""Hybrid PRD (Transition::uv with the per-ray rho interpolation, the JRest scatter of
intensity_core_opt, the scattering integral on JRest; Source/LwTransition.hpp:116-127,
Source/SimdFullIterationTemplates.hpp:397-408, Source/Prd.cpp:384-389, 697-946).

CPU: the oracle's restatement of configure_hprd_coeffs against the tables the real core builds (oracle/_ref), bit for
bit; the oracle's hybrid-PRD iterations against the core's, bit for bit.
GPU: the HIP path (lwhip_options.hprd) against the oracle at the one-call bound."""
import os

import numpy as np
import pytest

from lightweaver_amd import _abi as abi
from lightweaver_amd.harness import models
from oracle import bindings
from tests.helpers import TOL_ONE_CALL, rel_err

HAVE_REF = os.path.exists(bindings.REF_LIB)


def hprd_problem(Nrays=3, lineScale=0.3, vamp=8.0e3, seed=7, formalSolver=None):
    """FAL-C H + Ca II with Ca II H & K as PRD lines on a column with a few km/s of line-of-sight velocity (what makes
    the rest-frame and observer-frame grids differ)."""
    atmos = models.falc82()
    rng = np.random.default_rng(seed)
    k = np.arange(atmos.Nspace)
    atmos.vlos = vamp * np.sin(2.0 * np.pi * k / 37.0) + 0.2 * vamp * rng.standard_normal(atmos.Nspace)
    kw = {} if formalSolver is None else {'formalSolver': formalSolver}
    return models.falc_h_ca(Nrays=Nrays, lineScale=lineScale, prd=True, atmos=atmos, **kw)


def assert_tables_equal(a, b):
    np.testing.assert_array_equal(a.prdIdxs, b.prdIdxs)
    np.testing.assert_array_equal(a.hPrdIdxs, b.hPrdIdxs)
    np.testing.assert_array_equal(a.jCoeffOff, b.jCoeffOff)
    np.testing.assert_array_equal(a.jIdx, b.jIdx)
    np.testing.assert_array_equal(a.jFrac, b.jFrac)
    np.testing.assert_array_equal(a.lineAtom, b.lineAtom)
    np.testing.assert_array_equal(a.lineTrans, b.lineTrans)
    assert len(a.rho) == len(b.rho)
    for (i0, i1, f), (j0, j1, g) in zip(a.rho, b.rho):
        np.testing.assert_array_equal(i0, j0)
        np.testing.assert_array_equal(i1, j1)
        np.testing.assert_array_equal(f, g)


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built')
@pytest.mark.parametrize('vamp', [0.0, 8.0e3, 40.0e3])
def test_oracle_hprd_tables_match_core(vamp):
    prob = hprd_problem(vamp=vamp)
    with bindings.RefContext(prob.copy()) as rc, bindings.OracleContext(prob.copy()) as oc:
        tr = rc.configure_hprd()
        to = oc.build_hprd()
        assert tr.prdIdxs.size > 0 and tr.hPrdIdxs.size >= tr.prdIdxs.size
        assert_tables_equal(to, tr)
        to.close()
        tr.close()


def run_hprd(ctxFactory, prob, nIter=3, prdIter=3, tol=1e-2, Nthreads=None):
    """iterate_ctx_se with hybrid PRD (lightweaver/iterate_ctx.py:85-208; LwContext(hprd=True) configures the tables
    once at construction, Source/LwMiddleLayer.pyx:2822-2826): formal solution, statistical equilibrium, PRD
    sub-iterations; returns the problem, the per-iteration PRD results and JRest."""
    p = prob.copy()
    out = []
    with ctxFactory(p) as ctx:
        tables = ctx.configure_hprd() if hasattr(ctx, 'configure_hprd') else None
        if tables is None:
            tables = ctx.build_hprd()
            ctx.set_hprd(tables)
        for it in range(nIter):
            p.gamma_prefill()
            if Nthreads is None:
                ctx.formal_sol_gamma_matrices()
            else:
                ctx.formal_sol_gamma_matrices(Nthreads=Nthreads)
            if it >= 1:
                ctx.stat_equil()
            out.append(ctx.redistribute_prd(prdIter, tol))
        JRest = tables.JRest.copy()
        if hasattr(ctx, 'set_hprd'):
            ctx.set_hprd(None)
        tables.close()
    return p, out, JRest


def assert_same(pa, pb, tol):
    assert rel_err(pa.J, pb.J) <= tol
    assert rel_err(pa.I, pb.I) <= tol
    for a, b in zip(pa.atoms, pb.atoms):
        assert rel_err(a.n, b.n) <= tol
        if not a.detailed:
            assert rel_err(a.Gamma, b.Gamma) <= tol
        for ta, tb in zip(a.trans, b.trans):
            assert rel_err(ta.Rij, tb.Rij) <= tol and rel_err(ta.Rji, tb.Rji) <= tol
            if ta.rhoPrd is not None:
                assert rel_err(ta.rhoPrd, tb.rhoPrd) <= tol


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built')
def test_oracle_hprd_iterations_match_core_bitwise():
    prob = hprd_problem()
    pr, ur, Jr = run_hprd(bindings.RefContext, prob)
    po, uo, Jo = run_hprd(bindings.OracleContext, prob)
    assert np.abs(Jr).max() > 0.0
    np.testing.assert_array_equal(Jo, Jr)
    np.testing.assert_array_equal(po.J, pr.J)
    for a, b in zip(po.atoms, pr.atoms):
        np.testing.assert_array_equal(a.n, b.n)
        for ta, tb in zip(a.trans, b.trans):
            np.testing.assert_array_equal(ta.Rij, tb.Rij)
            if ta.rhoPrd is not None:
                np.testing.assert_array_equal(ta.rhoPrd, tb.rhoPrd)
    for x, y in zip(uo, ur):
        assert x['NprdSubIter'] == y['NprdSubIter']
        np.testing.assert_array_equal(x['dRho'], y['dRho'])
        np.testing.assert_array_equal(x['dJPrdMax'], y['dJPrdMax'])


def test_hybrid_prd_differs_from_plain_prd():
    """The velocities matter: with them the hybrid scheme's rho differs from angle-averaged PRD's (sanity of the fixture)."""
    prob = hprd_problem()
    ph, _, _ = run_hprd(bindings.OracleContext, prob, nIter=1)
    q = prob.copy()
    with bindings.OracleContext(q) as oc:
        q.gamma_prefill()
        oc.formal_sol_gamma_matrices()
        oc.redistribute_prd(3, 1e-2)
    lines = lambda p: [t for a in p.atoms for t in a.trans if t.rhoPrd is not None]
    assert max(np.abs(a.rhoPrd - b.rhoPrd).max() for a, b in zip(lines(ph), lines(q))) > 1e-3


# ---- the HIP path ------------------------------------------------------------------------------------------------------
def run_hprd_hip(prob, nIter=3, prdIter=3, tol=1e-2):
    """The same loop through the C ABI: the tables come from the oracle's builder (pinned to the core above)."""
    from lightweaver_amd.context import Context
    p = prob.copy()
    out = []
    with bindings.OracleContext(p.copy()) as oc:
        tables = oc.build_hprd()
        with Context(p, hprd=tables) as ctx:
            assert ctx.sweep_kind() == 'lanes'
            for it in range(nIter):
                p.gamma_prefill()
                ctx.formal_sol_gamma_matrices()
                if it >= 1:
                    ctx.stat_equil()
                out.append(ctx.redistribute_prd(prdIter, tol))
            ctx.download(abi.J)
        JRest = tables.JRest.copy()
        tables.close()
    return p, out, JRest


@pytest.mark.gpu
def test_hip_hprd_single_call_matches_oracle(gpu):
    """One formal solution + one PRD call from identical inputs: the 1e-9 one-call bound on J, JRest, rho, Gamma, rates."""
    prob = hprd_problem()
    po, uo, Jo = run_hprd(bindings.OracleContext, prob, nIter=1, prdIter=2)
    ph, uh, Jh = run_hprd_hip(prob, nIter=1, prdIter=2)
    assert uo[0]['NprdSubIter'] == uh[0].NprdSubIter == 2
    assert np.allclose(uo[0]['dRho'], uh[0].dRho, rtol=1e-6)
    assert rel_err(Jh, Jo) <= TOL_ONE_CALL
    assert_same(po, ph, TOL_ONE_CALL)


@pytest.mark.gpu
@pytest.mark.parametrize('solver', ['linear', 'besser'])
def test_hip_hprd_other_solvers_match_oracle(gpu, solver):
    """Hybrid PRD under the other two 1D formal solvers (the lane sweep's linear / BESSER coefficients): one formal solution +
    one PRD call against the oracle at the one-call bound."""
    from lightweaver_amd import _abi as abi
    prob = hprd_problem(formalSolver={'linear': abi.FS_LINEAR_1D, 'besser': abi.FS_BESSER_1D}[solver])
    po, uo, Jo = run_hprd(bindings.OracleContext, prob, nIter=1, prdIter=2)
    ph, uh, Jh = run_hprd_hip(prob, nIter=1, prdIter=2)
    assert uo[0]['NprdSubIter'] == uh[0].NprdSubIter == 2
    assert rel_err(Jh, Jo) <= TOL_ONE_CALL
    assert_same(po, ph, TOL_ONE_CALL)


@pytest.mark.gpu
@pytest.mark.parametrize('vamp', [8.0e3, 40.0e3])
def test_hip_hprd_iterations_match_oracle(gpu, vamp):
    prob = hprd_problem(vamp=vamp)
    po, uo, Jo = run_hprd(bindings.OracleContext, prob)
    ph, uh, Jh = run_hprd_hip(prob)
    for a, b in zip(uo, uh):
        assert a['NprdSubIter'] == b.NprdSubIter
        assert np.allclose(a['dRho'], b.dRho, rtol=1e-5, atol=1e-12)
        assert np.allclose(a['dJPrdMax'], b.dJPrdMax, rtol=1e-7, atol=1e-12)
    assert rel_err(Jh, Jo) <= 1e-8
    assert_same(po, ph, 1e-8)  # three full iterations deep: the one-call bound compounded


@pytest.mark.gpu
def test_hip_hprd_on_wavelength_shards(gpu):
    """Hybrid PRD on two wavelength shards (round 4; the reference's hybrid path has no such restriction,
    SimdFullIterationTemplates.hpp:397-408, Prd.cpp:384-389): two contexts owning [0, m) and [m, Nlambda) on one device, m
    inside Ca II K's grid, the two all-reduces of a sub-iteration emulated through the library's own buffers.  Every rank keeps
    rho of the lines' whole grids (a ray's rho is interpolated across the cut) and its partial JRest rides in the all-reduce of
    the gather buffer: the sharded iteration + sub-iterations must reproduce the unsharded ones."""
    import torch
    from lightweaver_amd.context import Context
    from lightweaver_amd.distributed import DeviceReduceBuffer

    def allreduce(ctxs, getbuf):
        views = []
        for c in ctxs:
            ptr, n = getbuf(c)
            c.synchronize()
            views.append(torch.as_tensor(DeviceReduceBuffer(ptr, n), device='cuda'))
        total = sum(v.clone() for v in views)
        for v in views:
            v.copy_(total)
        torch.cuda.synchronize()

    prob = hprd_problem()
    lines = lambda p: [t for a in p.atoms for t in a.trans if t.rhoPrd is not None]
    last = lines(prob)[-1]
    m = (last.Nblue + last.Nred) // 2
    with bindings.OracleContext(prob.copy()) as oc:
        tables = oc.build_hprd()
        whole = prob.copy()
        with Context(whole, hprd=tables) as ctx:
            ctx.formal_sol_gamma_matrices(deviceResident=True)
            uw = ctx.prd_redistribute(3, 1e-30, deviceResident=True)
            ctx.download(abi.J | abi.RATES | abi.RHOPRD)
        pa, pb = prob.copy(), prob.copy()
        with Context(pa, laStart=0, laEnd=m, worldSize=2, worldRank=0, hprd=tables) as A, \
                Context(pb, laStart=m, laEnd=prob.Nlambda, worldSize=2, worldRank=1, hprd=tables) as B:
            ctxs = (A, B)
            for c in ctxs:
                assert c.sweep_kind() == 'lanes'
                c.gamma_prefill_from_C(1.0)
                c.fs_partial()
            allreduce(ctxs, lambda c: c.reduce_buffer())
            for c in ctxs:
                c.fs_finalise()
            for it in range(3):
                allreduce(ctxs, lambda c: c.prd_pack())
                for c in ctxs:
                    c.prd_partial()
                allreduce(ctxs, lambda c: c.reduce_buffer())
                res = [c.prd_finalise() for c in ctxs]
                for dRho, idx, dJ, dJIdx in res:         # every rank sees the same global maxima
                    assert np.allclose(dRho, uw.dRho[it], rtol=1e-7)
                    assert dJ == pytest.approx(uw.dJPrdMax[it], rel=1e-7)
            for c in ctxs:
                c.download(abi.J | abi.RATES | abi.RHOPRD)
        tables.close()
    assert rel_err(pa.J[:m], whole.J[:m]) <= 1e-9 and rel_err(pb.J[m:], whole.J[m:]) <= 1e-9
    for ta, tb, tw in zip(lines(pa), lines(pb), lines(whole)):
        # (rho of the whole grid on both ranks)
        assert rel_err(ta.rhoPrd, tw.rhoPrd) <= 1e-9 and rel_err(tb.rhoPrd, tw.rhoPrd) <= 1e-9
        for t in (ta, tb):
            assert rel_err(t.Rij, tw.Rij) <= 1e-9 and rel_err(t.Rji, tw.Rji) <= 1e-9

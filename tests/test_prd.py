"""PRD sub-iterations (BASELINE configs[2]).

CPU: the C restatement (oracle/lw_oracle_prd.inc) against the real core compiled from
/root/reference (oracle/_ref, lwref_redistribute_prd) and against the committed golden outputs.
GPU: lwhip_redistribute_prd through the C ABI against the oracle on the same inputs.
Tolerance: 1e-9 relative after one call (BASELINE.json north_star), max-norm scaled per array."""
import os

import numpy as np
import pytest

from lightweaver_amd import _abi as abi
from lightweaver_amd.harness import models
from oracle import bindings
from tests.helpers import GOLDEN, TOL_ONE_CALL, load_fixture, rebuild, rel_err

HAVE_REF = os.path.exists(bindings.REF_LIB)


def prd_problem(**kw):
    args = dict(Nrays=3, lineScale=0.3, prd=True)
    args.update(kw)
    return models.falc_h_ca(**args)


def run_iterations(ctxFactory, prob, nIter=3, prdIter=3, tol=1e-2):
    """The loop of iterate_ctx_se with PRD (lightweaver/iterate_ctx.py:85-208): formal solution,
    statistical equilibrium, then PRD sub-iterations."""
    p = prob.copy()
    out = []
    with ctxFactory(p) as ctx:
        for it in range(nIter):
            p.gamma_prefill()
            ctx.formal_sol_gamma_matrices()
            if it >= 1:
                ctx.stat_equil()
            out.append(ctx.redistribute_prd(prdIter, tol))
    return p, out


def apply_prd_delta(base, d):
    """The PRD problem of fixture falc_h_ca_prd: fixture falc_h_ca_small with the listed lines made
    PRD (rhoPrd = 1, their Qelast) and depth data off (see tests/golden/make_golden.py)."""
    p = base.copy()
    p.storeDepthData = False
    for q in range(int(d['prd/N'])):
        ia, kr = (int(x) for x in d[f'prd/line{q}'])
        t = p.atoms[ia].trans[kr]
        t.Qelast = np.array(d[f'prd/Qelast{q}'], dtype=np.float64)
        t.rhoPrd = np.ones((t.Nred - t.Nblue, p.Nspace))
    return rebuild(p)


def golden_prd_problem():
    base, _ = load_fixture('falc_h_ca_small')
    with np.load(os.path.join(GOLDEN, 'falc_h_ca_prd.npz')) as z:
        d = {k: z[k] for k in z.files}
    return apply_prd_delta(base, d), d


def prd_lines(p):
    return [t for a in p.atoms if not a.detailed for t in a.trans if t.type == 0 and t.rhoPrd is not None]


def assert_same_state(pa, pb, tol):
    assert rel_err(pa.J, pb.J) <= tol
    assert rel_err(pa.I, pb.I) <= tol  # (the rates pass of a sub-iteration rewrites spect.I of its wavelengths)
    for a, b in zip(pa.atoms, pb.atoms):
        assert rel_err(a.n, b.n) <= tol
        for ta, tb in zip(a.trans, b.trans):
            assert rel_err(ta.Rij, tb.Rij) <= tol
            assert rel_err(ta.Rji, tb.Rji) <= tol
            if ta.rhoPrd is not None:
                assert rel_err(ta.rhoPrd, tb.rhoPrd) <= tol


def test_GII_known_values():
    lib = bindings.OracleContext.lib_()
    # outside the support (Prd.cpp:71-73, 101-103)
    assert lib.lworacle_GII(1e-3, 1.0, -4.5) == 0.0
    assert lib.lworacle_GII(1e-3, 6.0, 0.0) == 0.0
    # symmetric under (qEmit, qAbs) -> (-qEmit, -qAbs) (Prd.cpp:56-60)
    for qe, qa in [(0.3, 1.2), (2.5, -1.0), (5.0, 7.5)]:
        assert lib.lworacle_GII(2e-3, qe, qa) == lib.lworacle_GII(2e-3, -qe, -qa)
    # core branch, |qAbs| <= qEmit < 2: G0(qEmit) = 1 / (|q| + sqrt(q^2 + 4/pi))
    q = 1.5
    assert lib.lworacle_GII(1e-3, q, 0.5) == pytest.approx(1.0 / (q + np.sqrt(q * q + 1.273239545)), rel=1e-15)


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
def test_oracle_GII_matches_reference():
    ref = bindings.RefContext.lib_()
    orc = bindings.OracleContext.lib_()
    rng = np.random.default_rng(5)
    for _ in range(2000):
        a = 10 ** rng.uniform(-4, -1)
        qe = rng.uniform(-8, 8)
        qa = qe + rng.uniform(-6, 6)
        assert orc.lworacle_GII(a, qe, qa) == ref.lwref_GII(a, qe, qa)


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
@pytest.mark.parametrize('solver', [2, 0])
def test_oracle_prd_matches_reference(solver):
    prob = prd_problem()
    prob.formalSolver = solver
    pr, ur = run_iterations(bindings.RefContext, prob)
    po, uo = run_iterations(bindings.OracleContext, prob)
    for a, b in zip(ur, uo):
        assert a['NprdSubIter'] == b['NprdSubIter'] and a['Nprd'] == b['Nprd'] == 2
        np.testing.assert_array_equal(a['dRho'], b['dRho'])
        np.testing.assert_array_equal(a['dRhoMaxIdx'], b['dRhoMaxIdx'])
        np.testing.assert_array_equal(a['dJPrdMax'], b['dJPrdMax'])
        np.testing.assert_array_equal(a['dJPrdMaxIdx'], b['dJPrdMaxIdx'])
    assert_same_state(pr, po, 0.0)
    rho = np.concatenate([t.rhoPrd.ravel() for t in prd_lines(po)])
    assert np.isfinite(rho).all() and np.abs(rho - 1.0).max() > 1e-3  # PRD actually did something


def detailed_prd_problem():
    """An active atom with a PRD line (H Lyman-alpha made PRD) next to a DETAILED atom with PRD lines (Ca II H & K):
    what `include_detailed_atoms` (the default of LwContext.prd_redistribute, LwMiddleLayer.pyx:3678-3680) is about."""
    p, _ = golden_prd_problem()
    h, ca = p.atoms
    t = next(t for t in h.trans if t.type == abi.LINE)
    t.rhoPrd = np.ones((t.Nred - t.Nblue, p.Nspace))
    ca.detailed = True
    return rebuild(p)


def run_detailed(ctxFactory, prob, include, nIter=2):
    p = prob.copy()
    out = []
    with ctxFactory(p) as ctx:
        for it in range(nIter):
            p.gamma_prefill()
            ctx.formal_sol_gamma_matrices()
            ctx.stat_equil()
            out.append(ctx.redistribute_prd(3, 1e-3, include_detailed_atoms=include))
    return p, out


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
@pytest.mark.parametrize('include', [True, False])
def test_oracle_prd_detailed_atoms_matches_reference(include):
    """redistribute_prd_lines with and without include_detailed_atoms (PrdTemplates.hpp:25-29, 190-215): the
    detailed atom's PRD lines join the list after the active atoms' (their Pj + Qj has no collisional term: a
    detailed atom carries no C), their rates are zeroed and re-integrated by the PRD rates pass.  Bit for bit."""
    prob = detailed_prd_problem()
    pr, ur = run_detailed(bindings.RefContext, prob, include)
    po, uo = run_detailed(bindings.OracleContext, prob, include)
    for a, b in zip(ur, uo):
        assert a['NprdSubIter'] == b['NprdSubIter'] and a['Nprd'] == b['Nprd'] == (3 if include else 1)
        for k in ('dRho', 'dRhoMaxIdx', 'dJPrdMax', 'dJPrdMaxIdx'):
            np.testing.assert_array_equal(a[k], b[k])
    assert_same_state(pr, po, 0.0)
    ca = po.atoms[1]
    moved = max(np.abs(t.rhoPrd - 1.0).max() for t in ca.trans if t.rhoPrd is not None)
    assert (moved > 1e-3) == include       # the detailed lines are redistributed only when asked for


def test_oracle_prd_golden():
    """The oracle against the reference's committed outputs (generated by tests/golden/make_golden.py
    from the real core): bit-for-bit."""
    prob, gold = golden_prd_problem()
    p, upd = run_iterations(bindings.OracleContext, prob)
    assert [u['NprdSubIter'] for u in upd] == list(gold['NprdSubIter'])
    np.testing.assert_array_equal(np.concatenate([u['dRho'].ravel() for u in upd]), gold['dRho'])
    np.testing.assert_array_equal(np.concatenate([u['dRhoMaxIdx'].ravel() for u in upd]), gold['dRhoMaxIdx'])
    np.testing.assert_array_equal(np.concatenate([u['dJPrdMax'] for u in upd]), gold['dJPrdMax'])
    np.testing.assert_array_equal(p.J, gold['J'])
    for q, t in enumerate(prd_lines(p)):
        np.testing.assert_array_equal(t.rhoPrd, gold[f'rho{q}'])
        np.testing.assert_array_equal(t.Rij, gold[f'Rij{q}'])
        np.testing.assert_array_equal(t.Rji, gold[f'Rji{q}'])
    for ia, a in enumerate(p.atoms):
        np.testing.assert_array_equal(a.n, gold[f'n{ia}'])


@pytest.mark.gpu
def test_hip_prd_golden():
    """The HIP path against the reference's committed outputs."""
    from lightweaver_amd.context import Context
    prob, gold = golden_prd_problem()
    p, upd = run_iterations(Context, prob)
    assert [u.NprdSubIter for u in upd] == list(gold['NprdSubIter'])
    assert np.allclose(np.concatenate([u.dRho.ravel() for u in upd]), gold['dRho'], rtol=1e-5)
    assert rel_err(p.J, gold['J']) <= 1e-8
    for q, t in enumerate(prd_lines(p)):
        assert rel_err(t.rhoPrd, gold[f'rho{q}']) <= 1e-8
        assert rel_err(t.Rij, gold[f'Rij{q}']) <= 1e-8


def test_prd_tolerance_stops_early():
    prob = prd_problem()
    p, upd = run_iterations(bindings.OracleContext, prob, nIter=1, prdIter=5, tol=1e30)
    assert upd[0]['NprdSubIter'] == 1


@pytest.mark.gpu
@pytest.mark.parametrize('solver', [2, 0, 1])
def test_hip_prd_matches_oracle(solver):
    from lightweaver_amd.context import Context
    prob = prd_problem()
    prob.formalSolver = solver
    po, uo = run_iterations(bindings.OracleContext, prob)
    ph, uh = run_iterations(Context, prob)
    for a, b in zip(uo, uh):
        assert a['NprdSubIter'] == b.NprdSubIter
        # max |d rho / rho| is attained where rho passes close to 0 in this (deliberately rough)
        # start, which amplifies the 1e-11 differences in rho itself; the state below is the check
        assert np.allclose(a['dRho'], b.dRho, rtol=1e-5, atol=1e-12)
        assert np.allclose(a['dJPrdMax'], b.dJPrdMax, rtol=1e-7, atol=1e-12)
    assert_same_state(po, ph, 1e-8)  # three full iterations deep: the one-call bound compounded


@pytest.mark.gpu
def test_hip_prd_loop_stops_where_the_reference_loop_stops(monkeypatch):
    """redistribute_prd(maxIter, tol) stops after the first sub-iteration whose largest |d rho / rho| is below tol
    (PrdTemplates.hpp:250-275).  The library queues up to four sub-iterations' launches without a host round trip and keeps
    that rule on the device (the launches of sub-iterations after the stop return at once): same number of sub-iterations as
    the oracle for a tolerance that stops the loop in the MIDDLE of a queued group, in the next group and never; the state
    equals the oracle's, and equals what the library does with a host round trip per sub-iteration (LWHIP_PRD_PIPELINE=0)."""
    from lightweaver_amd.context import Context
    prob = prd_problem()

    def run(factory, maxIter, tol):
        p = prob.copy()
        with factory(p) as ctx:
            p.gamma_prefill()
            ctx.formal_sol_gamma_matrices()
            u = ctx.redistribute_prd(maxIter, tol)
        return p, u

    _, u0 = run(bindings.OracleContext, 7, 0.0)
    assert u0['NprdSubIter'] == 7
    m = np.max(np.asarray(u0['dRho']).reshape(7, -1), axis=1)   # the largest change of each sub-iteration
    assert np.all(m[1:] < m[:-1]), m   # (it shrinks from one to the next: a tolerance between two values stops at a known place)
    for stopAfter in (2, 3, 5, 6):     # 1-based sub-iteration that is the first below the tolerance
        tol = float(np.sqrt(m[stopAfter - 1] * m[stopAfter - 2]))
        po, uo = run(bindings.OracleContext, 7, tol)
        assert uo['NprdSubIter'] == stopAfter
        ph, uh = run(Context, 7, tol)
        assert uh.NprdSubIter == stopAfter, (stopAfter, uh.NprdSubIter)
        assert np.allclose(np.asarray(uo['dRho']).reshape(-1)[:uh.dRho.size], np.asarray(uh.dRho).reshape(-1), rtol=1e-5, atol=1e-12)
        assert_same_state(po, ph, 1e-9)
        monkeypatch.setenv('LWHIP_PRD_PIPELINE', '0')
        pn, un = run(Context, 7, tol)
        monkeypatch.delenv('LWHIP_PRD_PIPELINE')
        assert un.NprdSubIter == stopAfter
        assert_same_state(pn, ph, 1e-10)


@pytest.mark.gpu
def test_hip_prd_cached_kernel_gives_the_general_kernels_rho(monkeypatch):
    """Sub-iterations after the first run `prd_scatter_cached_kernel` (brackets of the slice's window straight from J, no
    transposed copy); `LWHIP_PRD_GENERAL=1` keeps every sub-iteration on the general kernel.  Both evaluate the interpolation
    through the same two functions: rho, and everything derived from it, is the same to the last bit (fixed-order contexts,
    so that the rates entering Pj + Qj carry no run-to-run difference of their own)."""
    from lightweaver_amd.context import Context
    prob = prd_problem()

    def run():
        p = prob.copy()
        with Context(p, deterministic=True) as ctx:
            for _ in range(2):
                p.gamma_prefill()
                ctx.formal_sol_gamma_matrices()
                u = ctx.redistribute_prd(4, 0.0)
        rho = [t.rhoPrd.copy() for a in p.atoms for t in a.trans if t.rhoPrd is not None]
        return rho, p.J.copy(), np.asarray(u.dRho).copy()

    rc, Jc, dc = run()
    monkeypatch.setenv('LWHIP_PRD_GENERAL', '1')
    rg, Jg, dg = run()
    assert len(rc) == len(rg) > 0
    for a, b in zip(rc, rg):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(Jc, Jg)
    np.testing.assert_array_equal(dc, dg)


@pytest.mark.gpu
@pytest.mark.parametrize('include', [True, False])
def test_hip_prd_detailed_atoms_matches_oracle(include):
    """include_detailed_atoms on the device: a Context created with prdDetailed redistributes the detailed atom's PRD
    lines too (no collisional term in their Pj + Qj), visits their wavelengths in the PRD rates pass and rewrites their
    rates; without it their sums are added on top, as the reference does."""
    from lightweaver_amd.context import Context
    prob = detailed_prd_problem()
    po, uo = run_detailed(bindings.OracleContext, prob, include)
    ph, uh = run_detailed(lambda p: Context(p, prdDetailed=include), prob, include)
    for a, b in zip(uo, uh):
        assert a['NprdSubIter'] == b.NprdSubIter and b.dRho.shape[1] == (3 if include else 1)
        assert np.allclose(a['dRho'], b.dRho, rtol=1e-5, atol=1e-12)
        assert np.allclose(a['dJPrdMax'], b.dJPrdMax, rtol=1e-7, atol=1e-12)
    assert_same_state(po, ph, 1e-8)


@pytest.mark.gpu
def test_hip_prd_single_call():
    """One formal solution + one PRD call from identical inputs: the 1e-9 one-call bound."""
    from lightweaver_amd.context import Context
    prob = prd_problem()
    po, uo = run_iterations(bindings.OracleContext, prob, nIter=1, prdIter=2)
    ph, uh = run_iterations(Context, prob, nIter=1, prdIter=2)
    assert uo[0]['NprdSubIter'] == uh[0].NprdSubIter == 2
    np.testing.assert_array_equal(uo[0]['dRhoMaxIdx'], uh[0].dRhoMaxIdx)
    assert np.allclose(uo[0]['dRho'], uh[0].dRho, rtol=1e-9)
    assert_same_state(po, ph, TOL_ONE_CALL)


@pytest.mark.gpu
def test_hip_prd_device_resident_matches_drop_in():
    from lightweaver_amd.context import Context
    prob = prd_problem()
    pa, _ = run_iterations(Context, prob, nIter=2)
    pb = prob.copy()
    with Context(pb) as ctx:
        for it in range(2):
            ctx.formal_sol_gamma_matrices(deviceResident=True)
            if it >= 1:
                ctx.stat_equil(deviceResident=True)
            ctx.prd_redistribute(3, 1e-2, deviceResident=True)
        ctx.download()
        ctx.download(abi.RHOPRD | abi.POPS)
    assert_same_state(pa, pb, 1e-8)  # fp64 LDS atomics: the summation order differs run to run


@pytest.mark.gpu
def test_hip_prd_single_call_refused_on_a_shard():
    from lightweaver_amd.context import Context, LwHipError
    prob = prd_problem()
    with Context(prob, laStart=0, laEnd=prob.wavelength.shape[0] // 2, worldSize=2, worldRank=0) as ctx:
        with pytest.raises(LwHipError):
            ctx.prd_redistribute()


@pytest.mark.gpu
def test_hip_prd_on_wavelength_shards(gpu):
    """Two contexts owning [0, m) and [m, Nlambda) on one device, the two all-reduces emulated on the host
    through the library's own buffers: the sharded sub-iterations must reproduce the unsharded ones
    (the split cuts Ca II K's grid in two)."""
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

    prob = prd_problem()
    lines = prd_lines(prob)
    m = (lines[-1].Nblue + lines[-1].Nred) // 2      # inside the last PRD line's grid
    whole = prob.copy()
    with Context(whole) as ctx:
        ctx.formal_sol_gamma_matrices(deviceResident=True)
        uw = ctx.prd_redistribute(3, 1e-30, deviceResident=True)
        ctx.download(abi.J | abi.RATES | abi.RHOPRD)
    pa, pb = prob.copy(), prob.copy()
    with Context(pa, laStart=0, laEnd=m, worldSize=2, worldRank=0) as A, \
            Context(pb, laStart=m, laEnd=prob.Nlambda, worldSize=2, worldRank=1) as B:
        ctxs = (A, B)
        for c in ctxs:
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
                assert np.allclose(dRho, uw.dRho[it], rtol=1e-9)
                np.testing.assert_array_equal(idx, uw.dRhoMaxIdx[it])
                assert dJ == pytest.approx(uw.dJPrdMax[it], rel=1e-9) and dJIdx == uw.dJPrdMaxIdx[it]
        for c in ctxs:
            c.download(abi.J | abi.RATES | abi.RHOPRD)
    assert rel_err(pa.J[:m], whole.J[:m]) <= 1e-10 and rel_err(pb.J[m:], whole.J[m:]) <= 1e-10
    for ta, tb, tw in zip(prd_lines(pa), prd_lines(pb), prd_lines(whole)):
        cut = min(max(m - tw.Nblue, 0), tw.Nred - tw.Nblue)
        assert rel_err(ta.rhoPrd[:cut], tw.rhoPrd[:cut]) <= 1e-10
        assert rel_err(tb.rhoPrd[cut:], tw.rhoPrd[cut:]) <= 1e-10
        for t in (ta, tb):
            assert rel_err(t.Rij, tw.Rij) <= 1e-10 and rel_err(t.Rji, tw.Rji) <= 1e-10

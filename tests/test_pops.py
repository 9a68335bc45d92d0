"""Population updates beside statistical equilibrium: the backward-Euler
time_dependent_update and the Newton-Raphson charge-conservation nr_post_update.

CPU: the C restatement (oracle/lw_oracle.c) against the real core compiled from /root/reference
(oracle/_ref).  GPU: lwhip_time_dep_update / lwhip_nr_post_update through the C ABI against the
oracle on the same inputs.  Tolerance 1e-9 relative (the solves are well conditioned here)."""
import os

import numpy as np
import pytest

from helpers import load_fixture, rel_err, variant_problem
from oracle import bindings

HAVE_REF = os.path.exists(bindings.REF_LIB)
TOL = 1e-9


def problem_after_one_iteration(ctxFactory, name='falc_h_ca_small'):
    base, d = load_fixture(name)
    p = variant_problem(base, d, 'iter1')
    ctx = ctxFactory(p)
    p.gamma_prefill()
    ctx.formal_sol_gamma_matrices()
    return p, ctx


def active(p):
    return [ia for ia, a in enumerate(p.atoms) if not a.detailed]


def nr_inputs(p, seed=3):
    """Deterministic stand-ins for what the Python layer hands to nr_post_update: ionisation stages
    (last level of each atom is the next ion), an electron density consistent with the ions plus a
    background, and a smooth dC/dne."""
    rng = np.random.default_rng(seed)
    Ns = p.Nspace
    stages = []
    ne = np.zeros(Ns)
    for ia in active(p):
        a = p.atoms[ia]
        s = np.zeros(a.Nlevel)
        s[-1] = 1.0
        stages.append(s)
        ne += a.n[-1]
    bg = 0.1 * ne * (1.0 + 0.1 * rng.random(Ns))
    ne = ne + bg
    ne *= 1.0 + 0.02 * rng.standard_normal(Ns)       # off equilibrium so that the step is non-trivial
    dC = []
    for ia in active(p):
        a = p.atoms[ia]
        dC.append(a.C / ne[None, None, :] * (0.5 + rng.random(a.C.shape)))
    return stages, bg, ne, dC


def run_time_dep(ctxFactory, dt):
    p, ctx = problem_after_one_iteration(ctxFactory)
    prev = [p.atoms[ia].n.copy() for ia in active(p)]
    for q, ia in enumerate(active(p)):
        assert ctx.time_dep_update(ia, prev[q], dt) == 0
    return p


def run_nr(ctxFactory, timeDep, fdC):
    p, ctx = problem_after_one_iteration(ctxFactory)
    stages, bg, ne, dC = nr_inputs(p)
    nPrev = [p.atoms[ia].n * (1.0 + 0.01 * np.cos(np.arange(p.Nspace)))[None, :] for ia in active(p)] if timeDep else None
    st = ctx.nr_post_update(active(p), stages, bg, ne, dC=dC if fdC else None, nPrev=nPrev, dt=0.05, crsw=1.0)
    assert st == 0
    return p, ne


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
@pytest.mark.parametrize('dt', [1e-3, 0.1, 10.0])
def test_oracle_time_dep_matches_reference(dt):
    pr = run_time_dep(bindings.RefContext, dt)
    po = run_time_dep(bindings.OracleContext, dt)
    for a, b in zip(pr.atoms, po.atoms):
        np.testing.assert_array_equal(a.n, b.n)


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
@pytest.mark.parametrize('timeDep', [False, True])
@pytest.mark.parametrize('fdC', [False, True])
def test_oracle_nr_post_update_matches_reference(timeDep, fdC):
    pr, ner = run_nr(bindings.RefContext, timeDep, fdC)
    po, neo = run_nr(bindings.OracleContext, timeDep, fdC)
    np.testing.assert_array_equal(ner, neo)
    for a, b in zip(pr.atoms, po.atoms):
        np.testing.assert_array_equal(a.n, b.n)
    # the step did something and conserved each atom's total (last row of each block)
    p0, _ = problem_after_one_iteration(bindings.OracleContext)
    assert max(rel_err(a.n, b.n) for a, b in zip(po.atoms, p0.atoms)) > 1e-6
    for a in po.atoms:
        assert np.allclose(a.n.sum(axis=0), a.nTotal, rtol=1e-10)


RANGE = (17, 49)   # spaceStart, spaceEnd: not aligned with any workgroup size


def run_ranged(ctxFactory, what, rng):
    """One population update restricted to the depth points [rng[0], rng[1]) (None: everywhere); returns the problem,
    the populations before it and ne."""
    p, ctx = problem_after_one_iteration(ctxFactory)
    before = [a.n.copy() for a in p.atoms]
    stages, bg, ne, dC = nr_inputs(p)
    ne0 = ne.copy()
    if rng is not None:
        ctx.set_depth_range(*rng)
    if what == 'stat_eq':
        assert ctx.stat_equil() == 0
    elif what == 'time_dep':
        for ia in active(p):
            assert ctx.time_dep_update(ia, before[ia], 0.1) == 0
    else:
        assert ctx.nr_post_update(active(p), stages, bg, ne, dC=dC, crsw=1.0) == 0
    if rng is not None:
        ctx.set_depth_range(-1, -1)
    return p, before, ne0, ne


def composed(full, before, rng):
    out = before.copy()
    out[..., rng[0]:rng[1]] = full[..., rng[0]:rng[1]]
    return out


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
@pytest.mark.parametrize('what', ['stat_eq', 'time_dep', 'nr_post'])
def test_reference_depth_range_is_the_full_update_on_that_range(what):
    """spaceStart / spaceEnd of the core (Source/UpdatePopulations.cpp:22, :135, :316): the depth points are
    independent, so a ranged call equals the full call inside the range and leaves the rest alone -- the property the
    GPU test below checks the HIP path against (the oracle has no ranged form of its own)."""
    pf, before, ne0, neF = run_ranged(bindings.RefContext, what, None)
    pr, _, _, neR = run_ranged(bindings.RefContext, what, RANGE)
    for a, b, b0 in zip(pr.atoms, pf.atoms, before):
        np.testing.assert_array_equal(a.n, composed(b.n, b0, RANGE))
        assert not np.array_equal(a.n, b.n) or a.detailed
    if what == 'nr_post':
        np.testing.assert_array_equal(neR, composed(neF, ne0, RANGE))


def test_time_dep_large_dt_tends_to_statistical_equilibrium():
    """(1 - dt Gamma) n = nOld with dt -> infinity has the kernel of Gamma as its solution: the
    stat_equil populations (up to normalisation)."""
    p, ctx = problem_after_one_iteration(bindings.OracleContext)
    q, ctx2 = problem_after_one_iteration(bindings.OracleContext)
    ia = active(p)[0]
    assert ctx.time_dep_update(ia, p.atoms[ia].n.copy(), 1e12) == 0
    assert ctx2.stat_equil(ia) == 0
    a, b = p.atoms[ia].n, q.atoms[ia].n
    assert rel_err(a / a.sum(axis=0), b / b.sum(axis=0)) < 1e-3


class _HipAdapter:
    """lightweaver_amd.Context behind the call shapes of the oracle bindings used above."""
    def __init__(self, p):
        from lightweaver_amd.context import Context
        self.p = p
        self.ctx = Context(p)

    def formal_sol_gamma_matrices(self):
        # identical inputs for the solves under test: Gamma comes from the oracle's formal solution
        # (the HIP formal solution is compared with it elsewhere; here its 1e-10 differences would be
        # amplified by the conditioning of (1 - dt Gamma) at large dt)
        with bindings.OracleContext(self.p) as oc:
            return oc.formal_sol_gamma_matrices()

    def set_depth_range(self, s, e):
        self.ctx.set_depth_range(s, e)

    def stat_equil(self):
        self.ctx.stat_equil()
        return 0

    def time_dep_update(self, atom, nOld, dt):
        prev = {ia: None for ia in active(self.p)}
        lst = [nOld if ia == atom else self.p.atoms[ia].n for ia in active(self.p)]
        self.ctx.time_dep_update(dt, lst, atom=atom)
        return 0

    def nr_post_update(self, atoms, stages, bg, ne, dC=None, nPrev=None, dt=0.0, crsw=1.0):
        self.ctx.crsw = crsw
        self.ctx.nr_post_update(stages, bg, ne, dC=dC, nPrev=nPrev, dt=dt, atoms=atoms)
        return 0


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [1e-3, 0.1, 10.0])
def test_hip_time_dep_matches_oracle(gpu, dt):
    po = run_time_dep(bindings.OracleContext, dt)
    ph = run_time_dep(_HipAdapter, dt)
    for a, b in zip(ph.atoms, po.atoms):
        assert rel_err(a.n, b.n) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize('timeDep', [False, True])
@pytest.mark.parametrize('fdC', [False, True])
def test_hip_nr_post_update_matches_oracle(gpu, timeDep, fdC):
    po, neo = run_nr(bindings.OracleContext, timeDep, fdC)
    ph, neh = run_nr(_HipAdapter, timeDep, fdC)
    assert rel_err(neh, neo) <= TOL
    for a, b in zip(ph.atoms, po.atoms):
        assert rel_err(a.n, b.n) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize('what', ['stat_eq', 'time_dep', 'nr_post'])
@pytest.mark.parametrize('rng', [RANGE, (0, 1), (81, 82), (5, 5)])
def test_hip_depth_range_matches_oracle(gpu, what, rng):
    """lwhip_set_depth_range: the three updates on [spaceStart, spaceEnd) only -- equal to the oracle's full update
    inside the range (1e-9), bit-identical to the input outside it."""
    po, before, ne0, neO = run_ranged(bindings.OracleContext, what, None)
    ph, _, _, neH = run_ranged(_HipAdapter, what, rng)
    sl = slice(rng[0], rng[1])
    for a, b, b0 in zip(ph.atoms, po.atoms, before):
        if a.detailed:
            continue
        outside = np.ones(a.n.shape[1], bool)
        outside[sl] = False
        np.testing.assert_array_equal(a.n[:, outside], b0[:, outside])
        if rng[1] > rng[0]:
            assert rel_err(a.n[:, sl], b.n[:, sl]) <= TOL
    if what == 'nr_post':
        outside = np.ones(neH.size, bool)
        outside[sl] = False
        np.testing.assert_array_equal(neH[outside], ne0[outside])
        if rng[1] > rng[0]:
            assert rel_err(neH[sl], neO[sl]) <= TOL


@pytest.mark.gpu
def test_hip_depth_range_rejects_bad_ranges(gpu):
    from lightweaver_amd.context import Context, LwHipError
    base, d = load_fixture('falc_h_ca_small')
    with Context(variant_problem(base, d, 'iter1')) as ctx:
        for s, e in ((-1, 10), (10, 5), (0, 10 ** 6)):
            with pytest.raises(LwHipError):
                ctx.set_depth_range(s, e)
        ctx.set_depth_range(-1, -1)


@pytest.mark.gpu
def test_hip_nr_rejects_detailed_atoms(gpu):
    from lightweaver_amd.context import Context, LwHipError
    base, d = load_fixture('falc_h_ca_small')
    p = variant_problem(base, d, 'prd_detailed')
    det = [ia for ia, a in enumerate(p.atoms) if a.detailed]
    if not det:
        pytest.skip('no detailed atom in this variant')
    with Context(p) as ctx:
        with pytest.raises(LwHipError):
            ctx.nr_post_update([np.zeros(p.atoms[det[0]].Nlevel)], np.zeros(p.Nspace), np.ones(p.Nspace), atoms=det)

"""The Lightweaver plugin (lightweaver_amd/LwHipIterScheme.so): loaded by the REAL core through its
own FsIterationFnsManager::load_fns_from_path (Source/FormalInterface.cpp:62-81), selected as the
iteration scheme of a reference `Context`, and driven through the reference entry points
formal_sol_gamma_matrices / formal_sol / stat_eq -- i.e. the drop-in path a Lightweaver user gets.
Needs oracle/_ref (built where the reference sources are; the .so files travel to the GPU box)."""
import os

import numpy as np
import pytest

from helpers import TOL_CONVERGED, TOL_ONE_CALL, collect, gamma_err_scaled, load_fixture, rel_err, variant_problem
from lightweaver_amd import _abi as abi
from lightweaver_amd import build
from oracle import bindings
from oracle.bindings import RefContext

PLUGIN = build.PLUGIN
needs_ref = pytest.mark.skipif(not (bindings.ref_available() and os.path.exists(PLUGIN)),
                               reason='reference build / plugin not present')


@needs_ref
def test_provider_is_loadable_and_named(hip_lib):
    base, _ = load_fixture('falc_h_vel')
    rc = RefContext(base.copy(), scheme=PLUGIN)
    assert rc.scheme_name == 'mali_full_precond_hip_gfx950'


@needs_ref
def test_plugin_fails_loudly_without_gpu(hip_lib, capfd):
    if hip_lib.lwhip_device_count() > 0:
        pytest.skip('a GPU is present')
    base, _ = load_fixture('falc_h_vel')
    p = base.copy()
    rc = RefContext(p, scheme=PLUGIN)
    J0 = p.J.copy()
    st = rc.lib.lwref_formal_sol_gamma_matrices(rc.h, 0, None)
    assert st != 0                      # the C++ exception surfaced through the driver
    assert 'no HIP device' in capfd.readouterr().err
    assert np.array_equal(p.J, J0)      # and nothing was computed on the CPU instead


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize('name', ['falc_h_ca_small', 'falc_h_vel'])
def test_dropin_matches_scalar_scheme(gpu, name):
    base, d = load_fixture(name)
    p, q = variant_problem(base, d, 'iter1'), variant_problem(base, d, 'iter1')
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    for it in range(3):
        p.gamma_prefill()
        q.gamma_prefill()
        r1 = hipc.formal_sol_gamma_matrices()
        r2 = ref.formal_sol_gamma_matrices()
        assert abs(r1[0] / r2[0] - 1.0) <= TOL_ONE_CALL
        a, b = collect(p, ('J', 'I', 'Gamma', 'R')), collect(q, ('J', 'I', 'Gamma', 'R'))
        worst = {k: rel_err(a[k], b[k]) for k in a}
        assert max(worst.values()) <= TOL_ONE_CALL * (it + 1), (it, worst)
        assert rel_err(p.depthI, q.depthI) <= TOL_ONE_CALL * (it + 1)
        assert hipc.stat_equil() == ref.stat_equil() == 0
        for x, y in zip(p.atoms, q.atoms):
            assert rel_err(x.n, y.n) <= 1e-7


@needs_ref
@pytest.mark.gpu
def test_dropin_is_bit_reproducible_by_default(gpu, monkeypatch):
    """What a Lightweaver user gets (round 5): the plugin creates its contexts in the fixed-order mode, so four iterations with
    stat_equil from the same inputs end in the SAME bits -- J, Gamma, rates, populations -- run after run, as the reference's
    own schemes do (SimdFullIterationTemplates.hpp:620-628, ThreadStorage.cpp:343-396); LWHIP_DETERMINISTIC=0 brings the
    atomic sums back (equal to rounding)."""
    base, d = load_fixture('falc_h_ca_small')
    runs = []
    for rep, det in enumerate((None, None, '0')):
        if det is None:
            monkeypatch.delenv('LWHIP_DETERMINISTIC', raising=False)
        else:
            monkeypatch.setenv('LWHIP_DETERMINISTIC', det)
        p = variant_problem(base, d, 'iter1')
        hipc = RefContext(p, scheme=PLUGIN)
        for it in range(4):
            p.gamma_prefill()
            hipc.formal_sol_gamma_matrices()
            assert hipc.stat_equil() == 0
        runs.append(p)
    a, b, c = runs
    assert np.array_equal(a.J, b.J) and np.array_equal(a.I, b.I)
    for x, y, z in zip(a.atoms, b.atoms, c.atoms):
        assert np.array_equal(x.Gamma, y.Gamma) and np.array_equal(x.n, y.n)
        for tx, ty in zip(x.trans, y.trans):
            assert np.array_equal(tx.Rij, ty.Rij) and np.array_equal(tx.Rji, ty.Rji)
        assert rel_err(x.n, z.n) <= 1e-9


@needs_ref
@pytest.mark.gpu
def test_dropin_resident_J(gpu, monkeypatch):
    """LWHIP_RESIDENT_J=1 (opt-in): J stays in HBM between the plugin's calls -- spect.J is neither
    uploaded nor written back while the variable is set, and the first call made without it brings the device's J down before
    anything goes up.  Four iterations with the variable set for the first three must end where four plain drop-in
    iterations end (to the run-to-run noise of the atomically summed Gamma), and spect.J must stay untouched in between."""
    base, d = load_fixture('falc_h_ca_small')
    p, q = variant_problem(base, d, 'iter1'), variant_problem(base, d, 'iter1')
    plain, res = RefContext(q, scheme=PLUGIN), RefContext(p, scheme=PLUGIN)
    for it in range(4):
        q.gamma_prefill()
        plain.formal_sol_gamma_matrices()
        plain.stat_equil()
    J0 = p.J.copy()
    monkeypatch.setenv('LWHIP_RESIDENT_J', '1')
    for it in range(3):
        p.gamma_prefill()
        r = res.formal_sol_gamma_matrices()
        res.stat_equil()
        assert np.array_equal(p.J, J0)          # not written back ...
        p.J[...] = -1.0                         # ... and not read either: garbage here must not matter
        J0 = p.J.copy()
    monkeypatch.delenv('LWHIP_RESIDENT_J')
    p.gamma_prefill()
    res.formal_sol_gamma_matrices()             # the device's J comes down before this call uploads anything
    res.stat_equil()
    # (Gamma is summed by atomics, so the populations -- and through them the later iterations' J -- agree to the last bits only)
    assert rel_err(p.J, q.J) <= 1e-9 and rel_err(p.I, q.I) <= 1e-9  # (4e-11 / 1e-10 seen: the populations' noise, below)
    for a, b in zip(p.atoms, q.atoms):
        assert rel_err(a.n, b.n) <= 1e-9        # (the solve amplifies Gamma's last-bit noise: 1.4e-10 seen)


@needs_ref
@pytest.mark.gpu
def test_dropin_formal_sol_and_solvers(gpu):
    base, d = load_fixture('falc_h_ca_small')
    for solver in (abi.FS_LINEAR_1D, abi.FS_BESSER_1D, abi.FS_BEZIER3_1D):
        p, q = base.copy(), base.copy()
        p.formalSolver = q.formalSolver = solver
        hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
        for upOnly in (True, False):
            hipc.formal_sol(upOnly=upOnly)
            ref.formal_sol(upOnly=upOnly)
            assert rel_err(p.I, q.I) <= TOL_ONE_CALL
        p.gamma_prefill()
        q.gamma_prefill()
        hipc.formal_sol_gamma_matrices()
        ref.formal_sol_gamma_matrices()
        assert rel_err(p.J, q.J) <= TOL_ONE_CALL


@needs_ref
@pytest.mark.gpu
def test_dropin_iterates_like_the_reference(gpu):
    """iterate_ctx_se-style loop (lightweaver/iterate_ctx.py:157-176) with the host arrays
    authoritative: 25 iterations, populations updated from iteration 3 on."""
    base, d = load_fixture('falc_h_ca_small')
    p, q = base.copy(), base.copy()
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    for it in range(25):
        for prob, c in ((p, hipc), (q, ref)):
            prob.gamma_prefill()
            c.formal_sol_gamma_matrices()
            if it >= 3:
                assert c.stat_equil() == 0
    assert rel_err(p.J, q.J) <= TOL_CONVERGED
    for x, y in zip(p.atoms, q.atoms):
        assert rel_err(x.n, y.n) <= TOL_CONVERGED


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize('sampled', [False, True])
def test_dropin_sees_host_side_changes(gpu, monkeypatch, sampled):
    """The plugin re-uploads what the host changed between calls (populations, J, profiles,
    background), as the reference shares memory with the Python layer -- with the exact fingerprints of the default and
    with the sampled ones of LWHIP_SAMPLED_FINGERPRINTS=1 (updates that rewrite a whole array are seen by both)."""
    if sampled:
        monkeypatch.setenv('LWHIP_SAMPLED_FINGERPRINTS', '1')
    base, d = load_fixture('falc_h_vel')
    p, q = base.copy(), base.copy()
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    rng = np.random.default_rng(5)
    for it in range(3):
        scale = 1.0 + 0.1 * rng.random(p.bgChi.shape)
        for prob in (p, q):
            prob.bgChi *= scale
            prob.atoms[0].n *= 1.01
            for t in prob.atoms[0].trans:
                if t.type == abi.LINE:
                    t.phi *= 1.0 + 1e-3 * it
            prob.gamma_prefill()
        hipc.formal_sol_gamma_matrices()
        ref.formal_sol_gamma_matrices()
        assert rel_err(p.J, q.J) <= TOL_ONE_CALL
        assert rel_err(p.atoms[0].Gamma, q.atoms[0].Gamma) <= TOL_ONE_CALL


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize('mapped', ['1', '0'])
def test_dropin_J_is_an_output_of_the_sweep_and_host_edits_of_J_are_honoured(gpu, monkeypatch, mapped):
    """Round 6: spect.J is page-locked and written by the sweep itself, and its upload is skipped while nobody has rewritten
    it (exact fingerprint of the array against the device's own).  The reference's contract holds: J is in spect.J when a
    call returns; a caller that rewrites spect.J between two calls -- all of it, or ONE element by one unit in the last place --
    gets an iteration that starts from what it wrote, exactly like the scalar scheme; so does a PRD call; and a slow-group
    change discovered by the speculative run (the background edited in place) re-runs the iteration from the RIGHT J although
    the first run has overwritten the host's copy (LWHIP_MAP_HOST_J=0: the copying path of rounds 1-5, same results)."""
    monkeypatch.setenv('LWHIP_MAP_HOST_J', mapped)
    monkeypatch.setenv('LWHIP_DETERMINISTIC', '1')
    base, d = load_fixture('falc_h_ca_small')
    p, q = base.copy(), base.copy()
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)

    def both(f):
        for pr, c in ((p, hipc), (q, ref)):
            f(pr, c)
        assert rel_err(p.J, q.J) <= TOL_ONE_CALL and rel_err(p.I, q.I) <= TOL_ONE_CALL
        for a, b in zip(p.atoms, q.atoms):
            assert gamma_err_scaled(a.Gamma, b) <= TOL_ONE_CALL

    def call(pr, c):
        pr.gamma_prefill()
        c.formal_sol_gamma_matrices()
    both(call)
    both(call)                                              # nothing touched in between: the upload of J is skipped
    def edit_all(pr, c):
        pr.J *= 1.0 + 1e-3 * np.cos(np.arange(pr.J.size)).reshape(pr.J.shape)
        call(pr, c)
    both(edit_all)
    def edit_one(pr, c):
        pr.J[pr.Nlambda // 2 + 1, pr.Nspace // 2] *= 3.0   # (a big change of one element: its effect is far above the tolerance)
        call(pr, c)
    both(edit_one)
    Jp, Jq = p.J.copy(), q.J.copy()
    def edit_bg_and_J(pr, c):
        pr.bgChi[pr.Nlambda // 3, 5] *= 1.5                 # seen only by the fingerprints that run beside the speculative kernels
        call(pr, c)
    both(edit_bg_and_J)
    assert not np.array_equal(Jp, p.J)
    both(call)


@needs_ref
@pytest.mark.gpu
def test_dropin_detects_single_element_edits(gpu):
    """Exact change detection: an in-place edit of ONE depth point of a slowly changing input (what update_deps()
    does after a local atmosphere change) must reach the device, also when the call that first sees it is stat_eq --
    which records the new fingerprints -- and the formal solution only comes afterwards."""
    base, d = load_fixture('falc_h_ca_small')
    p, q = base.copy(), base.copy()
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    for pr, c in ((p, hipc), (q, ref)):
        pr.gamma_prefill()
        c.formal_sol_gamma_matrices()
    Ns = p.Nspace
    # elements a strided sample of the arrays would not look at
    edits = [('bgChi', (p.Nlambda // 2 + 1, Ns // 2 + 1), 1.7), ('bgEta', (3, 5), 0.4), ('bgSca', (7, Ns - 2), 2.5)]
    line = next(t for t in p.atoms[0].trans if t.type == abi.LINE)
    kr = p.atoms[0].trans.index(line)
    for pr in (p, q):
        for name, idx, f in edits:
            getattr(pr, name)[idx] *= f
        t = pr.atoms[0].trans[kr]
        t.phi[t.phi.shape[0] // 2, 1, 1, Ns // 3] *= 1.3
        pr.atoms[0].nStar[1, Ns // 2 + 3] *= 1.2
        pr.atoms[0].C[0, 1, Ns // 2 + 3] *= 1.5
        pr.temperature[Ns // 2 + 1] *= 1.01
    assert hipc.stat_equil() == ref.stat_equil() == 0      # sees the edits first
    for it in range(2):
        for pr, c in ((p, hipc), (q, ref)):
            pr.gamma_prefill()
            c.formal_sol_gamma_matrices()
        a, b = collect(p, ('J', 'I', 'Gamma', 'R')), collect(q, ('J', 'I', 'Gamma', 'R'))
        worst = {k: rel_err(a[k], b[k]) for k in a}
        assert max(worst.values()) <= TOL_ONE_CALL * (it + 2), (it, worst)


@needs_ref
@pytest.mark.gpu
def test_dropin_detects_collision_edit_before_prd_call(gpu):
    """One odd-depth element of C changes before a redistribute_prd call (the call that first sees, and records, the
    new fingerprint); the next formal_sol_gamma_matrices must use the new C through its device-side pre-fill."""
    from test_prd import golden_prd_problem
    prob, _ = golden_prd_problem()
    p, q = prob.copy(), prob.copy()
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    for pr, c in ((p, hipc), (q, ref)):
        pr.gamma_prefill()
        c.formal_sol_gamma_matrices()
    k = p.Nspace // 2 | 1
    for pr in (p, q):
        pr.atoms[-1].C[1, 0, k] *= 3.0
        pr.atoms[-1].C[0, 0, k] = -pr.atoms[-1].C[1:, 0, k].sum()
    u1, u2 = hipc.redistribute_prd(2, 1e-2), ref.redistribute_prd(2, 1e-2)
    assert np.allclose(u1['dRho'], u2['dRho'], rtol=1e-5)
    for pr, c in ((p, hipc), (q, ref)):
        pr.gamma_prefill()
        c.formal_sol_gamma_matrices()
        assert c.stat_equil() == 0
    a, b = collect(p, ('J', 'Gamma', 'R')), collect(q, ('J', 'Gamma', 'R'))
    worst = {kk: rel_err(a[kk], b[kk]) for kk in a}
    assert max(worst.values()) <= 1e-8, worst
    for x, y in zip(p.atoms, q.atoms):
        assert rel_err(x.n, y.n) <= 1e-7


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize('include', [True, False])
def test_dropin_prd_include_detailed_atoms(gpu, include):
    """redistribute_prd of the plugin with ExtraParams include_detailed_atoms (what LwContext.prd_redistribute passes,
    True by default): the device context is rebuilt with the detailed atoms' PRD lines in its list.  (One value per
    Context, as the Python layer does: the core caches the PRD wavelength list of its first call in spect.hPrdIdxs --
    PrdTemplates.hpp:222-240 assigns through a reference -- so toggling the flag on a live Context is not meaningful.)"""
    from test_prd import assert_same_state, detailed_prd_problem
    prob = detailed_prd_problem()
    p, q = prob.copy(), prob.copy()
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    for it in range(2):
        for pr, c in ((p, hipc), (q, ref)):
            pr.gamma_prefill()
            c.formal_sol_gamma_matrices()
            assert c.stat_equil() == 0
        u1 = hipc.redistribute_prd(3, 1e-3, include_detailed_atoms=include)
        u2 = ref.redistribute_prd(3, 1e-3, include_detailed_atoms=include)
        assert u1['NprdSubIter'] == u2['NprdSubIter'] and u1['Nprd'] == u2['Nprd'] == (3 if include else 1)
        assert np.allclose(u1['dRho'], u2['dRho'], rtol=1e-5)
        assert np.allclose(u1['dJPrdMax'], u2['dJPrdMax'], rtol=1e-7)
    assert_same_state(p, q, 1e-8)


@needs_ref
@pytest.mark.gpu
def test_dropin_stat_eq_as_first_call(gpu):
    """stat_eq may be the first member the core calls on a fresh Context (e.g. after loading a pickled state with
    its Gamma): the plugin finds the owning Context from the atoms registered by alloc_global_scratch."""
    base, d = load_fixture('falc_h_ca_small')
    src = variant_problem(base, d, 'iter1')
    warm = RefContext(src)
    src.gamma_prefill()
    warm.formal_sol_gamma_matrices()
    p, q = src.copy(), src.copy()
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    assert hipc.stat_equil() == 0
    assert ref.stat_equil() == 0
    for x, y in zip(p.atoms, q.atoms):
        assert rel_err(x.n, y.n) <= 1e-9


@needs_ref
@pytest.mark.gpu
def test_dropin_prd_redistribute(gpu):
    """FsIterationFns::redistribute_prd of the plugin, driven by the real core's
    redistribute_prd_lines dispatch (Source/Prd.cpp), next to the core's own scalar scheme."""
    from test_prd import assert_same_state, golden_prd_problem
    prob, _ = golden_prd_problem()
    p, q = prob.copy(), prob.copy()
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    for it in range(3):
        for pr, c in ((p, hipc), (q, ref)):
            pr.gamma_prefill()
            c.formal_sol_gamma_matrices()
            if it >= 1:
                assert c.stat_equil() == 0
        u1, u2 = hipc.redistribute_prd(3, 1e-2), ref.redistribute_prd(3, 1e-2)
        assert u1['NprdSubIter'] == u2['NprdSubIter'] and u1['Nprd'] == u2['Nprd'] == 2
        assert np.allclose(u1['dRho'], u2['dRho'], rtol=1e-5)
        assert np.allclose(u1['dJPrdMax'], u2['dJPrdMax'], rtol=1e-7)
    assert_same_state(p, q, 1e-8)


@needs_ref
@pytest.mark.gpu
def test_dropin_hybrid_prd(gpu):
    """Hybrid PRD through the drop-in boundary: configure_hprd_coeffs runs in the real core (it is not part of the
    iteration scheme), the plugin flattens the tables it finds in the Context (spect.JCoeffs / JRest, the lines'
    hPrdCoeffs) into lwhip_options.hprd, and the device then interpolates rho per ray, scatters into JRest and runs the
    scattering integral on it -- next to the core's own scalar scheme on the same Context set-up."""
    from test_hprd import assert_same, hprd_problem
    prob = hprd_problem()
    p, q = prob.copy(), prob.copy()
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    th, tr = hipc.configure_hprd(), ref.configure_hprd()
    for it in range(3):
        for pr, c in ((p, hipc), (q, ref)):
            pr.gamma_prefill()
            c.formal_sol_gamma_matrices()
            if it >= 1:
                assert c.stat_equil() == 0
        assert rel_err(th.JRest, tr.JRest) <= 1e-8
        u1, u2 = hipc.redistribute_prd(3, 1e-2), ref.redistribute_prd(3, 1e-2)
        assert u1['NprdSubIter'] == u2['NprdSubIter'] and u1['Nprd'] == u2['Nprd'] == 2
        assert np.allclose(u1['dRho'], u2['dRho'], rtol=1e-5)
        assert np.allclose(u1['dJPrdMax'], u2['dJPrdMax'], rtol=1e-7)
    assert_same(p, q, 1e-8)
    th.close()
    tr.close()
    hipc.close()
    ref.close()


@needs_ref
@pytest.mark.gpu
def test_dropin_time_dep_and_nr_updates(gpu):
    """FsIterationFns::time_dep_update / nr_post_update of the plugin through the real core's
    trampolines (Source/UpdatePopulations.cpp:153-157, 473-483), next to the core's own."""
    from test_pops import active, nr_inputs
    base, d = load_fixture('falc_h_ca_small')
    p, q = variant_problem(base, d, 'iter1'), variant_problem(base, d, 'iter1')
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    for pr, c in ((p, hipc), (q, ref)):
        pr.gamma_prefill()
        c.formal_sol_gamma_matrices()
    # identical operators for the two solves (the formal solutions agree to 1e-10, the solves
    # amplify that; the formal solution itself is compared in the tests above)
    for x, y in zip(p.atoms, q.atoms):
        x.Gamma[...] = y.Gamma
    prev = [a.n.copy() for a in q.atoms]
    for ia in active(p):
        assert hipc.time_dep_update(ia, prev[ia], 0.1) == 0
        assert ref.time_dep_update(ia, prev[ia], 0.1) == 0
        assert rel_err(p.atoms[ia].n, q.atoms[ia].n) <= 1e-9
    stages, bg, ne, dC = nr_inputs(q)
    ne2 = ne.copy()
    assert hipc.nr_post_update(active(p), stages, bg, ne, dC=dC, crsw=1.0) == 0
    assert ref.nr_post_update(active(q), stages, bg, ne2, dC=dC, crsw=1.0) == 0
    assert rel_err(ne, ne2) <= 1e-9
    for x, y in zip(p.atoms, q.atoms):
        assert rel_err(x.n, y.n) <= 1e-9


@needs_ref
@pytest.mark.gpu
def test_dropin_partial_depth_ranges(gpu):
    """spaceStart / spaceEnd of FsIterationFns::stat_eq / time_dep_update / nr_post_update (the per-chunk form of
    Source/UpdatePopulations.cpp:22, :135, :316): the plugin solves the same range as the core and leaves the same
    points alone."""
    from test_pops import active, nr_inputs
    base, d = load_fixture('falc_h_ca_small')
    p, q = variant_problem(base, d, 'iter1'), variant_problem(base, d, 'iter1')
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    for pr, c in ((p, hipc), (q, ref)):
        pr.gamma_prefill()
        c.formal_sol_gamma_matrices()
    for x, y in zip(p.atoms, q.atoms):
        x.Gamma[...] = y.Gamma
    start = [a.n.copy() for a in q.atoms]
    for c in (hipc, ref):
        c.set_depth_range(11, 60)
    assert hipc.stat_equil() == ref.stat_equil() == 0
    for x, y, n0 in zip(p.atoms, q.atoms, start):
        assert rel_err(x.n, y.n) <= 1e-9
        np.testing.assert_array_equal(x.n[:, :11], n0[:, :11])
        np.testing.assert_array_equal(x.n[:, 60:], n0[:, 60:])
        assert x.detailed or not np.array_equal(x.n[:, 11:60], n0[:, 11:60])
    for c in (hipc, ref):
        c.set_depth_range(30, 82)
    for ia in active(p):
        assert hipc.time_dep_update(ia, start[ia], 0.1) == 0
        assert ref.time_dep_update(ia, start[ia], 0.1) == 0
        assert rel_err(p.atoms[ia].n, q.atoms[ia].n) <= 1e-9
    for c in (hipc, ref):
        c.set_depth_range(0, 40)
    stages, bg, ne, dC = nr_inputs(q)
    ne0, ne2 = ne.copy(), ne.copy()
    assert hipc.nr_post_update(active(p), stages, bg, ne, dC=dC, crsw=1.0) == 0
    assert ref.nr_post_update(active(q), stages, bg, ne2, dC=dC, crsw=1.0) == 0
    assert rel_err(ne, ne2) <= 1e-9
    np.testing.assert_array_equal(ne[40:], ne0[40:])
    for x, y in zip(p.atoms, q.atoms):
        assert rel_err(x.n, y.n) <= 1e-9
    # back to the whole atmosphere: the next call is a full one again
    for c in (hipc, ref):
        c.set_depth_range(-1, -1)
    assert hipc.stat_equil() == ref.stat_equil() == 0
    for x, y in zip(p.atoms, q.atoms):
        assert rel_err(x.n, y.n) <= 1e-9


@needs_ref
@pytest.mark.gpu
def test_dropin_2d_matches_scalar_scheme(gpu):
    """Ndim = 2 under the real core: the plugin flattens Atmosphere::intersections (built by the core's own
    build_intersection_list) into lwhip_grid2d and runs the device 2D pipeline; the built-in scalar scheme with
    piecewise_besser_2d + interp_linear_2d is the reference.  Second iteration re-uses the flattened table."""
    base, _ = load_fixture('falc2d_small')
    p, q = base.copy(), base.copy()
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    for it in range(2):
        p.gamma_prefill()
        q.gamma_prefill()
        r1 = hipc.formal_sol_gamma_matrices()
        r2 = ref.formal_sol_gamma_matrices()
        assert abs(r1[0] / r2[0] - 1.0) <= TOL_ONE_CALL
        assert rel_err(p.J, q.J) <= TOL_ONE_CALL * (it + 1) and rel_err(p.I, q.I) <= TOL_ONE_CALL * (it + 1)
        for x, y in zip(p.atoms, q.atoms):
            assert rel_err(x.Gamma, y.Gamma) <= 1e-8       # association of the sums, see tests/test_fs2d.py
            for ta, tb in zip(x.trans, y.trans):
                assert rel_err(ta.Rij, tb.Rij) <= TOL_ONE_CALL * (it + 1)
                assert rel_err(ta.Rji, tb.Rji) <= TOL_ONE_CALL * (it + 1)
        assert hipc.stat_equil() == ref.stat_equil() == 0
        for x, y in zip(p.atoms, q.atoms):
            assert rel_err(x.n, y.n) <= 1e-7


@needs_ref
@pytest.mark.gpu
def test_dropin_2d_prd_and_formal_sol(gpu):
    """The remaining plugin members on an Ndim = 2 Context of the real core: redistribute_prd (Ca II H & K as PRD
    lines) and simple_fs, next to the core's scalar scheme."""
    from test_fs2d import prd_problem_2d
    from test_prd import assert_same_state
    prob = prd_problem_2d()
    p, q = prob.copy(), prob.copy()
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    for it in range(2):
        for pr, c in ((p, hipc), (q, ref)):
            pr.gamma_prefill()
            c.formal_sol_gamma_matrices()
            assert c.stat_equil() == 0
        u1, u2 = hipc.redistribute_prd(3, 1e-2), ref.redistribute_prd(3, 1e-2)
        assert u1['NprdSubIter'] == u2['NprdSubIter'] and u1['Nprd'] == u2['Nprd'] == 2
        assert np.allclose(u1['dRho'], u2['dRho'], rtol=1e-5)
        assert np.allclose(u1['dJPrdMax'], u2['dJPrdMax'], rtol=1e-7)
    assert_same_state(p, q, 1e-8)
    for upOnly in (True, False):
        hipc.formal_sol(upOnly=upOnly)
        ref.formal_sol(upOnly=upOnly)
        assert rel_err(p.I, q.I) <= TOL_ONE_CALL * 10


@needs_ref
@pytest.mark.gpu
def test_dropin_2d_fixed_x_boundaries(gpu):
    """Ndim = 2 with CALLABLE x boundaries under the real core: the plugin flattens the core's own intersection table
    (no long characteristics) and the boundaries' bcData [Nlambda, Nmu, Nz]; the built-in scalar scheme is the
    reference."""
    from test_fs2d import fixed_x_problem
    base = fixed_x_problem()
    p, q = base.copy(), base.copy()
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    for it in range(2):
        p.gamma_prefill()
        q.gamma_prefill()
        r1 = hipc.formal_sol_gamma_matrices()
        r2 = ref.formal_sol_gamma_matrices()
        assert abs(r1[0] / r2[0] - 1.0) <= TOL_ONE_CALL
        assert rel_err(p.J, q.J) <= TOL_ONE_CALL * (it + 1) and rel_err(p.I, q.I) <= TOL_ONE_CALL * (it + 1)
        for x, y in zip(p.atoms, q.atoms):
            assert rel_err(x.Gamma, y.Gamma) <= 1e-8
        assert hipc.stat_equil() == ref.stat_equil() == 0
        for x, y in zip(p.atoms, q.atoms):
            assert rel_err(x.n, y.n) <= 1e-7
    # the boundary data is borrowed: an in-place change reaches the device on the next call
    for pr in (p, q):
        pr.grid2d.xLowerBc.bcData *= 1.5
        pr.gamma_prefill()
    hipc.formal_sol_gamma_matrices()
    ref.formal_sol_gamma_matrices()
    assert rel_err(p.J, q.J) <= 3 * TOL_ONE_CALL


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize('two_d', [False, True])
def test_dropin_zplane_decomposition(gpu, two_d):
    """ExtraParams ZPlaneDecomposition / ZPlaneDown / ZPlaneUp through the plugin: the caller's arrays are filled with
    the z-plane intensities of the device sweep (1D march, 2D pipeline), next to the built-in scalar scheme."""
    if two_d:
        base, _ = load_fixture('falc2d_small')
        shape = (base.Nlambda, base.Nrays, base.grid2d.Nx)
    else:
        base, _ = load_fixture('falc_h_ca_small')
        shape = (base.Nlambda, base.Nrays)
    p, q = base.copy(), base.copy()
    zh = [np.full(shape, -1.0) for _ in range(2)]
    zr = [np.full(shape, -1.0) for _ in range(2)]
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    hipc.set_zplane(zh[0], zh[1])
    ref.set_zplane(zr[0], zr[1])
    for it in range(2):
        p.gamma_prefill()
        q.gamma_prefill()
        hipc.formal_sol_gamma_matrices()
        ref.formal_sol_gamma_matrices()
        assert rel_err(zh[0], zr[0]) <= TOL_ONE_CALL * (it + 1) and rel_err(zh[1], zr[1]) <= TOL_ONE_CALL * (it + 1)
        assert rel_err(p.J, q.J) <= TOL_ONE_CALL * (it + 1)
    hipc.set_zplane(None, None)          # and off again
    p.gamma_prefill()
    hipc.formal_sol_gamma_matrices()


@needs_ref
@pytest.mark.gpu
def test_dropin_2d_callable_z_boundary(gpu):
    """A 2D Context irradiated from above (zUpperBc CALLABLE, bcData [Nlambda, Nmu, Nx]) under the real core."""
    from test_fs2d import callable_z_problem
    base = callable_z_problem()
    p, q = base.copy(), base.copy()
    hipc, ref = RefContext(p, scheme=PLUGIN), RefContext(q)
    for it in range(2):
        p.gamma_prefill()
        q.gamma_prefill()
        hipc.formal_sol_gamma_matrices()
        ref.formal_sol_gamma_matrices()
        assert rel_err(p.J, q.J) <= TOL_ONE_CALL * (it + 1) and rel_err(p.I, q.I) <= TOL_ONE_CALL * (it + 1)

#!/usr/bin/env python3
"""bench.py -- formal_sol_gamma_matrices iterations/second on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nlambda 10240] [--scaling weak|strong]

Workload (BASELINE.json north star): FAL-C, 82 depths, H(6) + Ca II(6) active, 5 mu-angles, fp64,
Bezier3 short characteristics, wavelength grid densified to ~10 240 points per GPU.  One step =
one full iteration (Gamma pre-fill from C on the device, sweep over all wavelengths and rays,
J / I / Gamma / rates updated, Gamma finalised, dJMax returned to the host), inputs resident in
HBM when the timed region starts.

N > 1 (one rank per GPU under torch.distributed.run; called WITHOUT a launcher, `python bench.py --gpus N` starts the ranks
itself as child processes before it touches the GPU): wavelengths are sharded contiguously; each
iteration does ONE all-reduce (RCCL) of the concatenated Gamma / Rij / Rji partial sums with every
rank's (dJMax, idx) riding in its tail.
  --scaling weak   (default) the global grid has N x 10 240 wavelengths: per-GPU work is fixed as N grows;
                   `value` counts 10 240-wavelength shard-iterations per second over all ranks.
  --scaling strong the 10 240-wavelength grid is split over the N ranks; `value` = whole-problem iterations/s.
`allreduce_us` is the mean host time per step spent issuing the collective (N > 1).

Beside the headline line's fields: `parity_checked` / `max_rel_err` compare a first iteration of the GPU path with
the first iteration of the `cpu_baseline` leg on identical inputs; `aux` carries the other BASELINE configs, each
with its own roofline block (not part of the metric): C3 the PRD sub-iteration, C4 a 512-column fused 1.5D batch,
C5 the 256 x 82 2D iteration.  `--no-aux` skips them, `--aux-only NAME` runs just one (profiling).

`--mode columns` is BASELINE configs[3] as a run of its own: `--columns` (512) perturbed FAL-C columns PER GPU, rank r owning
columns r, r + N, ... of the seeded set as one fused batch, no collective; value = column-iterations/s over all ranks.  With
N > 1 the default (lambda) line carries a short form of it as aux.C4_columns.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X fp64 vector (non-matrix) peak, same guide
# What the part actually issues (tools/micro/issue_bench.hip, profiles/r04_lanesweep.md section 1): one fp64 FMA per 8.5 clocks per
# SIMD with two resident wavefronts (the lane sweep's occupancy: 256 VGPRs), 6.9 with four -- against the data sheet's 4
FP64_ATTAINABLE_TFLOPS = {'two_waves_per_simd': FP64_VECTOR_PEAK_TFLOPS * 4.0 / 8.5, 'four_waves_per_simd': FP64_VECTOR_PEAK_TFLOPS * 4.0 / 6.9}
PARITY_TOL = 1e-9       # relative, element-wise, one call on identical inputs


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--nlambda', type=int, default=10240, help='wavelengths per GPU (weak) / in total (strong)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='strong',
                    help='strong (default): --nlambda wavelengths in total, split over the ranks: value = whole-problem '
                         'iterations/s, the metric of BASELINE.json; weak: --nlambda per rank, value = shard-iterations/s')
    ap.add_argument('--no-weak', action='store_true', help='N > 1: skip the secondary weak-scaling measurement')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-aux', action='store_true', help='skip the C3 / C4 / C5 side measurements')
    ap.add_argument('--aux', action='store_true', help='(accepted for compatibility: aux entries are on by default)')
    ap.add_argument('--aux-only', choices=('prd', 'columns', '2d', 'fixed', 'realistic', 'moving', 'ref500', 'dropin'), default=None,
                    help='run only this side measurement and print its entry (for rocprofv3)')
    ap.add_argument('--columns', type=int, default=512, help='columns of the C4 batch PER GPU (512 x 8 GPUs = the 4 096 of BASELINE configs[3])')
    ap.add_argument('--mode', choices=('lambda', 'columns'), default='lambda',
                    help="lambda (default): the headline -- one problem's wavelengths sharded over the ranks, one all-reduce per "
                         'step; columns: BASELINE configs[3] -- rank r owns columns r, r + N, ... of --columns x N perturbed FAL-C '
                         'columns as one fused batch, NO collective, value = column-iterations/s over all ranks (weak scaling)')
    ap.add_argument('--backend', default='nccl',
                    help="torch.distributed backend for N > 1 ('nccl' = RCCL, one rank per GPU; 'gloo' with --share-gpu "
                         'exercises the sharded path with several ranks on ONE GPU)')
    ap.add_argument('--share-gpu', action='store_true', help='every rank uses device 0 (functional check only)')
    return ap.parse_args()


PREWARM = int(os.environ.get('LWHIP_BENCH_PREWARM', '200'))  # untimed steps before the warm-up steps (clock settle; see main)


def profiled_traffic(sweep_kind='march'):
    """HBM bytes per sweep launch from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_summary.json, produced by tools/profile.sh: FETCH_SIZE and WRITE_SIZE in separate
    --pmc runs, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md).  Counters cannot be
    collected from inside this process, so the latest committed value is reported, or null."""
    import glob
    want = 'lanesweep_kernel' if sweep_kind == 'lanes' else 'raymarch_kernel'
    for fn in reversed(sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_summary.json')))):
        try:
            with open(fn) as f:
                d = json.load(f)
        except Exception:
            continue
        base = os.path.basename(fn)
        # the profile of this command: the same sweep kernel at the default 10 240 wavelengths
        if want in str(d.get('sweep_kernel', '')) and not any(tag in base for tag in ('_5120', '_2560', '_1280')):
            return d.get('sweep_traffic_bytes_per_launch'), base
    return None, None


def algorithmic_flops(prob, la0, la1):
    """fp64 operations one iteration needs on wavelengths [la0, la1), by SURVEY.md section 8(d)'s count:
    per (wavelength, ray, depth) ~120 for the Bezier3 short characteristic (incl. one exp and ~10
    divisions) + 12 per active transition in the gather + 22 per active transition in the rates."""
    import numpy as np
    active = np.zeros(prob.Nlambda)
    for a in prob.atoms:
        for t in a.trans:
            active[t.Nblue:t.Nred] += 1
    per_point = 120.0 + 34.0 * active[la0:la1]
    return float(per_point.sum() * 2 * prob.Nrays * prob.Nspace)


def snapshot(prob):
    """Copies of everything one iteration writes."""
    out = {'J': prob.J.copy(), 'I': prob.I.copy()}
    for ia, a in enumerate(prob.atoms):
        if not a.detailed:
            out[f'Gamma{ia}'] = a.Gamma.copy()
        for kr, t in enumerate(a.trans):
            out[f'Rij{ia}_{kr}'] = t.Rij.copy()
            out[f'Rji{ia}_{kr}'] = t.Rji.copy()
    return out


def rel_err(a, b):
    """max element-wise |a-b|/|b| (where the reference is exactly 0: |a|), as tests/helpers.py."""
    import numpy as np
    nz = b != 0.0
    err = float(np.max(np.abs(a[nz] - b[nz]) / np.abs(b[nz]))) if nz.any() else 0.0
    if (~nz).any():
        err = max(err, float(np.max(np.abs(a[~nz]))))
    return err


def first_iteration(p, ctx, step):
    """Outputs of one iteration from the problem's initial state, and the populations one stat_equil then gives
    (populations restored afterwards)."""
    p.gamma_prefill()
    step()
    first = snapshot(p)
    n0 = [a.n.copy() for a in p.atoms]
    ctx.stat_equil()
    first.update({f'n{ia}': a.n.copy() for ia, a in enumerate(p.atoms) if not a.detailed})
    for a, n in zip(p.atoms, n0):
        a.n[...] = n
    return first


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None: the GPU boxes of this pool show 256
    cpus and grant 16 -- a 64-thread CPU baseline then runs on 16 cpus' worth of time, and says so."""
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        return None if q == 'max' else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        p = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def cpu_baseline(prob, budget_s):
    """The reference core (oracle/_ref, kind 'reference') with its widest SIMD scheme -- or, where it is absent, the
    C oracle (kind 'port') -- timed on the host cores on a bounded number of iterations of the SAME workload.
    Reported beside the GPU number; not the optimisation target.

    Also returns the parity references: the first iteration (+ stat_equil) of the reference's SCALAR scheme
    (`mali_full_precond_scalar`, what SURVEY.md 8c names as the oracle; the C port where the reference is absent),
    and of the timed SIMD scheme -- the reference's own SIMD schemes differ from its scalar scheme by ~1e-6 in J
    near the lower boundary, which is reported, not hidden."""
    from oracle import bindings
    ncpu = os.cpu_count() or 1
    threads = min(ncpu, 64)
    p = prob.copy()
    simd_first = None
    if bindings.ref_available():
        isa = next((i for i in ('AVX512', 'AVX2FMA', 'SSE2') if bindings.simd_plugin(i)), None)
        ctx = bindings.RefContext(p, scheme=bindings.simd_plugin(isa) if isa else None,
                                  Nthreads=threads)
        kind, name = 'reference', ctx.scheme_name
        step = lambda: ctx.formal_sol_gamma_matrices()
        q = prob.copy()
        sc = bindings.RefContext(q, Nthreads=threads)
        scalar_first = first_iteration(q, sc, lambda: sc.formal_sol_gamma_matrices())
        scalar_name = sc.scheme_name
        simd_first = first_iteration(p, ctx, step)
    else:
        ctx = bindings.OracleContext(p)
        kind, name = 'port', 'lw_oracle.c (OpenMP)'
        step = lambda: ctx.formal_sol_gamma_matrices(Nthreads=threads)
        scalar_first = first_iteration(p, ctx, step)
        scalar_name = name
    n, t0 = 0, time.perf_counter()
    while True:
        p.gamma_prefill()
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 50:
            break
    quota = cpu_quota()
    return ({'value': n / dt, 'unit': 'iters/s', 'cores': threads, 'kind': kind, 'cpu_quota_cpus': quota,
             'sample': f'{n} iterations of the same {prob.Nlambda}-wavelength workload, scheme {name}, '
                       f'Nthreads={threads} of {ncpu} host cpus'
                       + (f" (the container's cpu quota is {quota:g} cpus' worth of time)" if quota else '')},
            {'name': scalar_name, 'first': scalar_first, 'simd_name': name, 'simd_first': simd_first})


def gpu_first_iteration(prob, device, stream):
    """First iteration + stat_equil of the HIP path on the inputs the CPU leg starts from (host profiles uploaded)."""
    from lightweaver_amd.context import Context
    g = prob.copy()
    with Context(g, device=device, stream=stream) as c:
        c.formal_sol_gamma_matrices()
        out = snapshot(g)
        c.stat_equil()
        out.update({f'n{ia}': a.n.copy() for ia, a in enumerate(g.atoms) if not a.detailed})
    return out


def roofline_block(kernel, alg_bytes, ms, basis):
    achieved = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {'bound': 'hbm', 'kernel': kernel, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS, 'traffic': None, 'algorithmic_bytes_per_launch': alg_bytes,
            'kernel_ms': ms, 'time_basis': basis}


class no_gc:
    """The interpreter's cyclic collector is not part of any workload timed here (main(): a full collection with torch imported
    is ~40 ms and lands wherever the allocation counts put it): collect before, keep it off inside."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.collect()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()


def timed(fn, n, torch):
    with no_gc():
        # (the collection above is tens of milliseconds with the device idle: warm-up calls AFTER it, so that the timed calls run
        # at the clocks of a loaded device, as the headline's do)
        for _ in range(max(2, min(n, 50))):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n


def aux_prd(args, device, stream, torch, models):
    """C3: FAL-C with Ca II H & K as PRD lines on the bench grid: one PRD sub-iteration = scattering integral of
    every PRD line + the rates pass over the PRD wavelengths (redistribute_prd_lines, Source/PrdTemplates.hpp:175-290)."""
    from lightweaver_amd.context import Context
    pp = models.throughput_grid(NlambdaTarget=args.nlambda, Nrays=5, prd=True)
    with Context(pp, device=device, stream=stream) as pc:
        for _ in range(3):
            pc.formal_sol_gamma_matrices(deviceResident=True)
        nsub = 3
        ms = timed(lambda: pc.prd_redistribute(nsub, 0.0, deviceResident=True), 5, torch) / nsub
        upd = pc.prd_redistribute(nsub, 0.0, deviceResident=True)
        lines = [t for a in pp.atoms if not a.detailed for t in a.trans if t.rhoPrd is not None]
        nla = sum(t.Nred - t.Nblue for t in lines)
        nprdLa = len({la for t in lines for la in range(t.Nblue, t.Nred)})
    Ns, Nmu = pp.Nspace, pp.Nrays
    # inputs and outputs only -- J in and rho out per PRD line, then the rates pass over the PRD wavelengths
    # (phi once, background, J in / out).  The library's own cache of the fine-grid gII weights and grid indices (10 B per fine point, <= 87
    # points per (lambda, depth); 20 B until round 4, which also cached the interpolation fraction) is an
    # intermediate: reported beside the roofline, not inside it.
    alg = 8.0 * 2 * nla * Ns + 8.0 * Ns * (2.0 * Nmu * nla + 5.0 * nprdLa)
    cache = 10.0 * 87 * nla * Ns
    out = {'config': 'C3: FALC H(6)+CaII(6), Ca II H & K PRD, same wavelength grid',
           'ms_per_subiter': ms, 'subiters_per_s': 1e3 / ms, 'prd_lines': int(upd.dRho.shape[1]),
           'Nlambda': pp.Nlambda, 'Nlambda_prd': nprdLa,
           'roofline': roofline_block('prd_scatter_kernel + the sweep in its PRD-rates mode', alg, ms,
                                      'whole sub-iteration (host clock)'),
           'cache_bytes_per_subiter': cache,
           'cache_stream_GBps': cache / (ms * 1e-3) / 1e9}
    if not args.no_cpu_baseline:
        # parity at the timed size: one formal solution + redistribute_prd(3, 0) from identical inputs, oracle vs HIP
        try:
            import numpy as np
            from oracle.bindings import OracleContext
            q = pp.copy()
            for t in (t for a in q.atoms for t in a.trans if t.rhoPrd is not None):
                t.rhoPrd[...] = 1.0
            p2 = q.copy()
            oc = OracleContext(q)
            q.gamma_prefill()
            oc.formal_sol_gamma_matrices(Nthreads=min(os.cpu_count() or 1, 64))
            oc.redistribute_prd(3, 0.0)
            with Context(p2, device=device, stream=stream) as c2:
                p2.gamma_prefill()
                c2.formal_sol_gamma_matrices()
                c2.redistribute_prd(3, 0.0)
            rel = lambda x, y: float(np.max(np.abs(x - y)) / np.max(np.abs(y)))
            errs = {'J': rel(p2.J, q.J)}
            for a, b in zip(p2.atoms, q.atoms):
                for ta, tb in zip(a.trans, b.trans):
                    if ta.rhoPrd is not None:
                        errs['rho'] = max(errs.get('rho', 0.0), rel(ta.rhoPrd, tb.rhoPrd))
                        errs['R'] = max(errs.get('R', 0.0), rel(ta.Rij, tb.Rij), rel(ta.Rji, tb.Rji))
            out['parity'] = {'against': 'oracle redistribute_prd (the reference scheme restated, pinned in tests/test_prd.py)',
                             'rel_err': errs, 'tolerance': 1e-9}
            out['max_rel_err'] = max(errs.values())
            out['parity_checked'] = bool(out['max_rel_err'] <= 1e-9)
        except Exception as e:  # the measurement stands without the check; say why it is missing
            out['parity_checked'] = False
            out['parity_error'] = repr(e)
    return out


def aux_columns(args, device, stream, torch, models):
    """C4: a fused 1.5D batch of seeded-perturbed FAL-C columns, H + Ca II at ~3 000 wavelengths each, every column
    resident in HBM, line profiles generated on the device; columns shard over GPUs with no collective."""
    from lightweaver_amd.batch import ColumnBatch
    ncol = args.columns
    base = models.falc82()
    t0 = time.perf_counter()
    probs = [models.falc_h_ca(Nrays=5, lineScale=3.1, atmos=models.perturbed(base, seed=1234 + c), computeProfiles=False)
             for c in range(ncol)]
    t1 = time.perf_counter()
    with ColumnBatch(probs, device=device, stream=stream) as batch:
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        batch.formal_sol_gamma_matrices(sync_host=False)
        ms = timed(lambda: batch.formal_sol_gamma_matrices(sync_host=False), 5, torch)
        alg = float(sum(c.algorithmic_bytes() for c in batch.contexts))
        fused = batch._batch is not None
        kind = batch.contexts[0].sweep_kind()
        # f1: phi / wphi of every line of every column regenerated on the device (what follows an atmosphere update)
        prof_ms = timed(batch.compute_profiles, 3, torch)
    return {'config': f'C4: {ncol} perturbed FALC columns, H(6)+CaII(6), {probs[0].Nlambda} wavelengths each, fused batch',
            'columns': ncol, 'fused': fused, 'ms_per_batch_iteration': ms, 'column_iters_per_s': ncol * 1e3 / ms,
            'host_build_s': t1 - t0, 'contexts_and_profiles_s': t2 - t1, 'device_profiles_ms': prof_ms,
            'sweep': kind,
            'roofline': roofline_block(('lanesweep_kernel<BATCH> (+ tile_pre / reduce / apply)' if kind == 'lanes'
                                        else 'raymarch_kernel<BATCH> (+ tile_pre / tile_post / reduce / apply)'), alg, ms,
                                       'whole batch iteration (host clock)')}


def aux_2d(args, device, stream, torch, models):
    """C5: 256 x 82 x-periodic atmosphere of perturbed FAL-C columns, H(6), BESSER short characteristics."""
    import numpy as np
    from lightweaver_amd.context import Context
    base = models.falc82()
    cols = [models.perturbed(base, seed=1234 + j) for j in range(256)]
    p2 = models.build_problem_2d(cols, np.linspace(0.0, 25.0e3 * 255, 256), [models.H_6(0.5)])
    with Context(p2, device=device, stream=stream) as c2:
        ms = timed(lambda: c2.formal_sol_gamma_matrices(deviceResident=True), 5, torch)
        alg = c2.algorithmic_bytes()
    return {'config': f'C5: {p2.grid2d.Nx}x{p2.grid2d.Nz}, H(6), {p2.Nlambda} wavelengths, {p2.Nrays}x2 rays, periodic x',
            'ms_per_iteration': ms, 'iters_per_s': 1e3 / ms,
            'roofline': roofline_block('fs2d_coef_kernel + rates2d_kernel + fs2d_scan_kernel + gather2d_kernel', alg, ms,
                                       'whole iteration (host clock)')}


def aux_fixed_order(args, device, stream, torch, models):
    """The same workload with Gamma / rates summed in a FIXED order (LWHIP_OPT_DETERMINISTIC: what the Lightweaver plugin
    selects by default): bit-reproducible iterations, as the reference's schemes are."""
    from lightweaver_amd.context import Context
    prob = models.throughput_grid(NlambdaTarget=args.nlambda, Nrays=5, computeProfiles=False)
    with Context(prob, device=device, stream=stream, deterministic=True) as c:
        c.compute_profiles(deviceResident=True)
        for _ in range(100):
            c.formal_sol_gamma_matrices(deviceResident=True)
        ms = timed(lambda: c.formal_sol_gamma_matrices(deviceResident=True), 100, torch)
        kind = c.sweep_kind()
    return {'config': 'the headline workload in the fixed-order mode (bit-reproducible Gamma / rates)', 'ms_per_step': ms,
            'iters_per_s': 1e3 / ms, 'sweep': kind}


def aux_realistic(args, device, stream, torch, models):
    """A grid with what real model atoms have and the stock H + Ca II pair lacks: overlapping lines WITH coupled continua --
    hydrogen and its deuterium-like twin (models.D_6: every Lyman / Balmer line sits inside the other atom's line, so more
    than half of the wavelengths carry two lines and the bound-free continua coupled to each), three active atoms'
    continua in the ultraviolet.  Round 4 sent such a context to the march as a whole; the lane sweep now serves it tile
    by tile (generic kind, wide finish program)."""
    from lightweaver_amd.context import Context
    prob = models.throughput_grid(NlambdaTarget=args.nlambda, Nrays=5, realistic=True, computeProfiles=False)
    import numpy as np
    lines = np.zeros(prob.Nlambda, dtype=int)
    for a in prob.atoms:
        for t in a.trans:
            if t.alpha is None:
                lines[t.Nblue:t.Nred] += 1
    with Context(prob, device=device, stream=stream) as c:
        c.compute_profiles(deviceResident=True)
        for _ in range(60):
            c.formal_sol_gamma_matrices(deviceResident=True)
        ms = timed(lambda: c.formal_sol_gamma_matrices(deviceResident=True), 60, torch)
        kind = c.sweep_kind()
        alg = c.algorithmic_bytes()
    return {'config': f'FALC 82-depth H(6)+D(6)+CaII(6), 5 mu, {prob.Nlambda} wavelengths: '
                      f'{int((lines >= 2).sum())} with two overlapping lines, {int((lines == 1).sum())} with one',
            'ms_per_step': ms, 'iters_per_s': 1e3 / ms, 'sweep': kind,
            'roofline': roofline_block('lanesweep_kernel (generic tiles: two lines + coupled continua)' if kind == 'lanes'
                                       else 'raymarch_kernel', alg, ms, 'whole step (host clock)')}


def column_problems(models, ncol_total, world, rank, lineScale=3.1):
    """The columns of rank `rank` (round robin: lightweaver_amd.batch.columns_of_rank) of the seeded set of C4: column c is
    FAL-C perturbed with seed 1234 + c, whatever the number of ranks."""
    from lightweaver_amd.batch import columns_of_rank
    base = models.falc82()
    mine = columns_of_rank(ncol_total, world, rank)
    return mine, [models.falc_h_ca(Nrays=5, lineScale=lineScale, atmos=models.perturbed(base, seed=1234 + c), computeProfiles=False)
                  for c in mine]


def run_columns(args, rank, world, local_rank, stream, torch, dist, models, steps, warmup):
    """BASELINE configs[3] across the ranks: every rank advances ITS columns as one fused batch on its GPU; the columns never
    exchange radiation (independent Contexts in the reference: Source/SimdFullIterationTemplates.hpp:588-719 per column), so
    there is no collective in the data path -- only the barrier that brackets the timed region.  Returns the whole-job figures
    on every rank (elapsed = max over ranks)."""
    from lightweaver_amd.batch import ColumnBatch
    ncol_total = args.columns * world
    t0 = time.perf_counter()
    mine, probs = column_problems(models, ncol_total, world, rank)
    t1 = time.perf_counter()
    with ColumnBatch(probs, device=local_rank, stream=stream.cuda_stream) as batch:
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        step = lambda: batch.formal_sol_gamma_matrices(sync_host=False)   # noqa: E731
        for _ in range(warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tt = time.perf_counter()
        for _ in range(steps):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - tt
        last = batch.formal_sol_gamma_matrices(sync_host=True)
        alg = float(sum(c.algorithmic_bytes() for c in batch.contexts))
        fused = batch._batch is not None
        kind = batch.contexts[0].sweep_kind()
        nla = probs[0].Nlambda
    own_ms = 1e3 * elapsed / steps
    if world > 1:
        # the slowest rank's clock; every rank's own time rides along (no data-path collective: these are the report's)
        tv = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(tv, op=dist.ReduceOp.MAX)
        elapsed = float(tv.item())
        per = torch.zeros(world, dtype=torch.float64, device='cuda')
        per[rank] = own_ms
        dist.all_reduce(per)
        per_rank_ms = [float(x) for x in per.tolist()]
    else:
        per_rank_ms = [own_ms]
    return {'columns_total': ncol_total, 'columns_per_gpu': len(mine), 'Nlambda_per_column': nla, 'fused': fused, 'sweep': kind,
            'elapsed': elapsed, 'ms_per_batch_iteration': 1e3 * elapsed / steps,
            'column_iters_per_s': ncol_total * steps / elapsed, 'per_rank_ms': per_rank_ms,
            'host_build_s': t1 - t0, 'contexts_and_profiles_s': t2 - t1,
            'final_dJMax_max': max(u.dJMax for u in last),
            'roofline': roofline_block(('lanesweep_kernel<BATCH> (+ tile_pre / reduce / apply)' if kind == 'lanes'
                                        else 'raymarch_kernel<BATCH> (+ tile_pre / tile_post / reduce / apply)'), alg, own_ms,
                                       "this rank's whole batch iteration (host clock); algorithmic bytes of ITS columns")}


def columns_line(args, c, world):
    """The bench line of --mode columns."""
    return {
        'metric': 'formal_sol_gamma_matrices column-iterations/sec (1.5D batch, BASELINE configs[3])',
        'value': c['column_iters_per_s'], 'unit': 'column-iters/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': c['ms_per_batch_iteration'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic (FAL-C perturbed per column with seed 1234 + c, own H/CaII model atoms, generated in-process)',
        'config': {'workload': f"1.5D batch: {c['columns_total']} perturbed FALC 82-depth columns, H(6)+CaII(6), 5 mu, Bezier3, "
                               f"{c['Nlambda_per_column']} wavelengths each; {c['columns_per_gpu']} columns per GPU as one fused batch",
                   'columns_total': c['columns_total'], 'columns_per_gpu': c['columns_per_gpu'],
                   'parallelism': f'column replicas x{world}, no collective' if world > 1 else 'single GPU',
                   'sweep': c['sweep'], 'fused': c['fused'], 'final_dJMax_max': c['final_dJMax_max']},
        'roofline': c['roofline'], 'per_rank_ms_per_step': c['per_rank_ms'],
        'host_build_s': c['host_build_s'], 'contexts_and_profiles_s': c['contexts_and_profiles_s'],
    }


def aux_moving(args, device, stream, torch, models):
    """The headline grid on a MOVING atmosphere (FAL-C perturbed with seed 1234: a smoothed 2 km/s line-of-sight velocity
    field, the case of every RADYN / flare run): the line profiles differ between the two directions of an angle, so the
    lane sweep cannot share a direction's gather and stencils between the rays of a pair as it does on the static headline
    atmosphere.  Its kernel time gives roofline.frac_general; parity of its first iteration against the oracle in the run."""
    import numpy as np
    from lightweaver_amd.context import Context
    atmos = models.perturbed(models.falc82(), seed=1234)
    prob = models.throughput_grid(NlambdaTarget=args.nlambda, Nrays=5, computeProfiles=False, atmos=atmos)
    with Context(prob, device=device, stream=stream) as c:
        c.compute_profiles(deviceResident=True)
        for _ in range(100):
            c.formal_sol_gamma_matrices(deviceResident=True)
        c.profile_enable(4)
        ms = timed(lambda: c.formal_sol_gamma_matrices(deviceResident=True), 100, torch)
        kms, nl = c.sweep_time()
        c.profile_enable(False)
        kind = c.sweep_kind()
        alg = c.algorithmic_bytes()
    out = {'config': f'the headline workload on a moving atmosphere (perturbed FAL-C, seed 1234, |v| up to '
                     f'{float(np.max(np.abs(atmos.vlos))) * 1e-3:.1f} km/s), {prob.Nlambda} wavelengths',
           'ms_per_step': ms, 'iters_per_s': 1e3 / ms, 'sweep': kind,
           'roofline': roofline_block('lanesweep_kernel (general path: no ray pairs)' if kind == 'lanes' else 'raymarch_kernel',
                                      alg, kms if kms > 0 else ms, f'HIP events around the sweep launch ({nl} timed)' if kms > 0 else 'whole step')}
    if not args.no_cpu_baseline:
        try:
            from oracle.bindings import OracleContext
            q = prob.copy()
            models.compute_profiles_host(q)
            g = q.copy()
            oc = OracleContext(q)
            q.gamma_prefill()
            oc.formal_sol_gamma_matrices(Nthreads=min(os.cpu_count() or 1, 64))
            with Context(g, device=device, stream=stream) as c2:
                g.gamma_prefill()
                c2.formal_sol_gamma_matrices()
            a, b = snapshot(g), snapshot(q)
            errs = {}
            for k in b:
                grp = ''.join(ch for ch in k.split('_')[0] if not ch.isdigit())
                errs[grp] = max(errs.get(grp, 0.0), rel_err(a[k], b[k]))
            out['parity'] = {'against': 'lw_oracle.c (pinned to the reference core bit for bit, tests/test_oracle_vs_reference.py)',
                             'rel_err': errs, 'tolerance': PARITY_TOL}
            out['max_rel_err'] = max(errs.values())
            out['parity_checked'] = bool(out['max_rel_err'] <= PARITY_TOL)
        except Exception as e:
            out['parity_checked'] = False
            out['parity_error'] = repr(e)
    return out


def aux_ref500(args, device, stream, torch, models):
    """The configuration of the reference's ONLY shipped benchmark harness (lightweaver/benchmark.py:19-45): FAL-C resampled
    to 500 depth points, H(6) + Ca II(6) active, quadrature(5), 50 x formal_sol_gamma_matrices; beside it the reference core
    with its widest SIMD scheme timed the way that harness does (Nthreads = 1) and on all host cores."""
    from lightweaver_amd.context import Context
    atmos = models.resample(models.falc82(), 500)
    prob = models.falc_h_ca(Nrays=5, lineScale=3.1, atmos=atmos, computeProfiles=False)
    with Context(prob, device=device, stream=stream) as c:
        c.compute_profiles(deviceResident=True)
        for _ in range(30):
            c.formal_sol_gamma_matrices(deviceResident=True)
        c.profile_enable(2)
        ms = timed(lambda: c.formal_sol_gamma_matrices(deviceResident=True), 50, torch)
        kms, nl = c.sweep_time()
        c.profile_enable(False)
        kind = c.sweep_kind()
        alg = c.algorithmic_bytes()
    out = {'config': f'reference benchmark.py configuration: FALC resampled to 500 depths, H(6)+CaII(6), 5 mu, {prob.Nlambda} wavelengths, '
                     '50 x formal_sol_gamma_matrices',
           'ms_per_step': ms, 'iters_per_s': 1e3 / ms, 'sweep': kind,
           'roofline': roofline_block('lanesweep_kernel' if kind == 'lanes' else 'raymarch_kernel', alg, kms if kms > 0 else ms,
                                      f'HIP events around the sweep launch ({nl} timed)' if kms > 0 else 'whole step')}
    if not args.no_cpu_baseline:
        try:
            from oracle import bindings
            if not bindings.ref_available():
                raise RuntimeError('oracle/_ref is not on this box')
            isa = next((i for i in ('AVX512', 'AVX2FMA', 'SSE2') if bindings.simd_plugin(i)), None)
            ncpu = os.cpu_count() or 1
            cpu = {}
            host = prob.copy()
            models.compute_profiles_host(host)   # (the CPU leg needs phi on the host)
            for label, nthr, budget in (('Nthreads_1_as_benchmark_py', 1, 6.0), ('all_cores', min(ncpu, 64), 4.0)):
                q = host.copy()
                rc = bindings.RefContext(q, scheme=bindings.simd_plugin(isa) if isa else None, Nthreads=nthr)
                q.gamma_prefill()
                rc.formal_sol_gamma_matrices()
                n, t0 = 0, time.perf_counter()
                while True:
                    q.gamma_prefill()
                    rc.formal_sol_gamma_matrices()
                    n += 1
                    dt = time.perf_counter() - t0
                    if dt >= budget or n >= 50:
                        break
                cpu[label] = {'iters_per_s': n / dt, 'ms_per_step': 1e3 * dt / n, 'threads': nthr, 'iterations_timed': n,
                              'scheme': rc.scheme_name}
            out['cpu_baseline'] = {'kind': 'reference', 'cpu_quota_cpus': cpu_quota(), **cpu}
            out['speedup_vs_reference_1_thread'] = cpu['Nthreads_1_as_benchmark_py']['ms_per_step'] / ms
        except Exception as e:
            out['cpu_baseline'] = {'failed': repr(e)}
    return out


def aux_dropin(args, device, stream, torch, models):
    """The drop-in call as a Lightweaver user gets it: the REAL core (oracle/_ref/liblwref.so) loads LwHipIterScheme.so through
    its own load_fns_from_path and calls ctx.iterFns.fs_iter -- formal_sol_gamma_matrices with every array the core owns kept
    host-authoritative (Source/LwMiddleLayer.pyx:3152-3207), default environment: 50 calls at the headline size."""
    from lightweaver_amd import build
    from oracle import bindings
    if not bindings.ref_available() or not build.PLUGIN or not os.path.exists(build.PLUGIN):
        return {'failed': 'oracle/_ref or LwHipIterScheme.so is not on this box'}
    prob = models.throughput_grid(NlambdaTarget=args.nlambda, Nrays=5)
    rc = bindings.RefContext(prob, scheme=build.PLUGIN)
    try:
        def call():
            prob.gamma_prefill()
            rc.formal_sol_gamma_matrices()
        # (the host side of a call -- a pool of threads that fingerprints 90 MB of the caller's arrays -- needs some tens of
        # milliseconds of load before the cores run at speed: 50 untimed calls, then 200 timed ones)
        with no_gc():
            for _ in range(50):
                call()
            each = []
            t0 = time.perf_counter()
            for _ in range(200):
                t1 = time.perf_counter()
                call()
                each.append(time.perf_counter() - t1)
            ms = 1e3 * (time.perf_counter() - t0) / 200
            each.sort()
            med, p90 = 1e3 * each[100], 1e3 * each[180]
            t0 = time.perf_counter()
            for _ in range(50):
                prob.gamma_prefill()
            pre = 1e3 * (time.perf_counter() - t0) / 50
        name = rc.scheme_name
    finally:
        rc.close() if hasattr(rc, 'close') else None
    return {'config': f'reference core -> load_fns_from_path(LwHipIterScheme.so) -> 200 x formal_sol_gamma_matrices (after 50 untimed), {prob.Nlambda} '
                      'wavelengths, default environment, Gamma pre-filled on the host before every call as LwContext does',
            'scheme': name, 'ms_per_call': ms - pre, 'ms_per_call_with_host_prefill': ms, 'calls_per_s': 1e3 / ms,
            # (half of a call is host work on a pool of threads: on a box whose host other tenants load, a few calls wait for a
            # descheduled thread and pull the mean up -- the median and the 90th percentile say which it was)
            'ms_per_call_median_with_host_prefill': med, 'ms_per_call_p90_with_host_prefill': p90,
            'host_load_average_1min': (os.getloadavg()[0] if hasattr(os, 'getloadavg') else None)}


AUX = {'prd': ('C3_prd', aux_prd), 'columns': ('C4_columns', aux_columns), '2d': ('C5_2d', aux_2d),
       'fixed': ('fixed_order', aux_fixed_order), 'realistic': ('realistic_1d', aux_realistic),
       'moving': ('moving_1d', aux_moving), 'dropin': ('dropin', aux_dropin), 'ref500': ('ref_benchmark_500', aux_ref500)}


def rank_launch_command(ngpus, argv, port=None):
    """`python bench.py --gpus N` called without a launcher: the command that starts the N ranks (one per GPU) as fresh
    child processes under torch.distributed.run, rendezvous on 127.0.0.1 (the container's hostname may not resolve)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={ngpus}',
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args):
    """Spawns the ranks and passes rank 0's JSON line and the job's exit code on.  Nothing in THIS process has touched the GPU
    (no torch import, no HIP call): the ranks are children, never a re-exec of a process that initialised the device."""
    import subprocess
    cmd = rank_launch_command(args.gpus, sys.argv[1:])
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # (dmabuf IPC: what RCCL needs on this driver)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // max(args.gpus, 1))))
    proc = subprocess.run(cmd, env=env, cwd=ROOT)
    sys.exit(proc.returncode)


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
            self_launch(args)
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    from lightweaver_amd.context import Context
    from lightweaver_amd.distributed import ShardedIteration, shard_bounds
    from lightweaver_amd.harness import models

    if not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    stream = torch.cuda.current_stream()
    if args.aux_only:
        name, fn = AUX[args.aux_only]
        print(json.dumps({name: fn(args, local_rank, stream.cuda_stream, torch, models)}))
        return
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend)

    if args.mode == 'columns':
        import gc
        gc.collect()
        gc.disable()
        c = run_columns(args, rank, world, local_rank, stream, torch, dist, models, args.steps, args.warmup)
        gc.enable()
        if rank == 0:
            print(json.dumps(columns_line(args, c, world)))
        if world > 1:
            dist.destroy_process_group()
        return

    Nla_total = args.nlambda * (world if args.scaling == 'weak' else 1)
    # the line profiles are generated on the device (lwhip_compute_profiles), shard by shard: the host
    # never computes or ships the [Nlambda_line, Nmu, 2, Nspace] profile arrays
    prob = models.throughput_grid(NlambdaTarget=Nla_total, Nrays=5, computeProfiles=False)
    if abs(prob.Nlambda - Nla_total) > 0.01 * Nla_total:
        sys.exit(f'workload generator produced {prob.Nlambda} wavelengths, wanted {Nla_total}')
    bounds = shard_bounds(prob, world)
    la0, la1 = bounds[rank], bounds[rank + 1]

    ctx = Context(prob, device=local_rank, laStart=la0, laEnd=la1, stream=stream.cuda_stream,
                  worldSize=world, worldRank=rank)
    ctx.compute_profiles(deviceResident=True)
    it = ShardedIteration(ctx, world, rank)

    def step():
        return it.formal_sol_gamma_matrices(deviceResident=True)

    # The interpreter's cyclic garbage collector is not part of the workload: with torch imported a full
    # collection takes ~40 ms (80 steps' worth) and, being triggered by allocation counts, lands at the same
    # spot of every run -- measured inside the timed region of a 100-step run (tools/trace_bench.sh: a 38 ms gap
    # with no HIP call before the closing synchronize).  Collect now, keep it off while the clock runs.
    import gc
    gc.collect()
    gc.disable()
    # (the collection above takes tens of milliseconds: it comes BEFORE the warm-up steps, so that no idle gap -- in which the
    # device drops its clocks -- separates the warm-up from the timed steps)
    # Clock settle: a fresh process reaches the device's sustained clocks only after ~30 ms of load.  Round 4, the same box, the
    # 20-step form (tools/prewarm_exp.sh): sweep kernel 0.137 ms / step 0.153 ms with no settle steps, 0.131 / 0.149 with 60
    # (8 ms), 0.127 / 0.142 with 200 (28 ms), the same with 500 and 1 000 -- and the same as a 100-step run measures.  PREWARM
    # untimed steps before the W warm-up steps, reported as "prewarm_steps"; the timed region is exactly K steps.
    for _ in range(PREWARM):
        step()
    for _ in range(args.warmup):
        step()
    ctx.profile_enable(int(os.environ.get('LWHIP_BENCH_PROF', '4')))   # HIP events around every 4th sweep launch
    it.allreduce_s = 0.0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        up = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    allreduce_us = 1e6 * it.allreduce_s / max(args.steps, 1)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    sweep_ms, nlaunch = ctx.sweep_time()
    sweep_kind = ctx.sweep_kind()
    ctx.profile_enable(False)
    alg_bytes = ctx.algorithmic_bytes()
    alg_flops = algorithmic_flops(prob, la0, la1)

    # N > 1, strong: the weak-scaling figure beside it (every rank keeps --nlambda wavelengths: shard-iterations/s)
    weak = None
    if world > 1 and args.scaling == 'strong' and not args.no_weak:
        ctx.close()
        probW = models.throughput_grid(NlambdaTarget=args.nlambda * world, Nrays=5, computeProfiles=False)
        bW = shard_bounds(probW, world)
        ctx = Context(probW, device=local_rank, laStart=bW[rank], laEnd=bW[rank + 1], stream=stream.cuda_stream,
                      worldSize=world, worldRank=rank)
        ctx.compute_profiles(deviceResident=True)
        itW = ShardedIteration(ctx, world, rank)
        gc.collect()
        gc.disable()
        for _ in range(PREWARM + args.warmup):
            itW.formal_sol_gamma_matrices(deviceResident=True)
        dist.barrier()
        torch.cuda.synchronize()
        tw = time.perf_counter()
        for _ in range(args.steps):
            itW.formal_sol_gamma_matrices(deviceResident=True)
        dist.barrier()
        torch.cuda.synchronize()
        ew = torch.tensor([time.perf_counter() - tw], dtype=torch.float64, device='cuda')
        gc.enable()
        dist.all_reduce(ew, op=dist.ReduceOp.MAX)
        weak = {'weak_shard_iters_per_s': world * args.steps / float(ew.item()), 'weak_ms_per_step': 1e3 * float(ew.item()) / args.steps,
                'weak_Nlambda_total': probW.Nlambda}

    # not part of the metric ("report stat_eq time separately"): the population solve, and the other
    # BASELINE configs
    aux = {}
    if world == 1:
        aux['stat_eq_ms'] = timed(lambda: ctx.stat_equil(deviceResident=True), 20, torch)

        # what an iterate_ctx_se loop pays per iteration: the formal solution, then stat_equil of every atom, each with its result
        # on the host (lightweaver/iterate_ctx.py:157-176; device-resident)
        def se_iteration():
            ctx.formal_sol_gamma_matrices(deviceResident=True)
            ctx.stat_equil(deviceResident=True)
        aux['iterate_se_ms'] = timed(se_iteration, 100, torch)
    ctx.close()
    if world > 1 and not args.no_aux:
        # the split that scales by construction, beside the strong-scaling headline: every rank its own fused batch of columns
        try:
            cc = run_columns(args, rank, world, local_rank, stream, torch, dist, models, steps=10, warmup=2)
            cc.pop('elapsed')
            cc['config'] = (f"C4: {cc['columns_total']} perturbed FALC columns over {world} GPUs, {cc['columns_per_gpu']} per GPU, "
                            'no collective (python bench.py --gpus N --mode columns is the full run)')
            aux['C4_columns'] = cc
        except Exception as e:
            aux['C4_columns'] = {'failed': repr(e)}
    if world == 1 and not args.no_aux:
        for key, (name, fn) in AUX.items():
            try:
                aux[name] = fn(args, local_rank, stream.cuda_stream, torch, models)
            except Exception as e:   # a side measurement must never break the bench line
                aux[name] = {'failed': repr(e)}

    if rank == 0:
        per_step = world if args.scaling == 'weak' else 1
        value = per_step * args.steps / elapsed
        achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
        traffic, traffic_src = profiled_traffic(sweep_kind) if (world == 1 and args.nlambda == 10240) else (None, None)
        out = {
            'metric': 'formal_sol_gamma_matrices iters/sec',
            'value': value,
            'unit': 'iters/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'prewarm_steps': PREWARM,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic (FAL-C model atmosphere, own H/CaII model atoms, seeded, generated in-process)',
            'config': {
                'workload': 'FALC 82-depth H(6)+CaII(6), 5 mu, Bezier3, densified wavelength grid',
                'Nlambda_per_gpu': la1 - la0,
                'Nlambda_total': prob.Nlambda,
                'Nspace': prob.Nspace,
                'Nrays': prob.Nrays,
                'active_transitions': sum(len(a.trans) for a in prob.atoms),
                'parallelism': f'lambda-shard x{world}' if world > 1 else 'single GPU',
                'final_dJMax': up.dJMax,
            },
            'roofline': {
                'bound': 'hbm',
                'kernel': ('lanesweep_kernel (depth-across-lanes sweep)' if sweep_kind == 'lanes'
                           else 'raymarch_kernel (ray-column march with the fused post-pass)'),
                'achieved': achieved,
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS,
                'traffic': traffic,
                'traffic_source': traffic_src,
                'algorithmic_bytes_per_launch': alg_bytes,
                'kernel_ms': sweep_ms,
                'launches_timed': nlaunch,
                # what actually limits the kernel: fp64 vector issue, not HBM
                'secondary': {
                    'bound': 'fp64 vector (no MFMA: stencil + recurrence, not a contraction)',
                    'achieved': alg_flops / (sweep_ms * 1e-3) / 1e12 if sweep_ms > 0 else 0.0,
                    'peak': FP64_VECTOR_PEAK_TFLOPS,
                    'unit': 'TFLOP/s',
                    'frac': (alg_flops / (sweep_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS) if sweep_ms > 0 else 0.0,
                    'algorithmic_flops_per_launch': alg_flops,
                    # the measured issue ceiling of the part at the kernel's occupancy (not the data-sheet peak)
                    'attainable': FP64_ATTAINABLE_TFLOPS['two_waves_per_simd'],
                    'attainable_four_waves': FP64_ATTAINABLE_TFLOPS['four_waves_per_simd'],
                    'frac_of_attainable': (alg_flops / (sweep_ms * 1e-3) / 1e12 / FP64_ATTAINABLE_TFLOPS['two_waves_per_simd'])
                    if sweep_ms > 0 else 0.0,
                    'attainable_source': 'tools/micro/issue_bench.hip: one fp64 FMA per 8.5 / 6.9 clocks per SIMD at 2 / 4 '
                                         'wavefronts per SIMD (profiles/r04_lanesweep.md)',
                },
            },
        }
        out['config']['sweep'] = sweep_kind
        if world > 1:
            out['allreduce_us'] = allreduce_us
        if weak:
            out.update(weak)
        mv = aux.get('moving_1d')
        if isinstance(mv, dict) and isinstance(mv.get('roofline'), dict):
            out['roofline']['frac_general'] = mv['roofline']['frac']
            out['roofline']['kernel_ms_general'] = mv['roofline']['kernel_ms']
        out['roofline']['algorithmic_bytes_note'] = (
            'SURVEY 8(d) counts phi of both directions of every angle; on this static atmosphere the lane sweep reads ONE '
            "direction's phi for one-line tiles (ray pairs), so `achieved` is an upper bound on the bytes the kernel moves: "
            'frac_general (aux.moving_1d: the same grid with a velocity field, both directions read) is the number a moving '
            'atmosphere gets')
        for name in ('fixed_order', 'realistic_1d', 'moving_1d'):
            if isinstance(aux.get(name), dict) and 'ms_per_step' in aux[name]:
                aux[name]['ratio_to_headline'] = aux[name]['ms_per_step'] / out['ms_per_step']
        if aux:
            out['aux'] = aux
        if world == 1 and not args.no_cpu_baseline:
            try:
                single = prob.copy()
                models.compute_profiles_host(single)   # the CPU leg needs phi on the host
                out['cpu_baseline'], par = cpu_baseline(single, args.cpu_seconds)
                # parity at the timed size: the HIP path's first iteration (+ stat_equil) on the same inputs
                gpu_first = gpu_first_iteration(single, local_rank, stream.cuda_stream)

                def group_errs(x, ref):
                    errs = {}
                    for k, r in ref.items():
                        grp = ''.join(ch for ch in k.split('_')[0] if not ch.isdigit())
                        errs[grp] = max(errs.get(grp, 0.0), rel_err(x[k], r))
                    return errs
                errs = group_errs(gpu_first, par['first'])
                tol = {k: (1e-8 if k == 'n' else PARITY_TOL) for k in errs}
                out['max_rel_err'] = max(errs.values())
                out['parity_checked'] = bool(all(errs[k] <= tol[k] for k in errs))
                out['parity'] = {'against': par['name'], 'rel_err': errs,
                                 'tolerance': {'one_iteration': PARITY_TOL, 'n_after_stat_equil': 1e-8},
                                 'what': 'first formal_sol_gamma_matrices + stat_equil from identical inputs, '
                                         f'{prob.Nlambda} wavelengths'}
                if par['simd_first'] is not None:
                    out['parity']['reference_simd_vs_scalar'] = {
                        'scheme': par['simd_name'], 'rel_err': group_errs(par['simd_first'], par['first'])}
            except Exception as e:  # the baseline must never break the bench line
                out['cpu_baseline'] = {'value': None, 'unit': 'iters/s', 'cores': 0, 'kind': 'port',
                                       'sample': f'failed: {e!r}'}
                out['parity_checked'] = False
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

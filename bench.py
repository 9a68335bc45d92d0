#!/usr/bin/env python3
"""bench.py -- formal_sol_gamma_matrices iterations/second on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nlambda 10240]

Workload (BASELINE.json north star): FAL-C, 82 depths, H(6) + Ca II(6) active, 5 mu-angles, fp64,
Bezier3 short characteristics, wavelength grid densified to ~10 240 points per GPU.  One step =
one full iteration (Gamma pre-fill from C on the device, sweep over all wavelengths and rays,
J / I / Gamma / rates updated, Gamma finalised, dJMax returned to the host), inputs resident in
HBM when the timed region starts.

N > 1 (launched by torch.distributed.run, one rank per GPU): the global grid has N x 10 240
wavelengths, sharded contiguously; each iteration does ONE all-reduce (RCCL) of the concatenated
Gamma / Rij / Rji partial sums and one tiny all-gather of (dJMax, idx).  Per-GPU work is fixed as N
grows ("weak"); `value` counts 10 240-wavelength shard-iterations per second over all ranks.

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--nlambda', type=int, default=10240, help='wavelengths per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--aux', action='store_true',
                    help='also time the PRD sub-iterations and the 2D iteration (not part of the metric)')
    return ap.parse_args()


def profiled_traffic():
    """HBM bytes per sweep launch from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_summary.json, produced by tools/profile.sh: FETCH_SIZE and WRITE_SIZE in separate
    --pmc runs, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md).  Counters cannot be
    collected from inside this process, so the latest committed value is reported, or null."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_summary.json')))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        return d.get('sweep_traffic_bytes_per_launch'), os.path.basename(files[-1])
    except Exception:
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


def cpu_baseline(prob, budget_s):
    """The reference core (oracle/_ref, kind 'reference') -- or, where it is absent, the C oracle
    (kind 'port') -- timed on the host cores on a bounded number of iterations of the SAME
    workload.  Reported beside the GPU number; not the optimisation target."""
    from oracle import bindings
    ncpu = os.cpu_count() or 1
    p = prob.copy()
    if bindings.ref_available():
        isa = next((i for i in ('AVX512', 'AVX2FMA', 'SSE2') if bindings.simd_plugin(i)), None)
        threads = min(ncpu, 64)
        ctx = bindings.RefContext(p, scheme=bindings.simd_plugin(isa) if isa else None,
                                  Nthreads=threads)
        kind, name = 'reference', ctx.scheme_name
        step = lambda: ctx.formal_sol_gamma_matrices()
    else:
        threads = min(ncpu, 64)
        ctx = bindings.OracleContext(p)
        kind, name = 'port', 'lw_oracle.c (OpenMP)'
        step = lambda: ctx.formal_sol_gamma_matrices(Nthreads=threads)
    p.gamma_prefill()
    step()  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        p.gamma_prefill()
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 50:
            break
    return {'value': n / dt, 'unit': 'iters/s', 'cores': threads, 'kind': kind,
            'sample': f'{n} iterations of the same {prob.Nlambda}-wavelength workload, scheme {name}, '
                      f'Nthreads={threads} of {ncpu} host cpus'}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)')
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    from lightweaver_amd.context import Context
    from lightweaver_amd.distributed import ShardedIteration, shard_bounds
    from lightweaver_amd.harness import models

    if not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    Nla_total = args.nlambda * world
    # the line profiles are generated on the device (lwhip_compute_profiles), shard by shard: the host
    # never computes or ships the [Nlambda_line, Nmu, 2, Nspace] profile arrays
    prob = models.throughput_grid(NlambdaTarget=Nla_total, Nrays=5, computeProfiles=False)
    if abs(prob.Nlambda - Nla_total) > 0.01 * Nla_total:
        sys.exit(f'workload generator produced {prob.Nlambda} wavelengths, wanted {Nla_total}')
    bounds = shard_bounds(prob, world)
    la0, la1 = bounds[rank], bounds[rank + 1]

    stream = torch.cuda.current_stream()
    ctx = Context(prob, device=local_rank, laStart=la0, laEnd=la1, stream=stream.cuda_stream,
                  worldSize=world, worldRank=rank)
    ctx.compute_profiles(deviceResident=True)
    it = ShardedIteration(ctx, world, rank)

    def step():
        return it.formal_sol_gamma_matrices(deviceResident=True)

    for _ in range(args.warmup):
        step()
    ctx.profile_enable(int(os.environ.get('LWHIP_BENCH_PROF', '4')))   # HIP events around every 4th sweep launch
    # The interpreter's cyclic garbage collector is not part of the workload: with torch imported a full
    # collection takes ~40 ms (80 steps' worth) and, being triggered by allocation counts, lands at the same
    # spot of every run -- measured inside the timed region of a 100-step run (tools/trace_bench.sh: a 38 ms gap
    # with no HIP call before the closing synchronize).  Collect now, keep it off while the clock runs.
    import gc
    gc.collect()
    gc.disable()
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
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    sweep_ms, nlaunch = ctx.sweep_time()
    ctx.profile_enable(False)
    alg_bytes = ctx.algorithmic_bytes()
    alg_flops = algorithmic_flops(prob, la0, la1)

    # not part of the metric ("report stat_eq time separately"): the population solve
    aux = {}
    if world == 1:
        ctx.stat_equil(deviceResident=True)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(20):
            ctx.stat_equil(deviceResident=True)
        torch.cuda.synchronize()
        aux['stat_eq_ms'] = 1e3 * (time.perf_counter() - ta) / 20
        if args.aux:
            pp = models.throughput_grid(NlambdaTarget=args.nlambda, Nrays=5, prd=True)
            with Context(pp, device=local_rank, stream=stream.cuda_stream) as pc:
                for _ in range(3):
                    pc.formal_sol_gamma_matrices(deviceResident=True)
                pc.prd_redistribute(3, 0.0, deviceResident=True)
                torch.cuda.synchronize()
                ta = time.perf_counter()
                for _ in range(5):
                    upd = pc.prd_redistribute(3, 0.0, deviceResident=True)
                torch.cuda.synchronize()
                aux['prd_ms_per_subiter'] = 1e3 * (time.perf_counter() - ta) / (5 * 3)
                aux['prd_lines'] = int(upd.dRho.shape[1])
                aux['prd_Nlambda'] = pp.Nlambda
            # BASELINE config 5 in miniature: 256 x 82 x-periodic atmosphere, H(6), BESSER short characteristics
            import numpy as np
            base = models.falc82()
            cols = [models.perturbed(base, seed=1234 + j) for j in range(256)]
            p2 = models.build_problem_2d(cols, np.linspace(0.0, 25.0e3 * 255, 256), [models.H_6(0.5)])
            with Context(p2, device=local_rank, stream=stream.cuda_stream) as c2:
                c2.formal_sol_gamma_matrices(deviceResident=True)
                torch.cuda.synchronize()
                ta = time.perf_counter()
                for _ in range(5):
                    c2.formal_sol_gamma_matrices(deviceResident=True)
                torch.cuda.synchronize()
                aux['iter2d_ms'] = 1e3 * (time.perf_counter() - ta) / 5
                aux['iter2d_config'] = f'{p2.grid2d.Nx}x{p2.grid2d.Nz}, {p2.Nlambda} wavelengths, {p2.Nrays}x2 rays'

    if rank == 0:
        value = world * args.steps / elapsed
        achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
        traffic, traffic_src = profiled_traffic() if (world == 1 and args.nlambda == 10240) else (None, None)
        out = {
            'metric': 'formal_sol_gamma_matrices iters/sec',
            'value': value,
            'unit': 'iters/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
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
                'kernel': 'sweep_atomic_kernel (sweep_body)',
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
                    'bound': 'fp64 vector (no MFMA: stencil + scan, not a contraction)',
                    'achieved': alg_flops / (sweep_ms * 1e-3) / 1e12 if sweep_ms > 0 else 0.0,
                    'peak': FP64_VECTOR_PEAK_TFLOPS,
                    'unit': 'TFLOP/s',
                    'frac': (alg_flops / (sweep_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS) if sweep_ms > 0 else 0.0,
                    'algorithmic_flops_per_launch': alg_flops,
                },
            },
        }
        if aux:
            out['aux'] = aux
        if world == 1 and not args.no_cpu_baseline:
            try:
                single = prob.copy()
                models.compute_profiles_host(single)   # the CPU leg needs phi on the host
                out['cpu_baseline'] = cpu_baseline(single, args.cpu_seconds)
            except Exception as e:  # the baseline must never break the bench line
                out['cpu_baseline'] = {'value': None, 'unit': 'iters/s', 'cores': 0, 'kind': 'port',
                                       'sample': f'failed: {e!r}'}
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""Driven by tools/oracle/hook_lifecycle.sh (which sets up the real Python framework under /tmp): the plugin under
lw.Context -- load, construct, pickle round trip, update_threads, first numerical call, destroy.

    hook_lifecycle.py <plugin.so>        run the lifecycle (trace lines of the plugin go to stderr)
    hook_lifecycle.py --check trace.log  alloc_global_scratch / free_global_scratch must pair up per Context
"""
import gc
import pickle
import re
import sys


def check(path):
    live, allocs, frees, calls = set(), 0, 0, 0
    for line in open(path):
        m = re.match(r'lwhip-plugin-trace (\w+) ctx=(\w+)', line)
        if not m:
            continue
        what, ctx = m.groups()
        if what == 'alloc_global_scratch':
            assert ctx not in live, f'double alloc for {ctx}'
            live.add(ctx)
            allocs += 1
        elif what == 'free_global_scratch':
            assert ctx in live, f'free without alloc for {ctx}'
            live.remove(ctx)
            frees += 1
        else:
            assert ctx in live, f'{what} on a Context without scratch'
            calls += 1
    assert allocs >= 3 and allocs == frees and not live, (allocs, frees, live)
    print(f'hooks pair up: {allocs} alloc_global_scratch / {frees} free_global_scratch, {calls} member calls in between')


def hydrogen3():
    from lightweaver.atomic_model import (AtomicLevel, AtomicModel, HydrogenicContinuum, LinearCoreExpWings,
                                          LineType, VoigtLine)
    from lightweaver.atomic_table import PeriodicTable
    from lightweaver.broadening import (HydrogenLinearStarkBroadening, LineBroadening, RadiativeBroadening,
                                        VdwUnsold)
    from lightweaver.collisional_rates import CE, CI
    T = [3000., 5000., 7000., 10000., 20000., 30000.]
    brd = lambda g: LineBroadening(natural=[RadiativeBroadening(gamma=g)],
                                   elastic=[VdwUnsold(vals=[1.0, 1.0]), HydrogenLinearStarkBroadening()])
    return AtomicModel(
        element=PeriodicTable['H'],
        levels=[AtomicLevel(E=0.0, g=2, label='H I 1S', stage=0), AtomicLevel(E=82258.211, g=8, label='H I 2', stage=0),
                AtomicLevel(E=97491.219, g=18, label='H I 3', stage=0), AtomicLevel(E=109677.617, g=1, label='H II', stage=1)],
        lines=[VoigtLine(j=1, i=0, f=4.162e-1, type=LineType.CRD,
                         quadrature=LinearCoreExpWings(qCore=15.0, qWing=600.0, Nlambda=40), broadening=brd(4.7e8)),
               VoigtLine(j=2, i=1, f=6.407e-1, type=LineType.CRD,
                         quadrature=LinearCoreExpWings(qCore=3.0, qWing=250.0, Nlambda=30), broadening=brd(9.98e7))],
        continua=[HydrogenicContinuum(j=3, i=0, NlambdaGen=10, alpha0=6.152e-22, minWavelength=22.794),
                  HydrogenicContinuum(j=3, i=1, NlambdaGen=10, alpha0=1.379e-21, minWavelength=91.176),
                  HydrogenicContinuum(j=3, i=2, NlambdaGen=10, alpha0=2.149e-21, minWavelength=205.147)],
        collisions=[CE(j=1, i=0, temperature=T, rates=[9.75e-16, 6.098e-16, 4.535e-16, 3.365e-16, 2.008e-16, 1.56e-16]),
                    CE(j=2, i=1, temperature=T, rates=[1.127e-14, 8.077e-15, 6.716e-15, 5.691e-15, 4.419e-15, 3.89e-15]),
                    CI(j=3, i=0, temperature=T, rates=[2.635e-17, 2.864e-17, 3.076e-17, 3.365e-17, 4.138e-17, 4.703e-17]),
                    CI(j=3, i=1, temperature=T, rates=[5.34e-16, 6.596e-16, 7.546e-16, 8.583e-16, 1.025e-15, 1.069e-15]),
                    CI(j=3, i=2, temperature=T, rates=[2.215e-15, 2.792e-15, 3.169e-15, 3.518e-15, 3.884e-15, 3.828e-15])])


def lifecycle(plugin):
    import numpy as np
    import lightweaver as lw
    from lightweaver.fal import Falc82
    from lightweaver.LwCompiled import FsIterationSchemes
    SCHEME = 'mali_full_precond_hip_gfx950'
    FsIterationSchemes.load_fns_from_path(plugin)
    assert SCHEME in FsIterationSchemes.names
    props = FsIterationSchemes.scheme_properties(SCHEME)
    print('registered', props)
    try:
        FsIterationSchemes.load_fns_from_path(plugin)
    except ValueError:
        print('second load of the same path is refused by the manager, as for the built-in plugins')

    def make(nthreads):
        atmos = Falc82()
        atmos.quadrature(3)
        aSet = lw.RadiativeSet([hydrogen3()])
        aSet.set_active('H')
        spect = aSet.compute_wavelength_grid()
        eqPops = aSet.compute_eq_pops(atmos)
        return lw.Context(atmos, spect, eqPops, Nthreads=nthreads, fsIterScheme=SCHEME)

    ctx = make(1)
    J0 = np.copy(ctx.spect.J)
    blob = pickle.dumps(ctx)
    ctx2 = pickle.loads(blob)                       # __setstate__ -> set_fs_iter_scheme -> setup_threads
    assert ctx2.kwargs['fsIterScheme'] == SCHEME
    ctx.update_threads()                            # free_global_scratch + alloc_global_scratch of the same Context
    ctx3 = make(2)                                  # threaded Context: hooks are per Context, not per thread
    ctx3.update_threads()
    try:
        ctx.formal_sol_gamma_matrices()
    except Exception as e:                          # the C++ exception of the plugin, surfaced by Cython's `except +`
        assert 'no HIP device' in str(e), e
        print('first numerical call without a GPU:', type(e).__name__, '-', str(e).splitlines()[0])
    else:
        import ctypes
        n = ctypes.CDLL(plugin.replace('LwHipIterScheme.so', 'liblwhip.so')).lwhip_device_count()
        assert n > 0, 'formal_sol_gamma_matrices returned without a device'
        print('a GPU is present: the call ran on it')
    if not np.array_equal(J0, ctx.spect.J):
        raise AssertionError('J changed although the call failed')
    del ctx, ctx2, ctx3
    gc.collect()
    print('destroyed')


if __name__ == '__main__':
    if sys.argv[1] == '--check':
        check(sys.argv[2])
    else:
        lifecycle(sys.argv[1])

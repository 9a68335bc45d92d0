"""oracle/bindings.py -- TEST INFRASTRUCTURE: ctypes bindings of the two checkers.

  RefLib    oracle/_ref/liblwref.so  -- the real Lightweaver core behind oracle/ref_driver.cpp
  OracleLib oracle/liblworacle.so    -- the plain-C restatement oracle/lw_oracle.c

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Both operate in place on the host arrays of a lightweaver_amd.model.Problem.
"""
import ctypes as C
import os

from lightweaver_amd import _abi as abi

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, '_ref')
REF_LIB = os.path.join(REF_DIR, 'liblwref.so')
ORACLE_LIB = os.path.join(HERE, 'liblworacle.so')


def ref_available():
    return os.path.exists(REF_LIB)


def oracle_available():
    return os.path.exists(ORACLE_LIB)


def simd_plugin(isa):
    """Path of a reference SIMD plugin ('SSE2' | 'AVX2FMA' | 'AVX512') if built and runnable."""
    path = os.path.join(REF_DIR, f'SimdImpl_{isa}.so')
    if not os.path.exists(path):
        return None
    flags = ''
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('flags'):
                    flags = line
                    break
    except OSError:
        pass
    need = {'SSE2': ['sse2'], 'AVX2FMA': ['avx2', 'fma'], 'AVX512': ['avx512f', 'avx512dq']}[isa]
    if not all((' ' + n + ' ') in (flags + ' ') for n in need):
        return None
    return path


class _Handle:
    def __init__(self, lib, prob, create, destroy):
        self.lib = lib
        self.prob = prob               # keep the Problem (and its arrays) alive
        self.desc = prob.descriptor()
        self._destroy = destroy
        self.h = create(self.desc)

    def close(self):
        if self.h:
            self._destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class HprdTables:
    """numpy views of an lwhip_hprd (borrowed: valid while the block that owns it lives) and the descriptor itself
    (`.ptr`: what lwhip_options.hprd / lworacle_set_hprd take)."""

    def __init__(self, ptr, prob, free=None):
        import numpy as np
        self.ptr, self._free = ptr, free
        h = ptr.contents
        Ns, Nrays = prob.Nspace, prob.Nrays
        as_np = np.ctypeslib.as_array
        self.prdIdxs = as_np(h.prdIdxs, (h.NprdLambda,))
        self.hPrdIdxs = as_np(h.hPrdIdxs, (max(h.NhPrd, 1),))[:h.NhPrd]
        self.JRest = as_np(h.JRest, (h.NprdLambda, Ns))
        ncell = h.NhPrd * Nrays * 2 * Ns
        self.jCoeffOff = as_np(h.jCoeffOff, (ncell + 1,))
        nj = int(self.jCoeffOff[-1])
        jdt = np.dtype([('frac', np.float64), ('idx', np.int32), ('_pad', np.int32)])
        jc = np.frombuffer(as_np(C.cast(h.jCoeffs, C.POINTER(C.c_uint8)), (max(nj, 1) * 16,)), dtype=jdt)[:nj]
        self.jFrac = jc['frac']
        self.jIdx = jc['idx']
        self.lineAtom = as_np(h.lineAtom, (h.Nlines,))
        self.lineTrans = as_np(h.lineTrans, (h.Nlines,))
        self.rho = []
        for q in range(h.Nlines):
            t = prob.atoms[int(self.lineAtom[q])].trans[int(self.lineTrans[q])]
            n = (t.Nred - t.Nblue) * Nrays * 2 * Ns
            rdt = np.dtype([('i0', np.int32), ('i1', np.int32), ('frac', np.float64)])
            raw = np.frombuffer(as_np(C.cast(h.rhoCoeffs[q], C.POINTER(C.c_uint8)), (n * 16,)), dtype=rdt)
            self.rho.append((raw['i0'], raw['i1'], raw['frac']))

    def close(self):
        if self._free and self.ptr:
            self._free(self.ptr)
            self.ptr = None


def _redistribute(fn, h, maxIter, tol, nprd_guess=16):
    import numpy as np
    nit = C.c_int32(0)
    nprd = C.c_int32(0)
    dRho = np.zeros(maxIter * nprd_guess)
    dRhoIdx = np.zeros(maxIter * nprd_guess, dtype=np.int32)
    dJ = np.zeros(maxIter)
    dJIdx = np.zeros(maxIter, dtype=np.int32)
    st = fn(h, maxIter, tol, C.byref(nit), C.byref(nprd), dRho.ctypes.data_as(abi.f64p),
            dRhoIdx.ctypes.data_as(abi.i32p), dJ.ctypes.data_as(abi.f64p), dJIdx.ctypes.data_as(abi.i32p))
    assert st == 0
    n, m = nit.value, nprd.value
    return {'NprdSubIter': n, 'Nprd': m, 'dRho': dRho[:n * m].reshape(n, m) if m else dRho[:0],
            'dRhoMaxIdx': dRhoIdx[:n * m].reshape(n, m) if m else dRhoIdx[:0],
            'dJPrdMax': dJ[:n], 'dJPrdMaxIdx': dJIdx[:n]}


class RefContext(_Handle):
    """The reference core driving `prob` (scheme: None = mali_full_precond_scalar, or a plugin path)."""
    _lib = None

    @classmethod
    def lib_(cls):
        if cls._lib is None:
            lib = C.CDLL(REF_LIB)
            lib.lwref_create.restype = C.c_void_p
            lib.lwref_create.argtypes = [C.POINTER(abi.lwhip_problem), C.c_char_p, C.c_int,
                                         C.c_char_p, C.c_int]
            lib.lwref_destroy.argtypes = [C.c_void_p]
            lib.lwref_scheme_name.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
            lib.lwref_compute_profiles.argtypes = [C.c_void_p]
            lib.lwref_formal_sol_gamma_matrices.argtypes = [C.c_void_p, C.c_int,
                                                            C.POINTER(abi.lwhip_iter_result)]
            lib.lwref_formal_sol.argtypes = [C.c_void_p, C.c_int]
            lib.lwref_stat_equil.argtypes = [C.c_void_p, C.c_int]
            lib.lwref_set_depth_range.argtypes = [C.c_void_p, C.c_int, C.c_int]
            lib.lwref_redistribute_prd.argtypes = [C.c_void_p, C.c_int, C.c_double, abi.i32p, abi.i32p,
                                                   abi.f64p, abi.i32p, abi.f64p, abi.i32p]
            lib.lwref_set_include_detailed.argtypes = [C.c_void_p, C.c_int]
            lib.lwref_set_include_detailed.restype = None
            lib.lwref_hprd_export.restype = C.POINTER(abi.lwhip_hprd)
            lib.lwref_hprd_export.argtypes = [C.c_void_p, C.c_int]
            lib.lwref_hprd_free.argtypes = [C.POINTER(abi.lwhip_hprd)]
            lib.lwref_hprd_free.restype = None
            lib.lwref_time_dep_update.argtypes = [C.c_void_p, C.c_int, abi.f64p, C.c_double]
            lib.lwref_nr_post_update.argtypes = [C.c_void_p, C.POINTER(abi.lwhip_nr_args)]
            lib.lwref_GII.restype = C.c_double
            lib.lwref_GII.argtypes = [C.c_double] * 3
            lib.lwref_w2.argtypes = [C.c_double, abi.f64p]
            lib.lwref_bezier3_coeffs.argtypes = [C.c_double, abi.f64p]
            lib.lwref_cent_deriv.restype = C.c_double
            lib.lwref_cent_deriv.argtypes = [C.c_double] * 5
            lib.lwref_voigt_H.restype = C.c_double
            lib.lwref_voigt_H.argtypes = [C.c_double, C.c_double]
            lib.lwref_planck_nu.argtypes = [C.c_int, abi.f64p, C.c_double, abi.f64p]
            lib.lwref_solve_lin_eq.argtypes = [C.c_int, abi.f64p, abi.f64p, C.c_int]
            lib.lwref_formal_solver_ray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                    abi.f64p, abi.f64p, abi.f64p, abi.f64p]
            cls._lib = lib
        return cls._lib

    def __init__(self, prob, scheme=None, Nthreads=1):
        lib = self.lib_()
        err = C.create_string_buffer(512)
        sch = scheme.encode() if scheme else None

        def create(desc):
            h = lib.lwref_create(C.byref(desc), sch, Nthreads, err, 512)
            if not h:
                raise RuntimeError('lwref_create: ' + err.value.decode())
            return h
        super().__init__(lib, prob, create, lib.lwref_destroy)

    @property
    def scheme_name(self):
        buf = C.create_string_buffer(128)
        self.lib.lwref_scheme_name(self.h, buf, 128)
        return buf.value.decode()

    def compute_profiles(self):
        assert self.lib.lwref_compute_profiles(self.h) == 0

    def formal_sol_gamma_matrices(self, lambdaIterate=False):
        res = abi.lwhip_iter_result()
        st = self.lib.lwref_formal_sol_gamma_matrices(self.h, int(lambdaIterate), C.byref(res))
        assert st == 0
        return res.dJMax, res.dJMaxIdx

    def formal_sol(self, upOnly=True):
        assert self.lib.lwref_formal_sol(self.h, int(upOnly)) == 0

    def set_depth_range(self, spaceStart=-1, spaceEnd=-1):
        """spaceStart / spaceEnd of the following stat_equil / time_dep_update / nr_post_update calls."""
        return self.lib.lwref_set_depth_range(self.h, int(spaceStart), int(spaceEnd))

    def stat_equil(self, atom=-1):
        return self.lib.lwref_stat_equil(self.h, atom)

    def set_zplane(self, down=None, up=None):
        """ExtraParams ZPlaneDecomposition / ZPlaneDown / ZPlaneUp of the formal solutions: float64 arrays
        [Nlambda, Nrays(, Nx)] filled in place."""
        self._zplane = (down, up)
        ptr = lambda a: a.ctypes.data_as(abi.f64p) if a is not None else None
        self.lib.lwref_set_zplane.argtypes = [C.c_void_p, abi.f64p, abi.f64p]
        self.lib.lwref_set_zplane.restype = None
        self.lib.lwref_set_zplane(self.h, ptr(down), ptr(up))

    def redistribute_prd(self, maxIter=3, tol=1e-2, include_detailed_atoms=False):
        self.lib.lwref_set_include_detailed(self.h, int(include_detailed_atoms))
        return _redistribute(self.lib.lwref_redistribute_prd, self.h, maxIter, tol)

    def configure_hprd(self, include_detailed_atoms=False):
        """configure_hprd_coeffs on the real Context (hybrid PRD from now on); its tables, flattened."""
        ptr = self.lib.lwref_hprd_export(self.h, int(include_detailed_atoms))
        assert ptr, 'configure_hprd_coeffs: no PRD lines'
        return HprdTables(ptr, self.prob, self.lib.lwref_hprd_free)

    def time_dep_update(self, atom, nOld, dt):
        import numpy as np
        nOld = np.ascontiguousarray(nOld, dtype=np.float64)
        return self.lib.lwref_time_dep_update(self.h, atom, nOld.ctypes.data_as(abi.f64p), dt)

    def nr_post_update(self, atoms, stages, backgroundNe, ne, dC=None, nPrev=None, dt=0.0, crsw=1.0):
        args, keep = abi.make_nr_args(atoms, stages, backgroundNe, ne, dC=dC, nPrev=nPrev, dt=dt, crsw=crsw)
        return self.lib.lwref_nr_post_update(self.h, C.byref(args))


class OracleContext(_Handle):
    """The plain-C restatement driving `prob`."""
    _lib = None

    @classmethod
    def lib_(cls):
        if cls._lib is None:
            lib = C.CDLL(ORACLE_LIB)
            lib.lworacle_create.restype = C.c_void_p
            lib.lworacle_create.argtypes = [C.POINTER(abi.lwhip_problem)]
            lib.lworacle_destroy.argtypes = [C.c_void_p]
            lib.lworacle_formal_sol_gamma_matrices.argtypes = [
                C.c_void_p, C.c_int, C.c_int, C.POINTER(abi.lwhip_iter_result), abi.f64p]
            lib.lworacle_formal_sol.argtypes = [C.c_void_p, C.c_int, C.c_int]
            lib.lworacle_compute_profiles.argtypes = [C.c_void_p]
            lib.lworacle_voigt_H.restype = C.c_double
            lib.lworacle_voigt_H.argtypes = [C.c_double, C.c_double]
            lib.lworacle_fs_partial.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, abi.f64p, abi.f64p]
            lib.lworacle_fs_finalise.argtypes = [C.c_void_p, abi.f64p]
            lib.lworacle_stat_equil.argtypes = [C.c_void_p, C.c_int]
            lib.lworacle_set_include_detailed.argtypes = [C.c_void_p, C.c_int]
            lib.lworacle_set_include_detailed.restype = None
            lib.lworacle_redistribute_prd.argtypes = [C.c_void_p, C.c_int, C.c_double, abi.i32p, abi.i32p,
                                                      abi.f64p, abi.i32p, abi.f64p, abi.i32p]
            lib.lworacle_prd_scatter_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, abi.f64p, abi.i32p]
            lib.lworacle_prd_rates_partial.argtypes = [C.c_void_p, C.c_int, C.c_int, abi.f64p, abi.f64p]
            lib.lworacle_prd_finalise.argtypes = [C.c_void_p, abi.f64p]
            lib.lworacle_time_dep_update.argtypes = [C.c_void_p, C.c_int, abi.f64p, C.c_double]
            lib.lworacle_nr_post_update.argtypes = [C.c_void_p, C.POINTER(abi.lwhip_nr_args)]
            lib.lworacle_GII.restype = C.c_double
            lib.lworacle_GII.argtypes = [C.c_double] * 3
            lib.lworacle_w2.argtypes = [C.c_double, abi.f64p]
            lib.lworacle_bezier3_coeffs.argtypes = [C.c_double, abi.f64p]
            lib.lworacle_cent_deriv.restype = C.c_double
            lib.lworacle_cent_deriv.argtypes = [C.c_double] * 5
            lib.lworacle_planck_nu.argtypes = [C.c_int, abi.f64p, C.c_double, abi.f64p]
            lib.lworacle_solve_lin_eq.argtypes = [C.c_int, abi.f64p, abi.f64p, C.c_int]
            lib.lworacle_formal_solver_ray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                       abi.f64p, abi.f64p, abi.f64p, abi.f64p]
            cls._lib = lib
        return cls._lib

    def __init__(self, prob):
        lib = self.lib_()

        def create(desc):
            h = lib.lworacle_create(C.byref(desc))
            if not h:
                raise RuntimeError('lworacle_create failed')
            return h
        super().__init__(lib, prob, create, lib.lworacle_destroy)

    def compute_profiles(self):
        assert self.lib.lworacle_compute_profiles(self.h) == 0

    def formal_sol_gamma_matrices(self, lambdaIterate=False, Nthreads=1, dJ=None):
        res = abi.lwhip_iter_result()
        dJp = dJ.ctypes.data_as(abi.f64p) if dJ is not None else C.cast(None, abi.f64p)
        st = self.lib.lworacle_formal_sol_gamma_matrices(self.h, int(lambdaIterate), Nthreads,
                                                         C.byref(res), dJp)
        assert st == 0
        return res.dJMax, res.dJMaxIdx

    def formal_sol(self, upOnly=True, Nthreads=1):
        assert self.lib.lworacle_formal_sol(self.h, int(upOnly), Nthreads) == 0

    def fs_partial_shard(self, laStart, laEnd, red, dJ, lambdaIterate=False):
        assert self.lib.lworacle_fs_partial(self.h, int(lambdaIterate), laStart, laEnd,
                                            red.ctypes.data_as(abi.f64p),
                                            dJ.ctypes.data_as(abi.f64p)) == 0

    def fs_finalise_shard(self, red):
        assert self.lib.lworacle_fs_finalise(self.h, red.ctypes.data_as(abi.f64p)) == 0

    def prd_scatter_shard(self, laStart, laEnd, dRho, dRhoFlatIdx):
        assert self.lib.lworacle_prd_scatter_shard(self.h, laStart, laEnd, dRho.ctypes.data_as(abi.f64p),
                                                   dRhoFlatIdx.ctypes.data_as(abi.i32p)) == 0

    def prd_rates_partial_shard(self, laStart, laEnd, red, dJ):
        assert self.lib.lworacle_prd_rates_partial(self.h, laStart, laEnd, red.ctypes.data_as(abi.f64p),
                                                   dJ.ctypes.data_as(abi.f64p)) == 0

    def prd_finalise_shard(self, red):
        assert self.lib.lworacle_prd_finalise(self.h, red.ctypes.data_as(abi.f64p)) == 0

    def stat_equil(self, atom=-1):
        return self.lib.lworacle_stat_equil(self.h, atom)

    def set_zplane(self, down=None, up=None):
        self._zplane = (down, up)
        ptr = lambda a: a.ctypes.data_as(abi.f64p) if a is not None else None
        self.lib.lworacle_set_zplane.argtypes = [C.c_void_p, abi.f64p, abi.f64p]
        assert self.lib.lworacle_set_zplane(self.h, ptr(down), ptr(up)) == 0

    def redistribute_prd(self, maxIter=3, tol=1e-2, include_detailed_atoms=False):
        self.lib.lworacle_set_include_detailed(self.h, int(include_detailed_atoms))
        return _redistribute(self.lib.lworacle_redistribute_prd, self.h, maxIter, tol)

    def build_hprd(self, include_detailed_atoms=False):
        """configure_hprd_coeffs restated (lworacle_hprd_build): the tables only, nothing is switched on."""
        self.lib.lworacle_hprd_build.restype = C.POINTER(abi.lwhip_hprd)
        self.lib.lworacle_hprd_build.argtypes = [C.c_void_p, C.c_int]
        self.lib.lworacle_hprd_free.argtypes = [C.POINTER(abi.lwhip_hprd)]
        self.lib.lworacle_hprd_free.restype = None
        ptr = self.lib.lworacle_hprd_build(self.h, int(include_detailed_atoms))
        assert ptr, 'lworacle_hprd_build: no PRD lines (or no vlosMu)'
        return HprdTables(ptr, self.prob, self.lib.lworacle_hprd_free)

    def set_hprd(self, tables):
        """Hybrid PRD with these tables from now on (None: plain PRD)."""
        self.lib.lworacle_set_hprd.argtypes = [C.c_void_p, C.POINTER(abi.lwhip_hprd)]
        self._hprd = tables
        assert self.lib.lworacle_set_hprd(self.h, tables.ptr if tables is not None else None) == 0

    def time_dep_update(self, atom, nOld, dt):
        import numpy as np
        nOld = np.ascontiguousarray(nOld, dtype=np.float64)
        return self.lib.lworacle_time_dep_update(self.h, atom, nOld.ctypes.data_as(abi.f64p), dt)

    def nr_post_update(self, atoms, stages, backgroundNe, ne, dC=None, nPrev=None, dt=0.0, crsw=1.0):
        args, keep = abi.make_nr_args(atoms, stages, backgroundNe, ne, dC=dC, nPrev=nPrev, dt=dt, crsw=crsw)
        return self.lib.lworacle_nr_post_update(self.h, C.byref(args))


class NgHandle:
    """Ng acceleration of one vector: kind='ref' is the reference's own struct Ng (Source/Ng.hpp) behind
    oracle/ref_driver.cpp, kind='oracle' the C restatement."""

    def __init__(self, kind, Norder=0, Nperiod=0, Ndelay=0, sol=None):
        import numpy as np
        self.np = np
        lib = RefContext.lib_() if kind == 'ref' else OracleContext.lib_()
        pre = 'lwref_ng_' if kind == 'ref' else 'lworacle_ng_'
        self._new, self._free = getattr(lib, pre + 'new'), getattr(lib, pre + 'free')
        self._acc, self._chg = getattr(lib, pre + 'accelerate'), getattr(lib, pre + 'max_change')
        self._new.restype = C.c_void_p
        self._new.argtypes = [C.c_int, C.c_int, C.c_int, abi.f64p, C.c_int]
        self._free.argtypes = [C.c_void_p]
        self._acc.argtypes = [C.c_void_p, abi.f64p, C.c_int]
        self._chg.argtypes = [C.c_void_p, abi.f64p, abi.i32p]
        if sol is not None:
            sol = np.ascontiguousarray(sol, dtype=np.float64).ravel()
            self.h = self._new(Norder, Nperiod, Ndelay, sol.ctypes.data_as(abi.f64p), sol.size)
        else:
            self.h = self._new(0, 0, 0, None, 0)

    def accelerate(self, sol):
        """In place on `sol` (a contiguous float64 array); returns whether it accelerated."""
        flat = sol.reshape(-1)
        assert flat.flags.c_contiguous and flat.dtype == self.np.float64
        r = self._acc(self.h, flat.ctypes.data_as(abi.f64p), flat.size)
        assert r >= 0
        return bool(r)

    def max_change(self):
        d, i = C.c_double(0.0), C.c_int32(0)
        self._chg(self.h, C.cast(C.byref(d), abi.f64p), C.cast(C.byref(i), abi.i32p))
        return d.value, i.value

    def __del__(self):
        try:
            if self.h:
                self._free(self.h)
                self.h = None
        except Exception:
            pass


class Ref2d:
    """The reference's 2D geometry (build_intersection_list) and solver (piecewise_besser_2d, interp_linear_2d)
    on an x-periodic grid, behind oracle/ref_driver.cpp."""

    def __init__(self, x, z, mux, muz, temperature, zLowerBc=abi.BC_THERMALISED, zUpperBc=abi.BC_ZERO, xLowerBc=None,
                 xUpperBc=None):
        """xLowerBc / xUpperBc: None (periodic) or model.Boundary with bcData [1, Nmu, Nz] (the primitive has one
        wavelength)."""
        import numpy as np
        self.np = np
        lib = RefContext.lib_()
        self.lib = lib
        ip = C.POINTER(abi.lwhip_intersection)
        lib.lwref_2d_create.restype = C.c_void_p
        lib.lwref_2d_create.argtypes = [abi.f64p, C.c_int, abi.f64p, C.c_int, abi.f64p, abi.f64p, C.c_int, C.c_int,
                                        C.c_int, abi.f64p]
        lib.lwref_2d_destroy.argtypes = [C.c_void_p]
        lib.lwref_2d_counts.argtypes = [C.c_void_p, abi.i32p, abi.i32p]
        lib.lwref_2d_stencils.argtypes = [C.c_void_p, ip, ip, abi.i32p, abi.i32p, ip]
        lib.lwref_2d_besser.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, abi.f64p, abi.f64p, abi.f64p, abi.f64p]
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self.x, self.z, self.mux, self.muz, self.T = f(x), f(z), f(mux), f(muz), f(temperature)
        self.bc = (zLowerBc, zUpperBc)
        p = lambda a: a.ctypes.data_as(abi.f64p)
        self.xbc = (xLowerBc, xUpperBc)
        if xLowerBc is None:
            self.h = lib.lwref_2d_create(p(self.x), self.x.size, p(self.z), self.z.size, p(self.mux), p(self.muz),
                                         self.mux.size, zLowerBc, zUpperBc, p(self.T))
        else:
            bp = C.POINTER(abi.lwhip_boundary)
            lib.lwref_2d_create_ex.restype = C.c_void_p
            lib.lwref_2d_create_ex.argtypes = lib.lwref_2d_create.argtypes + [bp, bp]
            self._keep = []

            def flat(b):
                out = abi.lwhip_boundary()
                out.type = abi.BC_CALLABLE
                b.idxs = np.ascontiguousarray(b.idxs, dtype=np.int32)
                b.bcData = np.ascontiguousarray(b.bcData, dtype=np.float64)
                out.Nmu = b.bcData.shape[1]
                out.idxs = b.idxs.ctypes.data_as(abi.i32p)
                out.bcData = b.bcData.ctypes.data_as(abi.f64p)
                self._keep.append(out)
                return C.byref(out)
            self.h = lib.lwref_2d_create_ex(p(self.x), self.x.size, p(self.z), self.z.size, p(self.mux), p(self.muz),
                                            self.mux.size, zLowerBc, zUpperBc, p(self.T), flat(xLowerBc), flat(xUpperBc))

    def grid(self):
        from lightweaver_amd.grid2d import INTERSECTION_DTYPE, Grid2d
        np = self.np
        nl, ns = C.c_int32(0), C.c_int32(0)
        self.lib.lwref_2d_counts(self.h, C.cast(C.byref(nl), abi.i32p), C.cast(C.byref(ns), abi.i32p))
        shape = (self.mux.size, 2, self.z.size, self.x.size)
        uw = np.zeros(shape, dtype=INTERSECTION_DTYPE)
        dw = np.zeros(shape, dtype=INTERSECTION_DTYPE)
        li = np.zeros(shape, dtype=np.int32)
        off = np.zeros(nl.value + 1, dtype=np.int32)
        sub = np.zeros(max(ns.value, 1), dtype=INTERSECTION_DTYPE)
        ip = C.POINTER(abi.lwhip_intersection)
        self.lib.lwref_2d_stencils(self.h, uw.ctypes.data_as(ip), dw.ctypes.data_as(ip), li.ctypes.data_as(abi.i32p),
                                   off.ctypes.data_as(abi.i32p), sub.ctypes.data_as(ip))
        return Grid2d(x=self.x, z=self.z, mux=self.mux, muz=self.muz, temperature=self.T.reshape(self.z.size, -1),
                      uw=uw, dw=dw, longCharIdx=li, substepOff=off, substeps=sub[:ns.value],
                      zLowerBc=self.bc[0], zUpperBc=self.bc[1], xLowerBc=self.xbc[0], xUpperBc=self.xbc[1])

    def besser(self, mu, toObs, wav, chi, S):
        np = self.np
        chi, S = np.ascontiguousarray(chi, dtype=np.float64), np.ascontiguousarray(S, dtype=np.float64)
        I, Psi = np.zeros_like(chi), np.zeros_like(chi)
        p = lambda a: a.ctypes.data_as(abi.f64p)
        assert self.lib.lwref_2d_besser(self.h, mu, int(toObs), wav, p(chi), p(S), p(I), p(Psi)) == 0
        return I, Psi

    def __del__(self):
        try:
            if self.h:
                self.lib.lwref_2d_destroy(self.h)
                self.h = None
        except Exception:
            pass


def oracle_2d_besser(grid, mu, toObs, wav, chi, S):
    """The C restatement of piecewise_besser_2d (oracle/lw_oracle_2d.inc) on a lightweaver_amd.grid2d.Grid2d."""
    import numpy as np
    lib = OracleContext.lib_()
    lib.lworacle_2d_besser.argtypes = [C.POINTER(abi.lwhip_grid2d), C.c_int, C.c_int, C.c_double, abi.f64p, abi.f64p,
                                       abi.f64p, abi.f64p]
    chi, S = np.ascontiguousarray(chi, dtype=np.float64), np.ascontiguousarray(S, dtype=np.float64)
    I, Psi = np.zeros_like(chi), np.zeros_like(chi)
    g = grid.descriptor()
    p = lambda a: a.ctypes.data_as(abi.f64p)
    assert lib.lworacle_2d_besser(C.byref(g), mu, int(toObs), wav, p(chi), p(S), p(I), p(Psi)) == 0
    return I, Psi

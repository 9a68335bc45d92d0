"""ctypes mirror of include/lwhip.h (the C-ABI drop-in boundary).

Field order and types must match the header exactly; tests/test_abi.py checks sizes and
offsets against a C program compiled from the header.
"""
import ctypes as C

ABI_VERSION = 4
AXIS_NONE, AXIS_X, AXIS_Z = 0, 1, 2

# enums (include/lwhip.h)
LINE, CONTINUUM = 0, 1
BC_UNINITIALISED, BC_ZERO, BC_THERMALISED, BC_PERIODIC, BC_CALLABLE = range(5)
FS_LINEAR_1D, FS_BESSER_1D, FS_BEZIER3_1D = range(3)
FS_NAMES = {
    'piecewise_linear_1d': FS_LINEAR_1D,
    'piecewise_besser_1d': FS_BESSER_1D,
    'piecewise_bezier3_1d': FS_BEZIER3_1D,
}
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_DEVICE, ERR_SINGULAR, ERR_BUSY = range(6)

ATMOS = 1 << 0
BACKGROUND = 1 << 1
PROFILES = 1 << 2
POPS = 1 << 3
NSTAR = 1 << 4
J = 1 << 5
GAMMA = 1 << 6
BC = 1 << 7
RHOPRD = 1 << 8
I = 1 << 9
RATES = 1 << 10
DEPTHDATA = 1 << 11
COLLISIONS = 1 << 12
ALL_INPUTS = ATMOS | BACKGROUND | PROFILES | POPS | NSTAR | J | GAMMA | BC | RHOPRD | COLLISIONS
ALL_OUTPUTS = J | GAMMA | I | RATES

f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)


class lwhip_transition(C.Structure):
    _fields_ = [
        ('type', C.c_int32), ('i', C.c_int32), ('j', C.c_int32),
        ('Nblue', C.c_int32), ('Nred', C.c_int32), ('prd', C.c_int32),
        ('Aji', C.c_double), ('Bji', C.c_double), ('Bij', C.c_double),
        ('lambda0', C.c_double), ('dopplerWidth', C.c_double),
        ('wavelength', f64p), ('alpha', f64p), ('phi', f64p), ('wphi', f64p),
        ('aDamp', f64p), ('rhoPrd', f64p), ('Rij', f64p), ('Rji', f64p), ('Qelast', f64p),
    ]


class lwhip_atom(C.Structure):
    _fields_ = [
        ('Nlevel', C.c_int32), ('Ntrans', C.c_int32), ('detailed', C.c_int32), ('_pad', C.c_int32),
        ('n', f64p), ('nStar', f64p), ('nTotal', f64p), ('vBroad', f64p),
        ('Gamma', f64p), ('C', f64p),
        ('trans', C.POINTER(lwhip_transition)),
    ]


class lwhip_boundary(C.Structure):
    _fields_ = [
        ('type', C.c_int32), ('Nmu', C.c_int32),
        ('idxs', i32p), ('bcData', f64p),
    ]


class lwhip_problem(C.Structure):
    _fields_ = [
        ('abiVersion', C.c_int32), ('Nspace', C.c_int32), ('Nrays', C.c_int32),
        ('Nlambda', C.c_int32), ('Natom', C.c_int32), ('formalSolver', C.c_int32),
        ('storeDepthData', C.c_int32), ('_pad', C.c_int32),
        ('height', f64p), ('temperature', f64p), ('vlosMu', f64p), ('muz', f64p),
        ('wmu', f64p), ('wavelength', f64p),
        ('zLowerBc', lwhip_boundary), ('zUpperBc', lwhip_boundary),
        ('bgChi', f64p), ('bgEta', f64p), ('bgSca', f64p),
        ('J', f64p), ('I', f64p),
        ('depthChi', f64p), ('depthEta', f64p), ('depthI', f64p),
        ('atoms', C.POINTER(lwhip_atom)),
        ('grid2d', C.c_void_p),
    ]


OPT_PRD_DETAILED = 1 << 16   # lwhip_options.flags: the PRD calls include the detailed atoms' PRD lines
OPT_DETERMINISTIC = 1 << 17  # ... Gamma / rates summed in a fixed order (bit-reproducible)


class lwhip_rho_coeff(C.Structure):
    _fields_ = [('i0', C.c_int32), ('i1', C.c_int32), ('frac', C.c_double)]


class lwhip_j_coeff(C.Structure):
    _fields_ = [('frac', C.c_double), ('idx', C.c_int32), ('_pad', C.c_int32)]


class lwhip_hprd(C.Structure):
    """Hybrid-PRD tables (what configure_hprd_coeffs leaves in Spectrum and the PRD lines), flattened."""
    _fields_ = [
        ('NprdLambda', C.c_int32), ('NhPrd', C.c_int32), ('Nlines', C.c_int32), ('_pad', C.c_int32),
        ('prdIdxs', i32p), ('hPrdIdxs', i32p), ('JRest', f64p),
        ('jCoeffOff', C.POINTER(C.c_int64)), ('jCoeffs', C.POINTER(lwhip_j_coeff)),
        ('lineAtom', i32p), ('lineTrans', i32p),
        ('rhoCoeffs', C.POINTER(C.POINTER(lwhip_rho_coeff))),
    ]


class lwhip_options(C.Structure):
    _fields_ = [
        ('device', C.c_int32), ('laStart', C.c_int32), ('laEnd', C.c_int32), ('flags', C.c_int32),
        ('worldSize', C.c_int32), ('worldRank', C.c_int32),
        ('stream', C.c_void_p),
        ('hprd', C.POINTER(lwhip_hprd)),
    ]


class lwhip_prd_result(C.Structure):
    _fields_ = [('NprdSubIter', C.c_int32), ('Nprd', C.c_int32), ('dRho', f64p), ('dRhoMaxIdx', i32p),
                ('dJPrdMax', f64p), ('dJPrdMaxIdx', i32p)]


class lwhip_intersection(C.Structure):
    _fields_ = [('axis', C.c_int32), ('_pad', C.c_int32), ('fracZ', C.c_double), ('fracX', C.c_double),
                ('distance', C.c_double)]


class lwhip_grid2d(C.Structure):
    _fields_ = [('Nx', C.c_int32), ('Nz', C.c_int32), ('Nrays', C.c_int32), ('periodic', C.c_int32),
                ('zLowerBc', C.c_int32), ('zUpperBc', C.c_int32), ('NlongChar', C.c_int32), ('_pad', C.c_int32),
                ('x', f64p), ('z', f64p), ('mux', f64p), ('muz', f64p), ('temperature', f64p),
                ('uw', C.POINTER(lwhip_intersection)), ('dw', C.POINTER(lwhip_intersection)),
                ('longCharIdx', i32p), ('substepOff', i32p), ('substeps', C.POINTER(lwhip_intersection)),
                ('xLowerBc', C.POINTER(lwhip_boundary)), ('xUpperBc', C.POINTER(lwhip_boundary))]


class lwhip_nr_args(C.Structure):
    _fields_ = [('Natoms', C.c_int32), ('_pad', C.c_int32), ('atoms', i32p),
                ('stages', C.POINTER(f64p)), ('dC', C.POINTER(f64p)), ('nPrev', C.POINTER(f64p)),
                ('backgroundNe', f64p), ('ne', f64p), ('dt', C.c_double), ('crsw', C.c_double)]


class lwhip_iter_result(C.Structure):
    _fields_ = [('updatedJ', C.c_int32), ('dJMaxIdx', C.c_int32), ('dJMax', C.c_double)]


_RAW = {}


def raw_view(struct):
    """The same memory as `struct` seen through a mirror type whose pointer fields are c_void_p: they take plain integer
    addresses (`arr.ctypes.data`) -- filling a typed POINTER field costs a ctypes.cast per array, twice the time, and a column
    batch fills a quarter of a million of them."""
    cls = type(struct)
    raw = _RAW.get(cls)
    if raw is None:
        fields = []
        for name, tp in cls._fields_:
            is_ptr = isinstance(tp, type) and issubclass(tp, C._Pointer)
            fields.append((name, C.c_void_p if is_ptr else tp))
        raw = type(cls.__name__ + '_raw', (C.Structure,), {'_fields_': fields})
        assert C.sizeof(raw) == C.sizeof(cls)
        _RAW[cls] = raw
    return raw.from_buffer(struct)


def addr(a):
    """Address of a numpy array's data, or None."""
    return None if a is None else a.ctypes.data


# every symbol include/lwhip.h declares: (name, restype, argtypes)
ctx_p = C.c_void_p
SYMBOLS = [
    ('lwhip_last_error', C.c_char_p, []),
    ('lwhip_abi_version', C.c_int, []),
    ('lwhip_device_count', C.c_int, []),
    ('lwhip_create', C.c_int, [C.POINTER(lwhip_problem), C.POINTER(lwhip_options), C.POINTER(ctx_p)]),
    ('lwhip_create_like', C.c_int, [C.POINTER(lwhip_problem), C.POINTER(lwhip_options), ctx_p, C.POINTER(ctx_p)]),
    ('lwhip_destroy', C.c_int, [ctx_p]),
    ('lwhip_upload', C.c_int, [ctx_p, C.c_uint32]),
    ('lwhip_download', C.c_int, [ctx_p, C.c_uint32]),
    ('lwhip_gamma_prefill_from_C', C.c_int, [ctx_p, C.c_double]),
    ('lwhip_iterate_from_C', C.c_int, [ctx_p, C.c_int, C.c_double, C.POINTER(lwhip_iter_result)]),
    ('lwhip_formal_sol_gamma_matrices', C.c_int, [ctx_p, C.c_int, C.POINTER(lwhip_iter_result)]),
    ('lwhip_fs_partial', C.c_int, [ctx_p, C.c_int]),
    ('lwhip_fs_finalise', C.c_int, [ctx_p, C.POINTER(lwhip_iter_result)]),
    ('lwhip_reduce_buffer', C.c_int, [ctx_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                     C.POINTER(C.c_size_t)]),
    ('lwhip_peer_window', C.c_int, [ctx_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ('lwhip_peer_export', C.c_int, [ctx_p, C.c_void_p]),
    ('lwhip_peer_attach', C.c_int, [ctx_p, C.c_void_p]),
    ('lwhip_peer_attach_pointers', C.c_int, [ctx_p, C.POINTER(C.c_void_p)]),
    ('lwhip_peer_detach', C.c_int, [ctx_p]),
    ('lwhip_peer_selftest', C.c_int, [ctx_p, C.c_int, C.POINTER(C.c_int32)]),
    ('lwhip_map_host_J', C.c_int, [ctx_p, C.c_int]),
    ('lwhip_fingerprint_J', C.c_int, [ctx_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    ('lwhip_host_fingerprint', C.c_uint64, [f64p, C.c_size_t]),
    ('lwhip_j_snapshot', C.c_int, [ctx_p]),
    ('lwhip_j_restore', C.c_int, [ctx_p]),
    ('lwhip_debug_check_quarantine', C.c_long, []),
    ('lwhip_debug_streams_created', C.c_long, []),
    ('lwhip_formal_sol', C.c_int, [ctx_p, C.c_int]),
    ('lwhip_stat_equil', C.c_int, [ctx_p, C.c_int]),
    ('lwhip_stat_equil_async', C.c_int, [ctx_p, C.c_int]),
    ('lwhip_stat_equil_report', C.c_int, [ctx_p, C.c_int, f64p, i32p]),
    ('lwhip_check_status', C.c_int, [ctx_p]),
    ('lwhip_compute_profiles', C.c_int, [ctx_p]),
    ('lwhip_formal_solver_2d', C.c_int, [C.c_int, C.POINTER(lwhip_grid2d), C.c_double, C.c_int, i32p, f64p, f64p,
                                          f64p, f64p]),
    ('lwhip_batch_create', C.c_int, [C.POINTER(ctx_p), C.c_int, C.POINTER(C.c_void_p)]),
    ('lwhip_batch_destroy', None, [C.c_void_p]),
    ('lwhip_batch_formal_sol_gamma_matrices', C.c_int, [C.c_void_p, C.c_int, C.c_double, C.POINTER(lwhip_iter_result)]),
    ('lwhip_batch_stat_equil', C.c_int, [C.c_void_p]),
    ('lwhip_batch_compute_profiles', C.c_int, [C.c_void_p]),
    ('lwhip_build_intersections', C.c_int, [C.POINTER(lwhip_grid2d), C.POINTER(lwhip_intersection),
                                             C.POINTER(lwhip_intersection), i32p, i32p, C.c_int32,
                                             C.POINTER(lwhip_intersection), C.c_int64, C.POINTER(C.c_int32),
                                             C.POINTER(C.c_int64)]),
    ('lwhip_redistribute_prd', C.c_int, [ctx_p, C.c_int, C.c_double, C.POINTER(lwhip_prd_result)]),
    ('lwhip_prd_pack', C.c_int, [ctx_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ('lwhip_prd_partial', C.c_int, [ctx_p]),
    ('lwhip_prd_finalise', C.c_int, [ctx_p, f64p, i32p, f64p, i32p]),
    ('lwhip_ng_configure', C.c_int, [ctx_p, C.c_int, C.c_int, C.c_int]),
    ('lwhip_ng_accelerate', C.c_int, [ctx_p, i32p, f64p, i32p]),
    ('lwhip_time_dep_update', C.c_int, [ctx_p, C.c_int, f64p, C.c_double]),
    ('lwhip_nr_post_update', C.c_int, [ctx_p, C.POINTER(lwhip_nr_args)]),
    ('lwhip_set_depth_range', C.c_int, [ctx_p, C.c_int, C.c_int]),
    ('lwhip_set_djmax_index_mode', C.c_int, [ctx_p, C.c_int]),
    ('lwhip_synchronize', C.c_int, [ctx_p]),
    ('lwhip_set_stream', C.c_int, [ctx_p, C.c_void_p]),
    ('lwhip_set_zplane_outputs', C.c_int, [ctx_p, f64p, f64p]),
    ('lwhip_profile_enable', C.c_int, [ctx_p, C.c_int]),
    ('lwhip_sweep_time', C.c_int, [ctx_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    ('lwhip_algorithmic_bytes', C.c_int, [ctx_p, C.POINTER(C.c_double)]),
    ('lwhip_sweep_kind', C.c_int, [ctx_p]),
]


def bind(lib):
    """Attach restype/argtypes for every declared symbol; raises AttributeError if one is missing."""
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def make_nr_args(atomIdxs, stages, backgroundNe, ne, dC=None, nPrev=None, dt=0.0, crsw=1.0):
    """lwhip_nr_args over numpy arrays; returns (struct, keepalive)."""
    import numpy as np
    n = len(atomIdxs)
    idx = np.ascontiguousarray(atomIdxs, dtype=np.int32)

    def ptr_array(arrs):
        if arrs is None:
            return None, None
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in arrs]
        arr = (f64p * n)(*[k.ctypes.data_as(f64p) for k in keep])
        return arr, keep
    st, k1 = ptr_array(stages)
    dc, k2 = ptr_array(dC)
    pv, k3 = ptr_array(nPrev)
    bg = np.ascontiguousarray(backgroundNe, dtype=np.float64)
    assert ne.dtype == np.float64 and ne.flags.c_contiguous
    args = lwhip_nr_args(n, 0, idx.ctypes.data_as(i32p), st, dc, pv, bg.ctypes.data_as(f64p),
                         ne.ctypes.data_as(f64p), float(dt), float(crsw))
    return args, (idx, st, dc, pv, k1, k2, k3, bg, ne)

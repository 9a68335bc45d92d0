"""Python host mirror of the reference's `LwContext` entry points for the hot path, over the C ABI.

Names, argument meaning and error behaviour follow Source/LwMiddleLayer.pyx:
    Context.formal_sol_gamma_matrices  <- LwContext.formal_sol_gamma_matrices  (:3152-3210)
    Context.formal_sol                 <- LwContext.formal_sol                  (:3212-3242)
    Context.stat_equil                 <- LwContext.stat_equil                  (:3461-3531)
    Context.prd_redistribute           <- LwContext.prd_redistribute            (:3647-3684)
    ExplodingMatrixError               <- lightweaver.utils.ExplodingMatrixError (raised :3509-3514)

The HIP library is mandatory: if it cannot be loaded, or no gfx950 device is visible, construction
raises -- there is no CPU path behind this class.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _abi as abi
from .model import Problem

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('LWHIP_LIB') or os.path.join(_PKG, 'liblwhip.so')  # LWHIP_LIB: an experiment build of the same library
_lib = None


class LwHipError(RuntimeError):
    pass


class ExplodingMatrixError(Exception):
    """Singular Gamma in stat_equil (lightweaver.utils.ExplodingMatrixError)."""


def load_library(path: Optional[str] = None):
    """dlopen liblwhip.so and bind every symbol include/lwhip.h declares."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise LwHipError(f'{path} is missing: build it with `python -m lightweaver_amd.build` '
                         '(hipcc --offload-arch=gfx950); there is no CPU fallback')
    lib = abi.bind(C.CDLL(path))
    if lib.lwhip_abi_version() != abi.ABI_VERSION:
        raise LwHipError('liblwhip.so ABI version does not match lightweaver_amd/_abi.py')
    _lib = lib
    return lib


def _check(lib, st, what):
    if st == abi.OK:
        return
    msg = lib.lwhip_last_error().decode(errors='replace')
    if st == abi.ERR_SINGULAR:
        raise ExplodingMatrixError(msg or 'Singular Matrix')
    raise LwHipError(f'{what} failed ({st}): {msg}')


@dataclass
class IterationUpdate:
    """The members of lightweaver.iteration_update.IterationUpdate this path fills."""
    updatedJ: bool = False
    dJMax: float = 0.0
    dJMaxIdx: int = 0
    crsw: float = 1.0
    updatedPops: bool = False
    dPops: Optional[list] = None          # per active atom: max |dn / n| of the last population update
    dPopsMaxIdx: Optional[list] = None    # and its flattened [level, depth] index
    ngAccelerated: bool = False
    updatedRho: bool = False
    NprdSubIter: int = 0
    dRho: Optional[np.ndarray] = None         # [NprdSubIter, Nprd]
    dRhoMaxIdx: Optional[np.ndarray] = None
    dJPrdMax: Optional[np.ndarray] = None     # [NprdSubIter]
    dJPrdMaxIdx: Optional[np.ndarray] = None


class Context:
    """Device-resident context for one `Problem` (optionally one wavelength shard of it)."""

    def __init__(self, prob: Problem, device: int = 0, laStart: int = 0, laEnd: int = 0,
                 stream: Optional[int] = None, lib_path: Optional[str] = None,
                 worldSize: int = 1, worldRank: int = 0, batchHint: int = 0, prdDetailed: bool = False, hprd=None,
                 deterministic: bool = False, like: Optional['Context'] = None):
        """like: a Context of a problem of the same structure (model atoms, wavelength grid, rays, solver, shard) whose
        structure tables this one borrows instead of building its own (lwhip_create_like: the columns of a batch); the
        library compares the structures and this constructor falls back to tables of its own when they differ.
        hprd: hybrid PRD -- the tables configure_hprd_coeffs leaves in the reference's Context
        (Source/Prd.cpp:697-946), as a ctypes pointer to lwhip_hprd or any object with such a `.ptr` (what
        LwContext(hprd=True) sets up, Source/LwMiddleLayer.pyx:2822-2826); borrowed for the life of the context.
        JRest comes back with download(J)."""
        self.lib = load_library(lib_path)
        self.prob = prob
        self._desc = prob.descriptor()
        opts = abi.lwhip_options()
        opts.device = device
        opts.laStart, opts.laEnd = laStart, laEnd
        opts.stream = stream
        opts.worldSize, opts.worldRank = worldSize, worldRank
        # prdDetailed: ExtraParams 'include_detailed_atoms' of the reference's PRD calls (LwMiddleLayer.pyx:3678-3680)
        opts.flags = (min(max(int(batchHint), 0), 0xffff) | (abi.OPT_PRD_DETAILED if prdDetailed else 0)
                      | (abi.OPT_DETERMINISTIC if deterministic else 0))
        self.prdDetailed = bool(prdDetailed)
        self._hprd = hprd   # keep the tables alive
        if hprd is not None:
            opts.hprd = getattr(hprd, 'ptr', hprd)
        h = C.c_void_p()
        self._like = None      # (kept alive: its tables are in use)
        self._pending = False  # close() asked for while borrowers were alive
        st = abi.ERR_INVALID
        if like is not None and getattr(like, '_h', None):
            st = self.lib.lwhip_create_like(C.byref(self._desc), C.byref(opts), like._h, C.byref(h))
            if st == abi.OK:
                while getattr(like, '_like', None) is not None:   # (a borrower's tables are its owner's)
                    like = like._like
                self._like = like
        if st != abi.OK:
            st = self.lib.lwhip_create(C.byref(self._desc), C.byref(opts), C.byref(h))
        _check(self.lib, st, 'lwhip_create')
        self._h = h
        self._res = abi.lwhip_iter_result()
        self._res_ref = C.byref(self._res)
        self.laStart = laStart
        self.laEnd = laEnd if laEnd else prob.Nlambda
        self.crsw = 1.0
        self._ng = None
        self.peers_attached = False

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, '_h', None):
            st = self.lib.lwhip_destroy(self._h)
            if st == abi.ERR_BUSY:
                self._pending = True   # contexts made with like=self still use its tables: the last of them closes it
                return
            if st != abi.OK:
                _check(self.lib, st, 'lwhip_destroy')
            self._h = None
            like, self._like = getattr(self, '_like', None), None
            if like is not None and like._pending:
                like._pending = False
                like.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- data movement ------------------------------------------------------------------------------
    def upload(self, mask=abi.ALL_INPUTS):
        _check(self.lib, self.lib.lwhip_upload(self._h, mask), 'lwhip_upload')

    def download(self, mask=abi.ALL_OUTPUTS):
        _check(self.lib, self.lib.lwhip_download(self._h, mask), 'lwhip_download')

    def set_zplane(self, down=None, up=None):
        """ExtraParams ZPlaneDecomposition (Source/SimdFullIterationTemplates.hpp:253-281, 351-384): every formal solution
        from now on also records I of the down rays in plane Nz - 2 (`down`) and of the up rays in plane 1 (`up`),
        float64 arrays [Nlambda, Nrays] (1D) / [Nlambda, Nrays, Nx] (2D) filled by download(I); None turns one off."""
        self._zplane = (down, up)
        ptr = lambda a: a.ctypes.data_as(abi.f64p) if a is not None else None
        _check(self.lib, self.lib.lwhip_set_zplane_outputs(self._h, ptr(down), ptr(up)), 'lwhip_set_zplane_outputs')

    def set_depth_range(self, spaceStart=-1, spaceEnd=-1):
        """spaceStart / spaceEnd of the reference's stat_eq / time_dep_update / nr_post_update
        (Source/UpdatePopulations.cpp:7-47, 120-151, 294-394): the depth points [spaceStart, spaceEnd) the following
        population updates solve; the others keep their values.  (-1, -1): the whole atmosphere."""
        _check(self.lib, self.lib.lwhip_set_depth_range(self._h, int(spaceStart), int(spaceEnd)), 'lwhip_set_depth_range')

    def set_djmax_index_mode(self, mode=0):
        """0: dJMaxIdx = the first wavelength that attains dJMax (the reference's threaded schemes); 1: the index the
        reference's single-thread loop records (Source/SimdFullIterationTemplates.hpp:627: the last wavelength whose dJ is
        below the running maximum)."""
        _check(self.lib, self.lib.lwhip_set_djmax_index_mode(self._h, int(mode)), 'lwhip_set_djmax_index_mode')

    def synchronize(self):
        _check(self.lib, self.lib.lwhip_synchronize(self._h), 'lwhip_synchronize')

    def set_stream(self, stream: Optional[int]):
        _check(self.lib, self.lib.lwhip_set_stream(self._h, stream), 'lwhip_set_stream')

    # -- the hot path ---------------------------------------------------------------------------------
    def formal_sol_gamma_matrices(self, fixCollisionalRates=True, lambdaIterate=False,
                                  sync_host=True, deviceResident=False) -> IterationUpdate:
        """One iteration.  Default (drop-in) mode mirrors the reference exactly: the host arrays
        are authoritative -- Gamma is pre-filled with crsw*C on the host, inputs are uploaded, the
        kernels run, outputs are downloaded.  With deviceResident=True nothing crosses PCIe: the
        pre-fill happens on the device from the resident C and results stay in HBM (call
        `download()` when the host needs them).  Collisional rates are an input here
        (`fixCollisionalRates` is accepted for signature parity; the harness never recomputes C)."""
        lib = self.lib
        if deviceResident and sync_host:
            # (one ABI call and a reused result record: every microsecond between the previous result and this launch is a
            # microsecond the device idles)
            res = self._res
            st = lib.lwhip_iterate_from_C(self._h, int(lambdaIterate), self.crsw, self._res_ref)
            if st != abi.OK:
                _check(lib, st, 'lwhip_iterate_from_C')
            return IterationUpdate(updatedJ=True, dJMax=res.dJMax, dJMaxIdx=res.dJMaxIdx, crsw=self.crsw)
        if deviceResident:
            _check(lib, lib.lwhip_gamma_prefill_from_C(self._h, self.crsw), 'lwhip_gamma_prefill_from_C')
        else:
            self.prob.gamma_prefill(self.crsw)
            self.upload(abi.GAMMA | abi.J | abi.POPS | abi.BC | abi.RHOPRD)
        res = abi.lwhip_iter_result()
        st = lib.lwhip_formal_sol_gamma_matrices(self._h, int(lambdaIterate),
                                                 C.byref(res) if sync_host else None)
        _check(lib, st, 'lwhip_formal_sol_gamma_matrices')
        if not deviceResident:
            self.download(abi.ALL_OUTPUTS | (abi.DEPTHDATA if self.prob.storeDepthData else 0))
        if not sync_host:
            # nothing was read back: a convergence test must not mistake "not known yet" for dJMax = 0
            return IterationUpdate(updatedJ=True, dJMax=float('nan'), dJMaxIdx=-1, crsw=self.crsw)
        return IterationUpdate(updatedJ=True, dJMax=res.dJMax, dJMaxIdx=res.dJMaxIdx, crsw=self.crsw)

    def formal_sol(self, upOnly=True, deviceResident=False):
        if not deviceResident:
            self.upload(abi.J | abi.POPS | abi.BC | abi.RHOPRD)
        _check(self.lib, self.lib.lwhip_formal_sol(self._h, int(upOnly)), 'lwhip_formal_sol')
        if not deviceResident:
            self.download(abi.I)

    def stat_equil(self, deviceResident=False, atom=-1, sync_host=True):
        """Statistical equilibrium for every active atom; raises ExplodingMatrixError on a
        singular matrix as the reference does.  With deviceResident and sync_host=False the solve is
        only queued; `check_status()` later waits and raises."""
        if not deviceResident:
            self.upload(abi.GAMMA | abi.POPS)
        if deviceResident and not sync_host:
            _check(self.lib, self.lib.lwhip_stat_equil_async(self._h, atom), 'lwhip_stat_equil_async')
            return IterationUpdate(updatedPops=True)
        nAct = sum(1 for a in self.prob.atoms if not a.detailed)
        dPops = np.zeros(max(nAct, 1))
        dIdx = np.zeros(max(nAct, 1), dtype=np.int32)
        accelerated = False
        if self._ng is not None:
            # LwContext.stat_equil -> rel_diff_ng_accelerate (Source/LwMiddleLayer.pyx:3318-3346)
            _check(self.lib, self.lib.lwhip_stat_equil(self._h, atom), 'lwhip_stat_equil')
            acc = np.zeros(max(nAct, 1), dtype=np.int32)
            _check(self.lib, self.lib.lwhip_ng_accelerate(self._h, acc.ctypes.data_as(abi.i32p),
                                                          dPops.ctypes.data_as(abi.f64p), dIdx.ctypes.data_as(abi.i32p)),
                   'lwhip_ng_accelerate')
            accelerated = bool(acc[:nAct].any())
        else:
            _check(self.lib, self.lib.lwhip_stat_equil_report(self._h, atom, dPops.ctypes.data_as(abi.f64p),
                                                              dIdx.ctypes.data_as(abi.i32p)), 'lwhip_stat_equil')
        if not deviceResident:
            self.download(abi.POPS)
        return IterationUpdate(updatedPops=True, dPops=list(dPops[:nAct]), dPopsMaxIdx=list(dIdx[:nAct]),
                               ngAccelerated=accelerated)

    def configure_ng(self, Norder=0, Nperiod=0, Ndelay=0):
        """Ng acceleration of the populations (NgOptions of lightweaver.Context, Source/LwMiddleLayer.pyx:2900;
        struct Ng, Source/Ng.hpp), kept on the device: call once the initial populations are uploaded."""
        _check(self.lib, self.lib.lwhip_ng_configure(self._h, int(Norder), int(Nperiod), int(Ndelay)),
               'lwhip_ng_configure')
        self._ng = (Norder, Nperiod, Ndelay)

    def prd_redistribute(self, maxIter=3, tol=1e-2, deviceResident=False, include_detailed_atoms=None) -> IterationUpdate:
        """PRD sub-iterations, LwContext.prd_redistribute (Source/LwMiddleLayer.pyx:3647-3684): updates rhoPrd of
        every PRD line of the active atoms -- and, on a Context created with prdDetailed=True (ExtraParams
        include_detailed_atoms, PrdTemplates.hpp:25-29, 190-215), of the detailed atoms --, J over the PRD wavelengths
        and those lines' Rij/Rji.  The host arrays are uploaded/downloaded around the call unless deviceResident.
        (`include_detailed_atoms` is accepted for signature parity with the test doubles and must match the Context.)"""
        if include_detailed_atoms is not None and bool(include_detailed_atoms) != self.prdDetailed:
            raise LwHipError('include_detailed_atoms is a property of the device context: create it with '
                             f'prdDetailed={bool(include_detailed_atoms)}')
        if not deviceResident:
            self.upload(abi.J | abi.POPS | abi.BC | abi.RHOPRD | abi.RATES | abi.COLLISIONS)
        Nprd = self.Nprd
        n = max(maxIter, 1)
        dRho = np.zeros(n * max(Nprd, 1))
        dRhoIdx = np.zeros(n * max(Nprd, 1), dtype=np.int32)
        dJ = np.zeros(n)
        dJIdx = np.zeros(n, dtype=np.int32)
        res = abi.lwhip_prd_result(0, 0, dRho.ctypes.data_as(abi.f64p), dRhoIdx.ctypes.data_as(abi.i32p),
                                   dJ.ctypes.data_as(abi.f64p), dJIdx.ctypes.data_as(abi.i32p))
        _check(self.lib, self.lib.lwhip_redistribute_prd(self._h, int(maxIter), float(tol), C.byref(res)),
               'lwhip_redistribute_prd')
        if not deviceResident:
            self.download(abi.J | abi.I | abi.RATES | abi.RHOPRD)  # (the rates pass rewrites spect.I of its wavelengths too)
        it, m = res.NprdSubIter, res.Nprd
        return IterationUpdate(updatedRho=it > 0, updatedJ=it > 0, NprdSubIter=it,
                               dRho=dRho[:it * m].reshape(it, m), dRhoMaxIdx=dRhoIdx[:it * m].reshape(it, m),
                               dJPrdMax=dJ[:it], dJPrdMaxIdx=dJIdx[:it],
                               dJMax=float(dJ[it - 1]) if it else 0.0,
                               dJMaxIdx=int(dJIdx[it - 1]) if it else 0)

    redistribute_prd = prd_redistribute

    # -- PRD sub-iteration split around its two collectives (wavelength shards) ------------------------
    def prd_pack(self):
        """(device pointer, count) of the PRD J gather buffer, this shard's rows filled."""
        ptr, n = C.c_void_p(), C.c_size_t()
        _check(self.lib, self.lib.lwhip_prd_pack(self._h, C.byref(ptr), C.byref(n)), 'lwhip_prd_pack')
        return ptr.value, n.value

    def prd_partial(self):
        _check(self.lib, self.lib.lwhip_prd_partial(self._h), 'lwhip_prd_partial')

    def prd_finalise(self):
        """-> (dRho [Nprd], dRhoMaxIdx [Nprd], dJMax, dJMaxIdx) of the sub-iteration."""
        Nprd = self.Nprd
        dRho = np.zeros(max(Nprd, 1))
        idx = np.zeros(max(Nprd, 1), dtype=np.int32)
        dJ, dJIdx = C.c_double(0.0), C.c_int32(0)
        _check(self.lib, self.lib.lwhip_prd_finalise(self._h, dRho.ctypes.data_as(abi.f64p), idx.ctypes.data_as(abi.i32p),
                                                     C.cast(C.byref(dJ), abi.f64p), C.cast(C.byref(dJIdx), abi.i32p)),
               'lwhip_prd_finalise')
        return dRho[:Nprd], idx[:Nprd], dJ.value, dJIdx.value

    @property
    def Nprd(self):
        return sum(1 for a in self.prob.atoms if (self.prdDetailed or not a.detailed) for t in a.trans
                   if t.type == abi.LINE and t.rhoPrd is not None)

    def time_dep_update(self, dt, prevTimePops, deviceResident=False, atom=None) -> IterationUpdate:
        """Backward-Euler population update, LwContext.time_dep_update (Source/LwMiddleLayer.pyx:3348-3420
        drives Source/UpdatePopulations.cpp:120-151 per atom): solve (1 - dt Gamma) n = nOld per depth.
        prevTimePops: list of [Nlevel, Nspace] arrays, one per active atom (the reference's
        `prevTimePops`); Gamma must hold the operator of the current iterate."""
        active = [ia for ia, a in enumerate(self.prob.atoms) if not a.detailed]
        before = None
        if not deviceResident:
            before = [np.array(self.prob.atoms[ia].n, dtype=np.float64, copy=True) for ia in active]
            self.upload(abi.GAMMA | abi.POPS)
        for q, ia in enumerate(active):
            if atom is not None and ia != atom:
                continue
            nOld = np.ascontiguousarray(prevTimePops[q], dtype=np.float64)
            _check(self.lib, self.lib.lwhip_time_dep_update(self._h, ia, nOld.ctypes.data_as(abi.f64p), float(dt)),
                   'lwhip_time_dep_update')
        if not deviceResident:
            self.download(abi.POPS)
            return self._rel_diff_pops(active, before)
        return IterationUpdate(updatedPops=True)

    def _rel_diff_pops(self, active, before) -> IterationUpdate:
        """LwContext.rel_diff_pops (Source/LwMiddleLayer.pyx:3294-3316 -> Ng::relative_change_from_prev,
        Source/Ng.hpp:116-136): per active atom max |(n - nPrev) / n| over the entries with n != 0 and the
        flattened [level, depth] index of the first maximum; nPrev = the populations before this update."""
        dPops, dIdx = [], []
        for ia, prev in zip(active, before):
            n = np.asarray(self.prob.atoms[ia].n, dtype=np.float64).reshape(-1)
            p = prev.reshape(-1)
            ch = np.zeros_like(n)
            nz = n != 0.0
            ch[nz] = np.abs((n[nz] - p[nz]) / n[nz])
            i = int(np.argmax(ch)) if ch.size else 0
            dPops.append(float(ch[i]) if ch.size else 0.0)
            dIdx.append(i)
        return IterationUpdate(updatedPops=True, dPops=dPops, dPopsMaxIdx=dIdx)

    def nr_post_update(self, stages, backgroundNe, ne, dC=None, nPrev=None, dt=0.0, atoms=None,
                       deviceResident=False) -> IterationUpdate:
        """One Newton-Raphson charge-conservation step for the listed active atoms, the device part
        of LwContext._nr_post_update_impl (Source/LwMiddleLayer.pyx:3533-3603 ->
        Source/UpdatePopulations.cpp:294-394).  `ne` [Nspace] is updated in place; `stages`, `dC`,
        `nPrev` are per-atom lists (dC / nPrev optional)."""
        if atoms is None:
            atoms = [ia for ia, a in enumerate(self.prob.atoms) if not a.detailed]
        before = None
        if not deviceResident:
            before = [np.array(self.prob.atoms[ia].n, dtype=np.float64, copy=True) for ia in atoms]
            self.upload(abi.GAMMA | abi.POPS | abi.COLLISIONS)
        args, keep = abi.make_nr_args(atoms, stages, backgroundNe, ne, dC=dC, nPrev=nPrev, dt=dt, crsw=self.crsw)
        _check(self.lib, self.lib.lwhip_nr_post_update(self._h, C.byref(args)), 'lwhip_nr_post_update')
        if not deviceResident:
            self.download(abi.POPS)
            return self._rel_diff_pops(atoms, before)
        return IterationUpdate(updatedPops=True)

    def check_status(self):
        _check(self.lib, self.lib.lwhip_check_status(self._h), 'lwhip_check_status')

    def compute_profiles(self, deviceResident=False):
        _check(self.lib, self.lib.lwhip_compute_profiles(self._h), 'lwhip_compute_profiles')
        if not deviceResident:
            self.download(abi.PROFILES)

    # -- multi-GPU split ----------------------------------------------------------------------------------
    def fs_partial(self, lambdaIterate=False):
        _check(self.lib, self.lib.lwhip_fs_partial(self._h, int(lambdaIterate)), 'lwhip_fs_partial')

    def fs_finalise(self) -> IterationUpdate:
        res = abi.lwhip_iter_result()
        _check(self.lib, self.lib.lwhip_fs_finalise(self._h, C.byref(res)), 'lwhip_fs_finalise')
        return IterationUpdate(updatedJ=True, dJMax=res.dJMax, dJMaxIdx=res.dJMaxIdx, crsw=self.crsw)

    def gamma_prefill_from_C(self, crsw=1.0):
        _check(self.lib, self.lib.lwhip_gamma_prefill_from_C(self._h, crsw), 'lwhip_gamma_prefill_from_C')

    def reduce_buffer(self):
        """(device pointer, total doubles to all-reduce) of the Gamma/R partial sums followed by
        the per-rank (dJMax, idx) slots."""
        ptr = C.c_void_p()
        nsum = C.c_size_t()
        ngather = C.c_size_t()
        _check(self.lib, self.lib.lwhip_reduce_buffer(self._h, C.byref(ptr), C.byref(nsum),
                                                      C.byref(ngather)), 'lwhip_reduce_buffer')
        return ptr.value, nsum.value + ngather.value

    # -- peer exchange of the sharded iteration (include/lwhip.h: lwhip_peer_*) ---------------------------------------
    def peer_window(self):
        """(device pointer, bytes) of this rank's exchange window."""
        ptr, n = C.c_void_p(), C.c_size_t()
        _check(self.lib, self.lib.lwhip_peer_window(self._h, C.byref(ptr), C.byref(n)), 'lwhip_peer_window')
        return ptr.value, n.value

    def peer_export(self) -> bytes:
        """The window's inter-process handle (64 opaque bytes) for the other ranks."""
        buf = C.create_string_buffer(64)
        _check(self.lib, self.lib.lwhip_peer_export(self._h, buf), 'lwhip_peer_export')
        return buf.raw

    def peer_attach(self, handles):
        """`handles`: every rank's peer_export() in rank order (own entry ignored).  From now on fs_partial / fs_finalise
        exchange through the windows: no all-reduce of the reduce buffer."""
        blob = b''.join(bytes(h) for h in handles)
        assert len(blob) == 64 * len(handles)
        _check(self.lib, self.lib.lwhip_peer_attach(self._h, C.c_char_p(blob)), 'lwhip_peer_attach')
        self.peers_attached = True

    def peer_attach_pointers(self, windows):
        """The same for windows this process can address (contexts of one process): device pointers in rank order."""
        arr = (C.c_void_p * len(windows))(*windows)
        _check(self.lib, self.lib.lwhip_peer_attach_pointers(self._h, arr), 'lwhip_peer_attach_pointers')
        self.peers_attached = True

    def peer_selftest(self, timeout_ms=200) -> int:
        """One exchange of a known pattern (all attached ranks together): 0 = every rank's slot arrived intact."""
        res = C.c_int32(-1)
        _check(self.lib, self.lib.lwhip_peer_selftest(self._h, int(timeout_ms), C.byref(res)), 'lwhip_peer_selftest')
        return res.value

    def peer_detach(self):
        _check(self.lib, self.lib.lwhip_peer_detach(self._h), 'lwhip_peer_detach')
        self.peers_attached = False

    # -- the caller's J array as an output of the sweep (include/lwhip.h: lwhip_map_host_J) ------------------------------
    def map_host_J(self, enable=True):
        """Page-lock prob.J and let the sweep store J straight into it (no copy on download); returns False where the
        library cannot (2D, shards, the march sweep, registration refused)."""
        st = self.lib.lwhip_map_host_J(self._h, int(enable))
        if st == abi.ERR_UNSUPPORTED:
            return False
        _check(self.lib, st, 'lwhip_map_host_J')
        return bool(enable)

    def fingerprint_J(self, array=None) -> int:
        """Fingerprint of the device's J as it would read in `array` (default: prob.J's rows of this shard), formed on the
        device; lightweaver_amd.context.host_fingerprint(array) is the host's side of the comparison."""
        import numpy as np
        ptr = (array if array is not None else self.prob.J[self.laStart:self.laEnd]).ctypes.data
        out = C.c_uint64()
        _check(self.lib, self.lib.lwhip_fingerprint_J(self._h, C.c_void_p(ptr), C.byref(out)), 'lwhip_fingerprint_J')
        return out.value

    # -- measurement -----------------------------------------------------------------------------------------
    def profile_enable(self, enable=True):
        _check(self.lib, self.lib.lwhip_profile_enable(self._h, int(enable)), 'lwhip_profile_enable')

    def sweep_time(self):
        ms = C.c_double()
        n = C.c_int()
        _check(self.lib, self.lib.lwhip_sweep_time(self._h, C.byref(ms), C.byref(n)), 'lwhip_sweep_time')
        return ms.value, n.value

    def sweep_kind(self):
        """'march' (ray-column march), 'lanes' (depth-across-lanes sweep) or '2d': which sweep kernel the library chose."""
        return {0: 'march', 1: 'lanes', 2: '2d'}.get(self.lib.lwhip_sweep_kind(self._h), '?')

    def algorithmic_bytes(self):
        b = C.c_double()
        _check(self.lib, self.lib.lwhip_algorithmic_bytes(self._h, C.byref(b)), 'lwhip_algorithmic_bytes')
        return b.value


def host_fingerprint(array) -> int:
    """include/lwhip.h lwhip_host_fingerprint of a C-contiguous float64 array."""
    import numpy as np
    assert array.dtype == np.float64 and array.flags.c_contiguous
    return load_library().lwhip_host_fingerprint(array.ctypes.data_as(abi.f64p), array.size)


def device_count():
    return load_library().lwhip_device_count()

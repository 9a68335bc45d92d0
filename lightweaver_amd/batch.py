"""1.5D column batches (BASELINE.json configs[3]): many independent plane-parallel columns, each a
Context-equivalent problem with its own atmosphere, profiles and populations.  Columns never
exchange radiation, so they shard across GPUs with NO collective ("replicas"): rank r simply owns
columns r, r + world, ....  On one GPU the columns' contexts are all resident in HBM (26 MB of phi
per H + Ca II column at ~3k wavelengths: thousands fit in 288 GB) and their iterations are queued
back to back on one stream, so the device never idles between columns; profiles are regenerated on
the device per column (lwhip_compute_profiles), so phi never crosses PCIe.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence

from . import _abi as abi
from .context import Context, _check
from .model import Problem


def columns_of_rank(ncolumns: int, world: int, rank: int) -> List[int]:
    """Round-robin column ownership; no data-path communication between ranks."""
    if not (0 <= rank < world):
        raise ValueError('bad rank')
    return list(range(rank, ncolumns, world))


class ColumnBatch:
    def __init__(self, problems: Sequence[Problem], device: int = 0, stream: Optional[int] = None,
                 device_profiles: bool = True, streams: Optional[Sequence[int]] = None, fused: bool = True):
        """`fused` (default): the columns advance in ONE set of launches per iteration (lwhip_batch_*: one grid
        slice per column), which is what fills the device when a single column's wavelength grid does not;
        it needs structurally identical columns on one stream and falls back to per-column launches otherwise.
        `streams`: optional HIP streams (handles) the columns are dealt to round-robin in the per-column mode, so
        that the small reduce / apply / solve kernels of one column overlap the sweep of the next."""
        self.problems = list(problems)
        self._batch = None
        if streams:
            self.contexts = [Context(p, device=device, stream=streams[i % len(streams)])
                             for i, p in enumerate(self.problems)]
        else:
            # (the columns of a fused batch share their structure: all but the first borrow the first one's tables)
            # Columns 2 .. n are created concurrently: a borrower's lwhip_create is ~3 ms of host work (its own bookkeeping of
            # the shared structure) + ~2 ms of allocations and uploads, ctypes releases the interpreter lock for the call, and
            # the library's create path shares nothing writable between contexts (the donor's borrower count is atomic, the
            # error text is per thread).  LWHIP_CREATE_THREADS=1 creates them one after the other (default 16).
            hint = len(self.problems) if fused else 0
            self.contexts = [Context(self.problems[0], device=device, stream=stream, batchHint=hint)] if self.problems else []
            donor = self.contexts[0] if (fused and self.contexts) else None
            rest = self.problems[1:]
            # (round 6, tools/create_threads_time.py: 64 borrowers take 5.8 / 1.8 / 1.35 / 0.99 ms each on 1 / 4 / 8 / 16 threads --
            # beyond that the HIP runtime's own locks around allocations and copies are what is left)
            nthr = max(1, min(int(os.environ.get('LWHIP_CREATE_THREADS', '16')), os.cpu_count() or 1, len(rest)))
            if os.environ.get('LWHIP_DEBUG') and any(os.environ.get(k) for k in ('LWHIP_PAD_SENTINEL', 'LWHIP_TRACE_ALLOC')):
                nthr = 1    # (those diagnoses number the allocations in creation order: one creating thread)
            make = lambda p: Context(p, device=device, stream=stream, batchHint=hint, like=donor)  # noqa: E731
            if nthr > 1:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=nthr) as ex:
                    futs = [ex.submit(make, p) for p in rest]
                    made, err = [], None
                    for f in futs:
                        try:
                            made.append(f.result())
                        except Exception as e:  # (close what was made, then report the first failure)
                            err = err or e
                    if err is not None:
                        for c in reversed(made):
                            c.close()
                        self.contexts[0].close()
                        raise err
                    self.contexts += made
            else:
                self.contexts += [make(p) for p in rest]
        if fused and not streams and self.contexts:
            import ctypes as C
            lib = self.contexts[0].lib
            arr = (C.c_void_p * len(self.contexts))(*[c._h for c in self.contexts])
            h = C.c_void_p()
            if lib.lwhip_batch_create(arr, len(self.contexts), C.byref(h)) == abi.OK:
                self._batch = h
        if device_profiles:
            self.compute_profiles()

    def __len__(self):
        return len(self.contexts)

    def compute_profiles(self):
        """phi / wphi of every line of every column on the device (lwhip_batch_compute_profiles: one launch pair for
        the whole batch; per column where the batch is not fused).  After an atmosphere upload the next iteration
        does this by itself for the columns concerned."""
        if self._batch is not None:
            lib = self.contexts[0].lib
            _check(lib, lib.lwhip_batch_compute_profiles(self._batch), 'lwhip_batch_compute_profiles')
            return
        for c in self.contexts:
            c.compute_profiles(deviceResident=True)

    def close(self):
        if self._batch is not None:
            self.contexts[0].lib.lwhip_batch_destroy(self._batch)
            self._batch = None
        for c in reversed(self.contexts):  # (borrowers of the first column's tables before the first column)
            c.close()
        self.contexts = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def formal_sol_gamma_matrices(self, lambdaIterate=False, sync_host=True):
        """One iteration of every column; kernels of all columns are queued before the first
        result is read back.  sync_host=False reads nothing back (no host wait per column): use it for
        all but the iterations whose dJMax is wanted."""
        if self._batch is not None:
            import ctypes as C
            from .context import IterationUpdate
            lib = self.contexts[0].lib
            crsw = self.contexts[0].crsw
            if not sync_host:
                _check(lib, lib.lwhip_batch_formal_sol_gamma_matrices(self._batch, int(lambdaIterate), crsw, None),
                       'lwhip_batch_formal_sol_gamma_matrices')
                return None
            res = (abi.lwhip_iter_result * len(self.contexts))()
            _check(lib, lib.lwhip_batch_formal_sol_gamma_matrices(self._batch, int(lambdaIterate), crsw, res),
                   'lwhip_batch_formal_sol_gamma_matrices')
            return [IterationUpdate(updatedJ=True, dJMax=r.dJMax, dJMaxIdx=r.dJMaxIdx, crsw=crsw) for r in res]
        for c in self.contexts:
            c.gamma_prefill_from_C(c.crsw)
            c.fs_partial(lambdaIterate)
        if not sync_host:
            for c in self.contexts:
                _check(c.lib, c.lib.lwhip_fs_finalise(c._h, None), 'lwhip_fs_finalise')
            return None
        return [c.fs_finalise() for c in self.contexts]

    def stat_equil(self):
        """Queued for every column without a host wait in between; one status check at the end."""
        if self._batch is not None:
            lib = self.contexts[0].lib
            _check(lib, lib.lwhip_batch_stat_equil(self._batch), 'lwhip_batch_stat_equil')
            return
        for c in self.contexts:
            c.stat_equil(deviceResident=True, sync_host=False)
        for c in self.contexts:
            c.check_status()

    def iterate(self, niter: int, nscatter: int = 3, callback: Optional[Callable] = None):
        """iterate_ctx_se-style loop (lightweaver/iterate_ctx.py:157-176) over the whole batch."""
        ups = []
        for it in range(niter):
            want = callback is not None or it == niter - 1
            ups = self.formal_sol_gamma_matrices(sync_host=want)
            if it >= nscatter:
                self.stat_equil()
            if callback is not None:
                callback(it, ups)
        return ups

    def download(self, mask=abi.ALL_OUTPUTS | abi.POPS):
        for c in self.contexts:
            c.download(mask)

"""Wavelength sharding across GPUs: one process per GPU, `torch.distributed` (backend "nccl" is RCCL
on ROCm) over xGMI.

Each wavelength is independent within an iteration (the reference parallelises over wavelength
across threads, Source/SimdFullIterationTemplates.hpp:675-698); a rank owns a contiguous,
cost-balanced block of wavelengths -- its J / I rows, background and phi live only in its HBM --
and contributes additively to every atom's Gamma and every transition's Rij/Rji.  One iteration
therefore needs exactly ONE collective: an all-reduce(sum) of the concatenated partial sums
[Ntrans, 4, Nspace] (80 KB for H + Ca II at 82 depths: latency-bound on xGMI, so a single fused
buffer, never one call per atom), whose tail carries each rank's (dJMax, idx) in its own slot so
the max-reduction rides along.  This replaces the per-thread accumulate of
Source/ThreadStorage.cpp:150-187,334-396.

The class is backend-agnostic: `ctx` only has to provide gamma_prefill_from_C / fs_partial /
reduce_tensor / fs_finalise, which lets the world_size-2 gloo tests drive the same code on CPU.
"""
from __future__ import annotations

import numpy as np

from . import _abi as abi


def wavelength_costs(prob) -> np.ndarray:
    """Relative cost of each wavelength: rays x (c0 + c1 * active lines + c2 * active continua);
    the same model lwhip_create uses to balance its chunks."""
    nl = np.zeros(prob.Nlambda)
    nc = np.zeros(prob.Nlambda)
    for a in prob.atoms:
        for t in a.trans:
            (nl if t.type == abi.LINE else nc)[t.Nblue:t.Nred] += 1
    return 1.0 + 0.243 * nl + 0.03 * nc


def shard_bounds(prob, world: int):
    """Contiguous cost-balanced wavelength blocks: bounds[r] .. bounds[r+1] belongs to rank r."""
    if world < 1 or world > prob.Nlambda:
        raise ValueError('need 1 <= world <= Nlambda')
    cum = np.cumsum(wavelength_costs(prob))
    bounds = [0]
    for r in range(1, world):
        b = int(np.searchsorted(cum, cum[-1] * r / world))
        b = min(max(b, bounds[-1] + 1), prob.Nlambda - (world - r))
        bounds.append(b)
    bounds.append(prob.Nlambda)
    return bounds


class DeviceReduceBuffer:
    """Zero-copy torch view of the library's reduce buffer (device pointer + length)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {'shape': (n,), 'typestr': '<f8', 'data': (ptr, False),
                                         'version': 2, 'strides': None}


class ShardedIteration:
    """One iteration on a wavelength shard: partial sums, ONE all-reduce, finalise.

    Stream contract.  `torch.distributed` orders a collective against torch's CURRENT stream only, while the
    library queues its kernels on the context's stream.  For a HIP context the two are therefore made the same
    stream before every collective phase (`ctx.set_stream(torch.cuda.current_stream())`): the sweep and reduce
    kernels, the all-reduce and the apply kernel are then ordered by stream order alone, with no host
    synchronisation in between.  A Context created with its own (library-owned) stream is re-bound here, and
    calling under `with torch.cuda.stream(s):` moves the context to `s` for that call.  One case needs no re-binding:
    torch's default stream is the legacy null stream (handle 0, which the C ABI reads as "the library's own stream");
    the library's own stream is a blocking stream, and the null stream synchronises implicitly with every blocking
    stream, in both directions -- so a collective on torch's default stream is ordered against the library's kernels
    by HIP's legacy default-stream semantics.
    """

    def __init__(self, ctx, world: int = 1, rank: int = 0, group=None, force_collective=False, transport='auto'):
        """`transport`: how the shards' partial sums meet --
          'collective'  one all-reduce of the reduce buffer per iteration (RCCL / gloo);
          'peer'        the library's peer exchange (include/lwhip.h lwhip_peer_*): every rank stores its sums into its slot
                        of every rank's window and the apply launch adds the slots in rank order -- no collective call and
                        no host in the loop, and the same bits on every rank.  The ranks must share a node (the windows are
                        exchanged as hipIpc handles through `all_gather_object`); PRD sub-iterations keep the collectives;
          'auto'        'peer' for HIP contexts when world > 1 and the handles can be opened on every rank, else 'collective'
                        (LWHIP_SHARD_TRANSPORT=collective|peer overrides)."""
        import os
        self.ctx = ctx
        self.world = world
        self.rank = rank
        self.group = group
        self.force_collective = force_collective  # run the all-reduce even for world == 1 (tests)
        self._buf = None
        self._bound = None      # the stream handle the context was last bound to
        self._side = None       # side stream of the host-authoritative mode (all-reduce overlapped with the J download)
        self.allreduce_s = 0.0  # host time spent issuing collectives (bench.py: allreduce_us)
        self._bind_stream()
        self.transport = 'collective'
        want = os.environ.get('LWHIP_SHARD_TRANSPORT', transport)
        if want in ('peer', 'auto') and world > 1 and self._is_hip() and hasattr(ctx, 'peer_export'):
            self._attach_peers(required=(want == 'peer'))

    def _attach_peers(self, required):
        """Exchange the windows' handles and attach; every rank learns whether ALL ranks succeeded (a partial attach would
        deadlock the exchange), and everybody falls back together otherwise."""
        import torch.distributed as dist
        try:
            mine = self.ctx.peer_export()
            err = None
        except Exception as e:   # noqa: BLE001
            mine, err = None, e
        got = [None] * self.world
        dist.all_gather_object(got, mine, group=self.group)
        ok = err is None and all(h is not None for h in got)
        if ok:
            try:
                self.ctx.peer_attach(got)
            except Exception as e:   # noqa: BLE001
                ok, err = False, e
        oks = [None] * self.world
        dist.all_gather_object(oks, bool(ok), group=self.group)
        if all(oks) and hasattr(self.ctx, 'peer_selftest'):
            # every rank could open every window: now ONE exchange of a known pattern on the actual topology -- a rank whose
            # kernels do not see a peer's stores (or whose flag never arrives) sends everybody back to the collective
            try:
                res = self.ctx.peer_selftest(200)
            except Exception as e:   # noqa: BLE001
                res, err = -1, e
            results = [None] * self.world
            dist.all_gather_object(results, int(res), group=self.group)
            if all(r == 0 for r in results):
                self.transport = 'peer'
                return
            err = err or RuntimeError(f'peer self-test results by rank: {results}')
            oks = [False]
        elif all(oks):
            self.transport = 'peer'
            return
        if ok:
            self.ctx.peer_detach()
        if required:
            raise RuntimeError(f'peer transport requested but not available on every rank (this rank: {err!r})')

    @property
    def _collective(self):
        return self.world > 1 or self.force_collective

    def _is_hip(self):
        return hasattr(self.ctx, 'lib') and hasattr(self.ctx, 'set_stream')

    def _bind_stream(self):
        """Make the context's stream torch's current stream (see the class docstring)."""
        if not (self._collective and self._is_hip()):
            return None
        import torch
        cur = torch.cuda.current_stream()
        if self._bound != cur.cuda_stream:
            self.ctx.set_stream(cur.cuda_stream)
            self._bound = cur.cuda_stream
        return cur

    def _reduce_tensor(self):
        if self._buf is None:
            if hasattr(self.ctx, 'reduce_tensor'):
                self._buf = self.ctx.reduce_tensor()
            else:
                import torch
                ptr, n = self.ctx.reduce_buffer()
                self._buf = torch.as_tensor(DeviceReduceBuffer(ptr, n), device='cuda')
        return self._buf

    def _all_reduce(self, t):
        import time
        import torch.distributed as dist
        t0 = time.perf_counter()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        self.allreduce_s += time.perf_counter() - t0

    def formal_sol_gamma_matrices(self, lambdaIterate=False, deviceResident=True, crsw=1.0):
        ctx = self.ctx
        # (decided per call: `force_collective` or `world` may be changed between calls)
        if not self._collective and hasattr(ctx, 'formal_sol_gamma_matrices') and hasattr(ctx, 'lib'):
            # no collective to interleave: the library's one-call form (which also folds the second reduce stage
            # and the dJMax reduction into the apply launch)
            ctx.crsw = crsw
            return ctx.formal_sol_gamma_matrices(lambdaIterate=lambdaIterate, deviceResident=deviceResident)
        cur = self._bind_stream()
        if deviceResident:
            ctx.gamma_prefill_from_C(crsw)
        else:
            ctx.prob.gamma_prefill(crsw)
            ctx.upload(abi.GAMMA | abi.J | abi.POPS | abi.BC | abi.RHOPRD)
        ctx.fs_partial(lambdaIterate)
        if self._collective and not deviceResident and cur is not None and self.transport != 'peer':
            # host-authoritative mode: J and I are final once the sweep has run (they do not depend on the sums), so
            # the all-reduce runs on a side stream while this stream copies them to the host
            import torch
            if self._side is None:
                self._side = torch.cuda.Stream(device=cur.device)
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                self._all_reduce(self._reduce_tensor())
            ctx.download(abi.J | abi.I)
            cur.wait_stream(self._side)
            up = ctx.fs_finalise()
            ctx.download(abi.ALL_OUTPUTS & ~(abi.J | abi.I))
            return up
        if self._collective and self.transport != 'peer':
            self._all_reduce(self._reduce_tensor())
        up = ctx.fs_finalise()   # (peer transport: its launch waits for every rank's slot and adds them in rank order)
        if not deviceResident:
            ctx.download(abi.ALL_OUTPUTS)
        return up

    def _prd_tensor(self):
        if hasattr(self.ctx, 'prd_tensor'):
            return self.ctx.prd_tensor()
        import torch
        ptr, n = self.ctx.prd_pack()
        return torch.as_tensor(DeviceReduceBuffer(ptr, n), device='cuda') if n else None

    def prd_redistribute(self, maxIter=3, tol=1e-2):
        """PRD sub-iterations on wavelength shards: per sub-iteration one all-reduce of
        the PRD lines' J rows (each rank contributes its own, ~Nlambda_PRD x Nspace doubles) before the
        scattering integral, and the iteration's usual all-reduce of the rate sums, whose tail also
        carries every rank's max |d rho / rho| per line.  Mirrors redistribute_prd_lines' loop
        (Source/PrdTemplates.hpp:250-290)."""
        from .context import IterationUpdate
        ctx = self.ctx
        collective = self._collective
        self._bind_stream()
        dRhoAll, idxAll, dJAll, dJIdxAll = [], [], [], []
        it = 0
        while it < maxIter:
            it += 1
            buf = self._prd_tensor()          # packs this rank's rows
            if collective and buf is not None:
                self._all_reduce(buf)
            ctx.prd_partial()
            if collective:
                self._all_reduce(self._reduce_tensor())
            dRho, idx, dJ, dJIdx = ctx.prd_finalise()
            dRhoAll.append(np.array(dRho))
            idxAll.append(np.array(idx))
            dJAll.append(dJ)
            dJIdxAll.append(dJIdx)
            if len(dRho) == 0 or dRho.max() < tol:
                break
        return IterationUpdate(updatedRho=it > 0, updatedJ=it > 0, NprdSubIter=it,
                               dRho=np.array(dRhoAll), dRhoMaxIdx=np.array(idxAll),
                               dJPrdMax=np.array(dJAll), dJPrdMaxIdx=np.array(dJIdxAll),
                               dJMax=dJAll[-1] if dJAll else 0.0, dJMaxIdx=dJIdxAll[-1] if dJIdxAll else 0)

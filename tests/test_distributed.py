"""The N > 1 path on CPU: world_size-2 gloo processes drive lightweaver_amd.distributed
(shard bounds, one fused all-reduce of the Gamma/R partial sums with the per-rank (dJMax, idx)
slots riding in its tail, finalise) with the C oracle standing in for the HIP context, and must
reproduce the unsharded iteration."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from helpers import load_fixture, rel_err  # noqa: E402
from lightweaver_amd import _abi as abi  # noqa: E402
from lightweaver_amd.context import IterationUpdate  # noqa: E402
from lightweaver_amd.distributed import ShardedIteration, shard_bounds, wavelength_costs  # noqa: E402
from lightweaver_amd.harness import models  # noqa: E402


class OracleShardContext:
    """Test double with the Context surface ShardedIteration needs, computing on the CPU oracle."""

    def __init__(self, prob, laStart, laEnd, world, rank):
        from oracle.bindings import OracleContext
        self.prob = prob
        self.oc = OracleContext(prob)
        self.laStart, self.laEnd, self.world, self.rank = laStart, laEnd, world, rank
        self.ntr = sum(len(a.trans) for a in prob.atoms)
        self.nsum = self.ntr * 4 * prob.Nspace
        # PRD lines of the active atoms and the layout of the J gather buffer (each line's full grid)
        self.prd = [t for a in prob.atoms if not a.detailed for t in a.trans
                    if t.type == abi.LINE and t.rhoPrd is not None]
        self.nprd = len(self.prd)
        self.prdOff = np.concatenate([[0], np.cumsum([t.Nred - t.Nblue for t in self.prd])]).astype(int)
        self.prdJ = torch.zeros(int(self.prdOff[-1]) * prob.Nspace, dtype=torch.float64)
        self.buf = torch.zeros(self.nsum + 2 * world * (1 + self.nprd), dtype=torch.float64)
        self.dJ = np.zeros(prob.Nlambda)

    def gamma_prefill_from_C(self, crsw=1.0):
        self.prob.gamma_prefill(crsw)

    def upload(self, mask):
        pass

    def download(self, mask):
        pass

    def reduce_tensor(self):
        return self.buf

    def fs_partial(self, lambdaIterate=False):
        red = np.zeros(self.nsum)
        self.oc.fs_partial_shard(self.laStart, self.laEnd, red, self.dJ, lambdaIterate)
        self.buf.zero_()
        self.buf[:self.nsum] = torch.from_numpy(red)
        shard = self.dJ[self.laStart:self.laEnd]
        i = int(np.argmax(shard))
        self.buf[self.nsum + 2 * self.rank] = float(shard[i])
        self.buf[self.nsum + 2 * self.rank + 1] = float(self.laStart + i)

    # -- PRD sub-iteration, same split as lwhip_prd_pack / _partial / _finalise -----------------------
    def prd_tensor(self):
        Ns = self.prob.Nspace
        self.prdJ.zero_()
        view = self.prdJ.numpy().reshape(-1, Ns)
        for q, t in enumerate(self.prd):
            lo, hi = max(t.Nblue, self.laStart), min(t.Nred, self.laEnd)
            if hi > lo:
                view[self.prdOff[q] + lo - t.Nblue:self.prdOff[q] + hi - t.Nblue] = self.prob.J[lo:hi]
        return self.prdJ

    def prd_partial(self):
        Ns = self.prob.Nspace
        view = self.prdJ.numpy().reshape(-1, Ns)
        for q, t in enumerate(self.prd):          # the gathered rows: J over every PRD line's whole grid
            self.prob.J[t.Nblue:t.Nred] = view[self.prdOff[q]:self.prdOff[q + 1]]
        dRho = np.zeros(max(self.nprd, 1))
        idx = np.zeros(max(self.nprd, 1), dtype=np.int32)
        self.oc.prd_scatter_shard(self.laStart, self.laEnd, dRho, idx)
        red = np.zeros(self.nsum)
        self.oc.prd_rates_partial_shard(self.laStart, self.laEnd, red, self.dJ)
        self.buf.zero_()
        self.buf[:self.nsum] = torch.from_numpy(red)
        shard = self.dJ[self.laStart:self.laEnd]
        i = int(np.argmax(shard))
        self.buf[self.nsum + 2 * self.rank] = float(shard[i])
        self.buf[self.nsum + 2 * self.rank + 1] = float(self.laStart + i) if shard[i] > 0 else 0.0
        base = self.nsum + 2 * self.world + 2 * self.rank * self.nprd
        for q in range(self.nprd):
            self.buf[base + 2 * q] = float(dRho[q])
            self.buf[base + 2 * q + 1] = float(idx[q])

    def prd_finalise(self):
        self.oc.prd_finalise_shard(self.buf[:self.nsum].numpy().copy())
        tail = self.buf[self.nsum:self.nsum + 2 * self.world].numpy().reshape(self.world, 2)
        best = max(range(self.world), key=lambda r: (tail[r, 0], -tail[r, 1]))
        ptail = self.buf[self.nsum + 2 * self.world:].numpy().reshape(self.world, self.nprd, 2)
        dRho = np.zeros(self.nprd)
        idx = np.zeros(self.nprd, dtype=np.int32)
        for q, t in enumerate(self.prd):
            r = max(range(self.world), key=lambda r: (ptail[r, q, 0], -ptail[r, q, 1]))
            dRho[q] = ptail[r, q, 0]
            idx[q] = int(ptail[r, q, 1]) % (t.Nred - t.Nblue)
        return dRho, idx, float(tail[best, 0]), int(tail[best, 1])

    def fs_finalise(self):
        self.oc.fs_finalise_shard(self.buf[:self.nsum].numpy().copy())
        tail = self.buf[self.nsum:self.nsum + 2 * self.world].numpy().reshape(self.world, 2)
        best = max(range(self.world), key=lambda r: (tail[r, 0], -tail[r, 1]))
        return IterationUpdate(updatedJ=True, dJMax=float(tail[best, 0]), dJMaxIdx=int(tail[best, 1]))


def _worker(rank, world, port, out, fixture='falc_h_ca_small'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        prob, _ = load_fixture(fixture)
        bounds = shard_bounds(prob, world)
        ctx = OracleShardContext(prob, bounds[rank], bounds[rank + 1], world, rank)
        it = ShardedIteration(ctx, world, rank)
        ups = []
        for _ in range(2):
            ups.append(it.formal_sol_gamma_matrices(deviceResident=True))
        la0, la1 = bounds[rank], bounds[rank + 1]
        np.savez(os.path.join(out, f'rank{rank}.npz'), J=prob.J[la0:la1], I=prob.I[la0:la1], la=[la0, la1],
                 G0=prob.atoms[0].Gamma, G1=prob.atoms[-1].Gamma,
                 R=np.stack([t.Rij for t in prob.atoms[0].trans]),
                 dJ=[u.dJMax for u in ups], idx=[u.dJMaxIdx for u in ups])
    finally:
        dist.destroy_process_group()


def test_shard_bounds_are_contiguous_and_balanced():
    prob = models.falc_h_ca(Nrays=3, lineScale=0.3, computeProfiles=False)
    cost = wavelength_costs(prob)
    for world in (1, 2, 3, 8):
        b = shard_bounds(prob, world)
        assert b[0] == 0 and b[-1] == prob.Nlambda and all(b[i] < b[i + 1] for i in range(world))
        parts = [cost[b[i]:b[i + 1]].sum() for i in range(world)]
        assert max(parts) <= 1.25 * cost.sum() / world + cost.max()
    with pytest.raises(ValueError):
        shard_bounds(prob, prob.Nlambda + 1)


def test_two_rank_gloo_matches_unsharded(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    # unsharded reference run of the same two iterations
    from oracle.bindings import OracleContext
    prob, _ = load_fixture('falc_h_ca_small')
    oc = OracleContext(prob)
    res = []
    dJ = np.zeros(prob.Nlambda)
    for _ in range(2):
        prob.gamma_prefill()
        oc.formal_sol_gamma_matrices(dJ=dJ)
        res.append((dJ.max(), int(np.argmax(dJ))))
    for rank in range(world):
        z = np.load(tmp_path / f'rank{rank}.npz')
        la0, la1 = z['la']
        assert np.array_equal(z['J'], prob.J[la0:la1]) and np.array_equal(z['I'], prob.I[la0:la1])
        assert rel_err(z['G0'], prob.atoms[0].Gamma) <= 1e-12
        assert rel_err(z['G1'], prob.atoms[1].Gamma) <= 1e-12
        assert rel_err(z['R'], np.stack([t.Rij for t in prob.atoms[0].trans])) <= 1e-12
        assert [float(x) for x in z['dJ']] == [r[0] for r in res]
        assert [int(x) for x in z['idx']] == [r[1] for r in res]


def test_two_rank_gloo_2d_matches_unsharded(tmp_path):
    """BASELINE config 5's split: the 2D problem (x-periodic, BESSER short characteristics) on two wavelength shards
    with one all-reduce of the Gamma / rate partial sums per iteration."""
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path), 'falc2d_small'), nprocs=world, join=True)
    from oracle.bindings import OracleContext
    prob, _ = load_fixture('falc2d_small')
    oc = OracleContext(prob)
    res = []
    dJ = np.zeros(prob.Nlambda)
    for _ in range(2):
        prob.gamma_prefill()
        oc.formal_sol_gamma_matrices(dJ=dJ)
        res.append((dJ.max(), int(np.argmax(dJ))))
    for rank in range(world):
        z = np.load(tmp_path / f'rank{rank}.npz')
        la0, la1 = z['la']
        assert np.array_equal(z['J'], prob.J[la0:la1]) and np.array_equal(z['I'], prob.I[la0:la1])
        assert rel_err(z['G0'], prob.atoms[0].Gamma) <= 1e-12
        assert rel_err(z['R'], np.stack([t.Rij for t in prob.atoms[0].trans])) <= 1e-12
        assert [float(x) for x in z['dJ']] == [r[0] for r in res]
        assert [int(x) for x in z['idx']] == [r[1] for r in res]


def _prd_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from test_prd import golden_prd_problem, prd_lines
        prob, _ = golden_prd_problem()
        lines = prd_lines(prob)
        # cut inside a PRD line's grid so that both ranks own part of it
        m = (lines[-1].Nblue + lines[-1].Nred) // 2
        bounds = [0, m, prob.Nlambda]
        ctx = OracleShardContext(prob, bounds[rank], bounds[rank + 1], world, rank)
        it = ShardedIteration(ctx, world, rank)
        ups = []
        for _ in range(2):
            it.formal_sol_gamma_matrices(deviceResident=True)
            ups.append(it.prd_redistribute(3, 1e-2))
        la0, la1 = bounds[rank], bounds[rank + 1]
        rho = [t.rhoPrd[max(la0 - t.Nblue, 0):max(min(la1, t.Nred) - t.Nblue, 0)] for t in lines]
        np.savez(os.path.join(out, f'prd{rank}.npz'), J=prob.J[la0:la1], la=[la0, la1],
                 R=np.stack([t.Rij for t in lines]), rho0=rho[0], rho1=rho[1],
                 n=[u.NprdSubIter for u in ups], dRho=np.concatenate([u.dRho.ravel() for u in ups]),
                 idx=np.concatenate([u.dRhoMaxIdx.ravel() for u in ups]),
                 dJ=np.concatenate([u.dJPrdMax for u in ups]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_prd_matches_unsharded(tmp_path):
    """PRD sub-iterations on two wavelength shards (J rows of the PRD lines all-reduced before the
    scattering integral, per-line max changes riding in the tail of the rate all-reduce) against the
    unsharded oracle loop."""
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_prd_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from oracle.bindings import OracleContext
    from test_prd import golden_prd_problem, prd_lines
    prob, _ = golden_prd_problem()
    oc = OracleContext(prob)
    ups = []
    for _ in range(2):
        prob.gamma_prefill()
        oc.formal_sol_gamma_matrices()
        ups.append(oc.redistribute_prd(3, 1e-2))
    lines = prd_lines(prob)
    for rank in range(world):
        z = np.load(tmp_path / f'prd{rank}.npz')
        la0, la1 = z['la']
        assert [int(x) for x in z['n']] == [u['NprdSubIter'] for u in ups]
        assert np.allclose(z['dRho'], np.concatenate([u['dRho'].ravel() for u in ups]), rtol=1e-9)
        assert np.array_equal(z['idx'], np.concatenate([u['dRhoMaxIdx'].ravel() for u in ups]))
        assert np.allclose(z['dJ'], np.concatenate([u['dJPrdMax'] for u in ups]), rtol=1e-12)
        assert rel_err(z['J'], prob.J[la0:la1]) <= 1e-12
        assert rel_err(z['R'], np.stack([t.Rij for t in lines])) <= 1e-12
        for q, t in enumerate(lines):
            want = t.rhoPrd[max(la0 - t.Nblue, 0):max(min(la1, t.Nred) - t.Nblue, 0)]
            assert rel_err(z[f'rho{q}'], want) <= 1e-10


# ---- the same split on the real device path: two processes share ONE GPU ------------------------------------------
def _gpu_worker(rank, world, port, out, resident, transport='collective'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lightweaver_amd.context import Context
        prob = models.falc_h_ca(Nrays=3, lineScale=0.3)
        bounds = shard_bounds(prob, world)
        la0, la1 = bounds[rank], bounds[rank + 1]
        # stream=None: the context starts on a library-owned stream; ShardedIteration must move it to torch's
        with Context(prob, device=0, laStart=la0, laEnd=la1, worldSize=world, worldRank=rank) as ctx:
            it = ShardedIteration(ctx, world, rank, transport=transport)
            assert it.transport == transport, (it.transport, transport)
            ups = []
            side = torch.cuda.Stream()
            for i in range(3):
                if i == 2:       # a different current stream: the context follows it
                    with torch.cuda.stream(side):
                        ups.append(it.formal_sol_gamma_matrices(deviceResident=resident))
                    side.synchronize()
                else:
                    ups.append(it.formal_sol_gamma_matrices(deviceResident=resident))
            if resident:
                ctx.download(abi.ALL_OUTPUTS)
        np.savez(os.path.join(out, f'gpu{rank}.npz'), J=prob.J[la0:la1], I=prob.I[la0:la1], la=[la0, la1],
                 G0=prob.atoms[0].Gamma, G1=prob.atoms[-1].Gamma,
                 R=np.stack([t.Rij for t in prob.atoms[0].trans]),
                 dJ=[u.dJMax for u in ups], idx=[u.dJMaxIdx for u in ups])
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('transport', ['collective', 'peer'])
@pytest.mark.parametrize('resident', [True, False])
def test_two_process_gloo_on_one_gpu_matches_unsharded(gpu, tmp_path, resident, transport):
    """The N > 1 data path with the real kernels: two processes, each with a wavelength-shard Context on the SAME
    GPU.  transport='collective': they all-reduce the zero-copy view of the library's device reduce buffer (gloo stages
    device tensors through the host; RCCL refuses two ranks on one device) between fs_partial and fs_finalise;
    transport='peer' (round 6, the default where it can be set up): no collective on the step -- each rank stores its sums
    into its slot of both ranks' windows (opened through hipIpc handles) and the apply launch adds the slots in rank order,
    so the two ranks' Gamma and rates are the same BITS.  Contexts start on their own stream, so this also covers the stream
    contract of ShardedIteration; resident=False covers the host-authoritative mode."""
    from lightweaver_amd.context import Context
    world = 2
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_gpu_worker, args=(world, port, str(tmp_path), resident, transport), nprocs=world, join=True)
    if transport == 'peer':
        za, zb = np.load(tmp_path / 'gpu0.npz'), np.load(tmp_path / 'gpu1.npz')
        for k in ('G0', 'G1', 'R'):
            assert np.array_equal(za[k], zb[k]), k
    prob = models.falc_h_ca(Nrays=3, lineScale=0.3)
    ups = []
    with Context(prob) as ctx:
        for _ in range(3):
            ups.append(ctx.formal_sol_gamma_matrices(deviceResident=True))
        ctx.download(abi.ALL_OUTPUTS)
    for rank in range(world):
        z = np.load(tmp_path / f'gpu{rank}.npz')
        la0, la1 = z['la']
        assert np.array_equal(z['J'], prob.J[la0:la1]) and np.array_equal(z['I'], prob.I[la0:la1])
        assert rel_err(z['G0'], prob.atoms[0].Gamma) <= 1e-12
        assert rel_err(z['G1'], prob.atoms[1].Gamma) <= 1e-12
        assert rel_err(z['R'], np.stack([t.Rij for t in prob.atoms[0].trans])) <= 1e-12
        assert [float(x) for x in z['dJ']] == [u.dJMax for u in ups]
        assert [int(x) for x in z['idx']] == [u.dJMaxIdx for u in ups]


# ---- PRD and hybrid PRD sub-iterations through the real collectives: two processes on one GPU ---------------------------
def _gpu_prd_worker(rank, world, port, out, hybrid):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lightweaver_amd.context import Context
        from oracle import bindings
        from test_hprd import hprd_problem
        prob = hprd_problem()
        bounds = shard_bounds(prob, world)
        la0, la1 = bounds[rank], bounds[rank + 1]
        tables = None
        if hybrid:
            oc = bindings.OracleContext(prob.copy())
            tables = oc.build_hprd()
        with Context(prob, device=0, laStart=la0, laEnd=la1, worldSize=world, worldRank=rank, hprd=tables) as ctx:
            it = ShardedIteration(ctx, world, rank)
            ups = []
            for _ in range(2):
                it.formal_sol_gamma_matrices(deviceResident=True)
                ups.append(it.prd_redistribute(2, 1e-30))
            ctx.download(abi.J | abi.RATES | abi.RHOPRD)
        lines = [t for a in prob.atoms for t in a.trans if t.rhoPrd is not None]
        np.savez(os.path.join(out, f'gprd{rank}.npz'), J=prob.J[la0:la1], la=[la0, la1],
                 R=np.stack([t.Rij for t in lines]), dRho=np.concatenate([u.dRho.ravel() for u in ups]),
                 **{f'rho{q}': t.rhoPrd for q, t in enumerate(lines)})
        if tables is not None:
            tables.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('hybrid', [False, True])
def test_two_process_gloo_prd_on_one_gpu(gpu, tmp_path, hybrid):
    """ShardedIteration.prd_redistribute with the real kernels and real (gloo) collectives: two processes, each a
    wavelength shard on the same GPU -- angle-averaged PRD, and hybrid PRD (round 4: rho of the lines' whole grids on every
    rank, the ranks' partial JRest sums in the all-reduce of the gather buffer) -- against the unsharded device run."""
    from lightweaver_amd.context import Context
    from oracle import bindings
    from test_hprd import hprd_problem
    world = 2
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_gpu_prd_worker, args=(world, port, str(tmp_path), hybrid), nprocs=world, join=True)
    prob = hprd_problem()
    tables = None
    if hybrid:
        oc = bindings.OracleContext(prob.copy())
        tables = oc.build_hprd()
    ups = []
    with Context(prob, hprd=tables) as ctx:
        for _ in range(2):
            ctx.formal_sol_gamma_matrices(deviceResident=True)
            ups.append(ctx.prd_redistribute(2, 1e-30, deviceResident=True))
        ctx.download(abi.J | abi.RATES | abi.RHOPRD)
    if tables is not None:
        tables.close()
    lines = [t for a in prob.atoms for t in a.trans if t.rhoPrd is not None]
    for rank in range(world):
        z = np.load(tmp_path / f'gprd{rank}.npz')
        la0, la1 = z['la']
        assert rel_err(z['J'], prob.J[la0:la1]) <= 1e-9
        assert rel_err(z['R'], np.stack([t.Rij for t in lines])) <= 1e-9
        assert np.allclose(z['dRho'], np.concatenate([u.dRho.ravel() for u in ups]), rtol=1e-6)
        for q, t in enumerate(lines):
            if hybrid:                   # the whole grid on every rank
                assert rel_err(z[f'rho{q}'], t.rhoPrd) <= 1e-9
            else:                        # the shard's rows
                lo, hi = max(la0 - t.Nblue, 0), max(min(la1, t.Nred) - t.Nblue, 0)
                assert rel_err(z[f'rho{q}'][lo:hi], t.rhoPrd[lo:hi]) <= 1e-9


# ---- the transport handshake of ShardedIteration (round 6) on CPU -----------------------------------------------------------
class _FakePeerCtx:
    """The surface _attach_peers touches, with a switch that makes one rank's attach fail."""
    lib = object()

    def __init__(self, rank, fail_attach=False, fail_export=False):
        self.rank, self.fail_attach, self.fail_export = rank, fail_attach, fail_export
        self.attached = self.detached = 0
        self.handles = None

    def set_stream(self, s):
        pass

    def peer_export(self):
        if self.fail_export:
            raise RuntimeError('no window')
        return bytes([self.rank]) * 64

    def peer_attach(self, handles):
        if self.fail_attach:
            raise RuntimeError('hipIpcOpenMemHandle failed')
        self.handles = [bytes(h) for h in handles]
        self.attached += 1

    def peer_detach(self):
        self.detached += 1

    def peer_selftest(self, timeout_ms=200):
        return self.selftest_result


def _handshake_worker(rank, world, port, out, scenario):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ctx = _FakePeerCtx(rank, fail_attach=(scenario == 'attach_fails_on_1' and rank == 1),
                           fail_export=(scenario == 'export_fails_on_0' and rank == 0))
        ctx.selftest_result = 2 if (scenario == 'selftest_fails_on_1' and rank == 1) else 0
        it = ShardedIteration.__new__(ShardedIteration)
        it.ctx, it.world, it.rank, it.group, it.transport = ctx, world, rank, None, 'collective'
        err = ''
        try:
            it._attach_peers(required=(scenario == 'required_but_fails' or False))
        except RuntimeError as e:
            err = str(e)
        np.savez(os.path.join(out, f'hs{rank}.npz'), transport=it.transport, attached=ctx.attached, detached=ctx.detached,
                 handles=np.frombuffer(b''.join(ctx.handles or []), dtype=np.uint8), err=err)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('scenario', ['ok', 'attach_fails_on_1', 'export_fails_on_0', 'selftest_fails_on_1'])
def test_peer_transport_handshake_is_all_or_nothing(tmp_path, scenario):
    """ShardedIteration sets the peer exchange up with two all_gather_object rounds: every rank's 64-byte window handle to every
    rank, then every rank's "I could open them all".  If ANY rank fails -- cannot export, cannot open a peer's window (ranks on
    different nodes), or the self-test exchange of a known pattern does not arrive intact on it (a peer's stores not visible
    to its kernels) -- EVERY rank stays on the collective (a rank that had attached detaches again): a mixed group would
    deadlock, half of it waiting for flags nobody raises."""
    world = 2
    port = 37500 + (os.getpid() % 2000)
    mp.spawn(_handshake_worker, args=(world, port, str(tmp_path), scenario), nprocs=world, join=True)
    z = [np.load(tmp_path / f'hs{r}.npz') for r in range(world)]
    if scenario == 'ok':
        for r in range(world):
            assert str(z[r]['transport']) == 'peer' and int(z[r]['attached']) == 1 and int(z[r]['detached']) == 0
            assert z[r]['handles'].tobytes() == bytes([0]) * 64 + bytes([1]) * 64      # rank order, every rank's own included
    else:
        for r in range(world):
            assert str(z[r]['transport']) == 'collective'
        if scenario == 'attach_fails_on_1':
            assert int(z[0]['attached']) == 1 and int(z[0]['detached']) == 1    # the rank that had attached let go again
            assert int(z[1]['attached']) == 0
        elif scenario == 'selftest_fails_on_1':
            # both had attached; the exchange of a known pattern did not arrive intact on rank 1: both detach
            assert [int(z[r]['attached']) for r in range(world)] == [1, 1] and [int(z[r]['detached']) for r in range(world)] == [1, 1]
        else:
            assert int(z[0]['attached']) == 0 and int(z[1]['attached']) == 0    # nobody tries with a handle missing

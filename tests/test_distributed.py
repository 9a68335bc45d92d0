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
        self.buf = torch.zeros(self.nsum + 2 * world, dtype=torch.float64)
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

    def fs_finalise(self):
        self.oc.fs_finalise_shard(self.buf[:self.nsum].numpy().copy())
        tail = self.buf[self.nsum:].numpy().reshape(self.world, 2)
        best = max(range(self.world), key=lambda r: (tail[r, 0], -tail[r, 1]))
        return IterationUpdate(updatedJ=True, dJMax=float(tail[best, 0]), dJMaxIdx=int(tail[best, 1]))


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        prob, _ = load_fixture('falc_h_ca_small')
        bounds = shard_bounds(prob, world)
        ctx = OracleShardContext(prob, bounds[rank], bounds[rank + 1], world, rank)
        it = ShardedIteration(ctx, world, rank)
        ups = []
        for _ in range(2):
            ups.append(it.formal_sol_gamma_matrices(deviceResident=True))
        la0, la1 = bounds[rank], bounds[rank + 1]
        np.savez(os.path.join(out, f'rank{rank}.npz'), J=prob.J[la0:la1], I=prob.I[la0:la1], la=[la0, la1],
                 G0=prob.atoms[0].Gamma, G1=prob.atoms[1].Gamma,
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

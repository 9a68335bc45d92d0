"""The fixed price of the peer exchange on ONE GPU: `world` shard contexts of this process (1 280 wavelengths each: the shard
of an 8-GPU run of the bench grid) attached to each other's windows by pointer, driven by one host thread in the order
partial(0..w-1), finalise(0..w-1).  Prints ms per round; under rocprofv3 --kernel-trace (tools/peer_timeline.sh) the kernel
timeline of the LAST rank gives sweep-end -> apply-start with every peer on time: reduce_stage2 + peer_publish + launch gaps.
usage: python3 tools/peer_probe.py [world=2] [Nlambda per shard=1280]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightweaver_amd.context import Context
from lightweaver_amd.distributed import shard_bounds
from lightweaver_amd.harness import models

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nla = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
prob = models.throughput_grid(world * nla, 5, computeProfiles=False)
b = shard_bounds(prob, world)
ctxs = [Context(prob.copy(), laStart=b[r], laEnd=b[r + 1], worldSize=world, worldRank=r) for r in range(world)]
for c in ctxs:
    c.compute_profiles(deviceResident=True)
wins = [c.peer_window()[0] for c in ctxs]
for c in ctxs:
    c.peer_attach_pointers(wins)

def round_():
    for c in ctxs:
        c.gamma_prefill_from_C(1.0)
        c.fs_partial()
    return [c.fs_finalise() for c in ctxs]

for _ in range(30):
    round_()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    ups = round_()
for c in ctxs:
    c.synchronize()
dt = time.perf_counter() - t0
print(f'{world} shard contexts x {[b[r + 1] - b[r] for r in range(world)]} wavelengths on one GPU, peer exchange: {1e3 * dt / n:.4f} ms per round '
      f'(dJMax {ups[0].dJMax:.6g})')
one = Context(prob.copy(), laStart=b[0], laEnd=b[1], worldSize=1, worldRank=0)
one.compute_profiles(deviceResident=True)
for _ in range(30):
    one.formal_sol_gamma_matrices(deviceResident=True)
t0 = time.perf_counter()
for _ in range(n):
    one.formal_sol_gamma_matrices(deviceResident=True)
one.synchronize()
print(f'one such shard alone, one-call form: {1e3 * (time.perf_counter() - t0) / n:.4f} ms per step')
one.close()
for c in ctxs:
    c.peer_detach()
    c.close()

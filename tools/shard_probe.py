"""Per-step time of the sharded iteration path on ONE GPU: a 1-rank RCCL ('nccl') group all-reduces the library's
reduce buffer between fs_partial and fs_finalise (what every rank of an N-GPU run does), next to the one-call form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.distributed import ShardedIteration
from lightweaver_amd.harness import models

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29611')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
NLA = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
prob = models.throughput_grid(NLA, 5, computeProfiles=False)
stream = torch.cuda.current_stream()
for force in (False, True):
    ctx = Context(prob.copy(), stream=stream.cuda_stream, worldSize=1, worldRank=0)
    ctx.compute_profiles(deviceResident=True)
    it = ShardedIteration(ctx, 1, 0, force_collective=force)
    for _ in range(20):
        it.formal_sol_gamma_matrices(deviceResident=True)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        it.formal_sol_gamma_matrices(deviceResident=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('Nlambda=%d' % NLA, 'sharded path with a 1-rank RCCL all-reduce' if force else 'one-call form', 'ms/step=%.4f' % (1e3 * dt / n),
          'allreduce host us/step=%.1f' % (1e6 * it.allreduce_s / (n + 20)))
    ctx.close()
dist.destroy_process_group()

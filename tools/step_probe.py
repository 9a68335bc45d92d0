"""Where the per-step time of bench.py goes: the same iteration driven through Context / ShardedIteration, on the
library's own stream or torch's current stream, with host or device generated profiles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.distributed import ShardedIteration
from lightweaver_amd.harness import models

def run(tag, prob, stream, devprof, sharded, n=100):
    kw = {} if stream is None else {'stream': stream}
    ctx = Context(prob, **kw)
    if devprof:
        ctx.compute_profiles(deviceResident=True)
    it = ShardedIteration(ctx, 1, 0) if sharded else ctx
    for _ in range(10):
        it.formal_sol_gamma_matrices(deviceResident=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        it.formal_sol_gamma_matrices(deviceResident=True)
    torch.cuda.synchronize()
    print(tag, 'ms/step=%.3f' % (1e3 * (time.perf_counter() - t0) / n))
    ctx.close()

ph = models.throughput_grid(10240, 5)
pd = models.throughput_grid(10240, 5, computeProfiles=False)
ts = torch.cuda.current_stream().cuda_stream
print('torch current stream handle', ts)
run('own stream, host phi, Context   ', ph, None, False, False)
run('own stream, device phi, Context ', pd.copy(), None, True, False)
run('torch stream, host phi, Context ', ph, ts, False, False)
run('torch stream, device phi, Shard ', pd.copy(), ts, True, True)
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    run('torch side stream, device phi   ', pd.copy(), s2.cuda_stream, True, True)

"""Where the drop-in call's time goes: each upload / download group timed alone (host clock, synchronised)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lightweaver_amd import _abi as abi
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models

def tm(f, n=30):
    for _ in range(3):
        f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return 1e3 * (time.perf_counter() - t0) / n

nla = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
prob = models.throughput_grid(nla, 5)
with Context(prob) as ctx:
    ctx.formal_sol_gamma_matrices()
    print(f'Nlambda={prob.Nlambda}')
    print(f'  host gamma_prefill      {tm(lambda: prob.gamma_prefill(ctx.crsw)):.3f} ms')
    for name in ('GAMMA', 'J', 'POPS', 'BC', 'RHOPRD'):
        m = getattr(abi, name)
        print(f'  upload   {name:10s}     {tm(lambda: ctx.upload(m)):.3f} ms')
    print(f'  upload   all five       {tm(lambda: ctx.upload(abi.GAMMA | abi.J | abi.POPS | abi.BC | abi.RHOPRD)):.3f} ms')
    for name in ('J', 'I', 'GAMMA', 'RATES', 'POPS'):
        m = getattr(abi, name)
        print(f'  download {name:10s}     {tm(lambda: ctx.download(m)):.3f} ms')
    print(f'  download ALL_OUTPUTS    {tm(lambda: ctx.download(abi.ALL_OUTPUTS)):.3f} ms')
    print(f'  resident iteration      {tm(lambda: ctx.formal_sol_gamma_matrices(deviceResident=True)):.3f} ms')
    print(f'  drop-in iteration       {tm(lambda: ctx.formal_sol_gamma_matrices()):.3f} ms')

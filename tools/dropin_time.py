"""Time the host-authoritative (drop-in) call: upload -> kernels -> download, per iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models
for nla in (3000, 10240):
    prob = models.throughput_grid(nla, 5)
    with Context(prob) as ctx:
        for _ in range(3):
            ctx.formal_sol_gamma_matrices()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            ctx.formal_sol_gamma_matrices()
        t1 = time.perf_counter()
        for _ in range(n):
            ctx.formal_sol_gamma_matrices(deviceResident=True)
        t2 = time.perf_counter()
        print(f'Nlambda={prob.Nlambda}: drop-in {1e3*(t1-t0)/n:.3f} ms/iter, device-resident {1e3*(t2-t1)/n:.3f} ms/iter')

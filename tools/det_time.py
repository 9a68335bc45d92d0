"""The deterministic mode (LWHIP_OPT_DETERMINISTIC) against the default: ms per device-resident iteration at the bench size and at a
shard size: python3 tools/det_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models

for nla in (2560, 10240):
    prob = models.throughput_grid(nla, 5)
    for det in (False, True):
        with Context(prob.copy(), deterministic=det) as ctx:
            f = lambda: ctx.formal_sol_gamma_matrices(deviceResident=True)
            for _ in range(100):
                f()
            t0 = time.perf_counter()
            for _ in range(100):
                f()
            print(f'Nlambda={prob.Nlambda} deterministic={det}: {10.0 * (time.perf_counter() - t0):.4f} ms per iteration')

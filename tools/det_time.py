"""The fixed-order mode (LWHIP_OPT_DETERMINISTIC) against the default: ms per device-resident iteration at the bench size and at a
shard size: python3 tools/det_time.py [default|det|both] [Nlambda ...]   (one mode per process under rocprofv3 --kernel-trace --stats)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models

mode = sys.argv[1] if len(sys.argv) > 1 else 'both'
sizes = [int(x) for x in sys.argv[2:]] or [2560, 10240]
for nla in sizes:
    prob = models.throughput_grid(nla, 5)
    for det in ((False, True) if mode == 'both' else (mode == 'det',)):
        with Context(prob.copy(), deterministic=det) as ctx:
            f = lambda: ctx.formal_sol_gamma_matrices(deviceResident=True)
            for _ in range(100):
                f()
            t0 = time.perf_counter()
            for _ in range(100):
                f()
            print(f'Nlambda={prob.Nlambda} deterministic={det}: {10.0 * (time.perf_counter() - t0):.4f} ms per iteration', flush=True)

"""Does a captured graph of one device-resident iteration (sweep -> reduce -> apply) run faster than the three stream launches?
python3 tools/graph_probe.py [Nlambda]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models

nla = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
prob = models.throughput_grid(nla, 5)
s = torch.cuda.Stream()
with Context(prob) as ctx:
    ctx.set_stream(s.cuda_stream)
    with torch.cuda.stream(s):
        def step():
            ctx.formal_sol_gamma_matrices(deviceResident=True, sync_host=False)
        for _ in range(80):
            step()
        s.synchronize()
        def timed(f, n=200):
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                f()
            s.synchronize()
            return 1e3 * (time.perf_counter() - t0) / n
        print(f'Nlambda={prob.Nlambda}: stream launches {timed(step):.4f} ms/step')
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
        for _ in range(20):
            g.replay()
        print(f'Nlambda={prob.Nlambda}: graph replay     {timed(g.replay):.4f} ms/step')
        print(f'Nlambda={prob.Nlambda}: stream launches {timed(step):.4f} ms/step')
        g10 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g10, stream=s):
            for _ in range(10):
                step()
        for _ in range(5):
            g10.replay()
        print(f'Nlambda={prob.Nlambda}: graph of 10      {timed(g10.replay, 40) / 10:.4f} ms/step')

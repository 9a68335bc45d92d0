"""Time the sweep with an alternative build of the library: python tools/bench_lib.py <lib.so> [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightweaver_amd import context
from lightweaver_amd.harness import models
lib = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
prob = models.throughput_grid(10240, 5)
ctx = context.Context(prob, lib_path=lib)
for _ in range(5):
    ctx.formal_sol_gamma_matrices(deviceResident=True)
ctx.profile_enable(True)
t0 = time.perf_counter()
for _ in range(steps):
    ctx.formal_sol_gamma_matrices(deviceResident=True)
dt = time.perf_counter() - t0
from lightweaver_amd import _abi as abi
ctx.download(abi.ALL_OUTPUTS)
print(os.path.basename(lib), 'it/s=%.0f' % (steps / dt), 'sweep_ms=%.3f' % ctx.sweep_time()[0],
      'J.sum=%.15e G.sum=%.15e' % (prob.J.sum(), prob.atoms[0].Gamma.sum()))
ctx.close()

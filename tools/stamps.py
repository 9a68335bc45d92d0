"""Diagnostic: run the stamped build of the sweep kernel and print per-phase cycle shares."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['LWHIP_DUMP_STAMPS'] = '1'
from lightweaver_amd import context
from lightweaver_amd.harness import models
from lightweaver_amd import build
lib = build.build_hip(defines=('LWHIP_STAMPS',), out=os.path.join(os.path.dirname(context.LIB_PATH), 'liblwhip_stamps.so'))
prob = models.throughput_grid(int(sys.argv[1]) if len(sys.argv) > 1 else 10240, 5)
ctx = context.Context(prob, lib_path=lib)
for _ in range(3):
    ctx.formal_sol_gamma_matrices(deviceResident=True)
ctx.close()

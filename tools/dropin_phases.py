"""The phases of the default drop-in call (LWHIP_PLUGIN_TIMING=1 prints them): python3 tools/dropin_phases.py [Nlambda]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['LWHIP_PLUGIN_TIMING'] = '1'
from lightweaver_amd import build
from lightweaver_amd.harness import models
from oracle.bindings import RefContext
nla = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
prob = models.throughput_grid(nla, 5)
rc = RefContext(prob, scheme=build.PLUGIN)
for _ in range(30):
    rc.formal_sol_gamma_matrices()
t0 = time.perf_counter()
for _ in range(20):
    rc.formal_sol_gamma_matrices()
print('ms per call (timing prints on): %.3f' % (1e3 * (time.perf_counter() - t0) / 20))

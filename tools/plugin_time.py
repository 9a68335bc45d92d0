"""The drop-in call as a Lightweaver user gets it -- the real core (oracle/_ref) driving the plugin -- per iteration, with
spect.J host-authoritative (the default) and with LWHIP_RESIDENT_J=1: python3 tools/plugin_time.py [Nlambda]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightweaver_amd import build
from lightweaver_amd.harness import models
from oracle.bindings import RefContext

nla = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
prob = models.throughput_grid(nla, 5)
rc = RefContext(prob, scheme=build.PLUGIN)


def timed(n=20):
    for _ in range(3):
        rc.formal_sol_gamma_matrices()
    t0 = time.perf_counter()
    for _ in range(n):
        rc.formal_sol_gamma_matrices()
    return 1e3 * (time.perf_counter() - t0) / n


print(f'Nlambda={prob.Nlambda}: plugin call, host-authoritative J  {timed():.3f} ms')
os.environ['LWHIP_RESIDENT_J'] = '1'
print(f'Nlambda={prob.Nlambda}: plugin call, LWHIP_RESIDENT_J=1     {timed():.3f} ms')
os.environ['LWHIP_SAMPLED_FINGERPRINTS'] = '1'
print(f'Nlambda={prob.Nlambda}: ... and LWHIP_SAMPLED_FINGERPRINTS=1  {timed():.3f} ms')
del os.environ['LWHIP_RESIDENT_J']
print(f'Nlambda={prob.Nlambda}: sampled fingerprints, host J        {timed():.3f} ms')
del os.environ['LWHIP_SAMPLED_FINGERPRINTS']
print(f'Nlambda={prob.Nlambda}: plugin call, host-authoritative J  {timed():.3f} ms')

"""The drop-in call as a Lightweaver user gets it -- the real core (oracle/_ref) driving the plugin -- per iteration, with
spect.J host-authoritative (the default), with the opt-in switches of INTEGRATION.md section 5, and with the statistical
equilibrium of all atoms after it: python3 tools/plugin_time.py [Nlambda]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightweaver_amd import build
from lightweaver_amd.harness import models
from oracle.bindings import RefContext

nla = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
prob = models.throughput_grid(nla, 5)
rc = RefContext(prob, scheme=build.PLUGIN)


def timed(n=20, se=False):
    def step():
        rc.formal_sol_gamma_matrices()
        if se:
            rc.stat_equil()
    for _ in range(3):
        step()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    return 1e3 * (time.perf_counter() - t0) / n


def show(what, **kw):
    print(f'Nlambda={prob.Nlambda}: {what:58s} {timed(**kw):.3f} ms')


show('plugin call, host-authoritative J')
show('formal solution + stat_equil of every atom', se=True)
os.environ['LWHIP_RESIDENT_J'] = '1'
show('LWHIP_RESIDENT_J=1')
os.environ['LWHIP_SAMPLED_FINGERPRINTS'] = '1'
show('LWHIP_RESIDENT_J=1 LWHIP_SAMPLED_FINGERPRINTS=1')
show('... + stat_equil of every atom', se=True)
del os.environ['LWHIP_RESIDENT_J']
show('LWHIP_SAMPLED_FINGERPRINTS=1')
del os.environ['LWHIP_SAMPLED_FINGERPRINTS']
show('plugin call, host-authoritative J')

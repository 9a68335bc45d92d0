import os, sys, time
sys.path.insert(0, os.getcwd())
import torch; torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models
from lightweaver_amd import _abi as abi
import ctypes as C
base = models.falc82()
n = 256
probs = [models.falc_h_ca(Nrays=5, lineScale=3.1, atmos=models.perturbed(base, seed=1234 + c), computeProfiles=False) for c in range(n)]
t0 = time.perf_counter()
for p in probs: p.descriptor()
t1 = time.perf_counter(); print('descriptors (serial)  %.3f s' % (t1 - t0))
c0 = Context(probs[0], batchHint=n)
t2 = time.perf_counter(); print('first context         %.3f s' % (t2 - t1))
from concurrent.futures import ThreadPoolExecutor
for nthr in (1, 4, 8, 16):
    t3 = time.perf_counter()
    if nthr == 1:
        cs = [Context(p, batchHint=n, like=c0) for p in probs[1:65]]
    else:
        with ThreadPoolExecutor(max_workers=nthr) as ex:
            cs = list(ex.map(lambda p: Context(p, batchHint=n, like=c0), probs[1:65]))
    t4 = time.perf_counter(); print('64 borrowers, %2d threads: %.3f s = %.2f ms each' % (nthr, t4 - t3, 1e3 * (t4 - t3) / 64))
    for c in cs: c.close()
c0.close()

# ---- the phases of a 512-column ColumnBatch (what bench.py's contexts_and_profiles_s is made of)
from lightweaver_amd.batch import ColumnBatch
import ctypes as C
n = 512
t0 = time.perf_counter()
probs = probs + [models.falc_h_ca(Nrays=5, lineScale=3.1, atmos=models.perturbed(base, seed=1234 + c), computeProfiles=False) for c in range(256, n)]
print('building %d more problems %.3f s' % (n - 256, time.perf_counter() - t0))
t0 = time.perf_counter()
c0 = Context(probs[0], batchHint=n)
t1 = time.perf_counter()
with ThreadPoolExecutor(max_workers=16) as ex:
    cs = list(ex.map(lambda p: Context(p, batchHint=n, like=c0), probs[1:]))
t2 = time.perf_counter()
lib = c0.lib
arr = (C.c_void_p * n)(*[c._h for c in [c0] + cs])
h = C.c_void_p()
assert lib.lwhip_batch_create(arr, n, C.byref(h)) == 0
t3 = time.perf_counter()
assert lib.lwhip_batch_compute_profiles(h) == 0
torch.cuda.synchronize()
t4 = time.perf_counter()
print('first %.3f | 511 borrowers %.3f | batch_create %.3f | profiles %.3f s' % (t1 - t0, t2 - t1, t3 - t2, t4 - t3))
lib.lwhip_batch_destroy(h)
for c in reversed(cs): c.close()
c0.close()

"""Which device allocations a borrower column still makes outside its arena (LWHIP_TRACE_ALLOC prints every hipMalloc of DevBuf).
usage: LWHIP_DEBUG=1 LWHIP_TRACE_ALLOC=1 python3 tools/borrower_allocs.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch; torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models
from lightweaver_amd.batch import ColumnBatch
base = models.falc82()
probs = [models.falc_h_ca(Nrays=5, lineScale=3.1, atmos=models.perturbed(base, seed=1234 + c), computeProfiles=False) for c in range(3)]
sys.stderr.write('== owner\n'); sys.stderr.flush()
c0 = Context(probs[0], batchHint=3)
sys.stderr.write('== borrower 1\n'); sys.stderr.flush()
c1 = Context(probs[1], batchHint=3, like=c0)
sys.stderr.write('== borrower 2\n'); sys.stderr.flush()
c2 = Context(probs[2], batchHint=3, like=c0)
sys.stderr.write('== batch create + profiles\n'); sys.stderr.flush()
import ctypes as C
lib = c0.lib
arr = (C.c_void_p * 3)(c0._h, c1._h, c2._h)
h = C.c_void_p()
assert lib.lwhip_batch_create(arr, 3, C.byref(h)) == 0
assert lib.lwhip_batch_compute_profiles(h) == 0
sys.stderr.write('== first batch iteration\n'); sys.stderr.flush()
res = None
assert lib.lwhip_batch_formal_sol_gamma_matrices(h, 0, 1.0, None) == 0
torch.cuda.synchronize()
sys.stderr.write('== done\n'); sys.stderr.flush()
lib.lwhip_batch_destroy(h)
for c in (c2, c1, c0): c.close()

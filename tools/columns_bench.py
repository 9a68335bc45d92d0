"""BASELINE configs[3] in miniature: a batch of independent, seeded-perturbed FAL-C columns (1.5D),
H + Ca II at ~3 000 wavelengths each, all resident in HBM, profiles generated on the device.
Reports column-iterations per second on one GPU (columns shard over GPUs with no collective)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from lightweaver_amd.batch import ColumnBatch
from lightweaver_amd.harness import models

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 1
base = models.falc82()
ls = 3.1  # lineScale giving ~3 000 wavelengths
t0 = time.perf_counter()
probs = [models.falc_h_ca(Nrays=5, lineScale=ls, atmos=models.perturbed(base, seed=1234 + c), computeProfiles=False)
         for c in range(ncol)]
t1 = time.perf_counter()
tstreams = [torch.cuda.Stream() for _ in range(nstreams)]
with ColumnBatch(probs, streams=[s.cuda_stream for s in tstreams] if nstreams > 1 else None) as batch:
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    batch.iterate(3)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    n = 10
    ups = batch.iterate(n)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    free, total = torch.cuda.mem_get_info()
print(f'streams={nstreams} columns={ncol} Nlambda={probs[0].Nlambda}: host build {t1-t0:.1f}s, contexts+device profiles {t2-t1:.1f}s; '
      f'{ncol*n/(t4-t3):.0f} column-iterations/s ({1e3*(t4-t3)/(ncol*n):.3f} ms each, incl. stat_equil from it 3); '
      f'HBM used {(total-free)/2**30:.1f} GiB; dJMax {max(u.dJMax for u in ups):.2e}')

import os, sys
sys.path.insert(0, os.getcwd())
import torch; torch.cuda.init()
from lightweaver_amd.context import Context
from lightweaver_amd.harness import models
base = models.falc82()
probs = [models.falc_h_ca(Nrays=5, lineScale=3.1, atmos=models.perturbed(base, seed=1234 + c), computeProfiles=False) for c in range(3)]
c0 = Context(probs[0], batchHint=512)
print('--- borrower', file=sys.stderr)
c1 = Context(probs[1], batchHint=512, like=c0)
c2 = Context(probs[2], batchHint=512, like=c0)
c2.close(); c1.close(); c0.close()

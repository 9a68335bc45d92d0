"""Control of tools/guard/hunt.sh: the process pattern of the flaky test WITHOUT liblwhip.so -- torch's HIP runtime does the
device work (streams made and destroyed, pinned blocks allocated and freed, small kernels, 4-byte and 1-KB copies to host),
then the problem is copied and the C oracle runs on the copy; its rates must equal those of an oracle run made BEFORE any
device work in the same process.  Exit code 1 = the deviation shows without the library (the writer is the runtime)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from lightweaver_amd.harness import models  # noqa: E402
from oracle.bindings import OracleContext  # noqa: E402


def oracle_rates(prob):
    q = prob.copy()
    q.gamma_prefill()
    oc = OracleContext(q)
    oc.formal_sol_gamma_matrices()
    oc.redistribute_prd(2, 0.0)
    return q


def main():
    assert 'lightweaver_amd.context' not in sys.modules
    atmos = models.perturbed(models.falc82(), seed=7)
    prob = models.falc_h_ca(Nrays=5, lineScale=0.4, prd=True, atmos=atmos)
    want = oracle_rates(prob)
    dev = torch.device('cuda:0')
    for rnd in range(3):            # (three "contexts")
        p = prob.copy()
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        bufs = []
        with torch.cuda.stream(streams[0]):
            for a in p.atoms:
                for t in a.trans:
                    bufs.append(torch.from_numpy(np.ascontiguousarray(t.wavelength)).to(dev, non_blocking=True))
            J = torch.from_numpy(p.J).to(dev)
            for _ in range(20):
                J = J * 1.0000001 + 1e-30
            pinned = torch.empty(1024, dtype=torch.float64).pin_memory()
            pinned.copy_(J.ravel()[:1024], non_blocking=True)
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            host_flag = flag.cpu()          # 4-byte copy to pageable memory
            small = J.ravel()[:128].cpu()   # 1 KB to pageable memory
        for s in streams:
            s.synchronize()
        del streams, bufs, J, pinned, flag, host_flag, small, p
        torch.cuda.empty_cache()
    got = oracle_rates(prob)
    bad = []
    for ia, (x, y) in enumerate(zip(got.atoms, want.atoms)):
        for kr, (tx, ty) in enumerate(zip(x.trans, y.trans)):
            if not (np.array_equal(tx.Rij, ty.Rij) and np.array_equal(tx.Rji, ty.Rji)):
                bad.append((ia, kr, float(np.max(np.abs(tx.Rij / ty.Rij - 1.0)))))
    if not np.array_equal(got.J, want.J):
        bad.append(('J',))
    if bad:
        print('GUARD control: the oracle deviates WITHOUT liblwhip.so in the process:', bad)
        return 1
    return 0


if __name__ == '__main__':
    sys.exit(main())

import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import torch; torch.cuda.init()
from tests.test_prd import prd_problem, prd_lines
from lightweaver_amd.context import Context
from lightweaver_amd import _abi as abi
hip = C.cdll.LoadLibrary('libamdhip64.so')
def d2h(ptr, n):
    host = np.zeros(n); hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(n * 8), 2); return host
def h2d(ptr, host):
    hip.hipMemcpy(C.c_void_p(ptr), host.ctypes.data_as(C.c_void_p), C.c_size_t(host.size * 8), 1)
def allreduce(ctxs, getbuf, show=None):
    bufs = []
    for c in ctxs:
        ptr, n = getbuf(c); c.synchronize(); bufs.append((ptr, d2h(ptr, n)))
    if show:
        for i, (_, b) in enumerate(bufs): print(show, 'rank', i, 'n', b.size, 'tail', b[-12:])
    total = sum(b for _, b in bufs)
    for ptr, _ in bufs: h2d(ptr, total)
prob = prd_problem(); lines = prd_lines(prob)
print('lines', [(t.Nblue, t.Nred) for t in lines], 'Nlambda', prob.Nlambda)
m = (lines[-1].Nblue + lines[-1].Nred) // 2
whole = prob.copy()
with Context(whole) as ctx:
    ctx.formal_sol_gamma_matrices(deviceResident=True)
    uw = ctx.prd_redistribute(1, 1e-30, deviceResident=True)
print('whole', uw.dRho, uw.dRhoMaxIdx, uw.dJPrdMax)
pa, pb = prob.copy(), prob.copy()
with Context(pa, laStart=0, laEnd=m, worldSize=2, worldRank=0) as A, Context(pb, laStart=m, laEnd=prob.Nlambda, worldSize=2, worldRank=1) as B:
    ctxs = (A, B)
    for c in ctxs:
        c.gamma_prefill_from_C(1.0); c.fs_partial()
    allreduce(ctxs, lambda c: c.reduce_buffer())
    for c in ctxs: c.fs_finalise()
    allreduce(ctxs, lambda c: c.prd_pack())
    for c in ctxs: c.prd_partial()
    allreduce(ctxs, lambda c: c.reduce_buffer(), show='after partial')
    for c in ctxs: print(c.prd_finalise())

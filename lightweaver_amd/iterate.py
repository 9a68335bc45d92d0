"""Host mirror of lightweaver.iterate_ctx_se (lightweaver/iterate_ctx.py:85-208) for `lightweaver_amd.Context`.

Same loop, same defaults and the same DefaultConvergenceCriteria (lightweaver/iterate_ctx.py:42-82): Nscatter
iterations of formal_sol_gamma_matrices alone, then formal solution + stat_equil (+ PRD sub-iterations) until
dJMax < JTol and max population change < popsTol (and, when rhoTol is given, the last sub-iteration's dRho <
rhoTol).  With deviceResident=True (the default here) J, populations, Gamma, rhoPrd never leave HBM; only
the three convergence numbers cross PCIe per iteration.
"""
from __future__ import annotations

from typing import Optional

from .context import Context, IterationUpdate


def dPopsMax(update: Optional[IterationUpdate]) -> float:
    """IterationUpdate.dPopsMax (lightweaver/iteration_update.py:87-98)."""
    if update is None or not update.dPops:
        return 0.0
    return max(update.dPops)


def dRhoMax(update: Optional[IterationUpdate]) -> float:
    """IterationUpdate.dRhoMax: the largest dRho of the final sub-iteration (:100-105)."""
    if update is None or update.NprdSubIter == 0 or update.dRho is None or update.dRho.size == 0:
        return 0.0
    return float(update.dRho[update.NprdSubIter - 1].max())


class DefaultConvergenceCriteria:
    """lightweaver.iterate_ctx.DefaultConvergenceCriteria (:42-82); crsw is fixed at 1 here."""

    def __init__(self, ctx: Context, JTol: float, popsTol: float, rhoTol: Optional[float]):
        self.ctx, self.JTol, self.popsTol, self.rhoTol = ctx, JTol, popsTol, rhoTol

    def is_converged(self, JUpdate, popsUpdate, prdUpdate) -> bool:
        updates = [JUpdate, popsUpdate] + ([prdUpdate] if prdUpdate is not None else [])
        ok = True
        for u in updates:
            ok = ok and (u.dJMax < self.JTol)
            ok = ok and (dPopsMax(u) < self.popsTol)
            if prdUpdate is not None and self.rhoTol is not None:
                ok = ok and (dRhoMax(u) < self.rhoTol)
        return ok


def iterate_ctx_se(ctx: Context, Nscatter: int = 3, NmaxIter: int = 2000, prd: bool = False,
                   JTol: float = 5e-3, popsTol: float = 1e-3, rhoTol: Optional[float] = None,
                   prdIterTol: float = 1e-2, maxPrdSubIter: int = 3, quiet: bool = True,
                   convergence=None, returnFinalConvergence: bool = False, deviceResident: bool = True):
    """Iterate `ctx` towards statistical equilibrium; returns the number of iterations taken (and the final
    updates when asked), like the reference.  Call ctx.download(...) afterwards when deviceResident."""
    conv = (convergence or DefaultConvergenceCriteria)(ctx, JTol, popsTol, rhoTol)
    it = 0
    for it in range(NmaxIter):
        JUpdate = ctx.formal_sol_gamma_matrices(deviceResident=deviceResident)
        if not quiet:
            print(f'-- Iteration {it}: dJ = {JUpdate.dJMax:.2e}')
        if it < Nscatter:
            continue
        popsUpdate = ctx.stat_equil(deviceResident=deviceResident)
        prdUpdate = None
        if prd:
            prdUpdate = ctx.prd_redistribute(maxIter=maxPrdSubIter, tol=prdIterTol, deviceResident=deviceResident)
        if not quiet:
            extra = f', PRD dRho = {dRhoMax(prdUpdate):.2e} ({prdUpdate.NprdSubIter} sub-iterations)' if prd else ''
            print(f'    pops delta = {dPopsMax(popsUpdate):.3e}{extra}')
        if conv.is_converged(JUpdate, popsUpdate, prdUpdate):
            final = [JUpdate, popsUpdate] + ([prdUpdate] if prdUpdate is not None else [])
            return (it, final) if returnFinalConvergence else it
    return (it, []) if returnFinalConvergence else it

// lwhip_api2d.hip -- host side of the C ABI (lwhip_host.h): the launch sequence of the 2D iteration and the 2D primitives
// (lwhip_formal_solver_2d, lwhip_build_intersections).
#include "lwhip_host.h"

namespace lwhip
{
// The 2D iteration of this context's wavelengths: batches of wavelengths through cont -> gather -> 2D formal
// solver -> rates (lwhip_2d.hip), the running sums in the reduce buffer, dJ in its tail.
// mode 0: the iteration (J, Gamma, rates); 1: formal_sol, all rays; 2: formal_sol, up-going rays only;
// 3: formal_sol_prd_update_rates (PrdTemplates.hpp:18-75) -- only the wavelengths that hold a PRD line, no depth data
int run_2d(lwhip_context* c, int lambdaIterate, int mode)
{
    const lwhip_grid2d& g = *c->prob.grid2d;
    const size_t Ns = c->Ns;
    const int nRaysActive = (mode == 2) ? c->Nrays : 2 * c->Nrays;
    const int32_t* rayList = (mode == 2) ? c->rayUp.p : c->rayAll.p;
    const size_t nRed = (size_t)c->Ntrans * 4 * Ns;
    const bool rates = (mode == 0 || mode == 3);
    if (rates)
    {
        HIP_TRY(hipMemsetAsync(c->red2d.p, 0, (size_t)c->groups2d * nRed * sizeof(double), c->stream));
        HIP_TRY(hipMemsetAsync(c->dJ.p, 0, (size_t)c->Nla * sizeof(double), c->stream)); // unvisited wavelengths: dJ = 0
    }
    Batch2dArgs a{};
    a.Ns = (int32_t)Ns;
    a.Nx = c->Nx;
    a.Nrays = c->Nrays;
    a.nRaysActive = nRaysActive;
    a.lambdaIterate = lambdaIterate;
    a.storeDepth = (c->prob.storeDepthData && mode != 3) ? 1 : 0;
    a.NlevTot = c->NlevTot;
    a.Natom = c->Natom;
    a.rayList = rayList;
    a.muz = c->muz.p;
    a.wmu = c->wmu.p;
    a.laHdr = c->laHdr.p;
    a.slots = c->slots.p;
    a.bgChi = c->bgChi.p;
    a.bgEta = c->bgEta.p;
    a.bgSca = c->bgSca.p;
    a.n = c->n.p;
    a.phi = c->phi.p;
    a.rho = c->rho.p;
    a.wphi = c->wphi.p;
    a.J = c->J.p;
    a.Iout = c->I.p;
    a.depthChi = c->depthChi.p;
    a.depthEta = c->depthEta.p;
    a.depthI = c->depthI.p;
    a.zDown = c->zDown.p;
    a.zUp = c->zUp.p;
    a.cs = (double2*)c->b2cs.p;
    a.I = c->b2I.p;
    a.PsiStar = c->b2Psi.p;
    a.red = c->red2d.p;
    a.redStride = nRed;
    a.nGroups = c->groups2d;
    a.maxRowsLa = c->maxRowsLa;
    a.slotTr = c->slotTrD.p;
    a.dJbits = (unsigned long long*)c->dJ.p;
    Fs2dArgs f{};
    f.Nx = g.Nx;
    f.rNx = 1.0f / (float)g.Nx;
    f.Nz = g.Nz;
    f.Nrays = g.Nrays;
    f.zLowerBc = g.zLowerBc;
    f.zUpperBc = g.zUpperBc;
    f.nRayCycle = nRaysActive;
    f.mux = c->g2mux.p;
    f.periodic = g.periodic ? 1 : 0;
    f.zNmuLow = c->prob.zLowerBc.Nmu;
    f.zNmuUp = c->prob.zUpperBc.Nmu;
    f.zbcLow = c->lowerBcData.p;
    f.zbcUp = c->upperBcData.p;
    f.zIdxLow = c->lowerIdx.p;
    f.zIdxUp = c->upperIdx.p;
    if (!g.periodic)
    {
        f.xNmuLow = g.xLowerBc->Nmu;
        f.xNmuUp = g.xUpperBc->Nmu;
        f.xbcLow = c->xbcLow.p;
        f.xbcUp = c->xbcUp.p;
        f.xIdxLow = c->xIdxLow.p;
        f.xIdxUp = c->xIdxUp.p;
    }
    f.temperature = c->temperature.p;
    f.uw = c->g2uw.p;
    f.dw = c->g2dw.p;
    f.uwS = c->g2uwS.p;
    f.dwS = c->g2dwS.p;
    f.uwA = c->g2uwA.p;
    f.dwA = c->g2dwA.p;
    f.nRec = (size_t)2 * g.Nrays * Ns;
    f.longCharIdx = c->g2long.p;
    f.substepOff = c->g2subOff.p;
    f.substeps = c->g2sub.p;
    f.NlongChar = g.NlongChar;
    f.lcOwner = c->g2lcOwner.p;
    f.lcBuf = g.NlongChar > 0 ? c->b2lc.p : nullptr;
    f.lcUpOnly = (mode == 2) ? 1 : 0;
    f.rays = rayList;
    f.cs = (const double2*)c->b2cs.p;
    f.I = c->b2I.p;
    f.PsiStar = c->b2Psi.p;
    f.coef = c->b2coef.p;
    f.cidx = c->b2idx.p;
    for (int la0 = 0; la0 < c->Nla;)
    {
        int nLa = std::min(c->batch2d, c->Nla - la0);
        if (mode == 3)
        {
            // the next run of wavelengths with a PRD line, at most one batch long
            while (la0 < c->Nla && !c->hdrHost[la0].hasPrd)
                ++la0;
            if (la0 >= c->Nla)
                break;
            nLa = 0;
            while (la0 + nLa < c->Nla && nLa < c->batch2d && c->hdrHost[la0 + nLa].hasPrd)
                ++nLa;
        }
        // continuum rows of the batch: a window of the global row numbering
        const auto lo = std::lower_bound(c->contLaHost.begin(), c->contLaHost.begin() + c->nContLa, la0);
        const auto hi = std::lower_bound(c->contLaHost.begin(), c->contLaHost.begin() + c->nContLa, la0 + nLa);
        const int contOff = (int)(lo - c->contLaHost.begin());
        const int nCont = (int)(hi - lo);
        double* rowsBase = c->rowsBuf.p;
        if (nCont > 0)
            rowsBase = c->rowsBuf.p - (size_t)c->hdrHost[*lo].rowBase * Ns;
        HIP_TRY(launch_cont_range(c->dargs.p, c->hargs, contOff, nCont, rowsBase, c->stream));
        a.la0 = la0;
        a.nLa = nLa;
        a.rows = rowsBase;
        HIP_TRY(launch_gather2d(a, c->maxL, c->stream));
        f.wavs = c->wavelength.p + la0;
        f.la0 = la0;
        f.nSolve = nLa * nRaysActive;
        HIP_TRY(launch_fs2d(f, nLa * nRaysActive, c->stream));
        if (rates)
            HIP_TRY(launch_rates2d(a, c->maxL, c->maxM, c->maxP, c->stream));
        else
            HIP_TRY(launch_iout2d(a, c->stream));
        la0 += nLa;
    }
    if (!rates)
        return LWHIP_OK;
    // the groups' sums, in a fixed order, into the reduce buffer
    HIP_TRY(launch_sum_groups(c->red2d.p, nRed, nRed, c->groups2d, c->stream));
    HIP_TRY(hipMemcpyAsync(c->red.p, c->red2d.p, nRed * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    // dJMax and its first index into this shard's slot of the tail (the slab stages have nothing to do)
    ReduceArgs r{};
    r.Ns = (int32_t)Ns;
    r.Ntrans = 0;
    r.red = c->red.p + (size_t)c->Ntrans * 4 * Ns;
    r.dJ = c->dJ.p;
    r.Nla = c->Nla;
    r.idxOffset = c->laStart;
    r.world = c->worldSize;
    r.rank = c->worldRank;
    HIP_TRY(launch_reduce_sum(r, c->stream));
    return LWHIP_OK;
}
}

extern "C" int lwhip_formal_solver_2d_impl(int device, const lwhip_grid2d* g, double wavelength, int nSolve,
                                           const int32_t* rays, const double* chi, const double* S, double* I,
                                           double* PsiStar, char* err, int errLen);

extern "C" int lwhip_build_intersections_impl(const lwhip_grid2d* grid, lwhip_intersection* uw, lwhip_intersection* dw,
                                              int32_t* longCharIdx, int32_t* substepOff, int32_t capLongChar,
                                              lwhip_intersection* substeps, int64_t capSubsteps, int32_t* nLongChar,
                                              int64_t* nSubsteps);

extern "C"
{
int lwhip_build_intersections(const lwhip_grid2d* grid, lwhip_intersection* uw, lwhip_intersection* dw,
                              int32_t* longCharIdx, int32_t* substepOff, int32_t capLongChar,
                              lwhip_intersection* substeps, int64_t capSubsteps, int32_t* nLongChar, int64_t* nSubsteps)
{
    if (!grid || !nLongChar || !nSubsteps || !grid->x || !grid->z || !grid->mux || !grid->muz)
        return fail(LWHIP_ERR_INVALID, "build_intersections: null argument");
    if (grid->Nx < 2 || grid->Nz < 2 || grid->Nrays < 1)
        return fail(LWHIP_ERR_INVALID, "build_intersections: need Nx >= 2, Nz >= 2, Nrays >= 1");
    if (uw && (!dw || !longCharIdx || !substepOff || (capSubsteps > 0 && !substeps)))
        return fail(LWHIP_ERR_INVALID, "build_intersections: incomplete output buffers");
    for (int m = 0; m < grid->Nrays; ++m)
        if (!(grid->muz[m] != 0.0))
            return fail(LWHIP_ERR_UNSUPPORTED, "build_intersections: a ray along x (muz = 0) never reaches a z plane");
    const int st = lwhip_build_intersections_impl(grid, uw, dw, longCharIdx, substepOff, capLongChar, substeps,
                                                  capSubsteps, nLongChar, nSubsteps);
    if (st == LWHIP_ERR_INVALID)
        return fail(st, "build_intersections: substepOff / substeps buffers too small");
    if (st != LWHIP_OK)
        return fail(st, "build_intersections: a long characteristic does not reach a z plane");
    return LWHIP_OK;
}

int lwhip_formal_solver_2d(int device, const lwhip_grid2d* grid, double wavelength, int nSolve, const int32_t* rays,
                           const double* chi, const double* S, double* I, double* PsiStar)
{
    if (!grid || !rays || !chi || !S || !I || !PsiStar || nSolve < 0)
        return fail(LWHIP_ERR_INVALID, "formal_solver_2d: null argument");
    if (!grid->periodic
        && (!grid->xLowerBc || !grid->xUpperBc || grid->xLowerBc->type != LWHIP_BC_CALLABLE || grid->xUpperBc->type != LWHIP_BC_CALLABLE
            || !grid->xLowerBc->idxs || !grid->xUpperBc->idxs || !grid->xLowerBc->bcData || !grid->xUpperBc->bcData))
        return fail(LWHIP_ERR_UNSUPPORTED, "formal_solver_2d: a grid without periodic x needs both x boundaries CALLABLE");
    if (grid->Nx < 2 || grid->Nz < 2 || grid->Nx > 1024)
        return fail(LWHIP_ERR_UNSUPPORTED, "formal_solver_2d: need 2 <= Nx <= 1024 and Nz >= 2");
    if (!grid->x || !grid->z || !grid->mux || !grid->muz || !grid->temperature || !grid->uw || !grid->dw
        || !grid->longCharIdx || !grid->substepOff || (grid->NlongChar > 0 && !grid->substeps))
        return fail(LWHIP_ERR_INVALID, "formal_solver_2d: incomplete grid description");
    if (grid->zLowerBc == LWHIP_BC_CALLABLE || grid->zUpperBc == LWHIP_BC_CALLABLE)
        return fail(LWHIP_ERR_UNSUPPORTED, "formal_solver_2d: CALLABLE z boundaries need a context (lwhip_problem carries their data)");
    if (!fs2d_long_chars_ok(grid))
        return fail(LWHIP_ERR_UNSUPPORTED, "formal_solver_2d: a long characteristic that does not end on a z plane");
    for (int p = 0; p < nSolve; ++p)
        if (rays[p] < 0 || rays[p] >= 2 * grid->Nrays)
            return fail(LWHIP_ERR_INVALID, "formal_solver_2d: ray index out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(LWHIP_ERR_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (nSolve == 0)
        return LWHIP_OK;
    char err[256] = { 0 };
    const int st = lwhip_formal_solver_2d_impl(device, grid, wavelength, nSolve, rays, chi, S, I, PsiStar, err, sizeof(err));
    if (st != LWHIP_OK)
        return fail(st, err);
    return LWHIP_OK;
}

}

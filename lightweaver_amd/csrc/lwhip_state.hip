// lwhip_state.hip -- host side of the C ABI (lwhip_host.h): device allocations of a context, the kernels' argument blocks,
// the launch sequence of one sweep, and the copies between the caller's arrays and HBM (lwhip_upload / lwhip_download).
#include "lwhip_host.h"

namespace lwhip
{
int alloc_state(lwhip_context* c)
{
    const size_t Ns = c->Ns, Nla = c->Nla, Nr = c->Nrays;
    // PRD lines of the active atoms in reference order (PrdTemplates.hpp:190-205) and the layout of the
    // J gather buffer: the full grid of every PRD line, line after line (the same on every shard)
    c->prdLines.clear();
    c->prdRowOff.clear();
    c->prdRowsTot = 0;
    for (int pass = 0; pass < (c->prdDetailed ? 2 : 1); ++pass) // the active atoms' lines, then the detailed atoms'
        for (int tr = 0; tr < c->Ntrans; ++tr)
        {
            const HostTrans& h = c->trans[tr];
            if (h.t.type == LWHIP_LINE && h.t.prd && h.rhoOff >= 0 && (c->atoms[h.atom].detailed != 0) == (pass == 1))
            {
                c->prdLines.push_back(tr);
                c->prdRowOff.push_back(c->prdRowsTot);
                c->prdRowsTot += h.t.Nred - h.t.Nblue;
            }
        }
    HIP_TRY(c->height.alloc(Ns));
    HIP_TRY(c->temperature.alloc(Ns));
    HIP_TRY(c->muz.alloc(2 * Nr)); // [muz | 1 / muz]
    HIP_TRY(c->wmu.alloc(Nr));
    HIP_TRY(c->wavelength.alloc(Nla));
    // (+ 8: the lane sweep reads a lane's block of depths with one wide load -- up to three doubles past the last row)
    HIP_TRY(c->bgChi.alloc(Nla * Ns + 8));
    HIP_TRY(c->bgEta.alloc(Nla * Ns + 8));
    HIP_TRY(c->bgSca.alloc(Nla * Ns + 8));
    HIP_TRY(c->J.alloc(Nla * Ns + 8));
    HIP_TRY(c->I.alloc(Nla * Nr * (size_t)c->Nx));
    if (c->prob.storeDepthData)
    {
        HIP_TRY(c->depthChi.alloc(Nla * Nr * 2 * Ns));
        HIP_TRY(c->depthEta.alloc(Nla * Nr * 2 * Ns));
        HIP_TRY(c->depthI.alloc(Nla * Nr * 2 * Ns));
    }
    HIP_TRY(c->vlosMu.alloc(Nr * Ns));
    HIP_TRY(c->vBroad.alloc((size_t)std::max(c->Natom, 1) * Ns));
    HIP_TRY(c->aDamp.alloc_zero((size_t)std::max(c->Nline, 1) * Ns)); // (zero = the value of a line without damping data)
    HIP_TRY(c->Qelast.alloc_zero((size_t)std::max(c->Nline, 1) * Ns));
    if (!c->is2d && c->batchHint > 1 && c->prob.vlosMu)
    {
        // a column of a 1.5D batch: the scratch of its device-side profiles (voigt_line_list, lwhip_api.hip -- same sizes) now,
        // out of the column's arena, instead of two allocations and a fill per column at the batch's first profile launch
        const size_t nLine = (size_t)std::max(c->Nline, 1);
        HIP_TRY(c->wphiScratch.alloc(nLine * 16 * Ns));
        HIP_TRY(c->wphiTicket.alloc_zero(nLine * ((Ns + 7) / 8 + 1)));
    }
    if (c->laneSweep)
    {
        // the depth-only pools the lane sweep reads, back to back (+ padding: a ray's last block reads past its row)
        const size_t nN = (size_t)c->NlevTot * Ns, nW = (size_t)std::max(c->Nline, 1) * Ns, nR = (size_t)std::max(c->Ncont, 1) * Ns;
        const size_t nG = (size_t)4 * (c->laneLR * c->laneD + 2);
        HIP_TRY(c->depArena.alloc(nN + nW + nR + nG + 16));
        HIP_TRY(c->depArena.clear_padding());
        c->n.view(c->depArena.p, nN);
        c->wphi.view(c->depArena.p + nN, nW);
        c->ratio.view(c->depArena.p + nN + nW, nR);
        c->geoT.view(c->depArena.p + nN + nW + nR, nG);
        HIP_TRY(c->bcPlanck.alloc(4 * Nla));
        HIP_TRY(c->bcPlanck.clear_padding());
    }
    else
    {
        HIP_TRY(c->n.alloc((size_t)c->NlevTot * Ns));
        HIP_TRY(c->ratio.alloc((size_t)std::max(c->Ncont, 1) * Ns));
        HIP_TRY(c->wphi.alloc((size_t)std::max(c->Nline, 1) * Ns));
    }
    HIP_TRY(c->nTotal.alloc((size_t)c->Natom * Ns));
    // (+ one ray's block: the lane sweep requests a ray's profile one ray ahead, after the last ray of the pool's last
    // wavelength that request reaches up to 2 Ns doubles past the end -- it is never used, but it must be mapped memory)
    HIP_TRY(c->phi.alloc((size_t)std::max<int64_t>(c->phiTot, 1) + 2 * (size_t)c->Ns + 8));
    HIP_TRY(c->rho.alloc((size_t)std::max<int64_t>(c->rhoTot, 1) + 8)); // (+ padding: reads past a row's end by a block of depths)
    HIP_TRY(c->Gamma.alloc((size_t)std::max<int64_t>(c->gammaTot, 1)));
    HIP_TRY(c->Cmat.alloc_zero((size_t)std::max<int64_t>(c->gammaTot, 1))); // (zero = no collisional pre-fill given)
    HIP_TRY(c->Rij.alloc((size_t)std::max(c->Ntrans, 1) * Ns));
    HIP_TRY(c->Rji.alloc((size_t)std::max(c->Ntrans, 1) * Ns));
    // reduce buffer: [Ntrans,4,Ns] partial sums followed by one (dJMax, idx) slot per shard
    HIP_TRY(c->red.alloc_zero((size_t)c->Ntrans * 4 * Ns + 2 * (size_t)c->worldSize * (1 + c->prdLines.size())));
    HIP_TRY(c->red8.alloc(c->is2d ? 8 : (size_t)std::max(c->Ntrans, 1) * LWHIP_REDUCE_SPLIT * 4 * Ns));
    c->gatherHost.assign(2 * (size_t)c->worldSize, 0.0);
    HIP_TRY(c->dJ.alloc(Nla));
    HIP_TRY(c->status.alloc(1));
    if (c->tiled)
    {
        HIP_TRY(c->geo.alloc(4 * Ns));
        c->ktStride = (4 + c->NlevTot + c->Nline + 1) / 2 * 2;
        HIP_TRY(c->kt.alloc((size_t)c->ktStride * (Ns + 2))); // one padding entry at each end
        HIP_TRY(c->rowsTile.alloc((size_t)std::max<int64_t>(c->rowsTileTot, 1)));
        if (c->laneSweep)
        {
            // the padding points of the row blocks (k >= Ns) are read: keep them finite
            HIP_TRY(c->rowsTile.clear_padding());
        }
        HIP_TRY(c->momTile.alloc((size_t)std::max<int64_t>(c->momTot, 1)));
        HIP_TRY(c->phiT.alloc((size_t)std::max<int64_t>(c->phiTTot, 1)));
        HIP_TRY(c->momTile.clear_padding());
    }
    if (c->is2d)
    {
        const lwhip_grid2d& g = *c->prob.grid2d;
        const size_t nSt = (size_t)g.Nrays * 2 * Ns;
        auto upv = [](auto& buf, const auto* host, size_t count) -> hipError_t {
            hipError_t e = buf.alloc(std::max<size_t>(count, 1));
            if (e != hipSuccess || !count)
                return e;
            return hipMemcpy(buf.p, host, count * sizeof(*host), hipMemcpyHostToDevice);
        };
        HIP_TRY(upv(c->g2mux, g.mux, (size_t)g.Nrays));
        HIP_TRY(upv(c->g2uw, g.uw, nSt));
        HIP_TRY(upv(c->g2dw, g.dw, nSt));
        {
            std::vector<double> su, sd;
            std::vector<int32_t> wu, wd;
            if (fs2d_records_packed(g.uw, nSt, g.Nx, g.Nz, su, wu) && fs2d_records_packed(g.dw, nSt, g.Nx, g.Nz, sd, wd))
            {
                HIP_TRY(c->g2uwS.upload(su));
                HIP_TRY(c->g2uwA.upload(wu));
                HIP_TRY(c->g2dwS.upload(sd));
                HIP_TRY(c->g2dwA.upload(wd));
            }
        }
        HIP_TRY(upv(c->g2long, g.longCharIdx, nSt));
        HIP_TRY(upv(c->g2subOff, g.substepOff, (size_t)g.NlongChar + 1));
        HIP_TRY(upv(c->g2sub, g.substeps, (size_t)g.substepOff[g.NlongChar]));
        if (g.NlongChar > 0)
        {
            // who owns each long characteristic: (ray, point)
            std::vector<int32_t> owner((size_t)2 * g.NlongChar, -1);
            for (size_t i = 0; i < nSt; ++i)
            {
                const int lc = g.longCharIdx[i];
                if (lc < 0)
                    continue;
                if (lc >= g.NlongChar)
                    return fail(LWHIP_ERR_INVALID, "grid2d: longCharIdx out of range");
                owner[2 * (size_t)lc] = (int32_t)(i / Ns);
                owner[2 * (size_t)lc + 1] = (int32_t)(i % Ns);
            }
            for (int lc = 0; lc < g.NlongChar; ++lc)
                if (owner[2 * (size_t)lc] < 0)
                    return fail(LWHIP_ERR_INVALID, "grid2d: a long characteristic belongs to no point");
            HIP_TRY(c->g2lcOwner.upload(owner));
        }
        // wavelengths per batch: per-ray chi, S, I, Psi* of a batch within ~8 GB of the 288.  Every kernel of a batch is
        // latency-bound per workgroup (82 sequential planes in fs2d), so the batch should fill the chip, and fewer,
        // larger batches mean fewer kernel tails (256 x 82 x 268 wavelengths: one batch of 3 GB, 0.1 ms less than two).
        const size_t perLa = (size_t)2 * Nr * Ns * (7 * sizeof(double) + sizeof(int32_t)); // + fs2d's point maps
        const size_t budget = (size_t)env_int("LWHIP_BATCH2D_MB", 8192) << 20;
        c->batch2d = (int)std::max<size_t>(1, std::min<size_t>(budget / perLa, Nla));
        c->batch2d = std::max(1, env_int("LWHIP_BATCH2D", c->batch2d));
        constexpr int R2D_WAVES_HOST = 3; // (= R2D_WAVES of lwhip_2d.hip)
        // rates2d: one thread per point per wavelength group; as many groups as keep every wavefront resident at the
        // kernel's 3 wavefronts per SIMD (256 x 82: 9 groups = 2 952 wavefronts on 3 072 slots; 7 were 2.2 per SIMD,
        // 10 spill into a second round)
        {
            int numCU = 256;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0)
                numCU = prop.multiProcessorCount;
            const size_t wavesPerGroup = 2 * ((Ns + 127) / 128);
            c->groups2d = (int)std::max<size_t>(1, std::min<size_t>({ (size_t)16, (size_t)c->batch2d, (size_t)numCU * 4 * R2D_WAVES_HOST / wavesPerGroup }));
        }
        c->groups2d = std::max(1, c->groups2d);
        HIP_TRY(c->red2d.alloc((size_t)c->groups2d * std::max(c->Ntrans, 1) * 4 * Ns));
        const size_t nb = (size_t)c->batch2d * 2 * Nr * Ns;
        HIP_TRY(c->b2cs.alloc(2 * nb)); // (chi, S) pairs
        HIP_TRY(c->b2I.alloc(nb));
        HIP_TRY(c->b2Psi.alloc(nb));
        HIP_TRY(c->b2coef.alloc(3 * nb));
        HIP_TRY(c->b2idx.alloc(nb));
        if (g.NlongChar > 0)
            HIP_TRY(c->b2lc.alloc((size_t)c->batch2d * g.NlongChar * 3));
        for (size_t la = 0; la < Nla; ++la)
            if (c->hdrHost[la].rowBase >= 0)
                c->maxRowsLa = std::max(c->maxRowsLa, c->hdrHost[la].nRows + c->hdrHost[la].nMixed + c->hdrHost[la].nPure);
        if ((size_t)c->maxRowsLa * 128 * sizeof(double) > 64 * 1024)
            return fail(LWHIP_ERR_UNSUPPORTED, "2D: more continuum rows at one wavelength than the rate kernel's LDS block holds");
        // continuum rows of one batch: the most any window of batch2d consecutive wavelengths holds (the PRD
        // rates pass starts its batches at the PRD lines, not at multiples of batch2d)
        int64_t maxRows = 1;
        {
            std::vector<int64_t> pre(Nla + 1, 0);
            for (size_t la = 0; la < Nla; ++la)
                pre[la + 1] = pre[la] + (c->hdrHost[la].rowBase >= 0 ? c->hdrHost[la].nRows + c->hdrHost[la].nMixed + c->hdrHost[la].nPure
                                                                     : 0);
            for (size_t la = 0; la < Nla; ++la)
                maxRows = std::max(maxRows, pre[std::min(Nla, la + (size_t)c->batch2d)] - pre[la]);
        }
        HIP_TRY(c->rowsBuf.alloc((size_t)maxRows * Ns));
    }
    if (c->prob.zLowerBc.type == LWHIP_BC_CALLABLE)
    {
        HIP_TRY(c->lowerBcData.alloc(Nla * c->prob.zLowerBc.Nmu * (size_t)c->Nx));
        HIP_TRY(c->lowerIdx.alloc(Nr * 2));
    }
    if (c->prob.zUpperBc.type == LWHIP_BC_CALLABLE)
    {
        HIP_TRY(c->upperBcData.alloc(Nla * c->prob.zUpperBc.Nmu * (size_t)c->Nx));
        HIP_TRY(c->upperIdx.alloc(Nr * 2));
    }
    return LWHIP_OK;
}

// argument block of the 2D pipeline's continuum-row kernel (lwhip_2d.hip)
int build_sweep_args(lwhip_context* c)
{
    if (!c->is2d)
        return LWHIP_OK;
    ContArgs a{};
    a.Ns = c->Ns;
    a.maxC = c->maxC;
    a.temperature = c->temperature.p;
    a.wavelength = c->wavelength.p;
    a.bgChi = c->bgChi.p;
    a.bgEta = c->bgEta.p;
    a.n = c->n.p;
    a.ratio = c->ratio.p;
    a.laHdr = c->laHdr.p;
    a.slots = c->slots.p;
    a.progs = c->progs.p;
    a.progRows = c->progRows.p;
    a.progEnts = c->progEnts.p;
    a.rows = c->rowsBuf.p;
    a.contLa = c->contLa.p;
    a.nContLa = c->nContLa;
    c->hargs = a;
    std::vector<ContArgs> v(1, a);
    HIP_TRY(c->dargs.upload(v));
    return LWHIP_OK;
}

int build_tile_args(lwhip_context* c)
{
    if (!c->tiled)
        return LWHIP_OK;
    TileArgs a{};
    a.Ns = c->Ns;
    a.Nmu = c->Nrays;
    a.L = c->tileL;
    a.nTiles = c->nTiles;
    a.Natom = c->Natom;
    a.NlevTot = c->NlevTot;
    a.Ntrans = c->Ntrans;
    a.maxCT = c->maxCTTile;
    a.maxC = c->maxC;
    a.lowerBc = c->prob.zLowerBc.type;
    a.upperBc = c->prob.zUpperBc.type;
    a.lowerNmu = c->prob.zLowerBc.Nmu;
    a.upperNmu = c->prob.zUpperBc.Nmu;
    a.storeDepth = c->prob.storeDepthData ? 1 : 0;
    a.Nla = c->Nla;
    a.Nline = c->Nline;
    a.ktStride = c->ktStride;
    a.kt = c->kt.p;
    a.tiles = c->tiles.p;
    a.tslots = c->tslots.p;
    a.copies = c->tcopies.p;
    a.tileRemap = c->tileRemap.p;
    a.chunkTile = c->chunkTile.p;
    a.chunkOrder = c->chunkOrderOn ? c->chunkOrder.p : nullptr;
    a.chunkSplit = c->chunkSplitOn ? c->chunkSplit.p : nullptr;
    a.tileList = nullptr;
    a.slotTr = c->tileSlotTr.p;
    a.postChunkTile = c->postChunkTile.p;
    a.postSlotTr = c->postSlotTr.p;
    a.postCs = c->postCs.p;
    a.contRec = c->contRec.p;
    a.maxCTPost = c->maxCTPost;
    a.preCols = c->preCols;
    a.geo = c->geo.p;
    a.temperature = c->temperature.p;
    a.muz = c->muz.p;
    a.wmu = c->wmu.p;
    a.wavelength = c->wavelength.p;
    a.lowerIdx = c->lowerIdx.p;
    a.upperIdx = c->upperIdx.p;
    a.lowerBcData = c->lowerBcData.p;
    a.upperBcData = c->upperBcData.p;
    a.bgChi = c->bgChi.p;
    a.bgEta = c->bgEta.p;
    a.bgSca = c->bgSca.p;
    a.J = c->J.p;
    a.I = c->I.p;
    a.depthChi = c->depthChi.p;
    a.depthEta = c->depthEta.p;
    a.depthI = c->depthI.p;
    a.zDown = c->zDown.p;
    a.zUp = c->zUp.p;
    a.n = c->n.p;
    a.ratio = c->ratio.p;
    a.wphi = c->wphi.p;
    a.phiT = c->phiT.p;
    a.rho = c->rho.p;
    a.rows = c->rowsTile.p;
    a.mom = c->momTile.p;
    a.laHdr = c->laHdr.p;
    a.slots = c->slots.p;
    a.progs = c->progs.p;
    a.progRows = c->progRows.p;
    a.progEnts = c->progEnts.p;
    a.red8 = c->red8.p;
    a.dJ = c->dJ.p;
    a.Jhost = c->JhostDev;
    a.depthSplit = c->depthSplit;
    if (c->laneSweep)
    {
        a.laneD = c->laneD;
        a.laneLR = c->laneLR;
        a.laneR = c->laneR;
        a.laneLRD = c->laneLR * c->laneD;
        a.laneSplit = c->laneSplit;
        a.phi = c->phi.p;
        a.geoT = c->geoT.p;
        a.postProg = c->postProg.p;
        a.phiDoubles = c->phi.n;
        a.rhoDoubles = c->rho.n;
        a.depBase = c->depArena.p;
        a.depBytes = (uint32_t)(c->depArena.n * sizeof(double));
        a.depN = (uint32_t)((c->n.p - c->depArena.p) * sizeof(double));
        a.depW = (uint32_t)((c->wphi.p - c->depArena.p) * sizeof(double));
        a.depRat = (uint32_t)((c->ratio.p - c->depArena.p) * sizeof(double));
        a.depG = (uint32_t)((c->geoT.p - c->depArena.p) * sizeof(double));
        a.hRho = c->hRho.p;
        a.hLa2h = c->hLa2h.p;
        a.hJOff = c->hJOff.p;
        a.hJCoef = c->hJCoef.p;
        a.JRest = c->JRest.p;
        a.detSlab = c->deterministic ? c->detSlab.p : nullptr;
        a.turnLate = c->deterministic ? c->turnLateDev : nullptr;
        a.laneWg = c->laneWg.p;
        a.laneTiles = c->laneTiles.p;
        a.laneRays = c->laneRays.p;
        a.laneFin = c->laneFin.p;
        a.laneFinPar = c->laneFinPar.p;
        a.bcPlanck = c->bcPlanck.p;
        a.laneFeedG = c->laneFeedG.p;
        a.momS = c->momScratch.p;
        a.momA = c->momA;
    }
    a.lsDbg = nullptr;
    if (std::getenv("LWHIP_LS_TIMING"))
    {
        // phase clocks of a -DLS_TIMING / -DRM_TIMING build: host-mapped, dumped by lwhip_destroy
        if (!c->lsDbgHost)
        {
            // [nTiles][8] per tile, then [workgroup][4]: prologue, tasks, epilogue clocks (lane sweep)
            HIP_TRY(hipHostMalloc((void**)&c->lsDbgHost, (size_t)c->nTiles * 24 * sizeof(double), hipHostMallocMapped));
            std::memset(c->lsDbgHost, 0, (size_t)c->nTiles * 24 * sizeof(double));
        }
        HIP_TRY(hipHostGetDevicePointer((void**)&a.lsDbg, c->lsDbgHost, 0));
    }
    // The device copies of the argument blocks keep their addresses for the life of the context (a fused column batch
    // holds them): the first call allocates, later ones (lwhip_set_zplane_outputs) overwrite in place on the stream.
    auto put = [&](DevBuf<TileArgs>& dev, TileArgs& host, const TileArgs& val) -> hipError_t {
        host = val;
        if (!dev.p)
        {
            hipError_t e = dev.alloc(1);
            if (e != hipSuccess)
                return e;
        }
        return h2d_copy(dev.p, &host, sizeof(TileArgs), c->stream);
    };
    HIP_TRY(put(c->dtargs, c->htargs, a));
    if (c->nTileChunksPrd > 0)
    {
        a.depthSplit = 1; // (the PRD rates pass keeps whole columns per wavefront)
        a.tslots = c->tslotsPrd.p;
        a.chunkTile = c->chunkTilePrd.p;
        a.tileList = c->tileListPrd.p;
        a.chunkOrder = nullptr;
        a.chunkSplit = nullptr;
        a.laneSplit = c->laneSplitPrd;
        a.slotTr = c->tileSlotTrPrd.p;
        a.postChunkTile = c->postChunkTilePrd.p;
        a.laneWg = c->laneWgPrd.p;
        a.laneTiles = c->laneTilesPrd.p;
        HIP_TRY(put(c->dtargsPrd, c->htargsPrd, a));
    }
    return LWHIP_OK;
}

// the tile-blocked copy of the line profiles follows every change of phi
bool retile_args(lwhip_context* c, RetileArgs& r)
{
    if (!c->tiled || c->Nline == 0 || c->laneSweep) // (the lane sweep reads the profiles where they lie)
        return false;
    r = RetileArgs{};
    r.Ns = c->Ns;
    r.Nmu = c->Nrays;
    r.L = c->tileL;
    r.nTiles = c->nTiles;
    r.tiles = c->tiles.p;
    r.tslots = c->tslots.p;
    r.trans = c->dtrans.p;
    r.phi = c->phi.p;
    r.phiT = c->phiT.p;
    return true;
}

int retile_profiles(lwhip_context* c)
{
    RetileArgs r;
    if (retile_args(c, r))
        HIP_TRY(launch_retile(r, c->stream));
    return LWHIP_OK;
}

TileDyn make_dyn(lwhip_context* c, bool upOnly, int lambdaIterate)
{
    TileDyn d{};
    d.nPass = upOnly ? 1 : 2;
    d.lambdaIterate = lambdaIterate;
    d.stopIter = 0;
    d.stopCtl = nullptr;
    d.phiSym = (c && c->phiSym && c->pairRays) ? 1 : 0;
    return d;
}

// one sweep over this context's wavelengths (rates = the full iteration, else formal_sol): pre-pass (the tiles' row
// blocks, the depth table), the march, and -- unless the sweep's workgroups finish their tiles themselves -- the post-pass
hipError_t run_sweep(lwhip_context* c, const TileDyn& dyn, bool rates, hipEvent_t e0, hipEvent_t e1)
{
    TileDyn td = dyn;
    const bool tprd = dyn.prdOnly != 0; // the PRD rates pass: its own tile list and chunking
    if (tprd && c->nTileChunksPrd == 0)
        return hipSuccess;
    if (!tprd)
        c->dJPrdClean = false; // (every wavelength's dJ is written)
    const TileArgs* da = tprd ? c->dtargsPrd.p : c->dtargs.p;
    const TileArgs& ha = tprd ? c->htargsPrd : c->htargs;
    const bool fuse = c->tileFuse && rates;
    if (rates && c->hprd && c->JRest.p)
    {
        // hybrid PRD: the rest-frame mean intensity is rebuilt by every pass that updates J
        // (SimdFullIterationTemplates.hpp:602-603, PrdTemplates.hpp:57-58)
        hipError_t ez = hipMemsetAsync(c->JRest.p, 0, c->JRest.n * sizeof(double), c->stream);
        if (ez != hipSuccess)
            return ez;
    }
    if (rates && c->zeroCheckDev)
    {
        // diagnosis: the sweep's workgroups ADD into the stage-1 buffer -- it has to hold zeros now
        hipError_t ec = launch_check_zero(c->red8.p, c->red8.n, c->zeroCheckDev, c->stream);
        if (ec != hipSuccess)
            return ec;
    }
    // (the lane sweep's tasks do their own pre-pass; the ray-column march has it as a launch of its own)
    hipError_t e = c->laneSweep ? hipSuccess : launch_tile_pre(da, ha, tprd ? c->nTilesPrd : c->nTiles, nullptr, 0, c->stream);
    if (e != hipSuccess)
        return e;
    if (e0) // lwhip_profile_enable: HIP events around the sweep kernel alone
        (void)hipEventRecord(e0, c->stream);
    if (c->laneSweep)
        e = launch_lane_sweep(da, ha, td, c->prob.formalSolver, rates, tprd ? c->nTileChunksPrd : c->nTileChunks, c->tileWaves, nullptr, 0,
                              c->stream);
    else
        e = launch_tile_sweep(da, ha, td, c->prob.formalSolver, c->tileCap, rates, fuse, tprd ? c->nTileChunksPrd : c->nTileChunks,
                              (fuse && ha.depthSplit > 1) ? 2 * ha.depthSplit : c->tileWaves, nullptr, 0, c->stream);
    if (e1)
        (void)hipEventRecord(e1, c->stream);
    if (e == hipSuccess && rates && c->laneSweep && c->deterministic)
        e = launch_det_reduce(c->detSlab.p, tprd ? c->detOffPrd.p : c->detOff.p, tprd ? c->detEntPrd.p : c->detEnt.p, c->detPart.p, c->Ns,
                              c->Ntrans, c->stream);
    if (e != hipSuccess || !rates || fuse || c->laneSweep)
        return e;
    return launch_tile_post(da, ha, td, tprd ? c->nPostChunksPrd : c->nPostChunks, nullptr, 0, c->stream);
}

int flush_prefill(lwhip_context* c)
{
    if (c->prefillPending && c->gammaTot > 0)
        HIP_TRY(launch_prefill(c->Gamma.p, c->Cmat.p, c->prefillCrsw, (size_t)c->gammaTot, c->stream));
    c->prefillPending = false;
    return LWHIP_OK;
}

// diagnosis (LWHIP_CHECK_ZERO=1): called where the host has just waited for the device
int verify_zero_check(lwhip_context* c)
{
    if (c->zeroCheck && *(volatile int*)c->zeroCheck != 0)
    {
        const int n = *(volatile int*)c->zeroCheck;
        *(volatile int*)c->zeroCheck = 0;
        return fail(LWHIP_ERR_DEVICE, "LWHIP_CHECK_ZERO: the stage-1 reduce buffer held " + std::to_string(n)
                                          + " non-zero words when a sweep started adding to it");
    }
    return LWHIP_OK;
}

int collect_profile(lwhip_context* c)
{
    for (auto& pr : c->pending)
    {
        HIP_TRY(hipEventSynchronize(pr.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        c->sweepMs += ms;
        c->sweepCount++;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    c->pending.clear();
    return LWHIP_OK;
}
}

extern "C"
{
namespace
{
// pinned staging blocks for the large uploads of column batches: taken from a pool for the duration of one lwhip_upload (the
// call waits for its stream before it returns), given back afterwards; at most one per concurrently creating thread exists
struct BigStage
{
    double* p = nullptr;
    size_t n = 0;
    static std::mutex& mtx()
    {
        static std::mutex m;
        return m;
    }
    static std::vector<std::pair<double*, size_t>>& pool()
    {
        static std::vector<std::pair<double*, size_t>>* v = new std::vector<std::pair<double*, size_t>>();
        return *v;
    }
    explicit BigStage(size_t doubles)
    {
        if (!doubles || doubles * sizeof(double) > ((size_t)64 << 20))
            return;
        {
            std::lock_guard<std::mutex> g(mtx());
            auto& v = pool();
            for (size_t i = 0; i < v.size(); ++i)
                if (v[i].second >= doubles)
                {
                    p = v[i].first;
                    n = v[i].second;
                    v.erase(v.begin() + (long)i);
                    return;
                }
        }
        if (hipHostMalloc((void**)&p, doubles * sizeof(double), hipHostMallocDefault) != hipSuccess)
        {
            (void)hipGetLastError();
            p = nullptr; // (no staging: the plain copies)
            return;
        }
        n = doubles;
    }
    ~BigStage()
    {
        if (!p)
            return;
        std::lock_guard<std::mutex> g(mtx());
        pool().emplace_back(p, n);
    }
};
}
static int upload_impl(lwhip_context* c, uint32_t mask);
int lwhip_upload(lwhip_context* c, uint32_t mask)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    // (the sources of the queued copies are the caller's arrays and this call's own temporaries: on an error path too the
    // stream has drained before either can go away)
    const int st = upload_impl(c, mask);
    if (st != LWHIP_OK)
        (void)hipStreamSynchronize(c->stream);
    return st;
}
static int upload_impl(lwhip_context* c, uint32_t mask)
{
    HIP_TRY(hipSetDevice(c->device));
    const lwhip_problem& p = c->prob;
    const size_t Ns = c->Ns, Nla = c->Nla, Nr = c->Nrays;
    const size_t l0 = c->laStart;
    std::vector<double> rmuzHost; // (lives until the synchronisation at the end)
    bool phiSymUp = true;         // (LWHIP_PROFILES: the uploaded profiles are the same for both directions of every angle)
    if (mask & LWHIP_GAMMA)
        c->prefillPending = false; // the host pre-fill being uploaded supersedes it
    if (mask & LWHIP_J)
        c->fpJValid = false;
    if (mask & LWHIP_ATMOS)
    {
        H2D(c->height.p, p.height, Ns);
        H2D(c->temperature.p, p.temperature, Ns);
        H2D(c->muz.p, p.muz, Nr);
        rmuzHost.resize(Nr);
        for (size_t m = 0; m < Nr; ++m)
            rmuzHost[m] = 1.0 / p.muz[m]; // zmu of the 1D solvers (FormalScalar.cpp:136-666), once instead of per ray and lane
        H2D(c->muz.p + Nr, rmuzHost.data(), Nr);
        H2D(c->wmu.p, p.wmu, Nr);
        H2D(c->wavelength.p, p.wavelength + l0, Nla);
        if (p.vlosMu)
        {
            H2D(c->vlosMu.p, p.vlosMu, Nr * Ns);
            c->vlosZero = true;
            for (size_t e = 0; e < Nr * Ns && c->vlosZero; ++e)
                c->vlosZero = p.vlosMu[e] == 0.0;
        }
        else
            c->vlosZero = false;
        if (c->tiled)
        {
            // per-depth geometry of the ray-column sweep (scalar loads): |h_k - h_k+1|, its reciprocal,
            // 1 / (|h_k-1 - h_k| + |h_k - h_k+1|), 1 / T_k
            std::vector<double> geo(4 * Ns, 0.0);
            for (size_t k = 0; k < Ns; ++k)
            {
                if (k + 1 < Ns)
                {
                    geo[4 * k] = std::fabs(p.height[k] - p.height[k + 1]);
                    geo[4 * k + 1] = 1.0 / geo[4 * k];
                }
                if (k >= 1 && k + 1 < Ns)
                    geo[4 * k + 2] = 1.0 / (std::fabs(p.height[k - 1] - p.height[k]) + std::fabs(p.height[k] - p.height[k + 1]));
                geo[4 * k + 3] = 1.0 / p.temperature[k];
            }
            HIP_TRY(h2d_copy(c->geo.p, geo.data(), geo.size() * sizeof(double), c->stream));
            std::vector<double> geoT;
            if (c->laneSweep)
            {
                // the same per quantity, entry k + 1 (zero padding at both ends): what a lane reads for its block of depths
                const size_t st = (size_t)c->laneLR * c->laneD + 2;
                geoT.assign(4 * st, 0.0);
                for (size_t k = 0; k < Ns; ++k)
                    for (int q = 0; q < 4; ++q)
                        geoT[q * st + k + 1] = geo[4 * k + q];
                HIP_TRY(h2d_copy(c->geoT.p, geoT.data(), geoT.size() * sizeof(double), c->stream));
            }
            std::vector<double> bcB;
            if (c->laneSweep)
            {
                // the thermalised boundaries' Planck functions (FormalScalar.cpp:471-666 -> planck_nu, LwMisc.hpp:29-46) of
                // every wavelength at the two uppermost and the two lowermost points: once per atmosphere, not per task
                auto planck = [](double T, double lambda) {
                    const double hc_kla = HC_K / lambda;
                    const double twohnu3_c2 = TWO_HC_NM3 / (lambda * lambda * lambda);
                    const double x = hc_kla / T;
                    return x <= 150.0 ? twohnu3_c2 / (std::exp(x) - 1.0) : 0.0;
                };
                bcB.resize(4 * Nla);
                const size_t kk[4] = { 0, 1, Ns - 2, Ns - 1 };
                for (size_t la = 0; la < Nla; ++la)
                    for (int q = 0; q < 4; ++q)
                        bcB[4 * la + q] = planck(p.temperature[kk[q]], p.wavelength[l0 + la]);
                if (c->bcPlanck.n < bcB.size())
                    HIP_TRY(c->bcPlanck.alloc(bcB.size()));
                HIP_TRY(h2d_copy(c->bcPlanck.p, bcB.data(), bcB.size() * sizeof(double), c->stream));
            }
            if (!h2d_batch()) // (a gathered upload has copied them already)
                HIP_TRY(hipStreamSynchronize(c->stream)); // `geo`, `bcB` are locals
        }
    }
    // Column batches (round 6): the four [Nla, Ns] arrays of a column -- 7.6 MB at 2 908 wavelengths -- go through a pinned
    // staging block of this call (a pool of them, one per creating thread at a time): the host copy runs in parallel on the
    // creating threads, while four pageable hipMemcpyAsync per column took ~1.5 ms each INSIDE the runtime's lock -- the
    // serial part of setting up 512 columns.
    BigStage big(c->batchHint > 1 && !h2d_batch() ? (size_t)4 * Nla * Ns : 0);
    auto h2d_big = [&](double* dst, const double* src, size_t count, size_t slot) -> hipError_t {
        if (h2d_batch())
            return h2d_copy(dst, src, count * sizeof(double), c->stream);
        if (!big.p)
            return hipMemcpyAsync(dst, src, count * sizeof(double), hipMemcpyHostToDevice, c->stream);
        double* st = big.p + slot * Nla * Ns;
        std::memcpy(st, src, count * sizeof(double));
        return hipMemcpyAsync(dst, st, count * sizeof(double), hipMemcpyHostToDevice, c->stream);
    };
    if (mask & LWHIP_BACKGROUND)
    {
        HIP_TRY(h2d_big(c->bgChi.p, p.bgChi + l0 * Ns, Nla * Ns, 0));
        HIP_TRY(h2d_big(c->bgEta.p, p.bgEta + l0 * Ns, Nla * Ns, 1));
        HIP_TRY(h2d_big(c->bgSca.p, p.bgSca + l0 * Ns, Nla * Ns, 2));
    }
    if (mask & LWHIP_J)
        HIP_TRY(h2d_big(c->J.p, p.J + l0 * Ns, Nla * Ns, 3));
    if (mask & LWHIP_BC)
    {
        if (p.zLowerBc.type == LWHIP_BC_CALLABLE)
        {
            H2D(c->lowerBcData.p, p.zLowerBc.bcData + l0 * p.zLowerBc.Nmu * c->Nx, Nla * p.zLowerBc.Nmu * (size_t)c->Nx);
            HIP_TRY(hipMemcpyAsync(c->lowerIdx.p, p.zLowerBc.idxs, Nr * 2 * sizeof(int32_t),
                                   hipMemcpyHostToDevice, c->stream));
        }
        if (p.zUpperBc.type == LWHIP_BC_CALLABLE)
        {
            H2D(c->upperBcData.p, p.zUpperBc.bcData + l0 * p.zUpperBc.Nmu * c->Nx, Nla * p.zUpperBc.Nmu * (size_t)c->Nx);
            HIP_TRY(hipMemcpyAsync(c->upperIdx.p, p.zUpperBc.idxs, Nr * 2 * sizeof(int32_t),
                                   hipMemcpyHostToDevice, c->stream));
        }
        if (c->is2d && !p.grid2d->periodic)
        {
            const lwhip_grid2d& g = *p.grid2d;
            const size_t nLow = (size_t)g.xLowerBc->Nmu * g.Nz, nUp = (size_t)g.xUpperBc->Nmu * g.Nz;
            // the plugin re-borrows these arrays on every call: the index tables are checked again against the (possibly
            // changed) Nmu of each side, and each side's buffer follows its own size
            for (int q = 0; q < 2 * g.Nrays; ++q)
                if (g.xLowerBc->idxs[q] < 0 || g.xLowerBc->idxs[q] >= g.xLowerBc->Nmu || g.xUpperBc->idxs[q] < 0
                    || g.xUpperBc->idxs[q] >= g.xUpperBc->Nmu)
                    return fail(LWHIP_ERR_INVALID, "2D grid: x boundary idxs out of range (\"Error in boundary condition indexing\")");
            if (c->xbcLow.n < Nla * nLow || c->xbcUp.n < Nla * nUp || !c->xIdxLow.p)
                HIP_TRY(hipStreamSynchronize(c->stream)); // nothing may still read a buffer about to be replaced
            if (c->xbcLow.n < Nla * nLow)
                HIP_TRY(c->xbcLow.alloc(Nla * nLow));
            if (c->xbcUp.n < Nla * nUp)
                HIP_TRY(c->xbcUp.alloc(Nla * nUp));
            if (!c->xIdxLow.p)
            {
                HIP_TRY(c->xIdxLow.alloc(Nr * 2));
                HIP_TRY(c->xIdxUp.alloc(Nr * 2));
            }
            H2D(c->xbcLow.p, g.xLowerBc->bcData + l0 * nLow, Nla * nLow);
            H2D(c->xbcUp.p, g.xUpperBc->bcData + l0 * nUp, Nla * nUp);
            HIP_TRY(hipMemcpyAsync(c->xIdxLow.p, g.xLowerBc->idxs, Nr * 2 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->xIdxUp.p, g.xUpperBc->idxs, Nr * 2 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        }
    }
    if (mask & LWHIP_NSTAR)
        std::fill(c->gIIValid.begin(), c->gIIValid.end(), 0); // aDamp / vBroad may have changed
    std::vector<double> ratioHost;
    if (mask & LWHIP_NSTAR)
        ratioHost.assign((size_t)std::max(c->Ncont, 1) * Ns, 1.0);
    // staging layout (doubles): [n pool][Gamma pool][C pool][Rij][Rji][nTotal][vBroad][aDamp][Qelast]
    // (round 6: the per-atom / per-line rows of the LWHIP_NSTAR group and C ride in it too -- a column of a 1.5D batch made
    // ~40 pageable 656-byte copies for them at creation, each a driver call)
    const size_t nPool = (size_t)c->NlevTot * Ns, gPool = (size_t)c->gammaTot, rPool = (size_t)c->Ntrans * Ns;
    const size_t aPool = (size_t)std::max(c->Natom, 1) * Ns, lPool = (size_t)std::max(c->Nline, 1) * Ns;
    if (!c->stage)
    {
        c->stageDoubles = nPool + 2 * gPool + 2 * rPool + 2 * aPool + 2 * lPool + 8;
        c->stage = (double*)pinned_acquire(c->device, c->stageDoubles * sizeof(double), nullptr);
        if (!c->stage)
            return fail(LWHIP_ERR_DEVICE, "lwhip_upload: no page-locked memory for the upload stage");
    }
    double* stN = c->stage;
    double* stG = stN + nPool;
    double* stC = stG + gPool;
    double* stRij = stC + gPool;
    double* stRji = stRij + rPool;
    double* stNT = stRji + rPool;
    double* stVB = stNT + aPool;
    double* stAD = stVB + aPool;
    double* stQE = stAD + lPool;
    bool anyR = false;
    // (a pool goes up as one copy when every row of it is given; otherwise row by row as before)
    bool allC = (mask & LWHIP_COLLISIONS) != 0, allAD = (mask & LWHIP_NSTAR) != 0, allQE = (mask & LWHIP_NSTAR) != 0;
    for (int ia = 0; ia < c->Natom; ++ia)
    {
        allC = allC && (c->atoms[ia].detailed || c->atoms[ia].C != nullptr);
        for (int kr = 0; kr < c->atoms[ia].Ntrans; ++kr)
        {
            const lwhip_transition& t = c->trans[c->atomTrOff[ia] + kr].t;
            if (t.type == LWHIP_LINE)
            {
                allAD = allAD && t.aDamp != nullptr;
                allQE = allQE && t.Qelast != nullptr;
            }
        }
    }
    if (allC)
        std::memset(stC, 0, gPool * sizeof(double));
    for (int ia = 0; ia < c->Natom; ++ia)
    {
        const lwhip_atom& a = c->atoms[ia];
        if (mask & LWHIP_POPS)
            std::memcpy(stN + (size_t)c->levelOff[ia] * Ns, a.n, (size_t)a.Nlevel * Ns * sizeof(double));
        if (mask & LWHIP_NSTAR)
        {
            std::memcpy(stNT + (size_t)ia * Ns, a.nTotal, Ns * sizeof(double));
            std::memcpy(stVB + (size_t)ia * Ns, a.vBroad, Ns * sizeof(double));
        }
        if (!a.detailed && (mask & LWHIP_GAMMA))
            std::memcpy(stG + c->gammaOff[ia], a.Gamma, (size_t)a.Nlevel * a.Nlevel * Ns * sizeof(double));
        if (!a.detailed && (mask & LWHIP_COLLISIONS) && a.C)
        {
            if (allC)
                std::memcpy(stC + c->gammaOff[ia], a.C, (size_t)a.Nlevel * a.Nlevel * Ns * sizeof(double));
            else // atoms without C keep whatever the pool holds: copy atom by atom, not the whole pool
                H2D(c->Cmat.p + c->gammaOff[ia], a.C, (size_t)a.Nlevel * a.Nlevel * Ns);
        }
        for (int kr = 0; kr < a.Ntrans; ++kr)
        {
            const HostTrans& h = c->trans[c->atomTrOff[ia] + kr];
            const lwhip_transition& t = h.t;
            const size_t nlt = h.NredLoc - h.NblueLoc;
            if ((mask & LWHIP_RATES) && t.Rij && t.Rji)
            {
                const size_t tr = c->atomTrOff[ia] + kr;
                std::memcpy(stRij + tr * Ns, t.Rij, Ns * sizeof(double));
                std::memcpy(stRji + tr * Ns, t.Rji, Ns * sizeof(double));
                anyR = true;
            }
            if (t.type == LWHIP_LINE)
            {
                if (mask & LWHIP_PROFILES)
                {
                    if (nlt)
                        H2D(c->phi.p + h.phiOff, t.phi + (size_t)h.ltStart * Nr * 2 * Ns, nlt * Nr * 2 * Ns);
                    // phi [lt][mu][down / up][k]: the same for both directions of every angle?  (one pass over what is uploaded)
                    for (size_t e = 0; e < nlt * Nr && phiSymUp; ++e)
                    {
                        const double* ph = t.phi + ((size_t)h.ltStart * Nr + e) * 2 * Ns;
                        phiSymUp = std::memcmp(ph, ph + Ns, Ns * sizeof(double)) == 0;
                    }
                    H2D(c->wphi.p + (size_t)h.row * Ns, t.wphi, Ns);
                }
                if ((mask & LWHIP_NSTAR) && t.aDamp)
                {
                    if (allAD)
                        std::memcpy(stAD + (size_t)h.row * Ns, t.aDamp, Ns * sizeof(double));
                    else
                        H2D(c->aDamp.p + (size_t)h.row * Ns, t.aDamp, Ns);
                }
                if ((mask & LWHIP_NSTAR) && t.Qelast)
                {
                    if (allQE)
                        std::memcpy(stQE + (size_t)h.row * Ns, t.Qelast, Ns * sizeof(double));
                    else
                        H2D(c->Qelast.p + (size_t)h.row * Ns, t.Qelast, Ns);
                }
                if ((mask & LWHIP_RHOPRD) && t.prd && h.rhoOff >= 0 && h.rhoRows > 0)
                    H2D(c->rho.p + h.rhoOff, t.rhoPrd + (size_t)h.rhoLt0 * Ns, (size_t)h.rhoRows * Ns);
            }
            else if (mask & LWHIP_NSTAR)
            {
                // nStar_i / nStar_j, evaluated exactly as LwAtom.hpp:112 does before the exp factor
                const double* ni = a.nStar + (size_t)t.i * Ns;
                const double* nj = a.nStar + (size_t)t.j * Ns;
                for (size_t k = 0; k < Ns; ++k)
                    ratioHost[(size_t)h.row * Ns + k] = ni[k] / nj[k];
            }
        }
    }
    if ((mask & LWHIP_NSTAR) && c->Natom > 0)
    {
        H2D(c->nTotal.p, stNT, (size_t)c->Natom * Ns);
        H2D(c->vBroad.p, stVB, (size_t)c->Natom * Ns);
        if (allAD && c->Nline > 0)
            H2D(c->aDamp.p, stAD, (size_t)c->Nline * Ns);
        if (allQE && c->Nline > 0)
            H2D(c->Qelast.p, stQE, (size_t)c->Nline * Ns);
    }
    if (allC && gPool)
        H2D(c->Cmat.p, stC, gPool);
    if ((mask & LWHIP_POPS) && nPool)
        H2D(c->n.p, stN, nPool);
    if ((mask & LWHIP_GAMMA) && gPool)
        H2D(c->Gamma.p, stG, gPool);
    if (anyR)
    {
        // only complete sets are uploaded this way (every transition has host Rij/Rji in this ABI)
        H2D(c->Rij.p, stRij, rPool);
        H2D(c->Rji.p, stRji, rPool);
    }
    if (mask & LWHIP_NSTAR)
        H2D(c->ratio.p, ratioHost.data(), ratioHost.size());
    if (mask & LWHIP_PROFILES)
    {
        if (H2DBatch* b = h2d_batch())
            HIP_TRY(b->flush()); // (kernels that read what was uploaded follow)
        const int str = retile_profiles(c);
        if (str != LWHIP_OK)
            return str;
        c->deviceProfiles = c->profilesStale = false; // the host's profiles are authoritative again
        c->phiSym = phiSymUp;
    }
    else if ((mask & (LWHIP_ATMOS | LWHIP_NSTAR)) && c->deviceProfiles)
        c->profilesStale = true; // phi / wphi follow the new velocities and widths before the next sweep
    if (!h2d_batch()) // (a gathered upload holds copies of the sources; lwhip_create sends it and waits)
        HIP_TRY(hipStreamSynchronize(c->stream));
    return LWHIP_OK;
}

static int download_impl(lwhip_context* c, uint32_t mask);
int lwhip_download(lwhip_context* c, uint32_t mask)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    // The caller's arrays are the targets of queued device-to-host copies: whatever path the call leaves by, none of them is
    // still in flight when the caller has its arrays back (it may free them the next moment).
    const int st = download_impl(c, mask);
    if (st != LWHIP_OK)
        (void)hipStreamSynchronize(c->stream);
    return st;
}
static int download_impl(lwhip_context* c, uint32_t mask)
{
    bool fpPending = false;
    HIP_TRY(hipSetDevice(c->device));
    const lwhip_problem& p = c->prob;
    const size_t Ns = c->Ns, Nla = c->Nla, Nr = c->Nrays;
    const size_t l0 = c->laStart;
    if (mask & LWHIP_GAMMA)
    {
        int stp = flush_prefill(c);
        if (stp != LWHIP_OK)
            return stp;
    }
    if (mask & LWHIP_J)
    {
        if (!c->JhostDev) // (lwhip_map_host_J: the sweep has stored J into the caller's array already)
            D2H(p.J + l0 * Ns, c->J.p, Nla * Ns);
        else if (!c->fpJValid)
        {
            // ... and the caller gets the fingerprint of what is there now without reading it (lwhip_fingerprint_J)
            const int stf = fingerprint_J_enqueue(c);
            if (stf != LWHIP_OK)
                return stf;
            fpPending = true;
        }
        if (c->hprd && c->JRest.p)
            D2H(c->hprd->JRest, c->JRest.p, c->JRest.n);
    }
    if (mask & LWHIP_I)
    {
        D2H(p.I + l0 * Nr * c->Nx, c->I.p, Nla * Nr * (size_t)c->Nx);
        // (an up-only formal solution does not trace the down rays: ZPlaneDown keeps what the caller has,
        // SimdFullIterationTemplates.hpp:351-361)
        if (c->zDownHost && !c->lastSweepUpOnly)
            D2H(c->zDownHost + l0 * Nr * c->Nx, c->zDown.p, Nla * Nr * (size_t)c->Nx);
        if (c->zUpHost)
            D2H(c->zUpHost + l0 * Nr * c->Nx, c->zUp.p, Nla * Nr * (size_t)c->Nx);
    }
    if ((mask & LWHIP_DEPTHDATA) && p.storeDepthData)
    {
        D2H(p.depthChi + l0 * Nr * 2 * Ns, c->depthChi.p, Nla * Nr * 2 * Ns);
        D2H(p.depthEta + l0 * Nr * 2 * Ns, c->depthEta.p, Nla * Nr * 2 * Ns);
        D2H(p.depthI + l0 * Nr * 2 * Ns, c->depthI.p, Nla * Nr * 2 * Ns);
    }
    const size_t nPool = (size_t)c->NlevTot * Ns, gPool = (size_t)c->gammaTot, rPool = (size_t)c->Ntrans * Ns;
    if (!c->stage)
    {
        // (the same size as lwhip_upload's layout: whoever comes first allocates)
        c->stageDoubles = nPool + 2 * gPool + 2 * rPool + 2 * (size_t)std::max(c->Natom, 1) * Ns + 2 * (size_t)std::max(c->Nline, 1) * Ns + 8;
        c->stage = (double*)pinned_acquire(c->device, c->stageDoubles * sizeof(double), nullptr);
        if (!c->stage)
            return fail(LWHIP_ERR_DEVICE, "lwhip_upload: no page-locked memory for the upload stage");
    }
    double* stN = c->stage;
    double* stG = stN + nPool;
    double* stRij = stG + 2 * gPool;
    double* stRji = stRij + rPool;
    if ((mask & LWHIP_POPS) && nPool)
        D2H(stN, c->n.p, nPool);
    if ((mask & LWHIP_GAMMA) && gPool)
        D2H(stG, c->Gamma.p, gPool);
    if ((mask & LWHIP_RATES) && rPool)
    {
        D2H(stRij, c->Rij.p, rPool);
        D2H(stRji, c->Rji.p, rPool);
    }
    for (int ia = 0; ia < c->Natom; ++ia)
    {
        for (int kr = 0; kr < c->atoms[ia].Ntrans; ++kr)
        {
            const int tr = c->atomTrOff[ia] + kr;
            const HostTrans& h = c->trans[tr];
            if ((mask & LWHIP_RHOPRD) && h.t.type == LWHIP_LINE && h.t.prd && h.rhoOff >= 0 && h.rhoRows > 0)
                D2H(h.t.rhoPrd + (size_t)h.rhoLt0 * Ns, c->rho.p + h.rhoOff, (size_t)h.rhoRows * Ns);
            if ((mask & LWHIP_PROFILES) && h.t.type == LWHIP_LINE)
            {
                const size_t nlt = h.NredLoc - h.NblueLoc;
                if (nlt)
                    D2H(h.t.phi + (size_t)h.ltStart * Nr * 2 * Ns, c->phi.p + h.phiOff, nlt * Nr * 2 * Ns);
                D2H(h.t.wphi, c->wphi.p + (size_t)h.row * Ns, Ns);
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (fpPending)
        fingerprint_J_fold(c, c->prob.J + (size_t)c->laStart * c->Ns);
    {
        const int stz = verify_zero_check(c);
        if (stz != LWHIP_OK)
            return stz;
    }
    // unpack the staged pools into the caller's arrays
    for (int ia = 0; ia < c->Natom; ++ia)
    {
        const lwhip_atom& a = c->atoms[ia];
        if (mask & LWHIP_POPS)
            std::memcpy(a.n, stN + (size_t)c->levelOff[ia] * Ns, (size_t)a.Nlevel * Ns * sizeof(double));
        if (!a.detailed && (mask & LWHIP_GAMMA))
            std::memcpy(a.Gamma, stG + c->gammaOff[ia], (size_t)a.Nlevel * a.Nlevel * Ns * sizeof(double));
        if (mask & LWHIP_RATES)
        {
            for (int kr = 0; kr < a.Ntrans; ++kr)
            {
                const size_t tr = c->atomTrOff[ia] + kr;
                const HostTrans& h = c->trans[tr];
                std::memcpy(h.t.Rij, stRij + tr * Ns, Ns * sizeof(double));
                std::memcpy(h.t.Rji, stRji + tr * Ns, Ns * sizeof(double));
            }
        }
    }
    return LWHIP_OK;
}
}

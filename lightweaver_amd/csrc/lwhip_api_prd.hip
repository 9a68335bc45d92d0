// lwhip_api_prd.hip -- host side of the C ABI (lwhip_host.h): the PRD sub-iterations.
#include "lwhip_host.h"

namespace
{
enum { PRD_PIPE_DEPTH = 4 }; // sub-iterations queued per host round trip (lwhip_redistribute_prd)
}
extern "C"
{
void prd_read_results(lwhip_context* c, const double* pinned, double* dRho, int32_t* dRhoMaxIdx, double* dJMax, int32_t* dJMaxIdx);
// ---- PRD sub-iterations.  One sub-iteration = pack (J of the PRD lines into the gather buffer) ->
// [all-reduce of that buffer across wavelength shards] -> partial (scattering integral for the shard's
// emission wavelengths, max change per line into the shard's slots of the reduce tail, the PRD rates
// pass + slab reduce) -> [all-reduce of the reduce buffer] -> finalise (rates of the PRD lines out,
// results read back).  lwhip_redistribute_prd runs the loop on one device.
int lwhip_prd_pack(lwhip_context* c, void** devPtr, size_t* count)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    if (c->partialPending || c->prdPending)
        return fail(LWHIP_ERR_INVALID, "lwhip_prd_pack inside an unfinished iteration / sub-iteration");
    HIP_TRY(hipSetDevice(c->device));
    const size_t Ns = c->Ns;
    const size_t n = (size_t)std::max<int64_t>(c->prdRowsTot, 1) * Ns;
    if (c->prdJ.n < n)
        HIP_TRY(c->prdJ.alloc(n));
    if (c->worldSize > 1)
        HIP_TRY(hipMemsetAsync(c->prdJ.p, 0, n * sizeof(double), c->stream)); // rows of other shards: the all-reduce sums
    for (size_t q = 0; q < c->prdLines.size(); ++q)
    {
        const HostTrans& h = c->trans[c->prdLines[q]];
        // the local mean intensity of the scattering integral: J of the line's wavelengths, or -- hybrid PRD -- the rows
        // of the rest-frame JRest they map to (consecutive: every wavelength of a PRD line is in prdIdxs; Prd.cpp:384-389).
        // A shard contributes its own wavelengths' J rows; of JRest, which every wavelength of the shard has added to, its
        // partial sums of ALL rows of the line (the all-reduce of the gather buffer then yields the totals)
        size_t nlt = h.NredLoc - h.NblueLoc, lt0 = h.ltStart;
        const double* src = c->J.p + (size_t)h.NblueLoc * Ns;
        if (c->hprd)
        {
            lt0 = 0;
            nlt = h.t.Nred - h.t.Nblue;
            src = c->JRest.p + (size_t)c->hLa2prdHost[h.t.Nblue] * Ns;
        }
        // (one device: nothing to exchange -- the scattering integral reads these rows where they lie, lwhip_prd_partial)
        if (nlt && c->worldSize > 1)
            HIP_TRY(hipMemcpyAsync(c->prdJ.p + ((size_t)c->prdRowOff[q] + lt0) * Ns, src, nlt * Ns * sizeof(double),
                                   hipMemcpyDeviceToDevice, c->stream));
    }
    if (devPtr)
        *devPtr = c->prdJ.p;
    if (count)
        *count = (size_t)c->prdRowsTot * Ns;
    return LWHIP_OK;
}

int lwhip_prd_partial(lwhip_context* c)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    if (c->partialPending || c->prdPending)
        return fail(LWHIP_ERR_INVALID, "lwhip_prd_partial inside an unfinished iteration / sub-iteration");
    HIP_TRY(hipSetDevice(c->device));
    c->fpJValid = false; // (the rates pass rewrites J at the PRD wavelengths)
    {
        const int stp = ensure_profiles(c);
        if (stp != LWHIP_OK)
            return stp;
    }
    const int Ns = c->Ns;
    const int Nprd = (int)c->prdLines.size();
    if (Nprd == 0)
        return LWHIP_OK;
    if (c->prdJ.n < (size_t)c->prdRowsTot * Ns)
        return fail(LWHIP_ERR_INVALID, "lwhip_prd_partial before lwhip_prd_pack");
    for (int tr : c->prdLines)
        if (!c->atoms[c->trans[tr].atom].C && !c->atoms[c->trans[tr].atom].detailed)
            return fail(LWHIP_ERR_INVALID, "redistribute_prd needs the collisional rates C of atom "
                                           + std::to_string(c->trans[tr].atom));
    if (c->prdChange.n < (size_t)Nprd * Ns * PRD_MAX_SLICES)
    {
        HIP_TRY(c->prdChange.alloc((size_t)Nprd * Ns * PRD_MAX_SLICES));
        HIP_TRY(c->prdChangeIdx.alloc((size_t)Nprd * Ns * PRD_MAX_SLICES));
    }
    // transposed J of every line side by side (the lines' scattering integrals run in one launch)
    std::vector<size_t> jtOff(Nprd + 1, 0);
    for (int q = 0; q < Nprd; ++q)
        jtOff[q + 1] = jtOff[q] + (size_t)(c->trans[c->prdLines[q]].t.Nred - c->trans[c->prdLines[q]].t.Nblue) * Ns;
    if (c->prdJt.n < jtOff[Nprd])
        HIP_TRY(c->prdJt.alloc(jtOff[Nprd]));
    std::vector<PrdLineArgs> lineArgs(Nprd);
    // PRD section of the reduce tail: [world][Nprd] x (max |d rho / rho|, flattened index)
    double* prdTail = c->red.p + (size_t)c->Ntrans * 4 * Ns + 2 * (size_t)c->worldSize;
    for (int q = 0; q < Nprd; ++q)
    {
        const int tr = c->prdLines[q];
        const HostTrans& h = c->trans[tr];
        const lwhip_atom& at = c->atoms[h.atom];
        PrdLineArgs a{};
        a.Ns = Ns;
        a.Nl = h.t.Nred - h.t.Nblue;
        a.Nblue = (int32_t)c->prdRowOff[q]; // rows of this line in the gather buffer
        a.Nlevel = at.Nlevel;
        a.tr = tr;
        a.gi = c->levelOff[h.atom] + h.t.i;
        a.gj = c->levelOff[h.atom] + h.t.j;
        a.lj = h.t.j;
        a.trBegin = c->atomTrOff[h.atom];
        a.trEnd = c->atomTrOff[h.atom + 1];
        a.lambda0 = h.t.lambda0;
        a.Bij = h.t.Bij;
        a.wave = c->lineWave.p + h.waveOff;
        a.J = c->prdJ.p;
        if (c->worldSize == 1)
        {
            // one device: the line's rows of J (hybrid PRD: of JRest) themselves, no copy into the gather buffer (the rates pass
            // that rewrites them comes after the scattering integral in the stream)
            a.Nblue = 0;
            a.J = c->hprd ? c->JRest.p + (size_t)c->hLa2prdHost[h.t.Nblue] * Ns : c->J.p + (size_t)h.NblueLoc * Ns;
        }
        a.Jt = c->prdJt.p + jtOff[q];
        a.vBroad = c->vBroad.p + (size_t)h.atom * Ns;
        a.aDamp = c->aDamp.p + (size_t)h.row * Ns;
        a.Qelast = c->Qelast.p + (size_t)h.row * Ns;
        a.Cmat = at.detailed ? nullptr : c->Cmat.p + c->gammaOff[h.atom];
        a.n = c->n.p;
        a.Rij = c->Rij.p;
        a.Rji = c->Rji.p;
        a.transLi = c->transLi.p;
        a.transLj = c->transLj.p;
        // this shard's emission wavelengths of the line: [laLo, laHi) in the line's own grid; its rho rows
        // (hybrid PRD: rho of the line's whole grid on every shard -- a ray's rho is interpolated between neighbouring
        // wavelengths of the line, wherever they are swept)
        a.laLo = h.rhoLt0;
        a.laHi = h.rhoLt0 + h.rhoRows;
        a.rho = c->rho.p + h.rhoOff;
        // enough workgroups to fill the chip: Ns x slices >= ~8 per CU, at least 128 wavelengths each
        {
            static const int perSlice = std::max(1, dbg_env_int("LWHIP_PRD_SLICE_LA", 128));
            a.slices = std::max(1, std::min({ (int)PRD_MAX_SLICES, (2048 * 128 / perSlice + Ns - 1) / Ns, (a.laHi - a.laLo + perSlice - 1) / perSlice }));
        }
        // gII weights depend only on aDamp, vBroad and the grid: cached across sub-iterations and
        // iterations (the reference's t->gII), dropped when lwhip_upload(LWHIP_NSTAR) refreshes those
        if (c->gII.size() != (size_t)c->Ntrans)
        {
            c->gII.resize(c->Ntrans);
            c->gIIValid.assign(c->Ntrans, 0);
        }
        a.gII = nullptr;
        a.gIIFill = 1;
        if (a.laHi > a.laLo)
        {
            // the end-corrected gII weights of the line's fine grids and the bracketing indices as 16 bits (10 bytes per fine
            // point; the interpolation fraction is re-derived in the kernel)
            const size_t per = (size_t)Ns * 88 * a.Nl;
            const size_t nWin = (size_t)Ns * PRD_MAX_SLICES; // (first, last) bracket per (depth, slice): 8 bytes
            const size_t need = per + (per + 3) / 4 + nWin;
            if (a.Nl > 32767)
                return fail(LWHIP_ERR_UNSUPPORTED, "a PRD line with more than 32 767 wavelengths");
            if (!c->gII[tr])
                c->gII[tr] = std::make_unique<DevBuf<double>>();
            if (c->gII[tr]->n < need)
            {
                if (c->gII[tr]->alloc(need) != hipSuccess)
                {
                    (void)hipGetLastError(); // out of memory for the cache: recompute every time
                    c->gII[tr].reset();
                }
                c->gIIValid[tr] = 0;
            }
            if (c->gII[tr])
            {
                a.gII = c->gII[tr]->p;
                a.gIT = (int32_t*)(a.gII + per);
                // (LWHIP_PRD_GENERAL=1 under LWHIP_DEBUG: no windows, every sub-iteration through the general kernel)
                a.gWin = c->prdGeneral ? nullptr : (int32_t*)(a.gII + per + (per + 3) / 4);
                a.gIIFill = c->gIIValid[tr] ? 0 : 1;
                c->gIIValid[tr] = 1;
            }
        }
        a.changeVal = c->prdChange.p + (size_t)q * Ns * PRD_MAX_SLICES;
        a.changeIdx = c->prdChangeIdx.p + (size_t)q * Ns * PRD_MAX_SLICES;
        a.tail = prdTail;
        a.world = c->worldSize;
        a.rank = c->worldRank;
        a.q = q;
        a.Nprd = Nprd;
        lineArgs[q] = a;
    }
    // the blocks change only when a buffer moves or a cache is filled: upload on change (stream-ordered copy from a
    // host copy that lives until the next change)
    if (c->prdArgsHost.size() != (size_t)Nprd
        || std::memcmp(c->prdArgsHost.data(), lineArgs.data(), (size_t)Nprd * sizeof(PrdLineArgs)) != 0)
    {
        HIP_TRY(hipStreamSynchronize(c->stream)); // the previous copy may still be read
        c->prdArgsHost = lineArgs;
        if (c->prdArgsDev.n < (size_t)Nprd)
            HIP_TRY(c->prdArgsDev.alloc(Nprd));
        HIP_TRY(hipMemcpyAsync(c->prdArgsDev.p, c->prdArgsHost.data(), (size_t)Nprd * sizeof(PrdLineArgs), hipMemcpyHostToDevice,
                               c->stream));
    }
    // one device, 1D: the per-line maxima of |d rho / rho| are reduced by the apply launch of lwhip_prd_finalise, which also
    // sums the stage-1 parts itself -- a sub-iteration is scatter -> rates pass -> apply (round 4: six launches)
    const bool fused = c->worldSize == 1 && !c->is2d && !c->deterministic;
    const bool piped = fused && c->prdPipeIter > 0;
    HIP_TRY(launch_prd_scatter(c->prdArgsDev.p, lineArgs.data(), Nprd, c->stream, !fused, piped ? c->prdCtl.p : nullptr, c->prdPipeIter));
    c->prdFused = fused;
    if (c->is2d)
    {
        // the same rates pass through the 2D pipeline, restricted to the runs of PRD wavelengths
        const int st2 = run_2d(c, 0, 3);
        if (st2 != LWHIP_OK)
            return st2;
        c->prdPending = true;
        return LWHIP_OK;
    }
    // formal_sol_prd_update_rates (PrdTemplates.hpp:18-75): J and the PRD lines' rates over the PRD wavelengths
    TileDyn dyn = make_dyn(c, false, 0);
    dyn.prdOnly = 1;
    if (piped)
    {
        dyn.stopCtl = c->prdCtl.p;
        dyn.stopIter = c->prdPipeIter;
    }
    // unvisited wavelengths: dJ = 0 -- after a full sweep; the sub-iterations that follow visit the same wavelengths
    if (!c->dJPrdClean)
        HIP_TRY(hipMemsetAsync(c->dJ.p, 0, (size_t)c->Nla * sizeof(double), c->stream));
    c->dJPrdClean = true;
    // the sweep adds its workgroup sums into the stage-1 buffer; stage 2 sums and zeroes the parts
    if (!c->red8Clean)
        HIP_TRY(hipMemsetAsync(c->red8.p, 0, c->red8.n * sizeof(double), c->stream));
    HIP_TRY(run_sweep(c, dyn, true));
    if (!fused)
    {
        ReduceArgs r = make_reduce_args(c);
        r.zeroParts = 1;
        HIP_TRY(launch_reduce_sum(r, c->stream));
    }
    c->red8Clean = true;
    c->prdPending = true;
    return LWHIP_OK;
}

int lwhip_prd_finalise(lwhip_context* c, double* dRho, int32_t* dRhoMaxIdx, double* dJMax, int32_t* dJMaxIdx)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    const int Nprd = (int)c->prdLines.size();
    if (Nprd == 0)
        return LWHIP_OK;
    if (!c->prdPending)
        return fail(LWHIP_ERR_INVALID, "lwhip_prd_finalise without lwhip_prd_partial");
    HIP_TRY(hipSetDevice(c->device));
    const int Ns = c->Ns;
    ApplyArgs ap{};
    ap.Ns = Ns;
    ap.Natom = c->Natom;
    ap.atomTrOff = c->atomTrOffD.p;
    ap.atomNlevel = c->atomNlevel.p;
    ap.atomDetailed = c->atomDetailed.p;
    ap.atomGammaOff = c->atomGammaOff.p;
    ap.trans = c->dtrans.p;
    ap.transLi = c->transLi.p;
    ap.transLj = c->transLj.p;
    ap.red = c->red.p;
    ap.Gamma = c->Gamma.p;
    ap.Rij = c->Rij.p;
    ap.Rji = c->Rji.p;
    ap.Cmat = c->Cmat.p;
    ap.prdOnly = 1;
    ap.prdDetailed = c->prdDetailed ? 1 : 0;
    if (c->prdFused)
    {
        // the sums straight from the stage-1 parts (zeroed as they are read), dJMax of the pass and the lines' changes in
        // extra workgroups of this launch
        ap.fromParts = 1;
        ap.zeroParts = 1;
        ap.red8 = c->red8.p;
        ap.dJ = c->dJ.p;
        ap.NlaDj = c->Nla;
        ap.idxOffset = c->laStart;
        ap.tail = c->red.p + (size_t)c->Ntrans * 4 * Ns;
        ap.prdList = c->prdArgsDev.p;
        ap.nPrdChange = Nprd;
    }
    const bool piped = c->prdFused && c->prdPipeIter > 0;
    if (piped)
    {
        ap.prdCtl = c->prdCtl.p;
        ap.prdTol = c->prdPipeTol;
        ap.prdIter = c->prdPipeIter;
    }
    HIP_TRY(launch_apply(ap, c->stream));
    c->prdFused = false;
    c->prdPending = false;
    const size_t nTail = 2 * (size_t)c->worldSize * (1 + Nprd);
    if (piped)
    {
        // (queued: the slot of this sub-iteration; lwhip_redistribute_prd waits once for the whole group and reads them)
        HIP_TRY(hipMemcpyAsync(c->prdPinnedPipe + (size_t)((c->prdPipeIter - 1) % PRD_PIPE_DEPTH) * nTail,
                               c->red.p + (size_t)c->Ntrans * 4 * Ns, nTail * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        return LWHIP_OK;
    }
    if (!c->prdPinned)
        HIP_TRY(hipHostMalloc((void**)&c->prdPinned, nTail * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(c->prdPinned, c->red.p + (size_t)c->Ntrans * 4 * Ns, nTail * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    prd_read_results(c, c->prdPinned, dRho, dRhoMaxIdx, dJMax, dJMaxIdx);
    return LWHIP_OK;
}

// the sub-iteration's results out of the reduce tail's copy: (dJMax, idx) per shard, then per line and shard (max |d rho / rho|, idx)
void prd_read_results(lwhip_context* c, const double* pinned, double* dRho, int32_t* dRhoMaxIdx, double* dJMax, int32_t* dJMaxIdx)
{
    const int Nprd = (int)c->prdLines.size();
    // max over shards, first (lowest) index on ties: max_idx, Constants.hpp:114-125
    auto best_of = [&](const double* slots, size_t stride, double& v, int& idx) {
        v = 0.0;
        idx = 0;
        bool any = false;
        for (int rk = 0; rk < c->worldSize; ++rk)
        {
            const double x = slots[rk * stride];
            const int i = (int)slots[rk * stride + 1];
            if (!any || x > v || (x == v && i < idx))
            {
                if (x > 0.0 || !any)
                {
                    v = x;
                    idx = i;
                    any = true;
                }
            }
        }
        if (!(v > 0.0))
        {
            v = 0.0;
            idx = 0;
        }
    };
    double v;
    int idx;
    best_of(pinned, 2, v, idx);
    if (dJMax)
        *dJMax = v;
    if (dJMaxIdx)
        *dJMaxIdx = idx;
    const double* prdTail = pinned + 2 * (size_t)c->worldSize;
    for (int q = 0; q < Nprd; ++q)
    {
        best_of(prdTail + 2 * q, 2 * (size_t)Nprd, v, idx);
        const int Nl = c->trans[c->prdLines[q]].t.Nred - c->trans[c->prdLines[q]].t.Nblue;
        if (dRho)
            dRho[q] = v;
        if (dRhoMaxIdx) // the flattened [la, k] index modulo Nlambda, as PrdTemplates.hpp:265 reports it
            dRhoMaxIdx[q] = idx % Nl;
    }
}

int lwhip_redistribute_prd(lwhip_context* c, int maxIter, double tol, lwhip_prd_result* res)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    if (c->worldSize > 1 || c->laStart != 0 || c->laEnd != c->prob.Nlambda)
        return fail(LWHIP_ERR_UNSUPPORTED, "redistribute_prd on a wavelength shard: use lwhip_prd_pack / _partial / "
                                           "_finalise around the two all-reduces");
    const int Nprd = (int)c->prdLines.size();
    if (res)
    {
        res->NprdSubIter = 0;
        res->Nprd = Nprd;
    }
    if (Nprd == 0 || maxIter <= 0)
        return LWHIP_OK;
    std::vector<double> dRho(Nprd);
    std::vector<int32_t> dRhoIdx(Nprd);
    int iter = 0;
    auto record = [&](int it, double dJ, int32_t dJIdx) {
        double dRhoMax = 0.0;
        for (int q = 0; q < Nprd; ++q)
        {
            dRhoMax = std::max(dRhoMax, dRho[q]);
            if (res && res->dRho)
                res->dRho[(size_t)(it - 1) * Nprd + q] = dRho[q];
            if (res && res->dRhoMaxIdx)
                res->dRhoMaxIdx[(size_t)(it - 1) * Nprd + q] = dRhoIdx[q];
        }
        if (res && res->dJPrdMax)
            res->dJPrdMax[it - 1] = dJ;
        if (res && res->dJPrdMaxIdx)
            res->dJPrdMaxIdx[it - 1] = dJIdx;
        return dRhoMax;
    };
    // One device, 1D, lane sweep, atomic sums: the launches of up to PRD_PIPE_DEPTH sub-iterations are queued back to back --
    // no host round trip between them (it was 20-25 us of an idle device per sub-iteration, of 155).  The stopping rule of the
    // loop (PrdTemplates.hpp:250-275: stop after the sub-iteration whose largest |d rho / rho| is below tol) is kept ON THE
    // DEVICE: the apply launch of a sub-iteration records that it met the tolerance, and every launch of a later
    // sub-iteration returns at once, so nothing is computed or written that the reference's loop would not have.
    const bool piped = c->worldSize == 1 && !c->is2d && !c->deterministic && c->laneSweep && !c->hprd && c->prdPipeline;
    if (piped)
    {
        const size_t nTail = 2 * (size_t)c->worldSize * (1 + Nprd);
        if (c->prdPinnedPipeN < (size_t)PRD_PIPE_DEPTH * nTail)
        {
            if (c->prdPinnedPipe)
            {
                HIP_TRY(hipStreamSynchronize(c->stream)); // (no queued copy may still aim at the block that goes)
                HIP_TRY(hipHostFree(c->prdPinnedPipe));
            }
            c->prdPinnedPipe = nullptr;
            c->prdPinnedPipeN = (size_t)PRD_PIPE_DEPTH * nTail;
            HIP_TRY(hipHostMalloc((void**)&c->prdPinnedPipe, c->prdPinnedPipeN * sizeof(double), hipHostMallocDefault));
        }
        if (c->prdCtl.n < 4)
            HIP_TRY(c->prdCtl.alloc_zero(4));
        HIP_TRY(hipMemsetAsync(c->prdCtl.p, 0, 4 * sizeof(int32_t), c->stream));
        int32_t* ctlHost = c->prdCtlHost; // (a window of the context's host block)
        int st = LWHIP_OK;
        while (iter < maxIter)
        {
            const int nb = std::min((int)PRD_PIPE_DEPTH, maxIter - iter);
            for (int j = 0; j < nb && st == LWHIP_OK; ++j)
            {
                c->prdPipeIter = iter + j + 1;
                c->prdPipeTol = tol;
                st = lwhip_prd_pack(c, nullptr, nullptr);
                if (st == LWHIP_OK)
                    st = lwhip_prd_partial(c);
                if (st == LWHIP_OK)
                    st = lwhip_prd_finalise(c, nullptr, nullptr, nullptr, nullptr);
            }
            c->prdPipeIter = 0;
            if (st != LWHIP_OK)
                return st;
            HIP_TRY(hipMemcpyAsync(ctlHost, c->prdCtl.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            const int stopAt = ctlHost[0]; // the sub-iteration that met the tolerance (0: none so far)
            bool stopped = false;
            for (int j = 0; j < nb; ++j)
            {
                const int it = iter + j + 1;
                double dJ = 0.0;
                int32_t dJIdx = 0;
                prd_read_results(c, c->prdPinnedPipe + (size_t)((it - 1) % PRD_PIPE_DEPTH) * nTail, dRho.data(), dRhoIdx.data(), &dJ, &dJIdx);
                record(it, dJ, dJIdx);
                if (stopAt == it)
                {
                    iter = it;
                    stopped = true;
                    break;
                }
            }
            if (stopped)
                break;
            iter += nb;
        }
        if (res)
            res->NprdSubIter = iter;
        return LWHIP_OK;
    }
    while (iter < maxIter)
    {
        ++iter;
        int st = lwhip_prd_pack(c, nullptr, nullptr);
        if (st == LWHIP_OK)
            st = lwhip_prd_partial(c);
        double dJ = 0.0;
        int32_t dJIdx = 0;
        if (st == LWHIP_OK)
            st = lwhip_prd_finalise(c, dRho.data(), dRhoIdx.data(), &dJ, &dJIdx);
        if (st != LWHIP_OK)
            return st;
        if (record(iter, dJ, dJIdx) < tol)
            break;
    }
    if (res)
        res->NprdSubIter = iter;
    return LWHIP_OK;
}
}

// lwhip_batch.hip -- host side of the C ABI (lwhip_host.h): 1.5D column batches -- one iteration of n structurally identical
// contexts in one set of launches.
#include "lwhip_host.h"

// ---- 1.5D column batches: one iteration of n structurally identical contexts in one set of launches ----------
struct lwhip_batch
{
    std::vector<lwhip_context*> ctxs;
    std::vector<hipStream_t> ownStreams; // what the columns ran on before they joined the batch
    DevBuf<const TileArgs*> apList;
    DevBuf<ReduceArgs> rList;
    DevBuf<ApplyArgs> aList;
    DevBuf<double> tail;       // [n][2] (dJMax, idx) of every column
    double* tailPinned = nullptr;
    std::vector<ApplyArgs> aHost;
    double aCrsw = 0.0;        // what the device copy of the apply blocks was built with
    bool aValid = false;
    DevBuf<StatEqArgs> seList; // stat_equil of all active atoms of every column
    StatEqArgs se0{};
    int seMaxNl = 0;
    DevBuf<VoigtLineArgs> voigtList; // lines of the columns whose profiles are being recomputed
    DevBuf<RetileArgs> retileList;   // ... and their retile arguments
};


// device profiles of the batch's columns: every line of every column that needs them (all = the explicit
// lwhip_batch_compute_profiles, else those whose atmosphere was uploaded since) in ONE launch pair on the batch's stream
static int batch_compute_profiles(lwhip_batch* b, bool all)
{
    std::vector<VoigtLineArgs> list;
    std::vector<lwhip_context*> todo;
    for (lwhip_context* c : b->ctxs)
    {
        if (!(all || c->profilesStale))
            continue;
        const int st = voigt_line_list(c, list);
        if (st != LWHIP_OK)
            return st;
        todo.push_back(c);
    }
    if (todo.empty())
        return LWHIP_OK;
    lwhip_context* c0 = b->ctxs[0];
    if (!list.empty())
    {
        if (b->voigtList.n < list.size())
        {
            HIP_TRY(hipStreamSynchronize(c0->stream)); // nothing may still read the buffer about to be replaced
            HIP_TRY(b->voigtList.alloc(list.size()));
        }
        HIP_TRY(hipMemcpyAsync(b->voigtList.p, list.data(), list.size() * sizeof(VoigtLineArgs), hipMemcpyHostToDevice, c0->stream));
        // the launch geometry allows 65 535 entries per grid dimension
        for (size_t off = 0; off < list.size(); off += 32768)
        {
            const int cnt = (int)std::min<size_t>(32768, list.size() - off);
            HIP_TRY(launch_voigt_lines(b->voigtList.p + off, list.data() + off, cnt, c0->stream));
        }
    }
    // ... and their tile-blocked copies, one grid slice per column
    std::vector<RetileArgs> rl;
    for (lwhip_context* c : todo)
    {
        c->deviceProfiles = true;
        c->profilesStale = false;
        c->phiSym = c->vlosZero;
        RetileArgs r;
        if (retile_args(c, r))
            rl.push_back(r);
    }
    if (!rl.empty())
    {
        if (b->retileList.n < rl.size())
        {
            HIP_TRY(hipStreamSynchronize(c0->stream));
            HIP_TRY(b->retileList.alloc(rl.size()));
        }
        HIP_TRY(hipMemcpyAsync(b->retileList.p, rl.data(), rl.size() * sizeof(RetileArgs), hipMemcpyHostToDevice, c0->stream));
        for (size_t off = 0; off < rl.size(); off += 32768)
            HIP_TRY(launch_retile_list(b->retileList.p + off, rl.data() + off, (int)std::min<size_t>(32768, rl.size() - off),
                                       c0->stream));
    }
    return LWHIP_OK;
}
static int batch_ensure_profiles(lwhip_batch* b) { return batch_compute_profiles(b, false); }

extern "C"
{
int lwhip_batch_create(lwhip_context* const* ctxs, int n, lwhip_batch** out)
{
    if (!ctxs || n < 1 || !out)
        return fail(LWHIP_ERR_INVALID, "batch_create: null argument");
    lwhip_context* c0 = ctxs[0];
    for (int i = 0; i < n; ++i)
    {
        lwhip_context* c = ctxs[i];
        if (!c)
            return fail(LWHIP_ERR_INVALID, "batch_create: null context");
        if (c->is2d || c->worldSize != 1)
            return fail(LWHIP_ERR_UNSUPPORTED, "batch_create: columns are 1D, unsharded contexts");
        if (c->deterministic)
            return fail(LWHIP_ERR_UNSUPPORTED, "batch_create: the fused batch sums by atomics (a context made with LWHIP_OPT_DETERMINISTIC "
                                               "iterates on its own)");
        if (c->device != c0->device || c->Ns != c0->Ns || c->Nla != c0->Nla
            || c->Nrays != c0->Nrays || c->Ntrans != c0->Ntrans || c->Natom != c0->Natom || c->nTiles != c0->nTiles
            || c->nTileChunks != c0->nTileChunks || c->nPostChunks != c0->nPostChunks || c->tileCap != c0->tileCap
            || c->tileFuse != c0->tileFuse || c->tileWaves != c0->tileWaves || c->maxCTTile != c0->maxCTTile
            || c->laneSweep != c0->laneSweep
            || c->maxL != c0->maxL || c->maxC != c0->maxC
            || c->NlevTot != c0->NlevTot || c->prob.formalSolver != c0->prob.formalSolver)
            return fail(LWHIP_ERR_INVALID, "batch_create: the columns must share device, model atoms, wavelength grid "
                                           "and solver");
    }
    HIP_TRY(hipSetDevice(c0->device));
    // the batch's launches go to the first column's stream; the other columns move onto it so that their own
    // uploads / downloads stay ordered with the batch
    std::vector<hipStream_t> before(n);
    for (int i = 0; i < n; ++i)
    {
        before[i] = ctxs[i]->stream;
        if (ctxs[i]->stream != c0->stream)
        {
            HIP_TRY(hipStreamSynchronize(ctxs[i]->stream));
            ctxs[i]->stream = c0->stream;
        }
    }
    auto b = std::make_unique<lwhip_batch>();
    b->ctxs.assign(ctxs, ctxs + n);
    b->ownStreams = before;
    std::vector<const TileArgs*> ap(n);
    std::vector<ReduceArgs> rl(n);
    HIP_TRY(b->tail.alloc((size_t)2 * n));
    for (int i = 0; i < n; ++i)
    {
        ap[i] = ctxs[i]->dtargs.p;
        rl[i] = make_reduce_args(ctxs[i]);
        rl[i].batchTail = b->tail.p + 2 * (size_t)i;
        rl[i].zeroParts = 1;
        // the columns' stage-1 buffers start clean (the sweep adds into them, stage 2 zeroes what it has summed)
        if (!ctxs[i]->red8Clean)
            HIP_TRY(hipMemsetAsync(ctxs[i]->red8.p, 0, ctxs[i]->red8.n * sizeof(double), c0->stream));
        ctxs[i]->red8Clean = true;
    }
    HIP_TRY(b->apList.upload(ap));
    HIP_TRY(b->rList.upload(rl));
    HIP_TRY(b->aList.alloc((size_t)n));
    b->aHost.resize(n);
    {
        std::vector<StatEqArgs> sl(n);
        for (int i = 0; i < n; ++i)
        {
            lwhip_context* c = ctxs[i];
            std::vector<NrAtom> atoms;
            int maxNl = 0;
            for (int ia = 0; ia < c->Natom; ++ia)
            {
                const lwhip_atom& a = c->atoms[ia];
                if (a.detailed)
                    continue;
                NrAtom at{};
                at.atom = ia;
                at.Nlevel = a.Nlevel;
                at.levelOff = c->levelOff[ia];
                at.gammaOff = c->gammaOff[ia];
                atoms.push_back(at);
                maxNl = std::max(maxNl, a.Nlevel);
            }
            if (!atoms.empty())
            {
                HIP_TRY(c->statEqAtoms.upload(atoms));
                c->statEqKey = -1;
            }
            StatEqArgs sa{};
            sa.Ns = c->Ns;
            sa.k0 = 0;
            sa.k1 = c->Ns;
            sa.Natoms = (int32_t)atoms.size();
            sa.atoms = c->statEqAtoms.p;
            sa.n = c->n.p;
            sa.nTotal = c->nTotal.p;
            sa.Gamma = c->Gamma.p;
            sa.status = c->statusDev;
            sa.change = nullptr;
            sl[i] = sa;
            if (i == 0)
            {
                b->se0 = sa;
                b->seMaxNl = maxNl;
            }
            else if (sa.Natoms != b->se0.Natoms || maxNl != b->seMaxNl)
                return fail(LWHIP_ERR_INVALID, "batch_create: the columns must share their active atoms");
        }
        HIP_TRY(b->seList.upload(sl));
    }
    HIP_TRY(hipHostMalloc((void**)&b->tailPinned, (size_t)2 * n * sizeof(double), hipHostMallocDefault));
    *out = b.release();
    return LWHIP_OK;
}

void lwhip_batch_destroy(lwhip_batch* b)
{
    if (!b)
        return;
    // the columns go back to their own streams (the first column's may be destroyed before the others)
    if (!b->ctxs.empty())
    {
        (void)hipSetDevice(b->ctxs[0]->device);
        (void)hipStreamSynchronize(b->ctxs[0]->stream);
        for (size_t i = 0; i < b->ctxs.size() && i < b->ownStreams.size(); ++i)
            b->ctxs[i]->stream = b->ownStreams[i];
    }
    if (b->tailPinned)
        (void)hipHostFree(b->tailPinned);
    delete b;
}

int lwhip_batch_formal_sol_gamma_matrices(lwhip_batch* b, int lambdaIterate, double crsw, lwhip_iter_result* results)
{
    if (!b || b->ctxs.empty())
        return fail(LWHIP_ERR_INVALID, "null batch");
    const int n = (int)b->ctxs.size();
    lwhip_context* c0 = b->ctxs[0];
    HIP_TRY(hipSetDevice(c0->device));
    {
        // columns whose atmosphere was updated: their phi / wphi first, all their lines in one launch pair
        const int stp = batch_ensure_profiles(b);
        if (stp != LWHIP_OK)
            return stp;
    }
    // Gamma <- crsw * C of every column is fused into its slice of the apply launch
    for (int i = 0; i < n; ++i)
    {
        lwhip_context* c = b->ctxs[i];
        if (c->partialPending || c->prdPending)
            return fail(LWHIP_ERR_INVALID, "batch iteration while a split iteration of a column is pending");
        c->prefillCrsw = crsw;
        c->dJPrdClean = false; // (the batch sweep writes every wavelength's dJ)
        c->prefillPending = c->gammaTot > 0 && c->Cmat.p != nullptr;
        b->aHost[i] = make_apply_args(c);
        c->prefillPending = false;
    }
    if (!b->aValid || b->aCrsw != crsw)
    {
        HIP_TRY(hipMemcpyAsync(b->aList.p, b->aHost.data(), (size_t)n * sizeof(ApplyArgs), hipMemcpyHostToDevice, c0->stream));
        b->aValid = true;
        b->aCrsw = crsw;
    }
    // one set of launches for all columns: pre-pass, sweep (each workgroup finishes its tile when fused), stage 2
    TileDyn dyn = make_dyn(c0, false, lambdaIterate);
    for (int q = 0; q < n && dyn.phiSym; ++q) // (one launch for all columns: pairs only if every column's profiles are symmetric)
        dyn.phiSym = b->ctxs[q]->phiSym ? 1 : 0;
    const bool fuse = c0->tileFuse;
    if (!c0->laneSweep) // (the lane sweep's tasks do their own pre-pass)
        HIP_TRY(launch_tile_pre(c0->dtargs.p, c0->htargs, c0->nTiles, b->apList.p, n, c0->stream));
    if (c0->laneSweep)
        HIP_TRY(launch_lane_sweep(c0->dtargs.p, c0->htargs, dyn, c0->prob.formalSolver, true, c0->nTileChunks, c0->tileWaves, b->apList.p, n,
                                  c0->stream));
    else
    {
        HIP_TRY(launch_tile_sweep(c0->dtargs.p, c0->htargs, dyn, c0->prob.formalSolver, c0->tileCap, true, fuse, c0->nTileChunks,
                                  c0->tileWaves, b->apList.p, n, c0->stream));
        if (!fuse)
            HIP_TRY(launch_tile_post(c0->dtargs.p, c0->htargs, dyn, c0->nPostChunks, b->apList.p, n, c0->stream));
    }
    {
        ReduceArgs r0 = make_reduce_args(c0);
        r0.zeroParts = 1;
        HIP_TRY(launch_reduce_sum(r0, c0->stream, b->rList.p, n));
    }
    if (c0->Natom > 0)
        HIP_TRY(launch_apply(b->aHost[0], c0->stream, b->aList.p, n));
    if (results)
    {
        HIP_TRY(hipMemcpyAsync(b->tailPinned, b->tail.p, (size_t)2 * n * sizeof(double), hipMemcpyDeviceToHost, c0->stream));
        HIP_TRY(hipStreamSynchronize(c0->stream));
        for (int i = 0; i < n; ++i)
        {
            results[i].updatedJ = 1;
            results[i].dJMax = b->tailPinned[2 * i];
            results[i].dJMaxIdx = (int32_t)b->tailPinned[2 * i + 1];
        }
    }
    return LWHIP_OK;
}

int lwhip_batch_compute_profiles(lwhip_batch* b)
{
    if (!b || b->ctxs.empty())
        return fail(LWHIP_ERR_INVALID, "null batch");
    HIP_TRY(hipSetDevice(b->ctxs[0]->device));
    return batch_compute_profiles(b, true);
}

int lwhip_batch_stat_equil(lwhip_batch* b)
{
    if (!b || b->ctxs.empty())
        return fail(LWHIP_ERR_INVALID, "null batch");
    lwhip_context* c0 = b->ctxs[0];
    HIP_TRY(hipSetDevice(c0->device));
    if (b->se0.Natoms <= 0)
        return LWHIP_OK;
    for (lwhip_context* c : b->ctxs)
    {
        const int stp = flush_prefill(c);
        if (stp != LWHIP_OK)
            return stp;
        *c->statusHost = 0;
    }
    HIP_TRY(launch_stat_eq(b->se0, b->seMaxNl, c0->stream, b->seList.p, (int)b->ctxs.size()));
    HIP_TRY(hipStreamSynchronize(c0->stream));
    for (lwhip_context* c : b->ctxs)
        if (*c->statusHost == LWHIP_ERR_SINGULAR)
        {
            *c->statusHost = 0;
            return fail(LWHIP_ERR_SINGULAR, "Singular Matrix");
        }
    return LWHIP_OK;
}
}

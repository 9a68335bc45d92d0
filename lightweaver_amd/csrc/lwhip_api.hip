// lwhip_api.hip -- host side of the C ABI declared in include/lwhip.h.
//
// Owns the HBM-resident copy of the problem, the per-wavelength activity tables the sweep kernel
// walks, and the launch sequence of one iteration:
//
//     sweep_kernel (per-chunk partial slabs, dJ) -> reduce_stage1/2 (fixed-order slab sum)
//     (stage 2 also reduces dJ to (dJMax, idx) in the gather tail)
//     [-> caller all-reduces the reduce buffer across GPUs] -> apply_kernel (Gamma pre-fill fused)
//
// There is no CPU fallback: without a HIP device every compute entry point fails with
// LWHIP_ERR_DEVICE.
#include "lwhip_internal.h"
#include "../../include/lwhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

using namespace lwhip;

namespace
{
thread_local std::string g_lastError;

int fail(int code, const std::string& msg)
{
    g_lastError = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do                                                                                                 \
    {                                                                                                  \
        hipError_t err__ = (expr);                                                                     \
        if (err__ != hipSuccess)                                                                       \
            return fail(LWHIP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(err__));       \
    } while (0)

template <typename T> struct DevBuf
{
    T* p = nullptr;
    size_t n = 0;
    bool owned = true;
    hipError_t alloc(size_t count)
    {
        release();
        n = count;
        if (count == 0)
            return hipSuccess;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        // diagnosis (LWHIP_POISON=1): fresh allocations hold NaN / -1, so that a read of memory nothing initialised shows up
        // in the results instead of depending on what the previous owner of the pages left there
        if (e == hipSuccess && poison_on())
            e = hipMemset(p, 0xFF, count * sizeof(T));
        static const bool trace = std::getenv("LWHIP_TRACE_ALLOC") != nullptr; // diagnosis: which buffer a faulting address follows
        if (trace)
            std::fprintf(stderr, "lwhip alloc %p .. %p (%zu x %zu B)\n", (void*)p, (void*)((char*)p + count * sizeof(T)), count, sizeof(T));
        return e;
    }
    static bool poison_on()
    {
        static const bool poison = std::getenv("LWHIP_POISON") != nullptr;
        return poison;
    }
    // a window of somebody else's allocation (the depth arena of the lane sweep)
    void view(T* ptr, size_t count)
    {
        release();
        p = ptr;
        n = count;
        owned = false;
    }
    hipError_t upload(const std::vector<T>& v)
    {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty())
            return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void release()
    {
        if (p && owned)
        {
            // (LWHIP_POISON=1: and a buffer is poisoned again when it is given back, with another pattern -- a stale device
            // pointer into it, or memory the allocator hands out behind this wrapper's back, then reads -nan / 0xFEFEFEFE)
            if (poison_on())
                (void)hipMemset(p, 0xFE, n * sizeof(T));
            (void)hipFree(p);
        }
        p = nullptr;
        n = 0;
        owned = true;
    }
    ~DevBuf() { release(); }
};

enum { PRD_MAX_SLICES = 32 }; // workgroups per depth point in the PRD scattering integral

struct HostTrans
{
    lwhip_transition t;   // borrowed host pointers
    int atom;
    int NblueLoc, NredLoc; // clipped to the shard, shard-local indices
    int ltStart;           // first own-grid index inside the shard
    int row;               // wphi row (lines) / ratio row (continua)
    int64_t parOff, phiOff, rhoOff;
    int64_t waveOff; // lines: offset of the full own grid / wlambda in the lineWave / lineWlam pools
};
}

struct lwhip_context
{
    lwhip_problem prob;             // copy of the descriptor (host pointers borrowed)
    std::vector<lwhip_atom> atoms;
    std::vector<HostTrans> trans;   // global transition list, reference order
    std::vector<int> levelOff, atomTrOff;
    std::vector<int64_t> gammaOff;
    int device = 0;
    int worldSize = 1, worldRank = 0;
    int laStart = 0, laEnd = 0, Nla = 0;
    int Ns = 0, Nrays = 0, Natom = 0, NlevTot = 0, Ntrans = 0, Nline = 0, Ncont = 0;
    int maxL = 0, maxC = 0;
    bool is2d = false;            // x-periodic 2D geometry (prob.grid2d): batched pipeline of lwhip_2d.hip
    int Nx = 1, batch2d = 1;
    std::vector<DevLaHeader> hdrHost;
    std::vector<int32_t> contLaHost;
    // the ray-column sweep (lwhip_raymarch.hip): tiles of structurally identical wavelengths
    bool tiled = false;
    int tileL = 12, nTiles = 0, nTileChunks = 0, nTileChunksPrd = 0, nPostChunks = 0, nPostChunksPrd = 0, nTilesPrd = 0;
    bool tileFuse = false; // the sweep's workgroup = one tile, finished by the post-pass inside the sweep launch
    // depth-across-lanes sweep (lwhip_lanesweep.hip): D points per lane, LR lanes per ray, R wavelengths per wavefront
    bool laneSweep = false;
    bool chunkOrderOn = false, chunkSplitOn = false;
    int laneSplitPrd = 1;          // ... in the PRD rates pass (its tile list is shorter)
    int laneSplit = 1;             // lane sweep: wavefronts a tile's rays are split over (1, 2, 4)
    int laneD = 0, laneLR = 0, laneR = 0;
    DevBuf<DevPostProg> postProg;
    DevBuf<double> geoT;
    // hybrid PRD (lwhip_options.hprd): the tables of configure_hprd_coeffs on the device
    // deterministic mode (LWHIP_OPT_DETERMINISTIC): per-workgroup slabs + the (workgroup, slot) lists of every transition
    bool deterministic = false;
    DevBuf<double> detSlab;
    DevBuf<int32_t> detOff, detEnt, detOffPrd, detEntPrd;
    const lwhip_hprd* hprd = nullptr;
    std::vector<int> hLa2prdHost, hLa2hHost;     // global wavelength -> row of JRest / ordinal among hPrdIdxs, or -1
    std::vector<int64_t> hRhoOffHost;             // per transition: offset of its rho-coefficient block, or -1
    DevBuf<lwhip_rho_coeff> hRho;
    DevBuf<int32_t> hLa2h;
    DevBuf<int64_t> hJOff;
    DevBuf<lwhip_j_coeff> hJCoef;
    DevBuf<double> JRest;
    double* lsDbgHost = nullptr; // LWHIP_LS_TIMING: phase clocks of the last sweep, [nTiles][8]
    DevBuf<double> depArena; // lane sweep: n | wphi | ratio | geoT in one allocation (one buffer resource in the kernel)
    int tileWaves = 4, maxSlotsTile = 0, maxCTTile = 1, maxCTPost = 1, tileCap = 0, ktStride = 4;
    int64_t rowsTileTot = 0, momTot = 0, phiTTot = 0;
    int preCols = 0;              // most level-sum columns a tile's rows need (LDS columns of the pre-pass)
    DevBuf<DevTile> tiles;
    DevBuf<DevTileSlot> tslots, tslotsPrd;
    DevBuf<DevTileCopy> tcopies;
    DevBuf<int32_t> tileRemap;
    DevBuf<DevContRec> contRec;
    DevBuf<int32_t> chunkOrder;   // lane sweep: dispatch order of the workgroups' chunks
    // lane sweep: the flat per-workgroup / per-tile records of a task's setup and finish (lwhip_internal.h)
    DevBuf<DevLaneWg> laneWg, laneWgPrd;
    DevBuf<DevLaneTile> laneTiles, laneTilesPrd;
    DevBuf<DevLaneRay> laneRays;
    DevBuf<DevLaneFin> laneFin;
    DevBuf<double> laneFinPar, bcPlanck;
    DevBuf<int32_t> chunkSplit;   // lane sweep: per chunk, the wavefronts its tiles' rays are split over (the launch's tail)
    DevBuf<int32_t> chunkTile, chunkTilePrd, tileListPrd, tileSlotTr, tileSlotTrPrd, postChunkTile, postChunkTilePrd, postSlotTr,
        postCs;
    DevBuf<double> geo, kt, rowsTile, momTile, phiT;
    DevBuf<TileArgs> dtargs, dtargsPrd;
    TileArgs htargs{}, htargsPrd{};
    int64_t rowsTot = 0;
    int nContLa = 0;
    int64_t gammaTot = 0, phiTot = 0, rhoTot = 0, parTot = 0;
    hipStream_t ownStream = nullptr, stream = nullptr;

    DevBuf<double> height, temperature, muz, wmu, wavelength, lowerBcData, upperBcData;
    DevBuf<int32_t> rayAll, rayUp, lowerIdx, upperIdx;
    DevBuf<double> bgChi, bgEta, bgSca, J, I, depthChi, depthEta, depthI;
    DevBuf<double> n, nTotal, ratio, wphi, phi, rho, par, Gamma, Cmat, Rij, Rji;
    DevBuf<double> vlosMu, vBroad, aDamp, Qelast, lineWave, lineWlam;
    DevBuf<double> prdChange, rowsBuf, popScratch, prdJt, prdJ;
    DevBuf<PrdLineArgs> prdArgsDev;   // argument blocks of the PRD lines of a sub-iteration (one launch for all lines)
    std::vector<PrdLineArgs> prdArgsHost; // what the device copy holds
    DevBuf<double> g2mux, b2cs, b2I, b2Psi, b2coef, red2d;
    DevBuf<int32_t> b2idx;
    int groups2d = 1, maxRowsLa = 1;
    int kLo = 0, kHi = -1; // depth range of the population updates (lwhip_set_depth_range); kHi < 0: to the end
    int djIdxMode = 0;     // lwhip_set_djmax_index_mode: 1 = the single-thread scheme's index bookkeeping
    double* tailMapped = nullptr; // host-mapped (dJMax, idx) of the one-call iteration
    double* tailMappedDev = nullptr;
    double tailTicket = 0.0;      // ticket of the last launch that reports through tailMapped
    // line profiles generated on the device (lwhip_compute_profiles): re-derived before the next sweep whenever the
    // inputs they depend on (vlosMu: LWHIP_ATMOS; vBroad, aDamp: LWHIP_NSTAR) are uploaded again
    bool deviceProfiles = false, profilesStale = false;
    DevBuf<VoigtLineArgs> voigtList; // the lines' argument blocks of lwhip_compute_profiles
    DevBuf<double> wphiScratch;   // [16, Ns] wavelength-slice sums of the profile normalisation
    DevBuf<int> wphiTicket;       // arrival counters of its point tiles
    bool lastSweepUpOnly = false; // the last sweep traced the up rays only (lwhip_formal_sol(upOnly))
    bool partsOnly = false;       // fs_partial ran stage 1 of the slab reduce only (one-call iteration)
    int batchHint = 0;            // lwhip_options.flags & 0xffff: contexts expected to share the device (column batch)
    bool prdDetailed = false;     // LWHIP_OPT_PRD_DETAILED: the PRD calls include the detailed atoms' PRD lines
    DevBuf<lwhip_intersection> g2uw, g2dw, g2sub;
    DevBuf<double> xbcLow, xbcUp;      // 2D, fixed x boundaries: [Nla, Nmu, Nz] of the shard
    DevBuf<double> zDown, zUp;         // ZPlaneDecomposition outputs [Nla, Nrays, Nx] (lwhip_set_zplane_outputs)
    double* zDownHost = nullptr;       // their host arrays [Nlambda, Nrays, Nx]
    double* zUpHost = nullptr;
    DevBuf<int32_t> xIdxLow, xIdxUp;   // [Nrays, 2]
    DevBuf<int32_t> g2long, g2subOff, slotTrD, g2lcOwner, g2uwA, g2dwA;
    DevBuf<double> g2uwS, g2dwS; // the records field by field
    DevBuf<double> b2lc; // [batch2d][NlongChar][3]
    std::vector<int> prdLines;        // PRD lines of the active atoms (global transition indices), reference order
    std::vector<int64_t> prdRowOff;   // first row of each line's grid in the prdJ gather buffer
    int64_t prdRowsTot = 0;
    bool prdPending = false;          // between lwhip_prd_partial and lwhip_prd_finalise
    std::vector<std::unique_ptr<DevBuf<double>>> gII; // per transition: cached PRD weights (lazily)
    std::vector<char> gIIValid;
    DevBuf<NrAtom> nrAtoms, statEqAtoms;
    DevBuf<NgAtom> ngAtoms;
    DevBuf<double> ngHistory;
    std::vector<NgAtom> ngAtomsHost;
    int ngOrder = 0, ngPeriod = 0, ngDelay = 0, ngCount = 0;
    bool ngConfigured = false;
    int statEqKey = -2;
    int32_t* statusHost = nullptr;
    int32_t* statusDev = nullptr;
    double* changeHost = nullptr; // host-mapped per-block population changes of the last reported solve
    double* changeDev = nullptr;
    size_t changeCount = 0;
    DevBuf<int32_t> transType;
    DevBuf<int32_t> contLa;
    DevBuf<int32_t> prdChangeIdx;
    double* prdPinned = nullptr;
    DevBuf<DevTrans> dtrans;
    DevBuf<DevLaHeader> laHdr;
    DevBuf<DevSlot> slots;
    DevBuf<DevProgram> progs;
    DevBuf<DevProgRow> progRows;
    DevBuf<DevProgEnt> progEnts;
    DevBuf<ContArgs> dargs;      // 2D: argument block of the continuum-row kernel
    ContArgs hargs{};
    bool atomicParts = false;     // the pending iteration's parts were accumulated by atomics
    bool red8Clean = false;       // red8 holds zeros (only the atomic path leaves it so)
    int* zeroCheck = nullptr;     // LWHIP_CHECK_ZERO=1 (diagnosis): host-mapped count of non-zero words found in red8 at sweep entry
    int* zeroCheckDev = nullptr;
    DevBuf<int32_t> atomTrOffD, atomNlevel, atomDetailed, transLi, transLj;
    DevBuf<int64_t> atomGammaOff;
    DevBuf<double> red, red8, dJ;
    std::vector<double> gatherHost;
    DevBuf<int32_t> status;

    bool profiling = false;
    int profEvery = 1, profCount = 0; // time every profEvery-th sweep launch (lwhip_profile_enable(ctx, n))
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double sweepMs = 0.0;
    int sweepCount = 0;
    bool partialPending = false;
    bool prefillPending = false; // lwhip_gamma_prefill_from_C deferred into the next apply_kernel
    double prefillCrsw = 1.0;
    double* gatherPinned = nullptr;
    // pinned staging for the many small per-atom / per-transition host arrays: they are packed here and
    // cross PCIe as a few large copies (a pageable hipMemcpy per 656-byte row costs ~12 us each)
    double* stage = nullptr;
    size_t stageDoubles = 0;

    ~lwhip_context()
    {
        for (auto& pr : pending)
        {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        if (ownStream)
            (void)hipStreamDestroy(ownStream);
        if (gatherPinned)
            (void)hipHostFree(gatherPinned);
        if (tailMapped)
            (void)hipHostFree(tailMapped);
        if (stage)
            (void)hipHostFree(stage);
        if (statusHost)
            (void)hipHostFree(statusHost);
        if (changeHost)
            (void)hipHostFree(changeHost);
        if (prdPinned)
            (void)hipHostFree(prdPinned);
        if (zeroCheck)
            (void)hipHostFree(zeroCheck);
    }
};

namespace
{
int env_int(const char* name, int dflt)
{
    const char* v = std::getenv(name);
    if (!v || !*v)
        return dflt;
    return std::atoi(v);
}

double trans_wlambda(const lwhip_transition& t, int lt)
{
    // Transition::wlambda, LwTransition.hpp:71-81
    const int len = t.Nred - t.Nblue;
    if (lt == 0)
        return 0.5 * (t.wavelength[1] - t.wavelength[0]) * t.dopplerWidth;
    if (lt == len - 1)
        return 0.5 * (t.wavelength[len - 1] - t.wavelength[len - 2]) * t.dopplerWidth;
    return 0.5 * (t.wavelength[lt + 1] - t.wavelength[lt - 1]) * t.dopplerWidth;
}

int validate(const lwhip_problem* p, std::string& why)
{
    if (!p)
    {
        why = "null problem";
        return LWHIP_ERR_INVALID;
    }
    if (p->abiVersion != LWHIP_ABI_VERSION)
    {
        why = "ABI version mismatch";
        return LWHIP_ERR_INVALID;
    }
    if (p->Nspace < 3 || p->Nrays < 1 || p->Nlambda < 1 || p->Natom < 0)
    {
        why = "need Nspace >= 3, Nrays >= 1, Nlambda >= 1";
        return LWHIP_ERR_INVALID;
    }
    if (p->grid2d)
    {
        const lwhip_grid2d& g = *p->grid2d;
        if (g.Nx < 2 || g.Nz < 2 || g.Nx > 1024 || g.Nx * g.Nz != p->Nspace || g.Nrays != p->Nrays)
        {
            why = "2D grid: 2 <= Nx <= 1024, Nz >= 2, Nx * Nz = Nspace, Nrays matching";
            return LWHIP_ERR_UNSUPPORTED;
        }
        if ((g.zLowerBc == LWHIP_BC_CALLABLE) != (p->zLowerBc.type == LWHIP_BC_CALLABLE)
            || (g.zUpperBc == LWHIP_BC_CALLABLE) != (p->zUpperBc.type == LWHIP_BC_CALLABLE))
        {
            why = "2D grid: a CALLABLE z boundary needs lwhip_problem.zLowerBc / zUpperBc of type CALLABLE (bcData [Nlambda, Nmu, Nx])";
            return LWHIP_ERR_INVALID;
        }
        for (int bt : { g.zLowerBc, g.zUpperBc })
            if (bt != LWHIP_BC_ZERO && bt != LWHIP_BC_THERMALISED && bt != LWHIP_BC_CALLABLE)
            {
                why = "2D grid: z boundaries are ZERO, THERMALISED or CALLABLE";
                return LWHIP_ERR_UNSUPPORTED;
            }
        if (!g.periodic)
        {
            // "Mixed boundary types not supported on x-axis!" (Source/FormalScalar2d.cpp:750-759)
            auto ok = [&](const lwhip_boundary* b) {
                return b && b->type == LWHIP_BC_CALLABLE && b->idxs && b->bcData && b->Nmu > 0;
            };
            if (!ok(g.xLowerBc) || !ok(g.xUpperBc) || g.NlongChar != 0)
            {
                why = "2D grid without periodic x: both x boundaries CALLABLE (idxs, bcData [Nlambda, Nmu, Nz]), no long characteristics";
                return LWHIP_ERR_UNSUPPORTED;
            }
            for (int q = 0; q < 2 * g.Nrays; ++q)
                if (g.xLowerBc->idxs[q] < 0 || g.xLowerBc->idxs[q] >= g.xLowerBc->Nmu || g.xUpperBc->idxs[q] < 0
                    || g.xUpperBc->idxs[q] >= g.xUpperBc->Nmu)
                {
                    why = "2D grid: x boundary idxs out of range (\"Error in boundary condition indexing\")";
                    return LWHIP_ERR_INVALID;
                }
        }
        if (!g.mux || !g.uw || !g.dw || !g.longCharIdx || !g.substepOff || (g.NlongChar > 0 && !g.substeps))
        {
            why = "2D grid: incomplete intersection table";
            return LWHIP_ERR_INVALID;
        }
        if (!fs2d_long_chars_ok(&g))
        {
            why = "2D grid: a long characteristic that does not end on a z plane";
            return LWHIP_ERR_UNSUPPORTED;
        }
        if (p->storeDepthData && (!p->depthChi || !p->depthEta || !p->depthI))
        {
            why = "storeDepthData set but depth arrays are null";
            return LWHIP_ERR_INVALID;
        }
    }
    else if (p->Nspace > 1024)
    {
        why = "Nspace > 1024 is not supported by the 1D sweep kernel";
        return LWHIP_ERR_UNSUPPORTED;
    }
    if (!p->grid2d && (p->formalSolver < LWHIP_FS_LINEAR_1D || p->formalSolver > LWHIP_FS_BEZIER3_1D))
    {
        why = "unknown formal solver";
        return LWHIP_ERR_INVALID;
    }
    if (!p->height || !p->temperature || !p->muz || !p->wmu || !p->wavelength || !p->bgChi
        || !p->bgEta || !p->bgSca || !p->J || !p->I)
    {
        why = "null atmosphere / spectrum / background array";
        return LWHIP_ERR_INVALID;
    }
    if (p->storeDepthData && (!p->depthChi || !p->depthEta || !p->depthI))
    {
        why = "storeDepthData set but depth arrays are null";
        return LWHIP_ERR_INVALID;
    }
    const lwhip_boundary* bcs[2] = { &p->zLowerBc, &p->zUpperBc };
    for (auto* b : bcs)
    {
        if (b->type == LWHIP_BC_CALLABLE && (!b->idxs || !b->bcData || b->Nmu < 1))
        {
            why = "CALLABLE boundary needs idxs, bcData, Nmu";
            return LWHIP_ERR_INVALID;
        }
    }
    bool seenDetailed = false;
    for (int ia = 0; ia < p->Natom; ++ia)
    {
        const lwhip_atom& a = p->atoms[ia];
        if (a.Nlevel < 1 || a.Nlevel > 32 || a.Ntrans < 0)
        {
            why = "atom needs 1 <= Nlevel <= 32";
            return a.Nlevel > 32 ? LWHIP_ERR_UNSUPPORTED : LWHIP_ERR_INVALID;
        }
        if (a.detailed)
            seenDetailed = true;
        else if (seenDetailed)
        {
            why = "active atoms must precede detailed atoms";
            return LWHIP_ERR_INVALID;
        }
        if (!a.n || !a.nStar || !a.nTotal || !a.vBroad || (!a.detailed && !a.Gamma))
        {
            why = "null atom array";
            return LWHIP_ERR_INVALID;
        }
        for (int kr = 0; kr < a.Ntrans; ++kr)
        {
            const lwhip_transition& t = a.trans[kr];
            if (t.i < 0 || t.j < 0 || t.i >= a.Nlevel || t.j >= a.Nlevel || t.i == t.j)
            {
                why = "bad transition levels";
                return LWHIP_ERR_INVALID;
            }
            if (t.Nblue < 0 || t.Nred > p->Nlambda || t.Nred - t.Nblue < 2)
            {
                why = "bad transition wavelength range (need >= 2 points inside the grid)";
                return LWHIP_ERR_INVALID;
            }
            if (!t.wavelength || !t.Rij || !t.Rji)
            {
                why = "null transition array";
                return LWHIP_ERR_INVALID;
            }
            if (t.type == LWHIP_LINE)
            {
                if (!t.phi || !t.wphi)
                {
                    why = "line without phi/wphi";
                    return LWHIP_ERR_INVALID;
                }
                if (t.prd && !t.rhoPrd)
                {
                    why = "prd line without rhoPrd";
                    return LWHIP_ERR_INVALID;
                }
            }
            else if (t.type == LWHIP_CONTINUUM)
            {
                if (!t.alpha)
                {
                    why = "continuum without alpha";
                    return LWHIP_ERR_INVALID;
                }
            }
            else
            {
                why = "unknown transition type";
                return LWHIP_ERR_INVALID;
            }
        }
    }
    return LWHIP_OK;
}

int build_tables(lwhip_context* c)
{
    const int Ns = c->Ns;
    // ---- global transition list, level rows, pools ------------------------------------------
    c->levelOff.assign(c->Natom + 1, 0);
    c->atomTrOff.assign(c->Natom + 1, 0);
    c->gammaOff.assign(c->Natom, 0);
    c->gammaTot = 0;
    for (int ia = 0; ia < c->Natom; ++ia)
    {
        const lwhip_atom& a = c->atoms[ia];
        c->levelOff[ia + 1] = c->levelOff[ia] + a.Nlevel;
        c->atomTrOff[ia + 1] = c->atomTrOff[ia] + a.Ntrans;
        c->gammaOff[ia] = c->gammaTot;
        if (!a.detailed)
            c->gammaTot += (int64_t)a.Nlevel * a.Nlevel * Ns;
    }
    c->NlevTot = c->levelOff[c->Natom];
    c->Ntrans = c->atomTrOff[c->Natom];
    // ---- hybrid PRD: the tables of configure_hprd_coeffs (lwhip_options.hprd) onto the device ----------------------
    c->hLa2prdHost.clear();
    c->hLa2hHost.clear();
    c->hRhoOffHost.assign(c->Ntrans, -1);
    if (c->hprd)
    {
        const lwhip_hprd& H = *c->hprd;
        const int Nlam = c->prob.Nlambda, Nr = c->Nrays;
        if (c->is2d || c->worldSize > 1 || c->laStart != 0 || c->laEnd != Nlam)
            return fail(LWHIP_ERR_UNSUPPORTED, "hybrid PRD: unsharded 1D contexts only (rho of the neighbouring wavelengths and "
                                               "JRest live on one device)");
        if (H.NprdLambda < 1 || H.Nlines < 1 || !H.prdIdxs || !H.JRest || !H.lineAtom || !H.lineTrans || !H.rhoCoeffs
            || (H.NhPrd > 0 && (!H.hPrdIdxs || !H.jCoeffOff || !H.jCoeffs)))
            return fail(LWHIP_ERR_INVALID, "hybrid PRD: incomplete lwhip_hprd");
        c->hLa2prdHost.assign(Nlam, -1);
        c->hLa2hHost.assign(Nlam, -1);
        for (int i = 0; i < H.NprdLambda; ++i)
        {
            if (H.prdIdxs[i] < 0 || H.prdIdxs[i] >= Nlam)
                return fail(LWHIP_ERR_INVALID, "hybrid PRD: prdIdxs out of range");
            c->hLa2prdHost[H.prdIdxs[i]] = i;
        }
        for (int i = 0; i < H.NhPrd; ++i)
        {
            if (H.hPrdIdxs[i] < 0 || H.hPrdIdxs[i] >= Nlam)
                return fail(LWHIP_ERR_INVALID, "hybrid PRD: hPrdIdxs out of range");
            c->hLa2hHost[H.hPrdIdxs[i]] = i;
        }
        std::vector<lwhip_rho_coeff> rho;
        for (int q = 0; q < H.Nlines; ++q)
        {
            const int ia = H.lineAtom[q], kr = H.lineTrans[q];
            if (ia < 0 || ia >= c->Natom || kr < 0 || kr >= c->atoms[ia].Ntrans || !H.rhoCoeffs[q])
                return fail(LWHIP_ERR_INVALID, "hybrid PRD: bad line list");
            const lwhip_transition& t = c->atoms[ia].trans[kr];
            if (t.type != LWHIP_LINE || !t.prd || !t.rhoPrd)
                return fail(LWHIP_ERR_INVALID, "hybrid PRD: a listed transition is not a PRD line");
            const int nlt = t.Nred - t.Nblue;
            const size_t n = (size_t)nlt * Nr * 2 * Ns;
            for (size_t i = 0; i < n; ++i)
                if (H.rhoCoeffs[q][i].i0 < 0 || H.rhoCoeffs[q][i].i1 >= nlt || H.rhoCoeffs[q][i].i0 > H.rhoCoeffs[q][i].i1)
                    return fail(LWHIP_ERR_INVALID, "hybrid PRD: rho interpolation index outside the line's grid");
            c->hRhoOffHost[c->atomTrOff[ia] + kr] = (int64_t)rho.size();
            rho.insert(rho.end(), H.rhoCoeffs[q], H.rhoCoeffs[q] + n);
        }
        rho.resize(rho.size() + 64, lwhip_rho_coeff{ 0, 0, 0.0 }); // (a ray's last block of depths reads past its row)
        HIP_TRY(c->hRho.upload(rho));
        std::vector<int32_t> l2h(c->hLa2hHost.begin(), c->hLa2hHost.end());
        HIP_TRY(c->hLa2h.upload(l2h));
        if (H.NhPrd > 0)
        {
            const size_t ncell = (size_t)H.NhPrd * Nr * 2 * Ns;
            std::vector<int64_t> off(H.jCoeffOff, H.jCoeffOff + ncell + 1);
            off.resize(off.size() + 8, off.back());
            const int64_t nj = off[ncell];
            for (int64_t e = 0; e < nj; ++e)
                if (H.jCoeffs[e].idx < 0 || H.jCoeffs[e].idx >= H.NprdLambda)
                    return fail(LWHIP_ERR_INVALID, "hybrid PRD: JCoeffs row outside JRest");
            std::vector<lwhip_j_coeff> jc(H.jCoeffs, H.jCoeffs + nj);
            jc.resize(jc.size() + 1, lwhip_j_coeff{ 0.0, 0, 0 });
            HIP_TRY(c->hJOff.upload(off));
            HIP_TRY(c->hJCoef.upload(jc));
        }
        HIP_TRY(c->JRest.alloc((size_t)H.NprdLambda * Ns));
        HIP_TRY(hipMemset(c->JRest.p, 0, c->JRest.n * sizeof(double)));
    }
    c->trans.clear();
    c->Nline = c->Ncont = 0;
    c->phiTot = c->rhoTot = c->parTot = 0;
    for (int ia = 0; ia < c->Natom; ++ia)
    {
        const lwhip_atom& a = c->atoms[ia];
        for (int kr = 0; kr < a.Ntrans; ++kr)
        {
            HostTrans h;
            h.t = a.trans[kr];
            h.atom = ia;
            const int b = std::max(h.t.Nblue, c->laStart);
            const int r = std::min(h.t.Nred, c->laEnd);
            if (r > b)
            {
                h.NblueLoc = b - c->laStart;
                h.NredLoc = r - c->laStart;
                h.ltStart = b - h.t.Nblue;
            }
            else
            {
                h.NblueLoc = h.NredLoc = 0;
                h.ltStart = 0;
            }
            const int64_t nlt = h.NredLoc - h.NblueLoc;
            h.parOff = c->parTot;
            c->parTot += 4 * nlt;
            h.phiOff = h.rhoOff = -1;
            if (h.t.type == LWHIP_LINE)
            {
                h.row = c->Nline++;
                h.phiOff = c->phiTot;
                c->phiTot += nlt * c->Nrays * 2 * Ns;
                if (h.t.prd)
                {
                    h.rhoOff = c->rhoTot;
                    c->rhoTot += nlt * Ns;
                }
            }
            else
            {
                h.row = c->Ncont++;
            }
            c->trans.push_back(h);
        }
    }

    // ---- per-(transition, lt) params -----------------------------------------------------------
    std::vector<double> par((size_t)c->parTot, 0.0);
    for (auto& h : c->trans)
    {
        const lwhip_transition& t = h.t;
        for (int l = 0; l < h.NredLoc - h.NblueLoc; ++l)
        {
            const int lt = h.ltStart + l;
            double* q = &par[(size_t)h.parOff + 4 * l];
            const double wlambda = trans_wlambda(t, lt);
            if (t.type == LWHIP_LINE)
            {
                const double hnu_4pi = HC_4PI * (t.lambda0 / t.wavelength[lt]);
                q[0] = hnu_4pi * t.Bij;
                q[1] = wlambda;
                q[2] = t.Bji / t.Bij;
                q[3] = t.Aji / t.Bji;
            }
            else
            {
                const double wl = t.wavelength[lt];
                q[0] = t.alpha[lt];
                q[1] = (wlambda / wl) * PI4_H;
                q[2] = TWO_HC_NM3 / (wl * wl * wl);
                q[3] = 0.0;
            }
        }
    }

    // ---- per-wavelength activity: [lines][mixed continua][pure continua], row programs --------------
    std::vector<DevLaHeader> hdr(c->Nla);
    std::vector<DevSlot> slots;
    std::vector<int> slotTr; // global transition of each slot record
    std::vector<DevProgram> progs;
    std::vector<DevProgRow> progRows;
    std::vector<DevProgEnt> progEnts;
    std::map<std::vector<int>, int> progIds;
    std::vector<int32_t> contLaHost;
    c->rowsTot = 0;
    c->maxL = c->maxC = 0;
    const int rowEtaA = 2, rowLevChi = 2 + c->Natom, rowLevU = 2 + c->Natom + c->NlevTot;
    for (int la = 0; la < c->Nla; ++la)
    {
        std::vector<int> lines, conts;
        for (int tr = 0; tr < c->Ntrans; ++tr)
        {
            const HostTrans& h = c->trans[tr];
            if (la < h.NblueLoc || la >= h.NredLoc)
                continue;
            (h.t.type == LWHIP_LINE ? lines : conts).push_back(tr);
        }
        auto isOp = [&](int tr) { return !c->atoms[c->trans[tr].atom].detailed; };
        // moment slots: the first two active atoms that have an active line here
        std::vector<int> momentAtoms;
        for (int tr : lines)
            if (isOp(tr) && std::find(momentAtoms.begin(), momentAtoms.end(), c->trans[tr].atom) == momentAtoms.end())
                momentAtoms.push_back(c->trans[tr].atom);
        auto msOf = [&](int atom) {
            for (int m = 0; m < (int)momentAtoms.size() && m < 2; ++m)
                if (momentAtoms[m] == atom)
                    return m;
            return -1;
        };
        auto atomHasLines = [&](int atom) {
            return std::find(momentAtoms.begin(), momentAtoms.end(), atom) != momentAtoms.end();
        };
        std::vector<int> mixed, pure;
        for (int tr : conts)
        {
            const HostTrans& h = c->trans[tr];
            bool isMixed = false;
            if (isOp(tr))
            {
                for (int lt : lines)
                {
                    const HostTrans& l = c->trans[lt];
                    if (l.atom != h.atom || !isOp(lt))
                        continue;
                    if (l.t.i == h.t.i || l.t.i == h.t.j || l.t.j == h.t.i || l.t.j == h.t.j)
                        isMixed = true;
                }
                if (atomHasLines(h.atom) && msOf(h.atom) < 0)
                    isMixed = true;
            }
            (isMixed ? mixed : pure).push_back(tr);
        }
        std::vector<int> order = lines;
        order.insert(order.end(), mixed.begin(), mixed.end());
        order.insert(order.end(), pure.begin(), pure.end());
        hdr[la].off = (int32_t)slots.size();
        hdr[la].nLine = (int16_t)lines.size();
        hdr[la].nMixed = (int16_t)mixed.size();
        hdr[la].nPure = (int16_t)pure.size();
        hdr[la].hasPrd = 0;
        if (c->hprd && c->hprd->NhPrd > 0)
            hdr[la].hasPrd = c->hLa2hHost[c->laStart + la] >= 0 ? 1 : 0; // hybrid PRD: the wavelengths that scatter into the PRD region (PrdTemplates.hpp:234-248)
        else
        for (int tr : lines)
            if (c->trans[tr].t.prd && c->trans[tr].rhoOff >= 0 && (isOp(tr) || c->prdDetailed))
                hdr[la].hasPrd = 1; // a wavelength the PRD rates pass visits (PrdTemplates.hpp:226-239)
        for (int tr : order)
        {
            const HostTrans& h = c->trans[tr];
            const int l = la - h.NblueLoc;
            DevSlot sl{};
            sl.cs = 0;
            sl.gi = c->levelOff[h.atom] + h.t.i;
            sl.gj = c->levelOff[h.atom] + h.t.j;
            sl.atom = h.atom;
            sl.flags = (isOp(tr) ? 1 : 0) | ((h.t.type == LWHIP_LINE && h.t.prd) ? 2 : 0);
            sl.row = h.row;
            sl.ms = isOp(tr) ? msOf(h.atom) : -1;
            for (int q = 0; q < 4; ++q)
                sl.p[q] = par[(size_t)h.parOff + 4 * l + q];
            sl.phiOff = (h.t.type == LWHIP_LINE) ? h.phiOff + (int64_t)l * c->Nrays * 2 * Ns : -1;
            sl.rhoOff = (h.rhoOff >= 0) ? h.rhoOff + (int64_t)l * Ns : -1;
            slots.push_back(sl);
            slotTr.push_back(tr);
        }
        {
            // position q of the reference's transition order among this wavelength's continua -> ordinal in the
            // [mixed][pure] slot order (kept in the q-th continuum record's _pad)
            std::vector<int> contOrd = mixed;
            contOrd.insert(contOrd.end(), pure.begin(), pure.end());
            std::vector<int> byTrQ(contOrd.size());
            for (size_t q = 0; q < byTrQ.size(); ++q)
                byTrQ[q] = (int)q;
            std::sort(byTrQ.begin(), byTrQ.end(), [&](int x, int y) { return contOrd[x] < contOrd[y]; });
            for (size_t q = 0; q < byTrQ.size(); ++q)
                slots[hdr[la].off + lines.size() + q]._pad = byTrQ[q];
        }
        c->maxL = std::max(c->maxL, (int)lines.size());
        c->maxC = std::max(c->maxC, (int)conts.size());

        // row program of this wavelength's continuum set (deduplicated by signature)
        std::vector<int> contOrder = mixed;
        contOrder.insert(contOrder.end(), pure.begin(), pure.end());
        std::vector<int> rowsNeeded = { 0, 1 };
        for (int tr : order)
        {
            if (!isOp(tr))
                continue;
            const HostTrans& h = c->trans[tr];
            const int gi = c->levelOff[h.atom] + h.t.i, gj = c->levelOff[h.atom] + h.t.j;
            for (int r : { rowEtaA + h.atom, rowLevChi + gi, rowLevChi + gj, rowLevU + gi, rowLevU + gj })
                if (std::find(rowsNeeded.begin(), rowsNeeded.end(), r) == rowsNeeded.end())
                    rowsNeeded.push_back(r);
        }
        std::sort(rowsNeeded.begin(), rowsNeeded.end());
        std::vector<int> key = contOrder;
        key.push_back(-1);
        key.insert(key.end(), rowsNeeded.begin(), rowsNeeded.end());
        auto it = progIds.find(key);
        if (it == progIds.end())
        {
            DevProgram pg{};
            pg.rowOff = (int32_t)progRows.size();
            pg.entOff = (int32_t)progEnts.size();
            // continua contribute in reference (transition index) order
            std::vector<int> byTr(contOrder.size());
            for (size_t q = 0; q < byTr.size(); ++q)
                byTr[q] = (int)q;
            std::sort(byTr.begin(), byTr.end(), [&](int x, int y) { return contOrder[x] < contOrder[y]; });
            for (int r : rowsNeeded)
            {
                DevProgRow pr{};
                pr.row = (int16_t)r;
                pr.entOff = (int32_t)(progEnts.size() - pg.entOff);
                for (int cq : byTr)
                {
                    const int tr = contOrder[cq];
                    const HostTrans& h = c->trans[tr];
                    const int gi = c->levelOff[h.atom] + h.t.i, gj = c->levelOff[h.atom] + h.t.j;
                    const bool op = isOp(tr);
                    if (r == 0)
                        progEnts.push_back(DevProgEnt{ (int16_t)cq, 0 });
                    else if (r == 1)
                        progEnts.push_back(DevProgEnt{ (int16_t)cq, 3 });
                    else if (op && r == rowEtaA + h.atom)
                        progEnts.push_back(DevProgEnt{ (int16_t)cq, 3 });
                    else if (op && r >= rowLevChi && r < rowLevU)
                    {
                        if (r == rowLevChi + gi)
                            progEnts.push_back(DevProgEnt{ (int16_t)cq, 0 });
                        if (r == rowLevChi + gj)
                            progEnts.push_back(DevProgEnt{ (int16_t)cq, 1 });
                    }
                    else if (op && r == rowLevU + gj)
                        progEnts.push_back(DevProgEnt{ (int16_t)cq, 2 });
                }
                pr.nEnt = (int16_t)(progEnts.size() - pg.entOff - pr.entOff);
                progRows.push_back(pr);
            }
            pg.nRows = (int32_t)(progRows.size() - pg.rowOff);
            pg.nEnt = (int32_t)(progEnts.size() - pg.entOff);
            if (pg.nRows > LWHIP_MAX_PROG_ROWS || pg.nEnt > LWHIP_MAX_PROG_ENTS || progs.size() >= 32767)
                return fail(LWHIP_ERR_UNSUPPORTED, "continuum row program exceeds the kernel's LDS tables");
            it = progIds.emplace(key, (int)progs.size()).first;
            progs.push_back(pg);
        }
        hdr[la].prog = (int16_t)it->second;
        // block of this wavelength in the continuum-row buffer (march kernel): the program's rows in
        // rowsNeeded order, then Vji of each continuum
        const int nContHere = (int)contOrder.size();
        hdr[la].nRows = (int32_t)rowsNeeded.size();
        hdr[la].rowBase = -1;
        if (nContHere > 0)
        {
            hdr[la].rowBase = (int32_t)c->rowsTot;
            c->rowsTot += (int64_t)rowsNeeded.size() + nContHere;
            contLaHost.push_back(la);
        }
        auto dense = [&](int r) -> int16_t {
            if (nContHere == 0)
                return -1;
            auto f = std::find(rowsNeeded.begin(), rowsNeeded.end(), r);
            return f == rowsNeeded.end() ? (int16_t)-1 : (int16_t)(f - rowsNeeded.begin());
        };
        for (size_t e = 0; e < order.size(); ++e)
        {
            DevSlot& sl = slots[hdr[la].off + e];
            const HostTrans& ht = c->trans[order[e]];
            const int gi = c->levelOff[ht.atom] + ht.t.i, gj = c->levelOff[ht.atom] + ht.t.j;
            sl.rEtaA = dense(rowEtaA + ht.atom);
            sl.rChiI = dense(rowLevChi + gi);
            sl.rChiJ = dense(rowLevChi + gj);
            sl.rUI = dense(rowLevU + gi);
            sl.rUJ = dense(rowLevU + gj);
            sl.rVji = (int16_t)((int)e >= (int)lines.size() ? (int)rowsNeeded.size() + ((int)e - (int)lines.size()) : -1);
            if (!isOp(order[e]))
                sl.rEtaA = sl.rChiI = sl.rChiJ = sl.rUI = sl.rUJ = -1;
        }
    }
    if (c->maxL > 8)
        return fail(LWHIP_ERR_UNSUPPORTED, "more than 8 lines overlap at one wavelength");
    if (c->maxC < 1)
        c->maxC = 1;

    int numCU = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0)
        numCU = prop.multiProcessorCount;

    // ---- tiles of the ray-column sweep (lwhip_raymarch.hip) -----------------------------------------------------
    // A tile = up to L consecutive wavelengths with the same slot records (same transitions in the same order, same
    // row program): one wavefront marches it, lane = (mu, wavelength).
    c->tiled = !c->is2d;
    if (c->tiled && (c->Nrays > 64 || Ns < 3))
        return fail(LWHIP_ERR_UNSUPPORTED, "the sweep needs Nrays <= 64 and Nspace >= 3");
    std::vector<DevTile> tiles;
    std::vector<int32_t> laneChunkSplit, laneChunkOrder; // lane sweep: the tail's shorter chunks (split factor, dispatch order)
    std::vector<DevTileSlot> tslots, tslotsPrd;
    std::vector<DevTileCopy> tcopies;
    std::vector<int32_t> tileRemap;
    std::vector<int32_t> chunkTile, chunkTilePrd, tileListPrd, tileSlotTr, tileSlotTrPrd, postChunkTile, postChunkTilePrd,
        postSlotTr, postCs(std::max<size_t>(slots.size(), 1), 0);
    std::function<bool(int, DevPostProg&)> lanePostProgram;
    std::vector<int> laneCsPure; // lane sweep: accumulator slot of every pure continuum inside its workgroup chunk
    if (c->tiled)
    {
        const int Nmu = c->Nrays;
        // The depth-across-lanes sweep serves the tiles whose line / mixed-continuum counts have a compiled kind and whose
        // pure continua fit its register program; anything else (and LWHIP_SWEEP=march) runs the ray-column march.
        c->laneSweep = false;
        // The pure continua of a wavelength as the lane sweep integrates them (DevPostProg): which level sums need a
        // register ("shared": several contributors, or a foreign one), which are the continuum's own term or zero.
        auto post_program = [&](int la, DevPostProg& pg) -> bool {
            pg = DevPostProg{};
            for (auto& w : pg.code)
                w = 0x9999u; // per term (+chi, -chi, U, eta) the entry it is added to: 1 .. 8 a shared sum, 9 none
            const DevLaHeader& h = hdr[la];
            const int base = h.off + h.nLine, nC = h.nMixed + h.nPure;
            if (nC > LWHIP_POST_MAXC)
                return false;
            // position q of the reference's transition order -> the continuum's slot record
            auto ordOf = [&](int q) { return slots[base + q]._pad; };
            auto rec = [&](int q) -> const DevSlot& { return slots[base + ordOf(q)]; };
            struct Contrib { int q, kind; };
            const int colChi = c->Natom, colU = c->Natom + c->NlevTot;
            auto contributors = [&](int col) {
                std::vector<Contrib> v;
                for (int q = 0; q < nC; ++q)
                {
                    const DevSlot& sl = rec(q);
                    if (!(sl.flags & 1))
                        continue;
                    if (col == sl.atom)
                        v.push_back({ q, 4 });
                    if (col == colChi + sl.gi)
                        v.push_back({ q, 1 });
                    if (col == colChi + sl.gj)
                        v.push_back({ q, 2 });
                    if (col == colU + sl.gj)
                        v.push_back({ q, 3 });
                }
                return v;
            };
            std::map<int, int> sumOf;
            for (int q = 0; q < nC; ++q)
            {
                const DevSlot& sl = rec(q);
                if (ordOf(q) < h.nMixed || !(sl.flags & 1))
                    continue; // mixed continua are integrated ray by ray; detailed atoms have rates only
                const int cols[5] = { sl.atom, colChi + sl.gi, colU + sl.gj, colChi + sl.gj, colU + sl.gi };
                for (int w = 0; w < 5; ++w)
                {
                    const auto v = contributors(cols[w]);
                    unsigned sel = 0;
                    if (v.size() == 1 && v[0].q == q)
                        sel = 8u + (unsigned)v[0].kind;
                    else if (!v.empty())
                    {
                        auto it = sumOf.find(cols[w]);
                        if (it == sumOf.end())
                        {
                            if ((int)sumOf.size() >= LWHIP_POST_SUMS)
                                return false;
                            const int m = (int)sumOf.size();
                            it = sumOf.emplace(cols[w], m).first;
                            for (const auto& cb : v) // (kind 1 .. 4 -> nibble kind - 1 of the contributor's word: the sum's index)
                                pg.code[cb.q] = (pg.code[cb.q] & ~(0xFu << (4 * (cb.kind - 1)))) | ((unsigned)(m + 1) << (4 * (cb.kind - 1)));
                        }
                        sel = (unsigned)it->second + 1u;
                    }
                    pg.ops[q] |= sel << (4 * w);
                }
            }
            return true;
        };
        lanePostProgram = post_program;
        {
            // Which sweep: the lane sweep wherever its compiled kinds cover the problem (measured on MI355X:
            // tools/size_sweep.sh, tools/cross_exp.sh, profiles/r03_size_sweep.md), the ray-column march otherwise.
            // LWHIP_SWEEP=lanes|march forces one.
            const char* sw = std::getenv("LWHIP_SWEEP");
            int D = 0, LR = 0, R = 0;
            const bool supported = lane_sweep_supported(Ns, Nmu, c->prob.formalSolver, &D, &LR, &R);
            const bool force = (sw && std::strcmp(sw, "lanes") == 0) || c->hprd != nullptr // (hybrid PRD lives in the lane sweep)
                               || c->deterministic;                                    // (and so does the fixed-order reduction)
            // (End of round 3 -- one tile per wavefront, rays split over two wavefronts for small launches and for the tail of
            // large ones, fixed-register finish: the lane sweep is the faster one at every size measured, 0.076 / 0.142 /
            // 0.187 / 0.217 / 0.315 / 0.426 ms per step at 1 280 / 5 120 / 8 192 / 10 240 / 15 360 / 20 480 wavelengths against
            // the march's 0.171 / 0.205 / 0.225 / 0.223-0.234 / 0.396 / 0.457, and 20 % faster on a 512-column batch.  The
            // march serves what the lane sweep's kinds do not cover, and LWHIP_SWEEP=march.)
            const bool prefer = force || !sw;
            if ((c->hprd || c->deterministic || !(sw && std::strcmp(sw, "march") == 0)) && env_int("LWHIP_TILE_GENERIC", 0) == 0 && supported && prefer)
            {
                bool ok = true;
                for (int la = 0; la < c->Nla && ok; ++la)
                {
                    const DevLaHeader& h = hdr[la];
                    const bool fastKind = (h.nLine == 0 && h.nMixed == 0) || (h.nLine == 1 && h.nMixed <= 1) || (h.nLine == 2 && h.nMixed == 0);
                    DevPostProg pg;
                    ok = fastKind && post_program(la, pg);
                }
                if (ok)
                {
                    c->laneSweep = true;
                    c->laneD = D;
                    c->laneLR = LR;
                    c->laneR = R;
                }
            }
        }
        if (c->hprd && !c->laneSweep)
            return fail(LWHIP_ERR_UNSUPPORTED, "hybrid PRD needs the depth-across-lanes sweep (1D, 13 <= Nspace <= 256, at most "
                                               "two lines or a line and a mixed continuum per wavelength)");
        const int L = c->laneSweep ? c->laneR : std::max(1, std::min({ 16, 64 / Nmu, env_int("LWHIP_TILE_L", 16) }));
        const int LRD = c->laneLR * c->laneD;
        c->tileL = L;
        auto pad16 = [](int64_t v) { return (v + 15) / 16 * 16; };
        auto same_structure = [&](int la, int lb) {
            const DevLaHeader &a = hdr[la], &b = hdr[lb];
            if (a.nLine != b.nLine || a.nMixed != b.nMixed || a.nPure != b.nPure || a.prog != b.prog || a.hasPrd != b.hasPrd)
                return false;
            const int n = a.nLine + a.nMixed + a.nPure;
            for (int e = 0; e < n; ++e)
                if (slotTr[a.off + e] != slotTr[b.off + e])
                    return false;
            return true;
        };
        c->rowsTileTot = c->momTot = c->phiTTot = 0;
        c->preCols = 0;
        c->maxSlotsTile = 0;
        c->tileCap = env_int("LWHIP_TILE_GENERIC", 0); // > 0: every tile through the guarded generic march (test hook)
        // (march, one problem: the launch is one round of workgroups -- one per tile, four per CU -- and lasts as long as its
        // slowest one, march + post-pass.  The tiles with two slots march longest; capping their wavelengths shortens their
        // post-pass, as long as the extra tiles still fit the round: measured 0.196 -> 0.184 ms at 10 240 wavelengths with 10
        // instead of 12, nothing beyond ~0.9 of the slots -- profiles/r03_march_bounds.md)
        auto count_tiles = [&](int capHeavy) {
            int n = 0;
            for (int la = 0; la < c->Nla; ++n)
            {
                const DevLaHeader& h = hdr[la];
                const int Lk = std::min(L, h.nLine + h.nMixed >= 2 ? capHeavy : L);
                int nl = 1;
                while (nl < Lk && la + nl < c->Nla && same_structure(la, la + nl))
                    ++nl;
                la += nl;
            }
            return n;
        };
        int autoHeavy = L;
        if (!c->laneSweep && c->batchHint <= 1 && L >= 6)
        {
            const int slots = 4 * numCU; // workgroups of two wavefronts at two wavefronts per SIMD
            for (int cap = L - 1; cap >= L - L / 6; --cap)
                if (count_tiles(cap) * 100 <= slots * 91)
                    autoHeavy = cap;
        }
        const int tileLHeavy = std::max(1, env_int("LWHIP_TILE_LH", autoHeavy)), tileLOne = std::max(1, env_int("LWHIP_TILE_L1", L));
        for (int la = 0; la < c->Nla;)
        {
            const DevLaHeader& h = hdr[la];
            // (march: a workgroup lasts its tile's march + post-pass, and the launch as long as its slowest workgroup: tiles
            // whose march is long -- two slots -- take fewer wavelengths, so that their post-pass is shorter)
            const int Lk = c->laneSweep ? L : std::min(L, h.nLine + h.nMixed >= 2 ? tileLHeavy : h.nLine + h.nMixed == 1 ? tileLOne : L);
            int nl = 1;
            while (nl < Lk && la + nl < c->Nla && same_structure(la, la + nl))
                ++nl;
            DevTile td{};
            td.la0 = la;
            td.nl = (int16_t)nl;
            td.nLine = h.nLine;
            td.nMixed = h.nMixed;
            td.hasPrd = (int16_t)h.hasPrd;
            td.slotOff = (int32_t)tslots.size();
            td.copyOff = (int32_t)tcopies.size();
            int nRowsT = 3;
            // the five continuum rows of an active slot (eta_A, chi_i, chi_j, U_i, U_j) are consecutive tile rows,
            // zero-filled where the wavelength has no such term: the march loads them unconditionally
            // which of a slot's five level sums no continuum of this wavelength feeds (the lane sweep does not read them)
            auto zero_mask = [&](const DevSlot& sl) -> int16_t {
                bool fed[5] = { false, false, false, false, false };
                for (int e = h.nLine; e < h.nLine + h.nMixed + h.nPure; ++e)
                {
                    const DevSlot& cs = slots[h.off + e];
                    if (!(cs.flags & 1))
                        continue;
                    fed[0] |= cs.atom == sl.atom;
                    fed[1] |= cs.gi == sl.gi || cs.gj == sl.gi;
                    fed[2] |= cs.gi == sl.gj || cs.gj == sl.gj;
                    fed[3] |= cs.gj == sl.gi;
                    fed[4] |= cs.gj == sl.gj;
                }
                int16_t zm = 0;
                for (int q = 0; q < 5; ++q)
                    zm |= fed[q] ? 0 : (int16_t)(1 << q);
                return zm;
            };
            auto five_rows = [&](const DevSlot& sl) -> int16_t {
                const int16_t r0 = (int16_t)nRowsT;
                // row ids as the row programs number them (the pre-pass forms the level sums directly)
                // (lane sweep: a row that no continuum of the wavelength feeds is never read -- its zero-mask bit -- and stays
                // the zero the allocation was cleared to: the pre-pass does not write it)
                const int16_t zmS = c->laneSweep ? zero_mask(sl) : (int16_t)0;
                int qi = 0;
                for (int id : { rowEtaA + sl.atom, rowLevChi + sl.gi, rowLevChi + sl.gj, rowLevU + sl.gi, rowLevU + sl.gj })
                {
                    if (zmS & (1 << qi++))
                    {
                        ++nRowsT;
                        continue;
                    }
                    if (h.nMixed + h.nPure > 0)
                        tcopies.push_back(DevTileCopy{ 0, (int16_t)id, (int16_t)nRowsT, 0 });
                    else
                        tcopies.push_back(DevTileCopy{ 3, 0, (int16_t)nRowsT, 0 });
                    ++nRowsT;
                }
                return r0;
            };
            for (int e = 0; e < h.nLine + h.nMixed; ++e)
            {
                const DevSlot& sl = slots[h.off + e];
                DevTileSlot ts{};
                ts.tr = slotTr[h.off + e];
                ts.cs = 0;
                ts.gi = sl.gi;
                ts.gj = sl.gj;
                ts.atom = sl.atom;
                ts.flags = sl.flags;
                ts.row = sl.row;
                ts.ms = sl.ms;
                ts.r0 = (sl.flags & 1) ? five_rows(sl) : (int16_t)0;
                ts.zmask = (sl.flags & 1) ? zero_mask(sl) : (int16_t)31;
                ts.rAux = -1;
                ts.e = (int16_t)e;
                ts.phiOff = -1;
                ts.rhoOff = sl.rhoOff;
                ts.hCoefOff = -1;
                if (c->hprd && e < h.nLine && c->hRhoOffHost[slotTr[h.off + e]] >= 0)
                {
                    // hybrid PRD: the coefficient block of the tile's first wavelength, and the line's first rho row
                    const HostTrans& ht = c->trans[slotTr[h.off + e]];
                    const int64_t lt0 = la - ht.NblueLoc;
                    ts.hCoefOff = c->hRhoOffHost[slotTr[h.off + e]] + lt0 * c->Nrays * 2 * Ns;
                    ts.rhoOff = ht.rhoOff; // row 0 of the line
                }
                if (e < h.nLine && c->laneSweep)
                    ts.phiOff = sl.phiOff; // the profiles are read where they lie
                else if (e < h.nLine)
                {
                    ts.phiOff = c->phiTTot;
                    c->phiTTot += (int64_t)2 * Ns * 64;
                }
                if (e < h.nLine)
                {
                    if ((sl.flags & 2) && !c->laneSweep) // (the lane sweep reads rho where it lies)
                    {
                        ts.rAux = (int16_t)nRowsT;
                        tcopies.push_back(DevTileCopy{ 2, (int16_t)e, (int16_t)nRowsT++, 0 });
                    }
                }
                else
                {
                    ts.rAux = (int16_t)nRowsT;
                    tcopies.push_back(DevTileCopy{ 1, (int16_t)(e - h.nLine), (int16_t)nRowsT++, 0 });
                }
                tslots.push_back(ts);
                const bool fastKind = (h.nLine == 0 && h.nMixed == 0) || (h.nLine == 1 && h.nMixed <= 1) || (h.nLine == 2 && h.nMixed == 0);
                if (!fastKind)
                    c->tileCap = std::max(c->tileCap, std::max((int)h.nLine, (int)h.nMixed));
            }
            td.nCopy = (int32_t)tcopies.size() - td.copyOff;
            {
                // distinct level-sum columns this tile's rows are copied from (the pre-pass sizes its LDS columns by the max)
                std::vector<int16_t> srcs;
                for (int q = td.copyOff; q < (int)tcopies.size(); ++q)
                    if (tcopies[q].kind == 0 && std::find(srcs.begin(), srcs.end(), tcopies[q].src) == srcs.end())
                        srcs.push_back(tcopies[q].src);
                c->preCols = std::max(c->preCols, (int)srcs.size());
                // ... and where each dense level-sum slot lives among them (what the pre-pass's workgroups used to
                // work out for themselves, one thread walking the copy list while 127 waited)
                const int nLev = c->Natom + 2 * c->NlevTot;
                const size_t base = tileRemap.size();
                tileRemap.resize(base + nLev + 1, (int32_t)srcs.size());
                for (size_t q = 0; q < srcs.size(); ++q)
                    if (srcs[q] >= 2 && srcs[q] - 2 < nLev)
                        tileRemap[base + (srcs[q] - 2)] = (int32_t)q;
            }
            td.nRowsT = nRowsT;
            td.rowBase = c->rowsTileTot;
            td.momBase = c->momTot;
            if (c->laneSweep)
                c->rowsTileTot += (int64_t)nRowsT * L * LRD; // [row][wavelength][k]
            else
            {
                c->rowsTileTot += (int64_t)nRowsT * Ns * 16; // rows of 16 doubles whatever L (RM_ROW)
                c->momTot += pad16((int64_t)8 * Ns * L);      // [direction][J, M1, Meta0, Meta1][Ns][L]
            }
            c->maxSlotsTile = std::max(c->maxSlotsTile, h.nLine + h.nMixed);
            tiles.push_back(td);
            la += nl;
        }
        c->nTiles = (int)tiles.size();
        if (std::getenv("LWHIP_VERBOSE"))
        {
            std::map<std::pair<int, int>, std::pair<int, int>> hist;
            for (auto& td : tiles)
            {
                auto& hh = hist[{ td.nLine, td.nMixed }];
                hh.first++;
                hh.second += td.nl;
            }
            for (auto& kv : hist)
                std::fprintf(stderr, "  tiles with %d lines + %d mixed continua: %d (%d wavelengths)\n", kv.first.first, kv.first.second,
                             kv.second.first, kv.second.second);
        }
        for (auto& td : tiles)
            if (td.nLine > 8 || td.nMixed > 8)
                return fail(LWHIP_ERR_UNSUPPORTED, "more than 8 lines (or 8 continua coupled to them) overlap at one wavelength");
    }
    if (c->tiled)
    {
        // workgroups of the sweep: contiguous, cost-balanced runs of tiles; the waves of a workgroup take its tiles in turn
        c->tileWaves = std::max(1, std::min(8, c->laneSweep ? env_int("LWHIP_LWAVES", 4) : env_int("LWHIP_TWAVES", 2)));
        if (c->laneSweep && c->deterministic)
            c->tileWaves = 1; // one wavefront per workgroup: its LDS sums are formed in program order
        else
            c->deterministic = false; // (not served by the ray-column march: the flag is ignored there)
        const int W = c->tileWaves;
        int wgPerCU = std::max(1, 8 / W);
        if (c->batchHint > 1)
            wgPerCU = 1;
        struct TileChunking
        {
            std::vector<int32_t> chunkTile, slotTr;
            std::vector<int> cs;     // per tile slot (mode 0, 2) or per wavelength slot record (mode 1)
            std::vector<int> csPure; // mode 2: per wavelength slot record, the pure continua
            int maxCT = 1;
        };
        auto tile_cost = [&](const DevTile& td) { return 1.0 + 0.35 * (td.nLine + td.nMixed); };
        // mode 0: accumulator slots for the tiles' lines and mixed continua; 1: for the pure continua (post-pass chunks);
        // 2: for both (the lane sweep's workgroups finish their tiles themselves)
        // fixedCount > 0: every chunk holds that many tiles (lane sweep: one per group of wavefronts -- a workgroup's rounds
        // are what a launch of one partial round lasts)
        auto chunk_tiles = [&](const std::vector<int32_t>& list, int tgt, int mode, int fixedCount = 0,
                               const std::vector<int32_t>* bounds = nullptr) {
            const bool pure = mode == 1;
            TileChunking ck;
            const int n = (int)list.size();
            tgt = std::max(1, std::min(tgt, n));
            if (fixedCount > 0)
                tgt = (n + fixedCount - 1) / fixedCount;
            double total = 0.0;
            for (int ti : list)
                total += tile_cost(tiles[ti]);
            ck.chunkTile.push_back(0);
            double acc = 0.0;
            int made = 0;
            if (bounds) // the chunks are given ([0, ..., n])
                ck.chunkTile.assign(bounds->begin(), bounds->end() - 1);
            for (int i = 0; i < n && !bounds; ++i)
            {
                acc += tile_cost(tiles[list[i]]);
                const int remaining = n - (i + 1), remainingChunks = tgt - (made + 1);
                const bool cut = fixedCount > 0 ? (i + 1) % fixedCount == 0 : (tgt >= n || acc >= total * (made + 1) / tgt || remaining <= remainingChunks);
                if (cut && i + 1 < n && made + 1 < tgt)
                {
                    ck.chunkTile.push_back(i + 1);
                    ++made;
                }
            }
            ck.chunkTile.push_back(n);
            const int nch = (int)ck.chunkTile.size() - 1;
            std::vector<std::vector<int>> chunkTrans(nch);
            ck.cs.assign(pure ? slots.size() : tslots.size(), 0);
            if (mode == 2)
                ck.csPure.assign(slots.size(), 0);
            for (int ch = 0; ch < nch; ++ch)
            {
                std::vector<int> slotOf(c->Ntrans, -1);
                auto slot_for = [&](int tr) {
                    if (slotOf[tr] < 0)
                    {
                        slotOf[tr] = (int)chunkTrans[ch].size();
                        chunkTrans[ch].push_back(tr);
                    }
                    return slotOf[tr];
                };
                for (int i = ck.chunkTile[ch]; i < ck.chunkTile[ch + 1]; ++i)
                {
                    const DevTile& td = tiles[list[i]];
                    if (!pure)
                        for (int e = 0; e < td.nLine + td.nMixed; ++e)
                            ck.cs[td.slotOff + e] = slot_for(tslots[td.slotOff + e].tr);
                    if (mode != 0)
                        for (int q = 0; q < td.nl; ++q)
                        {
                            const DevLaHeader& h = hdr[td.la0 + q];
                            for (int e = h.nLine + h.nMixed; e < h.nLine + h.nMixed + h.nPure; ++e)
                                (mode == 2 ? ck.csPure : ck.cs)[h.off + e] = slot_for(slotTr[h.off + e]);
                        }
                }
                ck.maxCT = std::max(ck.maxCT, (int)chunkTrans[ch].size());
            }
            ck.slotTr.assign((size_t)nch * ck.maxCT, -1);
            for (int ch = 0; ch < nch; ++ch)
                for (size_t q = 0; q < chunkTrans[ch].size(); ++q)
                    ck.slotTr[(size_t)ch * ck.maxCT + q] = chunkTrans[ch][q];
            return ck;
        };
        std::vector<int32_t> all(c->nTiles);
        for (int i = 0; i < c->nTiles; ++i)
        {
            all[i] = i;
            if (tiles[i].hasPrd)
                tileListPrd.push_back(i);
        }
        // (dispatching the lane sweep's long tasks -- tiles with more slots -- first was measured and is WORSE: 0.288 against
        // 0.204 ms at 10 240 wavelengths; co-resident long tasks slow each other down)
        int tgtSweep = std::min((2 * c->nTiles + W - 1) / W, numCU * wgPerCU); // two tasks per tile (down, up)
        // fused post-pass: the workgroup is the two wavefronts of one tile (and every tile is a workgroup)
        // (column batches too: 12.3k -> 16.0k column-iterations/s at 128 columns against a separate post launch)
        c->tileFuse = W == 2 && env_int("LWHIP_TILE_FUSE", 1) != 0 && !c->laneSweep;
        if (c->batchHint > 1 && !c->tileFuse)
            tgtSweep = std::max(1, std::min(tgtSweep, numCU * 4 / c->batchHint));
        if (c->tileFuse)
            tgtSweep = c->nTiles;
        if (!c->tileFuse)
            tgtSweep = std::max(1, tgtSweep);
        c->laneSplit = 1;
        if (c->laneSweep)
        {
            // Small problems: the launch is ONE partial round of tasks and lasts as long as its longest task (a tile with two
            // slots: ~0.13 ms alone).  While S wavefronts per tile still fit the round, a tile's rays are split over the S
            // wavefronts of a group (each finishes 1/S of the tile's points): shorter tasks, same round.
            if (!c->deterministic && c->batchHint <= 1 && W == 4)
            {
                const long slots = 2L * 4 * numCU;
                // (measured, tools/split_exp.sh: S = 2 wins while it fits; S = 4 has the same kernel time at 1 280 wavelengths
                // and a longer step -- four times the workgroups flushing their sums)
                c->laneSplit = 2L * c->nTiles <= slots ? 2 : 1;
                const int forced = env_int("LWHIP_LANE_SPLIT", 0);
                if (forced == 1 || forced == 2 || forced == 4)
                    c->laneSplit = forced;
            }
            tgtSweep = (c->nTiles * c->laneSplit + W - 1) / W; // one tile per group of wavefronts
            c->laneSplitPrd = c->laneSplit;
            if (!c->deterministic && c->batchHint <= 1 && W == 4 && !tileListPrd.empty() && env_int("LWHIP_LANE_SPLIT", 0) == 0)
                c->laneSplitPrd = 2L * (long)tileListPrd.size() <= 2L * 4 * numCU ? 2 : 1;
        }
        // post-pass workgroups: one resident set (LDS: accumulators + the threads' level columns; 8 waves per CU by registers)
        const size_t ldsPostEst = sizeof(double) * ((size_t)7 * 4 * Ns + (size_t)(c->Natom + 2 * c->NlevTot) * 128);
        const int postPerCU = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(150 * 1024) / ldsPostEst));
        int tgtPost = std::min(c->nTiles, numCU * postPerCU);
        if (c->batchHint > 1)
            tgtPost = std::max(1, std::min(tgtPost, numCU * 8 / c->batchHint));
        const int laneFixed = (c->laneSweep && env_int("LWHIP_LANE_UNIFORM", 1) != 0) ? std::max(1, W / c->laneSplit) : 0;
        TileChunking sw = chunk_tiles(all, tgtSweep, c->laneSweep ? 2 : 0, laneFixed);
        // Lane sweep, more workgroups than the chip holds at once (one problem, 2 workgroups of 4 wavefronts per CU): the
        // launch ends with workgroups that started late and have the chip to themselves -- whole tiles of 0.1 ms.  The
        // chunks that are dispatched last (the lightest: heaviest go first) are cut into smaller ones whose tiles' rays are
        // split over 2 or 4 wavefronts: the same work in shorter pieces, so the tail drains evenly.
        {
            const int slotsWg = 2 * numCU;
            const int nch0 = (int)sw.chunkTile.size() - 1;
            const int tailS = env_int("LWHIP_LANE_TAIL", 2);
            // (and in a launch of ONE partial round -- more than half a round of whole tiles, so that not every tile can be
            // split -- the heaviest chunks are the ones cut, as many as still fit the round: the launch lasts as long as its
            // longest workgroup)
            const bool oneRound = nch0 <= slotsWg;
            if (c->laneSweep && c->batchHint <= 1 && !c->deterministic && W == 4 && c->laneSplit == 1 && laneFixed == W
                && (nch0 > slotsWg || (oneRound && nch0 > slotsWg / 2)) && (tailS == 2 || tailS == 4))
            {
                auto tcost = [&](int i) { return 1.0 + 0.5 * tiles[i].nLine + 0.3 * tiles[i].nMixed + (tiles[i].nLine == 0 ? -0.25 : 0.0); };
                std::vector<double> cost(nch0, 0.0);
                for (int ch = 0; ch < nch0; ++ch)
                    for (int i = sw.chunkTile[ch]; i < sw.chunkTile[ch + 1]; ++i)
                        cost[ch] += tcost(i);
                std::vector<int32_t> ord(nch0);
                for (int ch = 0; ch < nch0; ++ch)
                    ord[ch] = ch;
                std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return cost[x] > cost[y]; });
                std::vector<char> cut(nch0, 0);
                if (nch0 > slotsWg)
                {
                    // the first `keep` chunks of the dispatch order stay whole; as many of the rest are cut in two as fill the
                    // launch's last round of workgroup slots exactly: whole + 2 cut = rounds x slots (measured at 10 240
                    // wavelengths, 857 chunks on 512 slots: keep 512 / 600 / 700 / 780 / all -> 0.152 / 0.150 / 0.139 / 0.148 /
                    // 0.155 ms; the rule gives 690)
                    const int rounds = (nch0 + slotsWg - 1) / slotsWg;
                    const int keepAuto = std::max(0, std::min(nch0, 2 * nch0 - rounds * slotsWg));
                    const int keep = std::min(nch0, std::max(0, env_int("LWHIP_LANE_KEEP", tailS == 2 ? keepAuto : slotsWg)));
                    for (int q = keep; q < nch0; ++q)
                        cut[ord[q]] = 1;
                }
                else
                {
                    // one partial round: a cut chunk becomes tailS / 1 workgroups; the heaviest first, while they fit.
                    // A round that is (nearly) full does not run as one: the dispatcher does not co-schedule every workgroup
                    // (6 144 wavelengths = 512 chunks on 512 slots took two task lengths, 0.115 ms) -- then EVERY chunk is cut
                    // and the launch is two rounds of half-length tasks
                    const int extra = tailS - 1;
                    const int fullPct = env_int("LWHIP_LANE_FULL", 85);
                    int k = std::min(nch0, std::max(0, (slotsWg - nch0) / extra));
                    if (tailS == 2 && nch0 * 100 > slotsWg * fullPct)
                        k = nch0;
                    for (int q = 0; q < k; ++q)
                        cut[ord[q]] = 1;
                }
                const int piece = W / tailS; // tiles per piece
                std::vector<int32_t> bounds, split, firstNew(nch0, 0), nNew(nch0, 0);
                bounds.push_back(0);
                for (int ch = 0; ch < nch0; ++ch)
                {
                    firstNew[ch] = (int)bounds.size() - 1;
                    const int b0 = sw.chunkTile[ch], b1 = sw.chunkTile[ch + 1];
                    if (!cut[ch])
                    {
                        bounds.push_back(b1);
                        split.push_back(1);
                    }
                    else
                        for (int b = b0; b < b1; b += piece)
                        {
                            bounds.push_back(std::min(b + piece, b1));
                            split.push_back(tailS);
                        }
                    nNew[ch] = (int)bounds.size() - 1 - firstNew[ch];
                }
                sw = chunk_tiles(all, 0, 2, 0, &bounds);
                laneChunkSplit = split;
                for (int q = 0; q < nch0; ++q)
                    for (int e = 0; e < nNew[ord[q]]; ++e)
                        laneChunkOrder.push_back(firstNew[ord[q]] + e);
            }
        }
        TileChunking po = chunk_tiles(all, tgtPost, 1);
        TileChunking swPrd, poPrd;
        if (!tileListPrd.empty())
        {
            // (fused post-pass: one tile per workgroup here too)
            swPrd = chunk_tiles(tileListPrd,
                                c->laneSweep ? ((int)tileListPrd.size() * c->laneSplitPrd + W - 1) / W
                                : c->tileFuse ? (int)tileListPrd.size()
                                              : std::min((2 * (int)tileListPrd.size() + W - 1) / W, numCU * wgPerCU),
                                0, c->laneSweep ? std::max(1, W / c->laneSplitPrd) : 0);
            poPrd = chunk_tiles(tileListPrd, std::min((int)tileListPrd.size(), numCU * postPerCU), 1);
        }
        // both chunkings share one accumulator-slot capacity per kind
        c->maxCTTile = std::max(sw.maxCT, swPrd.maxCT);
        c->maxCTPost = po.maxCT;
        auto widen = [&](TileChunking& ck, int maxCT) {
            if (ck.maxCT == maxCT)
                return;
            const int nch = (int)ck.chunkTile.size() - 1;
            std::vector<int32_t> w((size_t)nch * maxCT, -1);
            for (int ch = 0; ch < nch; ++ch)
                for (int q = 0; q < ck.maxCT; ++q)
                    w[(size_t)ch * maxCT + q] = ck.slotTr[(size_t)ch * ck.maxCT + q];
            ck.slotTr.swap(w);
            ck.maxCT = maxCT;
        };
        widen(sw, c->maxCTTile);
        if (!tileListPrd.empty())
            widen(swPrd, c->maxCTTile);
        for (size_t e = 0; e < tslots.size(); ++e)
            tslots[e].cs = sw.cs[e];
        laneCsPure = sw.csPure;
        chunkTile = sw.chunkTile;
        tileSlotTr = sw.slotTr;
        c->nTileChunks = (int)chunkTile.size() - 1;
        if (c->deterministic)
        {
            // per transition: the (workgroup, slot) pairs that hold its sums, in workgroup order
            auto lists = [&](const std::vector<int32_t>& slotTrTab, int nch, std::vector<int32_t>& off, std::vector<int32_t>& ent) {
                std::vector<std::vector<int32_t>> per(c->Ntrans);
                for (int ch = 0; ch < nch; ++ch)
                    for (int q = 0; q < c->maxCTTile; ++q)
                    {
                        const int tr = slotTrTab[(size_t)ch * c->maxCTTile + q];
                        if (tr >= 0)
                            per[tr].push_back(ch * c->maxCTTile + q);
                    }
                off.assign(1, 0);
                ent.clear();
                for (int tr = 0; tr < c->Ntrans; ++tr)
                {
                    ent.insert(ent.end(), per[tr].begin(), per[tr].end());
                    off.push_back((int32_t)ent.size());
                }
                if (ent.empty())
                    ent.push_back(0);
            };
            std::vector<int32_t> o, e;
            lists(tileSlotTr, c->nTileChunks, o, e);
            HIP_TRY(c->detOff.upload(o));
            HIP_TRY(c->detEnt.upload(e));
            if (!tileListPrd.empty())
            {
                lists(swPrd.slotTr, (int)swPrd.chunkTile.size() - 1, o, e);
                HIP_TRY(c->detOffPrd.upload(o));
                HIP_TRY(c->detEntPrd.upload(e));
            }
            const size_t nch = std::max<size_t>(c->nTileChunks, tileListPrd.empty() ? 0 : swPrd.chunkTile.size() - 1);
            HIP_TRY(c->detSlab.alloc(nch * c->maxCTTile * 4 * Ns));
        }
        postChunkTile = po.chunkTile;
        postSlotTr = po.slotTr;
        for (size_t e = 0; e < slots.size(); ++e)
            postCs[e] = po.cs[e];
        c->nPostChunks = (int)postChunkTile.size() - 1;
        c->nTilesPrd = (int)tileListPrd.size();
        c->nTileChunksPrd = c->nPostChunksPrd = 0;
        if (!tileListPrd.empty())
        {
            tslotsPrd = tslots;
            for (size_t e = 0; e < tslots.size(); ++e)
                tslotsPrd[e].cs = swPrd.cs[e];
            chunkTilePrd = swPrd.chunkTile;
            tileSlotTrPrd = swPrd.slotTr;
            postChunkTilePrd = poPrd.chunkTile;
            c->nTileChunksPrd = (int)chunkTilePrd.size() - 1;
            c->nPostChunksPrd = (int)postChunkTilePrd.size() - 1;
        }
        // LDS: the workgroup's accumulators + a 2 KB exchange row per wave (sweep); accumulators + continuum columns (post)
        const size_t ldsSweep = sizeof(double) * ((size_t)c->maxCTTile * 4 * Ns + std::max((size_t)W * 5 * 16 * 9, (size_t)16 + (size_t)(c->Natom + 2 * c->NlevTot) * 128));
        const size_t ldsPost = sizeof(double) * (16 + (size_t)(c->Natom + 2 * c->NlevTot) * 128);
        if (!c->laneSweep && (ldsSweep > 150 * 1024 || ldsPost > 150 * 1024))
            return fail(LWHIP_ERR_UNSUPPORTED, "problem does not fit the 160 KB LDS budget of one workgroup");
        if (c->laneSweep && sizeof(double) * (size_t)c->maxCTTile * 4 * (c->laneLR * c->laneD) > 64 * 1024)
            return fail(LWHIP_ERR_UNSUPPORTED, "more transitions per workgroup than the lane sweep's LDS accumulators hold");
    }

    const int NR2 = 2 * c->Nrays;
    // ---- device tables ---------------------------------------------------------------------------------
    std::vector<DevTrans> dt(c->Ntrans);
    std::vector<int32_t> li(c->Ntrans), lj(c->Ntrans);
    for (int tr = 0; tr < c->Ntrans; ++tr)
    {
        const HostTrans& h = c->trans[tr];
        DevTrans& d = dt[tr];
        d.type = h.t.type;
        d.atom = h.atom;
        d.gi = c->levelOff[h.atom] + h.t.i;
        d.gj = c->levelOff[h.atom] + h.t.j;
        d.Nblue = h.NblueLoc;
        d.Nred = h.NredLoc;
        d.prd = (h.t.type == LWHIP_LINE && h.t.prd) ? 1 : 0;
        d.op = c->atoms[h.atom].detailed ? 0 : 1;
        d.row = h.row;
        d._pad = 0;
        d.parOff = h.parOff;
        d.phiOff = h.phiOff;
        d.rhoOff = h.rhoOff;
        li[tr] = h.t.i;
        lj[tr] = h.t.j;
    }
    std::vector<int32_t> rayAll(NR2), rayUp(c->Nrays);
    for (int r = 0; r < NR2; ++r)
        rayAll[r] = r;
    for (int m = 0; m < c->Nrays; ++m)
        rayUp[m] = 2 * m + 1;
    std::vector<int32_t> aNl(c->Natom), aDet(c->Natom);
    for (int ia = 0; ia < c->Natom; ++ia)
    {
        aNl[ia] = c->atoms[ia].Nlevel;
        aDet[ia] = c->atoms[ia].detailed ? 1 : 0;
    }
    std::vector<int32_t> atomTrOff32(c->atomTrOff.begin(), c->atomTrOff.end());

    // full own grids and trapezoid weights of the lines (device Voigt profiles)
    {
        std::vector<double> lw, lq;
        for (auto& h : c->trans)
        {
            h.waveOff = -1;
            if (h.t.type != LWHIP_LINE)
                continue;
            h.waveOff = (int64_t)lw.size();
            const int len = h.t.Nred - h.t.Nblue;
            for (int lt = 0; lt < len; ++lt)
            {
                lw.push_back(h.t.wavelength[lt]);
                lq.push_back(trans_wlambda(h.t, lt));
            }
        }
        if (lw.empty())
        {
            lw.push_back(0.0);
            lq.push_back(0.0);
        }
        HIP_TRY(c->lineWave.upload(lw));
        HIP_TRY(c->lineWlam.upload(lq));
    }
    HIP_TRY(c->par.upload(par));
    HIP_TRY(c->dtrans.upload(dt));
    HIP_TRY(c->laHdr.upload(hdr));
    HIP_TRY(c->slots.upload(slots));
    {
        std::vector<int32_t> st32(slotTr.begin(), slotTr.end());
        if (st32.empty())
            st32.push_back(0);
        HIP_TRY(c->slotTrD.upload(st32));
    }
    c->hdrHost = hdr;
    c->contLaHost = contLaHost;
    HIP_TRY(c->progs.upload(progs));
    HIP_TRY(c->progRows.upload(progRows));
    HIP_TRY(c->progEnts.upload(progEnts));
    if (contLaHost.empty())
        contLaHost.push_back(0);
    c->nContLa = (c->rowsTot > 0) ? (int)contLaHost.size() : 0;
    HIP_TRY(c->contLa.upload(contLaHost));
    HIP_TRY(c->rayAll.upload(rayAll));
    HIP_TRY(c->rayUp.upload(rayUp));
    if (c->tiled)
    {
        auto up = [](auto& buf, auto& v) -> hipError_t {
            if (v.empty())
                v.resize(1);
            return buf.upload(v);
        };
        std::vector<DevContRec> cr(std::max<size_t>(slots.size(), 1), DevContRec{});
        std::vector<DevPostProg> pp;
        {
            // the continua of every wavelength in transition order, ready-to-use offsets and level-column slots
            for (int la = 0; la < c->Nla; ++la)
            {
                const DevLaHeader& h = hdr[la];
                const int base = h.off + h.nLine;
                for (int q = 0; q < h.nMixed + h.nPure; ++q)
                {
                    const int ord = slots[base + q]._pad;
                    const DevSlot& sl = slots[base + ord];
                    DevContRec r{};
                    r.nI = (uint32_t)((size_t)sl.gi * Ns * sizeof(double));
                    r.nJ = (uint32_t)((size_t)sl.gj * Ns * sizeof(double));
                    r.rat = (uint32_t)((size_t)sl.row * Ns * sizeof(double));
                    r.pOff = (uint32_t)((size_t)(h.nLine + ord) * sizeof(DevSlot) + offsetof(DevSlot, p));
                    r.sEta = (uint16_t)sl.atom;
                    r.sChiI = (uint16_t)(c->Natom + sl.gi);
                    r.sChiJ = (uint16_t)(c->Natom + sl.gj);
                    r.sUi = (uint16_t)(c->Natom + c->NlevTot + sl.gi);
                    r.sUj = (uint16_t)(c->Natom + c->NlevTot + sl.gj);
                    r.flags = (uint8_t)((sl.flags & 1) | (ord >= h.nMixed ? 2 : 0));
                    r.ms = (int8_t)sl.ms;
                    r.tr = (int16_t)slotTr[base + ord];
                    r.cs = (int16_t)((c->laneSweep && ord >= h.nMixed) ? laneCsPure[base + ord] : 0);
                    cr[base + q] = r;
                }
            }
            HIP_TRY(c->contRec.upload(cr));
        }
        if (c->laneSweep)
        {
            pp.assign(std::max<size_t>(tiles.size(), 1), DevPostProg{});
            for (size_t t = 0; t < tiles.size(); ++t)
                if (!lanePostProgram(tiles[t].la0, pp[t]))
                    return fail(LWHIP_ERR_UNSUPPORTED, "lane sweep: post program");
            HIP_TRY(c->postProg.upload(pp));
            if (std::getenv("LWHIP_VERBOSE"))
            {
                // which program words occur (the finish serves the common ones with fixed registers)
                std::map<std::pair<unsigned, unsigned>, int> hist;
                for (size_t t = 0; t < tiles.size(); ++t)
                {
                    const DevLaHeader& h = hdr[tiles[t].la0];
                    for (int q = 0; q < h.nMixed + h.nPure; ++q)
                        hist[{ pp[t].code[q], pp[t].ops[q] }]++;
                }
                for (const auto& kv : hist)
                    std::fprintf(stderr, "  finish program word code %#x ops %#x: %d continua\n", kv.first.first, kv.first.second, kv.second);
            }
        }
        HIP_TRY(up(c->tiles, tiles));
        HIP_TRY(up(c->tslots, tslots));
        HIP_TRY(up(c->tslotsPrd, tslotsPrd));
        HIP_TRY(up(c->tcopies, tcopies));
        HIP_TRY(up(c->tileRemap, tileRemap));
        HIP_TRY(up(c->chunkTile, chunkTile));
        c->chunkOrderOn = false;
        c->chunkSplitOn = false;
        if (!laneChunkOrder.empty())
        {
            HIP_TRY(up(c->chunkOrder, laneChunkOrder));
            HIP_TRY(up(c->chunkSplit, laneChunkSplit));
            c->chunkOrderOn = c->chunkSplitOn = true;
        }
        const int nchAll = (int)chunkTile.size() - 1;
        // more workgroups than the chip holds at once (2 per CU): the heavy ones go first, the light ones fill the tail
        // (measured: 8 192 wavelengths 0.188 -> 0.179 ms, 10 240: 0.203 -> 0.200; lightest first: 0.196 / 0.207)
        const int laneOrder = nchAll > 2 * numCU ? 1 : 0; // heaviest first
        if (c->laneSweep && c->batchHint <= 1 && laneOrder != 0 && laneChunkOrder.empty())
        {
            // the chunks stay what they are (neighbouring wavelengths: few accumulator slots); only the order in which the
            // workgroups take them changes: 1 = heaviest first, 2 = lightest first
            const int nch = nchAll;
            std::vector<double> cost(nch, 0.0);
            for (int ch = 0; ch < nch; ++ch)
                for (int i = chunkTile[ch]; i < chunkTile[ch + 1]; ++i)
                    cost[ch] += 1.0 + 0.5 * tiles[i].nLine + 0.3 * tiles[i].nMixed + (tiles[i].nLine == 0 ? -0.25 : 0.0);
            std::vector<int32_t> order(nch);
            for (int ch = 0; ch < nch; ++ch)
                order[ch] = ch;
            const bool heavyFirst = laneOrder == 1;
            std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return heavyFirst ? cost[x] > cost[y] : cost[x] < cost[y]; });
            HIP_TRY(up(c->chunkOrder, order));
            c->chunkOrderOn = true;
        }
        if (c->laneSweep)
        {
            // the flat records of the lane sweep's tasks (lwhip_internal.h: DevLaneWg, DevLaneTile, DevLaneRay, DevLaneFin)
            const int L = c->tileL;
            auto wg_records = [&](const std::vector<int32_t>& chTile, const std::vector<int32_t>* order, const std::vector<int32_t>* split,
                                  int launchSplit) {
                const int nch = (int)chTile.size() - 1;
                std::vector<DevLaneWg> w(std::max(nch, 1));
                for (int b = 0; b < nch; ++b)
                {
                    const int ch = order ? (*order)[b] : b;
                    const int S = split ? (*split)[ch] : launchSplit;
                    w[b] = DevLaneWg{ ch, S > 1 ? S : 1, chTile[ch], chTile[ch + 1] };
                }
                return w;
            };
            std::vector<int32_t> orderHost;
            if (c->chunkOrderOn)
            {
                orderHost.resize(c->chunkOrder.n);
                HIP_TRY(hipMemcpy(orderHost.data(), c->chunkOrder.p, orderHost.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            }
            auto wgs = wg_records(chunkTile, c->chunkOrderOn ? &orderHost : nullptr, c->chunkSplitOn ? &laneChunkSplit : nullptr, c->laneSplit);
            HIP_TRY(c->laneWg.upload(wgs));
            auto tile_records = [&](const std::vector<DevTileSlot>& ts) {
                std::vector<DevLaneTile> v(std::max<size_t>(tiles.size(), 1), DevLaneTile{});
                for (size_t t = 0; t < tiles.size(); ++t)
                {
                    const DevLaHeader& h = hdr[tiles[t].la0];
                    v[t].td = tiles[t];
                    v[t].nCont = h.nPure > 0 ? h.nMixed + h.nPure : 0;
                    v[t].nContAll = std::min(h.nMixed + h.nPure, (int)LWHIP_POST_MAXC);
                    {
                        // the finish's named-register form serves the tiles whose continua all carry the common words
                        bool fast = true;
                        const int nC = h.nMixed + h.nPure;
                        for (int q = 0; q < nC && q < LWHIP_POST_MAXC; ++q)
                        {
                            const DevContRec& r = cr[h.off + h.nLine + q];
                            const unsigned code = (r.flags & 1) ? pp[t].code[q] : 0x9999u, ops = pp[t].ops[q];
                            if (code != 0x9999u && code != 0x1239u && code != 0x4569u)
                                fast = false;
                            if ((r.flags & 2) && (r.flags & 1) && ops != 0x03291u && ops != 0x06594u && ops != 0x0AB9Cu)
                                fast = false;
                        }
                        v[t].finFast = (fast && env_int("LWHIP_FIN_FAST", 1) != 0) ? 1 : 0;
                    }
                    for (int e = 0; e < std::min(2, tiles[t].nLine + tiles[t].nMixed); ++e)
                        v[t].slot[e] = ts[tiles[t].slotOff + e];
                }
                return v;
            };
            {
                auto v = tile_records(tslots);
                HIP_TRY(c->laneTiles.upload(v));
                if (!tslotsPrd.empty() && tslotsPrd.size() == tslots.size() && !tileListPrd.empty())
                {
                    v = tile_records(tslotsPrd);
                    HIP_TRY(c->laneTilesPrd.upload(v));
                    auto wp = wg_records(chunkTilePrd, nullptr, nullptr, c->laneSplitPrd);
                    HIP_TRY(c->laneWgPrd.upload(wp));
                }
            }
            std::vector<DevLaneRay> rays(std::max<size_t>(tiles.size() * L, 1), DevLaneRay{});
            std::vector<DevLaneFin> fin(std::max<size_t>(tiles.size() * LWHIP_POST_MAXC, 1), DevLaneFin{});
            std::vector<double> finPar(std::max<size_t>(tiles.size() * LWHIP_POST_MAXC * L * 3, 1), 0.0);
            for (size_t t = 0; t < tiles.size(); ++t)
            {
                const DevTile& td = tiles[t];
                const DevLaHeader& h0 = hdr[td.la0];
                const int nSl = std::min(2, td.nLine + td.nMixed);
                for (int r = 0; r < L; ++r)
                {
                    const int la = td.la0 + std::min(r, td.nl - 1);
                    DevLaneRay& lr = rays[t * L + r];
                    lr.wav = c->prob.wavelength[c->laStart + la];
                    for (int e = 0; e < nSl; ++e)
                        for (int q = 0; q < 4; ++q)
                            lr.p[e][q] = slots[hdr[la].off + tslots[td.slotOff + e].e].p[q];
                }
                const int nC = h0.nMixed + h0.nPure;
                for (int q = 0; q < nC && q < LWHIP_POST_MAXC; ++q)
                {
                    const DevContRec& r = cr[h0.off + h0.nLine + q];
                    uint32_t w[8];
                    std::memcpy(w, &r, sizeof(w));
                    // the rows of the tile's slots this continuum feeds (zero_mask's matching, per continuum)
                    uint32_t feed = 0;
                    {
                        const int ord = slots[h0.off + h0.nLine + q]._pad; // transition-order position q -> slot ordinal
                        const DevSlot& cs = slots[h0.off + h0.nLine + ord];
                        for (int e = 0; e < nSl; ++e)
                        {
                            const DevTileSlot& us = tslots[td.slotOff + e];
                            uint32_t b = 0;
                            if ((cs.flags & 1) && (us.flags & 1))
                            {
                                b |= cs.atom == us.atom ? 1u : 0u;
                                b |= cs.gi == us.gi ? 2u : 0u;
                                b |= cs.gj == us.gi ? 4u : 0u;
                                b |= cs.gi == us.gj ? 8u : 0u;
                                b |= cs.gj == us.gj ? 16u : 0u;
                                b |= cs.gj == us.gi ? 32u : 0u;
                                b |= cs.gj == us.gj ? 64u : 0u;
                            }
                            if (e >= td.nLine && ord == e - td.nLine)
                                b |= 128u;
                            feed |= b << (8 * e);
                        }
                    }
                    fin[t * LWHIP_POST_MAXC + q] = DevLaneFin{ w[0], w[1], w[2], w[6], w[7], pp[t].code[q], pp[t].ops[q], feed };
                    for (int rr = 0; rr < L; ++rr)
                    {
                        const int la = td.la0 + std::min(rr, td.nl - 1);
                        const char* lsB = (const char*)&slots[hdr[la].off];
                        for (int k = 0; k < 3; ++k)
                        {
                            double v;
                            std::memcpy(&v, lsB + r.pOff + 8 * k, sizeof(double));
                            finPar[((t * LWHIP_POST_MAXC + q) * L + rr) * 3 + k] = v;
                        }
                    }
                }
            }
            HIP_TRY(c->laneRays.upload(rays));
            HIP_TRY(c->laneFin.upload(fin));
            HIP_TRY(c->laneFinPar.upload(finPar));
        }
        HIP_TRY(up(c->chunkTilePrd, chunkTilePrd));
        HIP_TRY(up(c->tileListPrd, tileListPrd));
        HIP_TRY(up(c->tileSlotTr, tileSlotTr));
        HIP_TRY(up(c->tileSlotTrPrd, tileSlotTrPrd));
        HIP_TRY(up(c->postChunkTile, postChunkTile));
        HIP_TRY(up(c->postChunkTilePrd, postChunkTilePrd));
        HIP_TRY(up(c->postSlotTr, postSlotTr));
        HIP_TRY(up(c->postCs, postCs));
    }
    HIP_TRY(c->transLi.upload(li));
    HIP_TRY(c->transLj.upload(lj));
    HIP_TRY(c->atomNlevel.upload(aNl));
    HIP_TRY(c->atomDetailed.upload(aDet));
    HIP_TRY(c->atomTrOffD.upload(atomTrOff32));
    HIP_TRY(c->atomGammaOff.upload(c->gammaOff));
    return LWHIP_OK;
}

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
    HIP_TRY(c->aDamp.alloc((size_t)std::max(c->Nline, 1) * Ns));
    HIP_TRY(hipMemset(c->aDamp.p, 0, c->aDamp.n * sizeof(double)));
    HIP_TRY(c->Qelast.alloc((size_t)std::max(c->Nline, 1) * Ns));
    HIP_TRY(hipMemset(c->Qelast.p, 0, c->Qelast.n * sizeof(double)));
    if (c->laneSweep)
    {
        // the depth-only pools the lane sweep reads, back to back (+ padding: a ray's last block reads past its row)
        const size_t nN = (size_t)c->NlevTot * Ns, nW = (size_t)std::max(c->Nline, 1) * Ns, nR = (size_t)std::max(c->Ncont, 1) * Ns;
        const size_t nG = (size_t)4 * (c->laneLR * c->laneD + 2);
        HIP_TRY(c->depArena.alloc(nN + nW + nR + nG + 16));
        HIP_TRY(hipMemset(c->depArena.p, 0, c->depArena.n * sizeof(double)));
        c->n.view(c->depArena.p, nN);
        c->wphi.view(c->depArena.p + nN, nW);
        c->ratio.view(c->depArena.p + nN + nW, nR);
        c->geoT.view(c->depArena.p + nN + nW + nR, nG);
        HIP_TRY(c->bcPlanck.alloc(4 * Nla));
        HIP_TRY(hipMemset(c->bcPlanck.p, 0, c->bcPlanck.n * sizeof(double)));
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
    HIP_TRY(c->Cmat.alloc((size_t)std::max<int64_t>(c->gammaTot, 1)));
    HIP_TRY(hipMemset(c->Cmat.p, 0, c->Cmat.n * sizeof(double)));
    HIP_TRY(c->Rij.alloc((size_t)std::max(c->Ntrans, 1) * Ns));
    HIP_TRY(c->Rji.alloc((size_t)std::max(c->Ntrans, 1) * Ns));
    // reduce buffer: [Ntrans,4,Ns] partial sums followed by one (dJMax, idx) slot per shard
    HIP_TRY(c->red.alloc((size_t)c->Ntrans * 4 * Ns + 2 * (size_t)c->worldSize * (1 + c->prdLines.size())));
    HIP_TRY(hipMemset(c->red.p, 0, c->red.n * sizeof(double)));
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
            HIP_TRY(hipMemset(c->rowsTile.p, 0, c->rowsTile.n * sizeof(double)));
        }
        HIP_TRY(c->momTile.alloc((size_t)std::max<int64_t>(c->momTot, 1)));
        HIP_TRY(c->phiT.alloc((size_t)std::max<int64_t>(c->phiTTot, 1)));
        HIP_TRY(hipMemset(c->momTile.p, 0, c->momTile.n * sizeof(double)));
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
        // rates2d: one thread per point per wavelength group; as many groups as keep every wavefront resident at the
        // kernel's 3 wavefronts per SIMD (256 x 82: 9 groups = 2 952 wavefronts on 3 072 slots; 7 were 2.2 per SIMD,
        // 10 spill into a second round)
        {
            int numCU = 256;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0)
                numCU = prop.multiProcessorCount;
            const size_t wavesPerGroup = 2 * ((Ns + 127) / 128);
            c->groups2d = (int)std::max<size_t>(1, std::min<size_t>({ (size_t)16, (size_t)c->batch2d, (size_t)numCU * 12 / wavesPerGroup }));
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

#define H2D(dst, src, count)                                                                           \
    HIP_TRY(hipMemcpyAsync((dst), (src), (size_t)(count) * sizeof(double), hipMemcpyHostToDevice, c->stream))
#define D2H(dst, src, count)                                                                           \
    HIP_TRY(hipMemcpyAsync((dst), (src), (size_t)(count) * sizeof(double), hipMemcpyDeviceToHost, c->stream))

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
        a.laneWg = c->laneWg.p;
        a.laneTiles = c->laneTiles.p;
        a.laneRays = c->laneRays.p;
        a.laneFin = c->laneFin.p;
        a.laneFinPar = c->laneFinPar.p;
        a.bcPlanck = c->bcPlanck.p;
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
        return hipMemcpyAsync(dev.p, &host, sizeof(TileArgs), hipMemcpyHostToDevice, c->stream);
    };
    HIP_TRY(put(c->dtargs, c->htargs, a));
    if (c->nTileChunksPrd > 0)
    {
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
static bool retile_args(lwhip_context* c, RetileArgs& r)
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
    (void)c;
    TileDyn d{};
    d.nPass = upOnly ? 1 : 2;
    d.lambdaIterate = lambdaIterate;
    d._pad = 0;
    return d;
}

// one sweep over this context's wavelengths (rates = the full iteration, else formal_sol): pre-pass (the tiles' row
// blocks, the depth table), the march, and -- unless the sweep's workgroups finish their tiles themselves -- the post-pass
hipError_t run_sweep(lwhip_context* c, const TileDyn& dyn, bool rates, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    TileDyn td = dyn;
    const bool tprd = dyn.prdOnly != 0; // the PRD rates pass: its own tile list and chunking
    if (tprd && c->nTileChunksPrd == 0)
        return hipSuccess;
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
                              c->tileWaves, nullptr, 0, c->stream);
    if (e1)
        (void)hipEventRecord(e1, c->stream);
    if (e == hipSuccess && rates && c->laneSweep && c->deterministic)
        e = launch_det_reduce(c->detSlab.p, tprd ? c->detOffPrd.p : c->detOff.p, tprd ? c->detEntPrd.p : c->detEnt.p, c->red8.p, c->Ns,
                              c->Ntrans, c->maxCTTile, c->stream);
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
const char* lwhip_last_error(void)
{
    return g_lastError.c_str();
}

int lwhip_abi_version(void)
{
    return LWHIP_ABI_VERSION;
}

int lwhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int lwhip_create(const lwhip_problem* prob, const lwhip_options* opts, lwhip_context** out)
{
    if (!out)
        return fail(LWHIP_ERR_INVALID, "null out pointer");
    *out = nullptr;
    std::string why;
    int st = validate(prob, why);
    if (st != LWHIP_OK)
        return fail(st, "lwhip_create: " + why);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(LWHIP_ERR_DEVICE, "lwhip_create: no HIP device visible (this library has no CPU path)");

    auto c = new lwhip_context();
    c->prob = *prob;
    c->atoms.assign(prob->atoms, prob->atoms + prob->Natom);
    c->device = opts ? opts->device : 0;
    c->batchHint = opts ? (opts->flags & 0xffff) : 0;
    c->prdDetailed = opts && (opts->flags & LWHIP_OPT_PRD_DETAILED);
    c->hprd = opts ? opts->hprd : nullptr;
    c->deterministic = ((opts && (opts->flags & LWHIP_OPT_DETERMINISTIC)) || env_int("LWHIP_DETERMINISTIC", 0) == 1)
                       && (!opts || (opts->flags & 0xffff) <= 1);
    if (c->device < 0 || c->device >= ndev)
    {
        delete c;
        return fail(LWHIP_ERR_INVALID, "lwhip_create: bad device ordinal");
    }
    c->laStart = 0;
    c->laEnd = prob->Nlambda;
    if (opts && !(opts->laStart == 0 && opts->laEnd == 0))
    {
        c->laStart = opts->laStart;
        c->laEnd = opts->laEnd;
        if (c->laStart < 0 || c->laEnd > prob->Nlambda || c->laEnd <= c->laStart)
        {
            delete c;
            return fail(LWHIP_ERR_INVALID, "lwhip_create: bad wavelength shard");
        }
    }
    if (opts && opts->worldSize > 1)
    {
        c->worldSize = opts->worldSize;
        c->worldRank = opts->worldRank;
        if (c->worldRank < 0 || c->worldRank >= c->worldSize)
        {
            delete c;
            return fail(LWHIP_ERR_INVALID, "lwhip_create: bad worldRank");
        }
    }
    c->is2d = prob->grid2d != nullptr;
    c->Nx = c->is2d ? prob->grid2d->Nx : 1;
    c->Nla = c->laEnd - c->laStart;
    c->Ns = prob->Nspace;
    c->Nrays = prob->Nrays;
    c->Natom = prob->Natom;

    auto bail = [&](int code) {
        delete c;
        return code;
    };
    if (hipSetDevice(c->device) != hipSuccess)
        return bail(fail(LWHIP_ERR_DEVICE, "hipSetDevice failed"));
    if (opts && opts->stream)
        c->stream = (hipStream_t)opts->stream;
    else
    {
        if (hipStreamCreate(&c->ownStream) != hipSuccess)
            return bail(fail(LWHIP_ERR_DEVICE, "hipStreamCreate failed"));
        c->stream = c->ownStream;
    }
    if (env_int("LWHIP_CHECK_ZERO", 0) == 1)
    {
        if (hipHostMalloc((void**)&c->zeroCheck, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess
            || hipHostGetDevicePointer((void**)&c->zeroCheckDev, c->zeroCheck, 0) != hipSuccess)
            return bail(fail(LWHIP_ERR_DEVICE, "hipHostMalloc failed"));
        *c->zeroCheck = 0;
    }
    const bool timing = std::getenv("LWHIP_CREATE_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    st = build_tables(c);
    if (st != LWHIP_OK)
        return bail(st);
    const auto t1 = now();
    st = alloc_state(c);
    if (st != LWHIP_OK)
        return bail(st);
    st = build_sweep_args(c);
    if (st != LWHIP_OK)
        return bail(st);
    st = build_tile_args(c);
    if (st != LWHIP_OK)
        return bail(st);
    const auto t2 = now();
    st = lwhip_upload(c, LWHIP_ALL_INPUTS);
    if (st != LWHIP_OK)
        return bail(st);
    if (timing)
        std::fprintf(stderr, "lwhip_create: tables %.2f ms, allocations %.2f ms, upload %.2f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, now()));
    if (std::getenv("LWHIP_VERBOSE"))
        std::fprintf(stderr,
                     "lwhip_create: Nla=%d Ns=%d Nmu=%d sweep=%s tiles=%d (L=%d, max slots %d, cap %d) chunks=%d x %d waves post=%d maxCT=%d/%d "
                     "rows=%.1f MB phiT=%.1f MB\n",
                     c->Nla, c->Ns, c->Nrays, c->is2d ? "2d" : c->laneSweep ? "depth-lanes" : "ray-column",
                     c->nTiles, c->tileL, c->maxSlotsTile, c->tileCap, c->nTileChunks, c->tileWaves, c->nPostChunks, c->maxCTTile, c->maxCTPost,
                     c->rowsTileTot * 8e-6, c->phiTTot * 8e-6);
    if (std::getenv("LWHIP_VERBOSE") && c->laneSweep)
        std::fprintf(stderr, "lwhip_create: lane sweep LDS per workgroup: accumulators %zu B + depth arena %zu B + per-wave blocks = %zu B\n",
                     sizeof(double) * (size_t)c->maxCTTile * 4 * (c->laneLR * c->laneD), (size_t)c->depArena.n * sizeof(double),
                     sizeof(double) * ((size_t)c->maxCTTile * 4 * (c->laneLR * c->laneD) + (size_t)c->tileWaves * 16 * 5 + 32
                                       + (size_t)c->tileWaves * 3 * 64 * c->laneD + (size_t)c->tileWaves * LWHIP_POST_MAXC * c->laneR * 3)
                         + (size_t)c->depArena.n * sizeof(double));
    *out = c;
    return LWHIP_OK;
}

int lwhip_destroy(lwhip_context* c)
{
    if (!c)
        return LWHIP_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->lsDbgHost && !c->laneSweep)
    {
        // -DRM_TIMING build of the march: per tile kind the count and the mean / max clocks of a direction's march and of the post-pass
        std::map<int, std::array<double, 6>> kinds; // n, sum march, max march, sum post, max post, -
        for (int t = 0; t < c->nTiles; ++t)
        {
            const double* o = c->lsDbgHost + (size_t)t * 8;
            auto& k = kinds[(int)o[0]];
            k[0] += 1.0;
            k[1] += 0.5 * (o[1] + o[2]);
            k[2] = std::max({ k[2], o[1], o[2] });
            k[3] += o[3];
            k[4] = std::max(k[4], o[3]);
        }
        for (const auto& kv : kinds)
            std::fprintf(stderr, "march clocks, tiles of kind (%d lines, %d mixed): %d tiles, march mean %.0f max %.0f, post-pass mean %.0f max %.0f\n",
                         kv.first / 10, kv.first % 10, (int)kv.second[0], kv.second[1] / kv.second[0], kv.second[2], kv.second[3] / kv.second[0], kv.second[4]);
        (void)hipHostFree(c->lsDbgHost);
        c->lsDbgHost = nullptr;
    }
    if (c->lsDbgHost)
    {
        double sum[8] = { 0 }, mx = 0.0;
        for (int t = 0; t < c->nTiles; ++t)
        {
            for (int q = 0; q < 8; ++q)
                sum[q] += c->lsDbgHost[(size_t)t * 8 + q];
            mx = std::max(mx, c->lsDbgHost[(size_t)t * 8 + 7]);
        }
        {
            std::map<int, std::array<double, 3>> kinds;
            for (int t = 0; t < c->nTiles; ++t)
            {
                const double tot = c->lsDbgHost[(size_t)t * 8 + 7];
                auto& k = kinds[(int)std::lround((tot - std::floor(tot)) * 100.0)];
                k[0] += 1.0;
                k[1] += std::floor(tot);
                k[2] = std::max(k[2], std::floor(tot));
            }
            for (const auto& kv : kinds)
                std::fprintf(stderr, "lane sweep clocks, tiles of kind (%d lines, %d mixed): %d tiles, task mean %.0f max %.0f\n", kv.first / 10,
                             kv.first % 10, (int)kv.second[0], kv.second[1] / kv.second[0], kv.second[2]);
        }
        {
            double w[3] = { 0, 0, 0 };
            int nw = 0;
            for (int b = 0; b < 4 * c->nTiles; ++b)
            {
                const double* o = c->lsDbgHost + (size_t)c->nTiles * 8 + (size_t)b * 4;
                if (o[1] > 0.0)
                {
                    ++nw;
                    for (int q = 0; q < 3; ++q)
                        w[q] += o[q];
                }
            }
            if (nw)
                std::fprintf(stderr, "lane sweep workgroup clocks (mean over %d workgroups): prologue %.0f tasks %.0f epilogue %.0f\n", nw, w[0] / nw,
                             w[1] / nw, w[2] / nw);
        }
        std::fprintf(stderr, "lane sweep phase clocks per tile (mean): setup %.0f gather %.0f stencils %.0f coefficients %.0f scan %.0f rates %.0f finish %.0f total %.0f (max %.0f) over %d tiles\n",
                     sum[0] / c->nTiles, sum[1] / c->nTiles, sum[2] / c->nTiles, sum[3] / c->nTiles, sum[4] / c->nTiles, sum[5] / c->nTiles,
                     sum[6] / c->nTiles, sum[7] / c->nTiles, mx, c->nTiles);
        (void)hipHostFree(c->lsDbgHost);
    }
    delete c;
    return LWHIP_OK;
}

int lwhip_upload(lwhip_context* c, uint32_t mask)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    const lwhip_problem& p = c->prob;
    const size_t Ns = c->Ns, Nla = c->Nla, Nr = c->Nrays;
    const size_t l0 = c->laStart;
    std::vector<double> rmuzHost; // (lives until the synchronisation at the end)
    if (mask & LWHIP_GAMMA)
        c->prefillPending = false; // the host pre-fill being uploaded supersedes it
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
            H2D(c->vlosMu.p, p.vlosMu, Nr * Ns);
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
            HIP_TRY(hipMemcpyAsync(c->geo.p, geo.data(), geo.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
            std::vector<double> geoT;
            if (c->laneSweep)
            {
                // the same per quantity, entry k + 1 (zero padding at both ends): what a lane reads for its block of depths
                const size_t st = (size_t)c->laneLR * c->laneD + 2;
                geoT.assign(4 * st, 0.0);
                for (size_t k = 0; k < Ns; ++k)
                    for (int q = 0; q < 4; ++q)
                        geoT[q * st + k + 1] = geo[4 * k + q];
                HIP_TRY(hipMemcpyAsync(c->geoT.p, geoT.data(), geoT.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
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
                HIP_TRY(hipMemcpyAsync(c->bcPlanck.p, bcB.data(), bcB.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
            }
            HIP_TRY(hipStreamSynchronize(c->stream)); // `geo`, `bcB` are locals
        }
    }
    if (mask & LWHIP_BACKGROUND)
    {
        H2D(c->bgChi.p, p.bgChi + l0 * Ns, Nla * Ns);
        H2D(c->bgEta.p, p.bgEta + l0 * Ns, Nla * Ns);
        H2D(c->bgSca.p, p.bgSca + l0 * Ns, Nla * Ns);
    }
    if (mask & LWHIP_J)
        H2D(c->J.p, p.J + l0 * Ns, Nla * Ns);
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
    // staging layout (doubles): [n pool][Gamma pool][C pool][Rij][Rji]
    const size_t nPool = (size_t)c->NlevTot * Ns, gPool = (size_t)c->gammaTot, rPool = (size_t)c->Ntrans * Ns;
    if (!c->stage)
    {
        c->stageDoubles = nPool + 2 * gPool + 2 * rPool + 8;
        HIP_TRY(hipHostMalloc((void**)&c->stage, c->stageDoubles * sizeof(double), hipHostMallocDefault));
    }
    double* stN = c->stage;
    double* stG = stN + nPool;
    double* stC = stG + gPool;
    double* stRij = stC + gPool;
    double* stRji = stRij + rPool;
    bool anyR = false;
    for (int ia = 0; ia < c->Natom; ++ia)
    {
        const lwhip_atom& a = c->atoms[ia];
        if (mask & LWHIP_POPS)
            std::memcpy(stN + (size_t)c->levelOff[ia] * Ns, a.n, (size_t)a.Nlevel * Ns * sizeof(double));
        if (mask & LWHIP_NSTAR)
        {
            H2D(c->nTotal.p + (size_t)ia * Ns, a.nTotal, Ns);
            H2D(c->vBroad.p + (size_t)ia * Ns, a.vBroad, Ns);
        }
        if (!a.detailed && (mask & LWHIP_GAMMA))
            std::memcpy(stG + c->gammaOff[ia], a.Gamma, (size_t)a.Nlevel * a.Nlevel * Ns * sizeof(double));
        if (!a.detailed && (mask & LWHIP_COLLISIONS) && a.C)
        {
            // atoms without C keep whatever the pool holds: copy atom by atom, not the whole pool
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
                    H2D(c->wphi.p + (size_t)h.row * Ns, t.wphi, Ns);
                }
                if ((mask & LWHIP_NSTAR) && t.aDamp)
                    H2D(c->aDamp.p + (size_t)h.row * Ns, t.aDamp, Ns);
                if ((mask & LWHIP_NSTAR) && t.Qelast)
                    H2D(c->Qelast.p + (size_t)h.row * Ns, t.Qelast, Ns);
                if ((mask & LWHIP_RHOPRD) && t.prd && nlt)
                    H2D(c->rho.p + h.rhoOff, t.rhoPrd + (size_t)h.ltStart * Ns, nlt * Ns);
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
        const int str = retile_profiles(c);
        if (str != LWHIP_OK)
            return str;
        c->deviceProfiles = c->profilesStale = false; // the host's profiles are authoritative again
    }
    else if ((mask & (LWHIP_ATMOS | LWHIP_NSTAR)) && c->deviceProfiles)
        c->profilesStale = true; // phi / wphi follow the new velocities and widths before the next sweep
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LWHIP_OK;
}

int lwhip_download(lwhip_context* c, uint32_t mask)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
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
        D2H(p.J + l0 * Ns, c->J.p, Nla * Ns);
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
        c->stageDoubles = nPool + 2 * gPool + 2 * rPool + 8;
        HIP_TRY(hipHostMalloc((void**)&c->stage, c->stageDoubles * sizeof(double), hipHostMallocDefault));
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
            if ((mask & LWHIP_RHOPRD) && h.t.type == LWHIP_LINE && h.t.prd && h.rhoOff >= 0 && h.NredLoc > h.NblueLoc)
                D2H(h.t.rhoPrd + (size_t)h.ltStart * Ns, c->rho.p + h.rhoOff, (size_t)(h.NredLoc - h.NblueLoc) * Ns);
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

int lwhip_gamma_prefill_from_C(lwhip_context* c, double crsw)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    c->prefillPending = true; // performed by the next apply_kernel (or flushed by download / stat_equil)
    c->prefillCrsw = crsw;
    return LWHIP_OK;
}

// The 2D iteration of this context's wavelengths: batches of wavelengths through cont -> gather -> 2D formal
// solver -> rates (lwhip_2d.hip), the running sums in the reduce buffer, dJ in its tail.
// mode 0: the iteration (J, Gamma, rates); 1: formal_sol, all rays; 2: formal_sol, up-going rays only;
// 3: formal_sol_prd_update_rates (PrdTemplates.hpp:18-75) -- only the wavelengths that hold a PRD line, no depth data
static int run_2d(lwhip_context* c, int lambdaIterate, int mode = 0)
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
            HIP_TRY(launch_rates2d(a, c->maxL, c->stream));
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

// argument blocks of the reduce / apply kernels of one context (also the per-column blocks of a batch)
static ReduceArgs make_reduce_args(lwhip_context* c)
{
    ReduceArgs r{};
    r.Ns = c->Ns;
    r.Ntrans = c->Ntrans;
    r.red8 = c->red8.p;
    r.red = c->red.p;
    r.dJ = c->dJ.p;
    r.Nla = c->Nla;
    r.idxOffset = c->laStart;
    r.world = c->worldSize;
    r.rank = c->worldRank;
    r.batchTail = nullptr;
    return r;
}

static ApplyArgs make_apply_args(lwhip_context* c)
{
    ApplyArgs a{};
    a.Ns = c->Ns;
    a.Natom = c->Natom;
    a.atomTrOff = c->atomTrOffD.p;
    a.atomNlevel = c->atomNlevel.p;
    a.atomDetailed = c->atomDetailed.p;
    a.atomGammaOff = c->atomGammaOff.p;
    a.trans = c->dtrans.p;
    a.transLi = c->transLi.p;
    a.transLj = c->transLj.p;
    a.red = c->red.p;
    a.Gamma = c->Gamma.p;
    a.Rij = c->Rij.p;
    a.Rji = c->Rji.p;
    a.Cmat = c->Cmat.p;
    a.crsw = c->prefillCrsw;
    a.prefill = c->prefillPending ? 1 : 0;
    return a;
}

static int ensure_profiles(lwhip_context* c);

static int fs_partial_impl(lwhip_context* c, int lambdaIterate, bool partsOnly)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    c->lastSweepUpOnly = false;
    {
        const int stp = ensure_profiles(c);
        if (stp != LWHIP_OK)
            return stp;
    }
    if (c->is2d)
    {
        const int st2 = run_2d(c, lambdaIterate);
        if (st2 != LWHIP_OK)
            return st2;
        if (c->worldSize > 1 && !c->prdLines.empty())
            HIP_TRY(hipMemsetAsync(c->red.p + (size_t)c->Ntrans * 4 * c->Ns + 2 * (size_t)c->worldSize, 0,
                                   2 * (size_t)c->worldSize * c->prdLines.size() * sizeof(double), c->stream));
        c->partialPending = true;
        return LWHIP_OK;
    }
    const TileDyn dyn = make_dyn(c, false, lambdaIterate);
    // the sweep's workgroups add their sums straight into the stage-1 buffer (atomics): the one-call iteration lets the
    // apply launch sum and zero it (no reduce launch), the split / sharded form runs stage 2
    if (!c->red8Clean)
        HIP_TRY(hipMemsetAsync(c->red8.p, 0, c->red8.n * sizeof(double), c->stream));
    c->red8Clean = true;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timeIt = c->profiling && (c->profCount++ % c->profEvery == 0);
    if (timeIt)
    {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
    }
    HIP_TRY(run_sweep(c, dyn, true, e0, e1));
    if (timeIt)
        c->pending.emplace_back(e0, e1);
    if (!partsOnly)
    {
        // fixed-order sum of the parts; the extra block writes this shard's (dJMax, global idx) into its slot of the
        // gather tail and zeros the other slots
        ReduceArgs r = make_reduce_args(c);
        r.zeroParts = 1;
        HIP_TRY(launch_reduce_sum(r, c->stream));
    }
    c->partsOnly = partsOnly;
    c->atomicParts = partsOnly;
    if (c->worldSize > 1 && !c->prdLines.empty()) // the PRD slots of the tail ride in the same all-reduce: keep them finite
        HIP_TRY(hipMemsetAsync(c->red.p + (size_t)c->Ntrans * 4 * c->Ns + 2 * (size_t)c->worldSize, 0,
                               2 * (size_t)c->worldSize * c->prdLines.size() * sizeof(double), c->stream));
    c->partialPending = true;
    return LWHIP_OK;
}

int lwhip_fs_partial(lwhip_context* c, int lambdaIterate)
{
    return fs_partial_impl(c, lambdaIterate, false);
}

int lwhip_fs_finalise(lwhip_context* c, lwhip_iter_result* res)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    if (!c->partialPending)
        return fail(LWHIP_ERR_INVALID, "lwhip_fs_finalise without lwhip_fs_partial");
    HIP_TRY(hipSetDevice(c->device));
    bool mapped = false; // (dJMax, idx) arrives through host-mapped memory: no read-back copy
    if (c->Natom > 0)
    {
        ApplyArgs a = make_apply_args(c);
        if (c->partsOnly)
        {
            a.fromParts = 1;
            a.zeroParts = c->atomicParts ? 1 : 0;
            a.red8 = c->red8.p;
            a.dJ = c->dJ.p;
            a.NlaDj = c->Nla;
            a.idxOffset = c->laStart;
            a.scalarIdx = c->djIdxMode;
            a.tail = c->red.p + (size_t)c->Ntrans * 4 * c->Ns;
            if (!c->tailMapped)
            {
                HIP_TRY(hipHostMalloc((void**)&c->tailMapped, 4 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
                HIP_TRY(hipHostGetDevicePointer((void**)&c->tailMappedDev, c->tailMapped, 0));
                c->tailMapped[2] = 0.0;
            }
            a.tailHost = c->tailMappedDev;
            c->tailTicket += 1.0;
            a.tailTicket = c->tailTicket;
            mapped = res != nullptr;
        }
        else if (res)
        {
            // split / sharded iteration: the (all-reduced) tail holds every shard's (dJMax, idx); the apply launch takes
            // their maximum and publishes it the same way -- no read-back copy, no wait for the end of the stream
            a.tail = c->red.p + (size_t)c->Ntrans * 4 * c->Ns;
            a.gatherWorld = c->worldSize;
            if (!c->tailMapped)
            {
                HIP_TRY(hipHostMalloc((void**)&c->tailMapped, 4 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
                HIP_TRY(hipHostGetDevicePointer((void**)&c->tailMappedDev, c->tailMapped, 0));
                c->tailMapped[2] = 0.0;
            }
            a.tailHost = c->tailMappedDev;
            c->tailTicket += 1.0;
            a.tailTicket = c->tailTicket;
            mapped = true;
        }
        HIP_TRY(launch_apply(a, c->stream));
        c->prefillPending = false;
        c->partsOnly = false;
    }
    c->partialPending = false;
    if (res && mapped)
    {
        // The dJMax block of the apply launch publishes (dJMax, idx) and then this launch's ticket in host-mapped memory:
        // poll for the ticket rather than for the end of the stream, so the caller can queue the next iteration while
        // the rest of the launch drains (everything downstream is stream-ordered).  The stream is queried now and then
        // so that a failed launch cannot hang the host.
        volatile double* tm = c->tailMapped;
        bool seen = false;
        for (long spin = 0; !(seen = (tm[2] == c->tailTicket)); ++spin)
        {
            __builtin_ia32_pause();
            if ((spin & 0x3fff) == 0x3fff && hipStreamQuery(c->stream) != hipErrorNotReady)
                break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!seen)
            HIP_TRY(hipStreamSynchronize(c->stream));
        res->updatedJ = 1;
        res->dJMax = c->tailMapped[0];
        res->dJMaxIdx = (int32_t)c->tailMapped[1];
        return LWHIP_OK;
    }
    if (res)
    {
        const double* tail = c->red.p + (size_t)c->Ntrans * 4 * c->Ns;
        if (!c->gatherPinned)
            HIP_TRY(hipHostMalloc((void**)&c->gatherPinned, c->gatherHost.size() * sizeof(double), hipHostMallocDefault));
        HIP_TRY(hipMemcpyAsync(c->gatherPinned, tail, c->gatherHost.size() * sizeof(double),
                               hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::memcpy(c->gatherHost.data(), c->gatherPinned, c->gatherHost.size() * sizeof(double));
        // max over shards, first (lowest) wavelength index on ties: max_idx, Constants.hpp:114-125
        double best = -1.0;
        int bestIdx = 0;
        for (int r = 0; r < c->worldSize; ++r)
        {
            const double v = c->gatherHost[2 * r];
            const int idx = (int)c->gatherHost[2 * r + 1];
            if (v > best || (v == best && idx < bestIdx))
            {
                best = v;
                bestIdx = idx;
            }
        }
        res->updatedJ = 1;
        res->dJMax = best < 0.0 ? 0.0 : best;
        res->dJMaxIdx = bestIdx;
    }
    return LWHIP_OK;
}

int lwhip_formal_sol_gamma_matrices(lwhip_context* c, int lambdaIterate, lwhip_iter_result* res)
{
    // the one-call form knows that nobody reads the reduce buffer between its two halves
    const bool partsOnly = c && !c->is2d && c->worldSize == 1 && c->Natom > 0 && c->Ntrans > 0;
    int st = fs_partial_impl(c, lambdaIterate, partsOnly);
    if (st != LWHIP_OK)
        return st;
    return lwhip_fs_finalise(c, res);
}

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

static int voigt_line_list(lwhip_context* c, std::vector<VoigtLineArgs>& out);

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
            if (!c->statusHost)
            {
                HIP_TRY(hipHostMalloc((void**)&c->statusHost, sizeof(int32_t), hipHostMallocMapped));
                HIP_TRY(hipHostGetDevicePointer((void**)&c->statusDev, c->statusHost, 0));
                *c->statusHost = 0;
            }
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
    const TileDyn dyn = make_dyn(c0, false, lambdaIterate);
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

int lwhip_reduce_buffer(lwhip_context* c, void** devPtr, size_t* nSum, size_t* nGather)
{
    if (!c || !devPtr || !nSum || !nGather)
        return fail(LWHIP_ERR_INVALID, "null argument");
    *devPtr = c->red.p;
    *nSum = (size_t)c->Ntrans * 4 * c->Ns;
    *nGather = 2 * (size_t)c->worldSize * (1 + c->prdLines.size());
    return LWHIP_OK;
}

int lwhip_formal_sol(lwhip_context* c, int upOnly)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    {
        const int stp = ensure_profiles(c);
        if (stp != LWHIP_OK)
            return stp;
    }
    c->lastSweepUpOnly = upOnly != 0;
    if (c->is2d)
        return run_2d(c, 0, upOnly ? 2 : 1);
    const TileDyn dyn = make_dyn(c, upOnly != 0, 0);
    HIP_TRY(run_sweep(c, dyn, false));
    return LWHIP_OK;
}

static int stat_equil_impl(lwhip_context* c, int atom, bool wait, double* dPops = nullptr, int32_t* dPopsMaxIdx = nullptr);

int lwhip_stat_equil(lwhip_context* c, int atom)
{
    return stat_equil_impl(c, atom, true);
}

int lwhip_stat_equil_async(lwhip_context* c, int atom)
{
    return stat_equil_impl(c, atom, false);
}

int lwhip_stat_equil_report(lwhip_context* c, int atom, double* dPops, int32_t* dPopsMaxIdx)
{
    return stat_equil_impl(c, atom, true, dPops, dPopsMaxIdx);
}

int lwhip_check_status(lwhip_context* c)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->statusHost && *c->statusHost == LWHIP_ERR_SINGULAR)
    {
        *c->statusHost = 0;
        return fail(LWHIP_ERR_SINGULAR, "Singular Matrix");
    }
    return LWHIP_OK;
}

static int stat_equil_impl(lwhip_context* c, int atom, bool wait, double* dPops, int32_t* dPopsMaxIdx)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    {
        int stp = flush_prefill(c);
        if (stp != LWHIP_OK)
            return stp;
    }
    // one launch for all requested atoms; the status word lives in host-mapped pinned memory, so the
    // call needs no memset kernel and no read-back copy, only the stream synchronisation
    if (!c->statusHost)
    {
        HIP_TRY(hipHostMalloc((void**)&c->statusHost, sizeof(int32_t), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&c->statusDev, c->statusHost, 0));
    }
    std::vector<NrAtom> atoms;
    int maxNl = 0;
    for (int ia = 0; ia < c->Natom; ++ia)
    {
        if (atom >= 0 && ia != atom)
            continue;
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
    if (atoms.empty())
        return LWHIP_OK;
    if (c->statEqKey != atom || c->statEqAtoms.n < atoms.size())
    {
        HIP_TRY(c->statEqAtoms.upload(atoms));
        c->statEqKey = atom;
    }
    if (wait)
        *c->statusHost = 0; // (the async form accumulates until lwhip_check_status)
    StatEqArgs sa{};
    sa.Ns = c->Ns;
    sa.k0 = c->kLo;
    sa.k1 = c->kHi < 0 ? c->Ns : c->kHi;
    sa.Natoms = (int32_t)atoms.size();
    sa.atoms = c->statEqAtoms.p;
    sa.n = c->n.p;
    sa.nTotal = c->nTotal.p;
    sa.Gamma = c->Gamma.p;
    sa.status = c->statusDev;
    const int nBlocks = stat_eq_blocks(c->Ns, maxNl);
    const bool report = wait && dPops;
    if (report)
    {
        const size_t need = atoms.size() * nBlocks * 2;
        if (c->changeCount < need)
        {
            if (c->changeHost)
                (void)hipHostFree(c->changeHost);
            c->changeHost = nullptr;
            HIP_TRY(hipHostMalloc((void**)&c->changeHost, need * sizeof(double), hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void**)&c->changeDev, c->changeHost, 0));
            c->changeCount = need;
        }
        sa.change = c->changeDev;
    }
    HIP_TRY(launch_stat_eq(sa, maxNl, c->stream));
    if (!wait)
        return LWHIP_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (*c->statusHost == LWHIP_ERR_SINGULAR)
        return fail(LWHIP_ERR_SINGULAR, "Singular Matrix");
    if (report)
    {
        // Ng::max_change (Source/Ng.hpp:138-156): the maximum and the flattened [level, depth] index of its
        // first occurrence, per solved atom; entries are indexed by position among the ACTIVE atoms
        int activePos = 0;
        size_t slot = 0;
        for (int ia = 0; ia < c->Natom; ++ia)
        {
            if (c->atoms[ia].detailed)
                continue;
            if (atom < 0 || ia == atom)
            {
                double v = 0.0;
                int idx = 0;
                for (int b = 0; b < nBlocks; ++b)
                {
                    const double x = c->changeHost[(slot * nBlocks + b) * 2];
                    const int i = (int)c->changeHost[(slot * nBlocks + b) * 2 + 1];
                    if (x > v || (x == v && x > 0.0 && i < idx))
                    {
                        v = x;
                        idx = i;
                    }
                }
                dPops[activePos] = v;
                if (dPopsMaxIdx)
                    dPopsMaxIdx[activePos] = idx;
                ++slot;
            }
            ++activePos;
        }
    }
    return LWHIP_OK;
}

int lwhip_ng_configure(lwhip_context* c, int Norder, int Nperiod, int Ndelay)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    if (Norder < 0 || Norder > LWHIP_NG_MAX_ORDER || (Norder > 0 && Nperiod < 1))
        return fail(LWHIP_ERR_INVALID, "Ng: 0 <= Norder <= 6 and Nperiod >= 1");
    HIP_TRY(hipSetDevice(c->device));
    c->ngOrder = Norder;
    c->ngPeriod = Nperiod;
    c->ngDelay = std::max(Ndelay, Nperiod + 2); // Ng.hpp:32
    c->ngCount = 1;                             // the constructor stores the current solution :34-38
    std::vector<NgAtom> atoms;
    int64_t hist = 0;
    for (int ia = 0; ia < c->Natom; ++ia)
    {
        if (c->atoms[ia].detailed)
            continue;
        NgAtom at{};
        at.len = c->atoms[ia].Nlevel * c->Ns;
        at.nOff = (int64_t)c->levelOff[ia] * c->Ns;
        at.histOff = hist;
        hist += (int64_t)(Norder + 2) * at.len;
        atoms.push_back(at);
    }
    c->ngAtomsHost = atoms;
    if (atoms.empty())
        return LWHIP_OK;
    HIP_TRY(c->ngAtoms.upload(atoms));
    HIP_TRY(c->ngHistory.alloc((size_t)hist));
    HIP_TRY(hipMemsetAsync(c->ngHistory.p, 0, (size_t)hist * sizeof(double), c->stream));
    for (const NgAtom& at : atoms)
        HIP_TRY(hipMemcpyAsync(c->ngHistory.p + at.histOff, c->n.p + at.nOff, (size_t)at.len * sizeof(double),
                               hipMemcpyDeviceToDevice, c->stream));
    c->ngConfigured = true;
    return LWHIP_OK;
}

int lwhip_ng_accelerate(lwhip_context* c, int32_t* accelerated, double* dPops, int32_t* dPopsMaxIdx)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    if (!c->ngConfigured)
        return fail(LWHIP_ERR_INVALID, "lwhip_ng_accelerate before lwhip_ng_configure");
    const int nAt = (int)c->ngAtomsHost.size();
    if (nAt == 0)
        return LWHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->statusHost)
    {
        HIP_TRY(hipHostMalloc((void**)&c->statusHost, sizeof(int32_t), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&c->statusDev, c->statusHost, 0));
    }
    if (c->changeCount < (size_t)2 * nAt)
    {
        if (c->changeHost)
            (void)hipHostFree(c->changeHost);
        c->changeHost = nullptr;
        HIP_TRY(hipHostMalloc((void**)&c->changeHost, (size_t)2 * nAt * sizeof(double), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&c->changeDev, c->changeHost, 0));
        c->changeCount = (size_t)2 * nAt;
    }
    const int No = c->ngOrder, nslots = No + 2;
    NgArgs a{};
    a.atoms = c->ngAtoms.p;
    a.n = c->n.p;
    a.history = c->ngHistory.p;
    a.change = c->changeDev;
    a.status = c->statusDev;
    a.Norder = No;
    a.storeSlot = c->ngCount % nslots; // Ng.hpp:62
    c->ngCount += 1;
    a.doAccel = (No > 0 && c->ngCount >= c->ngDelay && ((c->ngCount - c->ngDelay) % c->ngPeriod) == 0) ? 1 : 0;
    for (int i = 0; i < No + 2; ++i)
        a.slots[i] = ((c->ngCount - 1 - i) % nslots + nslots) % nslots;
    a.haveTwo = c->ngCount >= 2 ? 1 : 0;
    a.oldSlot = ((c->ngCount - 2) % nslots + nslots) % nslots;
    *c->statusHost = 0;
    HIP_TRY(launch_ng(a, nAt, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (*c->statusHost == LWHIP_ERR_SINGULAR)
        return fail(LWHIP_ERR_SINGULAR, "Singular Matrix");
    for (int q = 0; q < nAt; ++q)
    {
        if (accelerated)
            accelerated[q] = a.doAccel;
        if (dPops)
            dPops[q] = c->changeHost[2 * q];
        if (dPopsMaxIdx)
            dPopsMaxIdx[q] = (int32_t)c->changeHost[2 * q + 1];
    }
    return LWHIP_OK;
}

int lwhip_set_djmax_index_mode(lwhip_context* c, int mode)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    if (mode != 0 && mode != 1)
        return fail(LWHIP_ERR_INVALID, "lwhip_set_djmax_index_mode: mode is 0 or 1");
    // (served where the iteration's own apply launch reduces dJ: unsharded 1D contexts; elsewhere the first arg-max stays)
    c->djIdxMode = (mode == 1 && !c->is2d && c->worldSize == 1) ? 1 : 0;
    return LWHIP_OK;
}

int lwhip_set_depth_range(lwhip_context* c, int spaceStart, int spaceEnd)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    if (spaceStart < 0 && spaceEnd < 0) // the reference's "whole atmosphere" (UpdatePopulations.cpp:12-16)
    {
        c->kLo = 0;
        c->kHi = -1;
        return LWHIP_OK;
    }
    if (spaceStart < 0 || spaceEnd > c->Ns || spaceStart > spaceEnd)
        return fail(LWHIP_ERR_INVALID, "set_depth_range: need 0 <= spaceStart <= spaceEnd <= Nspace (or -1, -1)");
    c->kLo = spaceStart;
    c->kHi = spaceEnd;
    return LWHIP_OK;
}

int lwhip_time_dep_update(lwhip_context* c, int atom, const double* nOld, double dt)
{
    if (!c || !nOld)
        return fail(LWHIP_ERR_INVALID, "null argument");
    if (atom < 0 || atom >= c->Natom || c->atoms[atom].detailed)
        return fail(LWHIP_ERR_INVALID, "time_dep_update: not an active atom");
    HIP_TRY(hipSetDevice(c->device));
    {
        int stp = flush_prefill(c);
        if (stp != LWHIP_OK)
            return stp;
    }
    const lwhip_atom& a = c->atoms[atom];
    const size_t cnt = (size_t)a.Nlevel * c->Ns;
    if (c->popScratch.n < cnt)
        HIP_TRY(c->popScratch.alloc(cnt));
    HIP_TRY(hipMemcpyAsync(c->popScratch.p, nOld, cnt * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->status.p, 0, sizeof(int32_t), c->stream));
    HIP_TRY(launch_time_dep(a.Nlevel, c->Ns, c->kLo, c->kHi < 0 ? c->Ns : c->kHi, c->n.p + (size_t)c->levelOff[atom] * c->Ns, c->popScratch.p,
                            c->Gamma.p + c->gammaOff[atom], dt, c->status.p, c->stream));
    int32_t st = 0;
    HIP_TRY(hipMemcpyAsync(&st, c->status.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (st == LWHIP_ERR_SINGULAR)
        return fail(LWHIP_ERR_SINGULAR, "Singular Matrix");
    return LWHIP_OK;
}

int lwhip_nr_post_update(lwhip_context* c, const lwhip_nr_args* args)
{
    if (!c || !args || !args->atoms || !args->stages || !args->backgroundNe || !args->ne || args->Natoms <= 0)
        return fail(LWHIP_ERR_INVALID, "nr_post_update: null argument");
    HIP_TRY(hipSetDevice(c->device));
    {
        int stp = flush_prefill(c);
        if (stp != LWHIP_OK)
            return stp;
    }
    const size_t Ns = c->Ns;
    std::vector<NrAtom> atoms(args->Natoms);
    int eq = 0;
    int64_t dcRows = 0;
    for (int q = 0; q < args->Natoms; ++q)
    {
        const int ia = args->atoms[q];
        if (ia < 0 || ia >= c->Natom || c->atoms[ia].detailed)
            return fail(LWHIP_ERR_INVALID, "nr_post_update: not an active atom");
        if (!c->atoms[ia].C)
            return fail(LWHIP_ERR_INVALID, "nr_post_update needs the collisional rates C of every listed atom");
        NrAtom& at = atoms[q];
        at.atom = ia;
        at.Nlevel = c->atoms[ia].Nlevel;
        at.levelOff = c->levelOff[ia];
        at.eqOff = eq;
        at.trBegin = c->atomTrOff[ia];
        at.trEnd = c->atomTrOff[ia + 1];
        at.gammaOff = c->gammaOff[ia];
        at.dcOff = dcRows;
        eq += at.Nlevel;
        dcRows += (int64_t)at.Nlevel * at.Nlevel;
    }
    const int Neqn = eq + 1;
    if (Neqn > 64)
        return fail(LWHIP_ERR_UNSUPPORTED, "nr_post_update: more than 63 coupled levels");
    // scratch: [stages eq] [ne Ns] [bgNe Ns] [nPrev eq*Ns] [dC dcRows*Ns]
    const bool timeDep = args->nPrev != nullptr;
    const bool fdC = args->dC != nullptr;
    const size_t need = (size_t)eq + 2 * Ns + (timeDep ? (size_t)eq * Ns : 0) + (fdC ? (size_t)dcRows * Ns : 0);
    if (c->popScratch.n < need)
        HIP_TRY(c->popScratch.alloc(need));
    if (c->nrAtoms.n < atoms.size())
        HIP_TRY(c->nrAtoms.alloc(atoms.size()));
    HIP_TRY(hipMemcpyAsync(c->nrAtoms.p, atoms.data(), atoms.size() * sizeof(NrAtom), hipMemcpyHostToDevice, c->stream));
    double* dStages = c->popScratch.p;
    double* dNe = dStages + eq;
    double* dBg = dNe + Ns;
    double* dPrev = dBg + Ns;
    double* dDC = dPrev + (timeDep ? (size_t)eq * Ns : 0);
    for (int q = 0; q < args->Natoms; ++q)
    {
        const NrAtom& at = atoms[q];
        if (!args->stages[q])
            return fail(LWHIP_ERR_INVALID, "nr_post_update: null stages");
        H2D(dStages + at.eqOff, args->stages[q], at.Nlevel);
        if (timeDep)
        {
            if (!args->nPrev[q])
                return fail(LWHIP_ERR_INVALID, "nr_post_update: null nPrev");
            H2D(dPrev + (size_t)at.eqOff * Ns, args->nPrev[q], (size_t)at.Nlevel * Ns);
        }
        if (fdC)
        {
            if (!args->dC[q])
                return fail(LWHIP_ERR_INVALID, "nr_post_update: null dC");
            H2D(dDC + (size_t)at.dcOff * Ns, args->dC[q], (size_t)at.Nlevel * at.Nlevel * Ns);
        }
    }
    H2D(dNe, args->ne, Ns);
    H2D(dBg, args->backgroundNe, Ns);
    if (c->transType.n < (size_t)std::max(c->Ntrans, 1))
    {
        std::vector<int32_t> tt((size_t)std::max(c->Ntrans, 1), 0);
        for (int tr = 0; tr < c->Ntrans; ++tr)
            tt[tr] = c->trans[tr].t.type;
        HIP_TRY(c->transType.upload(tt));
    }
    HIP_TRY(hipMemsetAsync(c->status.p, 0, sizeof(int32_t), c->stream));
    NrArgs a{};
    a.Ns = (int32_t)Ns;
    a.Natoms = args->Natoms;
    a.Neqn = Neqn;
    a.timeDep = timeDep ? 1 : 0;
    a.atoms = c->nrAtoms.p;
    a.Gamma = c->Gamma.p;
    a.Cmat = c->Cmat.p;
    a.n = c->n.p;
    a.nTotal = c->nTotal.p;
    a.stages = dStages;
    a.nPrev = timeDep ? dPrev : nullptr;
    a.dC = fdC ? dDC : nullptr;
    a.backgroundNe = dBg;
    a.ne = dNe;
    a.transType = c->transType.p;
    a.transLi = c->transLi.p;
    a.transLj = c->transLj.p;
    a.dt = args->dt;
    a.crsw = args->crsw;
    a.status = c->status.p;
    a.k0 = c->kLo;
    a.k1 = c->kHi < 0 ? (int32_t)Ns : c->kHi;
    HIP_TRY(launch_nr_post(a, c->stream));
    int32_t st = 0;
    HIP_TRY(hipMemcpyAsync(&st, c->status.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    D2H(args->ne, dNe, Ns);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (st == LWHIP_ERR_SINGULAR)
        return fail(LWHIP_ERR_SINGULAR, "Singular Matrix");
    return LWHIP_OK;
}

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
        const size_t nlt = h.NredLoc - h.NblueLoc;
        // the local mean intensity of the scattering integral: J of the line's wavelengths, or -- hybrid PRD -- the rows
        // of the rest-frame JRest they map to (consecutive: every wavelength of a PRD line is in prdIdxs; Prd.cpp:384-389)
        const double* src = c->hprd ? c->JRest.p + (size_t)c->hLa2prdHost[h.t.Nblue + h.ltStart] * Ns : c->J.p + (size_t)h.NblueLoc * Ns;
        if (nlt)
            HIP_TRY(hipMemcpyAsync(c->prdJ.p + ((size_t)c->prdRowOff[q] + h.ltStart) * Ns, src, nlt * Ns * sizeof(double),
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
        a.laLo = h.ltStart;
        a.laHi = h.ltStart + (h.NredLoc - h.NblueLoc);
        a.rho = c->rho.p + h.rhoOff;
        // enough workgroups to fill the chip: Ns x slices >= ~8 per CU, at least 128 wavelengths each
        a.slices = std::max(1, std::min({ (int)PRD_MAX_SLICES, (2048 + Ns - 1) / Ns, (a.laHi - a.laLo + 127) / 128 }));
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
            const size_t need = per + (per + 3) / 4;
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
                a.gTT = nullptr;
                a.gIT = (int32_t*)(a.gII + per);
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
    HIP_TRY(launch_prd_scatter(c->prdArgsDev.p, lineArgs.data(), Nprd, c->stream));
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
    HIP_TRY(hipMemsetAsync(c->dJ.p, 0, (size_t)c->Nla * sizeof(double), c->stream)); // unvisited wavelengths: dJ = 0
    // the sweep adds its workgroup sums into the stage-1 buffer; stage 2 sums and zeroes the parts
    if (!c->red8Clean)
        HIP_TRY(hipMemsetAsync(c->red8.p, 0, c->red8.n * sizeof(double), c->stream));
    HIP_TRY(run_sweep(c, dyn, true));
    ReduceArgs r = make_reduce_args(c);
    r.zeroParts = 1;
    HIP_TRY(launch_reduce_sum(r, c->stream));
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
    HIP_TRY(launch_apply(ap, c->stream));
    c->prdPending = false;
    const size_t nTail = 2 * (size_t)c->worldSize * (1 + Nprd);
    if (!c->prdPinned)
        HIP_TRY(hipHostMalloc((void**)&c->prdPinned, nTail * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(c->prdPinned, c->red.p + (size_t)c->Ntrans * 4 * Ns, nTail * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
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
    best_of(c->prdPinned, 2, v, idx);
    if (dJMax)
        *dJMax = v;
    if (dJMaxIdx)
        *dJMaxIdx = idx;
    const double* prdTail = c->prdPinned + 2 * (size_t)c->worldSize;
    for (int q = 0; q < Nprd; ++q)
    {
        best_of(prdTail + 2 * q, 2 * (size_t)Nprd, v, idx);
        const int Nl = c->trans[c->prdLines[q]].t.Nred - c->trans[c->prdLines[q]].t.Nblue;
        if (dRho)
            dRho[q] = v;
        if (dRhoMaxIdx) // the flattened [la, k] index modulo Nlambda, as PrdTemplates.hpp:265 reports it
            dRhoMaxIdx[q] = idx % Nl;
    }
    return LWHIP_OK;
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
        double dRhoMax = 0.0;
        for (int q = 0; q < Nprd; ++q)
        {
            dRhoMax = std::max(dRhoMax, dRho[q]);
            if (res && res->dRho)
                res->dRho[(size_t)(iter - 1) * Nprd + q] = dRho[q];
            if (res && res->dRhoMaxIdx)
                res->dRhoMaxIdx[(size_t)(iter - 1) * Nprd + q] = dRhoIdx[q];
        }
        if (res && res->dJPrdMax)
            res->dJPrdMax[iter - 1] = dJ;
        if (res && res->dJPrdMaxIdx)
            res->dJPrdMaxIdx[iter - 1] = dJIdx;
        if (dRhoMax < tol)
            break;
    }
    if (res)
        res->NprdSubIter = iter;
    return LWHIP_OK;
}

extern "C" int lwhip_formal_solver_2d_impl(int device, const lwhip_grid2d* g, double wavelength, int nSolve,
                                           const int32_t* rays, const double* chi, const double* S, double* I,
                                           double* PsiStar, char* err, int errLen);

extern "C" int lwhip_build_intersections_impl(const lwhip_grid2d* grid, lwhip_intersection* uw, lwhip_intersection* dw,
                                              int32_t* longCharIdx, int32_t* substepOff, int32_t capLongChar,
                                              lwhip_intersection* substeps, int64_t capSubsteps, int32_t* nLongChar,
                                              int64_t* nSubsteps);

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

// the argument blocks of the context's lines (wavelength slices and tickets of their own: the lines run side by side)
static int voigt_line_list(lwhip_context* c, std::vector<VoigtLineArgs>& out)
{
    if (!c->prob.vlosMu)
        return fail(LWHIP_ERR_INVALID, "lwhip_compute_profiles needs vlosMu in the descriptor");
    const int slices = 16;
    const size_t tiles = (size_t)(c->Ns + 7) / 8 + 1;
    const size_t nLine = (size_t)std::max(c->Nline, 1);
    if (!c->is2d && c->wphiScratch.n < nLine * slices * c->Ns)
    {
        HIP_TRY(c->wphiScratch.alloc(nLine * slices * c->Ns));
        HIP_TRY(c->wphiTicket.alloc(nLine * tiles));
        HIP_TRY(hipMemsetAsync(c->wphiTicket.p, 0, c->wphiTicket.n * sizeof(int), c->stream));
    }
    size_t q = 0;
    for (int tr = 0; tr < c->Ntrans; ++tr)
    {
        const HostTrans& h = c->trans[tr];
        if (h.t.type != LWHIP_LINE)
            continue;
        if (!h.t.aDamp)
            return fail(LWHIP_ERR_INVALID, "lwhip_compute_profiles needs aDamp for every line");
        VoigtLineArgs a{};
        a.Ns = c->Ns;
        a.Nrays = c->Nrays;
        a.nlt = h.NredLoc - h.NblueLoc;
        a.ltStart = h.ltStart;
        a.nltFull = h.t.Nred - h.t.Nblue;
        a.lambda0 = h.t.lambda0;
        a.wave = c->lineWave.p + h.waveOff;
        a.wlam = c->lineWlam.p + h.waveOff;
        a.vlosMu = c->vlosMu.p;
        a.wmu = c->wmu.p;
        a.vBroad = c->vBroad.p + (size_t)h.atom * c->Ns;
        a.aDamp = c->aDamp.p + (size_t)h.row * c->Ns;
        a.phi = c->phi.p + (a.nlt > 0 ? h.phiOff : 0);
        a.wphi = c->wphi.p + (size_t)h.row * c->Ns;
        a.wphiScratch = c->is2d ? nullptr : c->wphiScratch.p + q * slices * c->Ns;
        a.wphiTicket = c->is2d ? nullptr : c->wphiTicket.p + q * tiles;
        a.wphiSlices = slices;
        out.push_back(a);
        ++q;
    }
    return LWHIP_OK;
}

// phi and wphi of every line from (aDamp, vBroad, vlosMu) on the device: compute_profile's Voigt loop
// (Source/LwMiddleLayer.pyx:3244-3288 -> LwTransition.hpp:155-190), one launch pair for the context's lines
static int compute_profiles_impl(lwhip_context* c)
{
    std::vector<VoigtLineArgs> list;
    const int st = voigt_line_list(c, list);
    if (st != LWHIP_OK)
        return st;
    if (list.size() == 1)
        HIP_TRY(launch_voigt_line(list[0], c->stream));
    else if (!list.empty())
    {
        // (the copy is queued behind the previous launch pair that read the buffer: same stream)
        if (c->voigtList.n < list.size())
            HIP_TRY(c->voigtList.alloc(list.size()));
        HIP_TRY(hipMemcpyAsync(c->voigtList.p, list.data(), list.size() * sizeof(VoigtLineArgs), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(launch_voigt_lines(c->voigtList.p, list.data(), (int)list.size(), c->stream));
    }
    c->deviceProfiles = true;
    c->profilesStale = false;
    return retile_profiles(c);
}

// before a sweep: device-generated profiles whose inputs were uploaded again are brought up to date
static int ensure_profiles(lwhip_context* c)
{
    return c->profilesStale ? compute_profiles_impl(c) : LWHIP_OK;
}

int lwhip_compute_profiles(lwhip_context* c)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    return compute_profiles_impl(c);
}

int lwhip_set_zplane_outputs(lwhip_context* c, double* zPlaneDown, double* zPlaneUp)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t n = (size_t)c->Nla * c->Nrays * c->Nx;
    c->zDownHost = zPlaneDown;
    c->zUpHost = zPlaneUp;
    if (zPlaneDown && c->zDown.n < n)
        HIP_TRY(c->zDown.alloc(n));
    if (zPlaneUp && c->zUp.n < n)
        HIP_TRY(c->zUp.alloc(n));
    if (!zPlaneDown)
        c->zDown.release();
    if (!zPlaneUp)
        c->zUp.release();
    if (zPlaneDown)
        HIP_TRY(hipMemsetAsync(c->zDown.p, 0, n * sizeof(double), c->stream));
    if (zPlaneUp)
        HIP_TRY(hipMemsetAsync(c->zUp.p, 0, n * sizeof(double), c->stream));
    return build_tile_args(c); // the sweep's argument block carries the pointers
}

int lwhip_synchronize(lwhip_context* c)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return verify_zero_check(c);
}

int lwhip_set_stream(lwhip_context* c, void* stream)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (stream)
        c->stream = (hipStream_t)stream;
    else
    {
        if (!c->ownStream)
            HIP_TRY(hipStreamCreate(&c->ownStream));
        c->stream = c->ownStream;
    }
    return LWHIP_OK;
}

int lwhip_profile_enable(lwhip_context* c, int enable)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    int st = collect_profile(c);
    if (st != LWHIP_OK)
        return st;
    c->profiling = enable != 0;
    c->profEvery = enable > 1 ? enable : 1; // n > 1: a sample of every n-th launch (two event records cost ~5 us)
    c->profCount = 0;
    c->sweepMs = 0.0;
    c->sweepCount = 0;
    return LWHIP_OK;
}

int lwhip_sweep_time(lwhip_context* c, double* meanMs, int* nLaunches)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    int st = collect_profile(c);
    if (st != LWHIP_OK)
        return st;
    if (meanMs)
        *meanMs = c->sweepCount ? c->sweepMs / c->sweepCount : 0.0;
    if (nLaunches)
        *nLaunches = c->sweepCount;
    return LWHIP_OK;
}

int lwhip_sweep_kind(lwhip_context* c)
{
    if (!c)
        return -1;
    return c->is2d ? 2 : c->laneSweep ? 1 : 0;
}

int lwhip_algorithmic_bytes(lwhip_context* c, double* bytes)
{
    if (!c || !bytes)
        return fail(LWHIP_ERR_INVALID, "null argument");
    // SURVEY.md 8(d) / BASELINE.md 4: phi read once, background chi/eta/sca read, J read + write,
    // emergent I write, Gamma read + write and n read, Rij/Rji write, small replicated terms.
    const double Ns = c->Ns, Nla = c->Nla, Nmu = c->Nrays;
    double sumLineLa = 0.0, gam = 0.0, lev = 0.0;
    for (auto& h : c->trans)
        if (h.t.type == LWHIP_LINE)
            sumLineLa += h.NredLoc - h.NblueLoc;
    for (auto& a : c->atoms)
    {
        lev += a.Nlevel;
        if (!a.detailed)
            gam += 2.0 * a.Nlevel * a.Nlevel;
    }
    double b = 8.0 * Ns * (2.0 * Nmu * sumLineLa + 3.0 * Nla + 2.0 * Nla);
    b += 8.0 * Nla * Nmu * c->Nx; // emergent I
    b += 8.0 * Ns * (gam + lev);
    b += 8.0 * Ns * 2.0 * c->Ntrans;
    b += 8.0 * Ns * (lev + 3.0 + Nmu);
    *bytes = b;
    return LWHIP_OK;
}
}

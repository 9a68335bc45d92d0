// lwhip_host.h -- what the host-side translation units of the C ABI (include/lwhip.h) share: the device-buffer wrapper,
// the context (the HBM-resident copy of a problem, its tables and launch state), the error plumbing and the functions that
// cross file boundaries.
//
//     lwhip_api.hip      create / destroy, one iteration (sweep -> reduce -> [all-reduce by the caller] -> apply), populations
//     lwhip_tables.hip   validation of the descriptor and the per-wavelength activity tables of the sweeps (build_tables)
//     lwhip_state.hip    device allocations, kernel argument blocks, the sweep's launch sequence, upload / download
//     lwhip_api2d.hip    the 2D iteration's launch sequence, the 2D primitives
//     lwhip_batch.hip    1.5D column batches
//     lwhip_api_prd.hip  PRD sub-iterations
//
// There is no CPU fallback: without a HIP device every compute entry point fails with LWHIP_ERR_DEVICE.
#pragma once
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
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

namespace lwhip
{
// records the message lwhip_last_error() returns (thread-local) and passes the code through
int fail(int code, const std::string& msg);

#define HIP_TRY(expr)                                                                                  \
    do                                                                                                 \
    {                                                                                                  \
        hipError_t err__ = (expr);                                                                     \
        if (err__ != hipSuccess)                                                                       \
            return fail(LWHIP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(err__));       \
    } while (0)

// > 0: the clearing fills of DevBuf::alloc are queued, not waited for: whoever raised it synchronises the null stream before
// anything touches the buffers (lwhip_create: 0.7 s less per 512 columns than a wait per allocation)
inline bool debug_knobs_on(); // (LWHIP_DEBUG=1: the diagnosis knobs below are read only then)
inline int& devbuf_defer_fill_sync()
{
    static thread_local int depth = 0;
    return depth;
}

// > 0: alloc(count) leaves out its safety-net clearing (alloc_zero still clears): the state buffers of a context made with
// lwhip_create_like -- a column of a 1.5D batch: ~40 fills of a few KB each per column, a driver call apiece.  Every byte a
// kernel reads of such a buffer is written first by an upload or a kernel (tests/test_padding.py runs the column batches under
// the finite sentinel fills); the owner of the tables, and every ordinary context, keeps the net.
inline int& devbuf_skip_safety_clear()
{
    static thread_local int depth = 0;
    return depth;
}

// The state allocations of a context made with lwhip_create_like come out of ONE device allocation (round 6): the owner of the
// tables counts the bytes its own alloc_state asks for (count mode), a borrower gets an arena of that size, cleared with one
// fill, and its DevBufs take consecutive 256-byte-aligned pieces of it -- ~60 hipMalloc and ~13 fills per column of a 1.5D
// batch become one of each.  A piece that does not fit (or any allocation outside lwhip_create) takes the ordinary path.
struct DevArena
{
    unsigned char* base = nullptr;
    size_t size = 0, used = 0;
    size_t counted = 0;   // count mode: bytes asked for
    bool counting = false;
    bool overflowed = false; // a piece did not fit and took the ordinary path (with its clearing fill on the null stream)
};
inline DevArena*& devbuf_arena()
{
    static thread_local DevArena* a = nullptr;
    return a;
}

// Host-to-device copies of a batch column's creation, gathered (round 6): while a borrower of a 1.5D batch is created, every
// upload -- the H2D macro, the argument blocks, DevBuf::upload -- is copied into ONE page-locked stage and described by a record;
// lwhip_create then sends the stage to a device inbox with one copy and lets one kernel (h2d_scatter_kernel) move the pieces to
// their buffers.  A column made ~55 copies (several of them from pageable memory, which the runtime stages while holding its
// lock) and one synchronous hipMemcpy that waited 6 ms for the other creating threads' streams; now one copy and one launch.
struct H2DBatch
{
    hipStream_t stream = nullptr;
    unsigned char* host = nullptr;  // page-locked stage
    unsigned char* inbox = nullptr; // its landing place on the device
    size_t cap = 0, used = 0;
    std::vector<H2DRec> recs;
    bool inFlight = false; // a flush was queued and the stage not yet known to be free again
    long copies = 0;
    bool add(void* dst, const void* src, size_t bytes);
    hipError_t flush();
};
inline H2DBatch*& h2d_batch()
{
    static thread_local H2DBatch* b = nullptr;
    return b;
}
// the copy every upload path calls: into the thread's batch when one is open on this stream, else the ordinary copy
inline hipError_t h2d_copy(void* dst, const void* src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0)
        return hipSuccess;
    H2DBatch* b = h2d_batch();
    if (b && b->stream == stream && b->add(dst, src, bytes))
        return hipSuccess;
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
}

template <typename T> struct DevBuf
{
    T* p = nullptr;
    size_t n = 0;
    bool owned = true;
    uint64_t sig = 0; // content fingerprint of what upload() put there (see upload_or_borrow)
    // What a fresh allocation holds.  Round 4 handed out CLEARED memory because some kernel read bytes no upload had written and
    // hipMalloc returns whatever the pages' previous owner left there.  Round 5 located such reads with a FINITE sentinel
    // (LWHIP_PAD_SENTINEL, below; NaN is swallowed by the fmin / fmax of the Steffen derivative, a finite 6.7e299 is not) and
    // fixed them; what is left of the clearing is listed in DESIGN.md section 4.
    //   alloc(count)         state buffer: every byte a kernel reads is written first by an upload or a kernel (asserted by the
    //                        sentinel test); cleared all the same as a safety net unless LWHIP_NO_CLEAR is set
    //   alloc(count, false)  the caller overwrites all of it at once (upload)
    //   alloc_zero(count)    a buffer that is ACCUMULATED into, or whose zero is a value: always cleared
    // LWHIP_PAD_SENTINEL=all | N | A-B (diagnosis): the fp64 state allocations numbered so (per context, LWHIP_TRACE_ALLOC prints
    // the numbers) are filled with the sentinel instead of zeros -- a result that depends on never-written memory is then off by
    // hundreds of orders of magnitude, deterministically (tests/test_padding.py runs the parity problems this way).
    hipError_t alloc(size_t count, bool clear = true) { return alloc_impl(count, clear ? 1 : 0); }
    hipError_t alloc_zero(size_t count) { return alloc_impl(count, 2); }
    hipError_t alloc_impl(size_t count, int init)
    {
        release();
        n = count;
        if (count == 0)
            return hipSuccess;
        if (DevArena* ar = devbuf_arena())
        {
            const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
            if (ar->counting)
                ar->counted += bytes;
            else if (ar->used + bytes <= ar->size)
            {
                // (the arena was cleared as a whole: zeros serve alloc_zero and the safety net alike)
                p = (T*)(ar->base + ar->used);
                ar->used += bytes;
                owned = false;
                return hipSuccess;
            }
            else
                ar->overflowed = true;
        }
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        const int seq = alloc_seq()++;
        static const bool noClear = debug_knobs_on() && std::getenv("LWHIP_NO_CLEAR") != nullptr; // (no safety net: what hipMalloc returns)
        const bool sentinel = init == 1 && std::is_same<T, double>::value && sentinel_for(seq);
        const char* what = "";
        if (e == hipSuccess && poison_on() && init != 2)
        {
            // diagnosis (LWHIP_POISON=1): fresh allocations hold NaN / -1 (not alloc_zero's: their zero is a value -- counters,
            // accumulators --, which round 6's peer-exchange arrival counter was the first to rely on without a clear of its own)
            e = hipMemset(p, 0xFF, count * sizeof(T));
            what = " poisoned";
        }
        else if (e == hipSuccess && sentinel)
        {
            // LWHIP_PAD_SENTINEL_VALUE=<x>: another fill value (a huge one is "optically thick" and may be as harmless as zero
            // where a plausible one is not: tests/test_padding.py tries several)
            static const char* sv = std::getenv("LWHIP_PAD_SENTINEL_VALUE");
            if (sv && *sv)
            {
                std::vector<double> h(count * sizeof(T) / sizeof(double) + 1, std::atof(sv));
                if (sv[0] == 'r')
                {
                    // "rand" / "randsign": DIFFERENT plausible values everywhere (log-uniform in 1e-8 .. 1e8) -- what stale data
                    // of a destroyed context looks like.  A uniform fill cannot show a dependence that goes through a
                    // difference or a ratio of two padding values (a slope, S = eta / chi): those are 0 and 1 for any constant.
                    // (one table of 2^20 values, made once, laid end to end from an offset that follows the allocation number)
                    const bool sign = std::strncmp(sv, "randsign", 8) == 0;
                    static std::vector<double> table;
                    if (table.empty())
                    {
                        table.resize(1u << 20);
                        uint64_t x = 0x9E3779B97F4A7C15ull;
                        for (auto& v : table)
                        {
                            x ^= x << 13;
                            x ^= x >> 7;
                            x ^= x << 17;
                            const double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
                            v = std::pow(10.0, 16.0 * u - 8.0) * ((sign && (x & 1)) ? -1.0 : 1.0);
                        }
                    }
                    size_t off = ((size_t)(seq + 1) * 7919u) & (table.size() - 1);
                    for (size_t i = 0; i < h.size();)
                    {
                        const size_t nCopy = std::min(h.size() - i, table.size() - off);
                        std::memcpy(h.data() + i, table.data() + off, nCopy * sizeof(double));
                        i += nCopy;
                        off = 0;
                    }
                }
                e = hipMemcpy(p, h.data(), count * sizeof(T), hipMemcpyHostToDevice);
            }
            else
                e = hipMemsetD32((hipDeviceptr_t)p, (int)0x7E377E37, count * sizeof(T) / 4); // 0x7E377E377E377E37 = 6.7e299
            what = " sentinel";
        }
        else if (e == hipSuccess && init == 1 && !std::is_same<T, double>::value && sentinel_on() && std::getenv("LWHIP_PAD_SENTINEL_INT"))
        {
            // (the other state allocations -- index and argument tables -- hold this 32-bit word: stale but valid-looking
            // indices are what neither zeros nor the 0xFF of LWHIP_POISON imitate)
            e = hipMemset(p, 0, count * sizeof(T));
            if (e == hipSuccess && count * sizeof(T) >= 4)
                e = hipMemsetD32((hipDeviceptr_t)p, std::atoi(std::getenv("LWHIP_PAD_SENTINEL_INT")), count * sizeof(T) / 4);
            what = " int-sentinel";
        }
        else if (e == hipSuccess && (init == 2 || (init == 1 && !noClear && devbuf_skip_safety_clear() == 0)))
        {
            e = hipMemset(p, 0, count * sizeof(T));
            what = " cleared";
        }
        static const bool syncAlloc = debug_knobs_on() && std::getenv("LWHIP_SYNC_ALLOC") != nullptr; // (diagnosis: round 4's clearing minus the clearing)
        // (the caller's stream may be a non-blocking one: no implicit order with the fill -- wait for it, or, inside lwhip_create,
        // let the ~60 fills of a context queue up and wait once before the first upload: defer_fill_sync)
        if (e == hipSuccess && ((*what && devbuf_defer_fill_sync() == 0) || syncAlloc))
            e = hipStreamSynchronize(nullptr);
        static const bool trace = std::getenv("LWHIP_TRACE_ALLOC") != nullptr; // diagnosis: which buffer a number / an address is
        if (trace)
            std::fprintf(stderr, "lwhip alloc #%d %p .. %p (%zu x %zu B)%s\n", seq, (void*)p, (void*)((char*)p + count * sizeof(T)), count, sizeof(T),
                         what);
        return e;
    }
    static std::atomic<int>& alloc_seq() // (diagnosis only: the numbers of LWHIP_TRACE_ALLOC / LWHIP_PAD_SENTINEL)
    {
        static std::atomic<int> seq{ 0 };
        return seq;
    }
    static bool poison_on()
    {
        static const bool poison = debug_knobs_on() && std::getenv("LWHIP_POISON") != nullptr;
        return poison;
    }
    // LWHIP_PAD_SENTINEL: "all", one number, or a range A-B (inclusive) of the fp64 allocation numbers of a context
    static bool sentinel_on()
    {
        static const bool on = debug_knobs_on() && std::getenv("LWHIP_PAD_SENTINEL") != nullptr && *std::getenv("LWHIP_PAD_SENTINEL");
        return on;
    }
    static bool sentinel_for(int seq)
    {
        if (!sentinel_on())
            return false;
        static int lo = 0, hi = -1;
        static bool parsed = false;
        if (!parsed)
        {
            const char* v = std::getenv("LWHIP_PAD_SENTINEL");
            if (v[0] == 'a')
            {
                lo = 0;
                hi = 1 << 30;
            }
            else
            {
                lo = hi = std::atoi(v);
                const char* dash = std::strchr(v, '-');
                if (dash)
                    hi = std::atoi(dash + 1);
            }
            parsed = true;
        }
        return seq >= lo && seq <= hi;
    }
    // the clearing of a buffer whose zeros only keep its padding finite (not a value): left out under the sentinel
    hipError_t clear_padding()
    {
        if (!p || sentinel_on())
            return hipSuccess;
        if (const DevArena* ar = devbuf_arena())
            if (!ar->counting && (unsigned char*)p >= ar->base && (unsigned char*)p < ar->base + ar->size)
                return hipSuccess; // (a piece of the borrower's arena, cleared as a whole a moment ago)
        hipError_t e = hipMemset(p, 0, n * sizeof(T));
        return (e == hipSuccess && devbuf_defer_fill_sync() == 0) ? hipStreamSynchronize(nullptr) : e;
    }
    // a window of somebody else's allocation (the depth arena of the lane sweep)
    void view(T* ptr, size_t count)
    {
        release();
        p = ptr;
        n = count;
        owned = false;
    }
    // the table of a structurally identical context instead of a copy of one's own (lwhip_create_like): same size, or upload
    // ... and only if it holds what this context would upload: the structure fingerprint of lwhip_create_like covers the
    // problem, not the layout choices build_tables reads from the environment (LWHIP_SWEEP, the tile / split / tail knobs of
    // tests and tools) -- a borrower made under another environment computes another layout and must not run on the donor's
    // tables.  The fingerprint samples the bytes (size, both ends, ~4 000 words in between): a different layout differs in
    // sizes or in the first records.
    static uint64_t fingerprint(const std::vector<T>& v)
    {
        const size_t nw = v.size() * sizeof(T) / 4;
        const unsigned char* b = (const unsigned char*)v.data();
        uint64_t h = 1469598103934665603ull ^ (uint64_t)(v.size() * sizeof(T));
        auto word = [&](size_t w) {
            uint32_t x;
            std::memcpy(&x, b + 4 * w, 4);
            h = (h ^ x) * 1099511628211ull;
        };
        const size_t step = nw > 8192 ? nw / 4096 : 1;
        for (size_t w = 0; w < nw; w += step)
            word(w);
        for (size_t w = nw > 16 ? nw - 16 : 0; w < nw; ++w)
            word(w);
        return h | 1ull;
    }
    hipError_t upload_or_borrow(const std::vector<T>& v, const DevBuf<T>* from)
    {
        if (from && from->p && from->n == v.size() && !v.empty() && from->sig == fingerprint(v))
        {
            view(from->p, from->n);
            sig = from->sig;
            return hipSuccess;
        }
        // (a structure table: the owner's copy is what borrowers use, so it is neither counted into the size of a borrower's
        // arena nor -- where a borrower has to keep a copy of its own after all -- taken out of one)
        DevArena* const ar = devbuf_arena();
        devbuf_arena() = nullptr;
        const hipError_t e = upload(v);
        devbuf_arena() = ar;
        return e;
    }
    hipError_t upload(const std::vector<T>& v)
    {
        hipError_t e = alloc(v.size(), poison_on());
        if (e != hipSuccess || v.empty())
            return e;
        sig = fingerprint(v);
        if (H2DBatch* b = h2d_batch())
            if (b->add(p, v.data(), v.size() * sizeof(T)))
                return hipSuccess;
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
        sig = 0;
    }
    ~DevBuf() { release(); }
};


struct HostTrans
{
    lwhip_transition t;   // borrowed host pointers
    int atom;
    int NblueLoc, NredLoc; // clipped to the shard, shard-local indices
    int ltStart;           // first own-grid index inside the shard
    int rhoLt0 = 0, rhoRows = 0; // PRD lines: the own-grid rows held in the rho pool (the shard's; hybrid PRD: the whole grid)
    int row;               // wphi row (lines) / ratio row (continua)
    int64_t parOff, phiOff, rhoOff;
    int64_t waveOff; // lines: offset of the full own grid / wlambda in the lineWave / lineWlam pools
};
}
using namespace lwhip;

struct lwhip_context;
namespace lwhip
{
int host_block_init(lwhip_context* c);
void host_block_release(lwhip_context* c);
hipError_t stream_acquire(int device, hipStream_t* out);
void stream_release(int device, hipStream_t s);
hipError_t stream_acquire_shared(int device, hipStream_t* out); // the columns of a batch made by one thread share a stream
void stream_release_shared(int device, hipStream_t s);
void* pinned_acquire(int device, size_t bytes, void** devPtr);  // page-locked, device-mapped host memory out of pooled slabs
void pinned_release(int device, void* p, size_t bytes);
void peer_release(lwhip_context* c);
int peer_publish(lwhip_context* c);
int fingerprint_J_enqueue(lwhip_context* c);
void fingerprint_J_fold(lwhip_context* c, const void* p);
void peer_apply_args(lwhip_context* c, struct ApplyArgs& a);
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
    int maxM = 0, maxP = 0; // most mixed / pure continua at one wavelength
    bool is2d = false;            // x-periodic 2D geometry (prob.grid2d): batched pipeline of lwhip_2d.hip
    int Nx = 1, batch2d = 1;
    std::vector<DevLaHeader> hdrHost;
    std::vector<int32_t> contLaHost;
    // the ray-column sweep (lwhip_raymarch.hip): tiles of structurally identical wavelengths
    bool tiled = false;
    int tileL = 12, nTiles = 0, nTileChunks = 0, nTileChunksPrd = 0, nPostChunks = 0, nPostChunksPrd = 0, nTilesPrd = 0;
    bool tileFuse = false; // the sweep's workgroup = one tile, finished by the post-pass inside the sweep launch
    int depthSplit = 1;    // the march on deep columns: a direction's depth points split over this many wavefronts (lwhip_raymarch.hip)
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
    DevBuf<double> detSlab, detPart; // (detPart: partial sums of the fixed-shape reduction, lwhip_lanesweep.hip)
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
    DevBuf<uint8_t> laneFeedG;    // generic tiles of the lane sweep: continuum -> slot feed bytes
    int nGenTiles = 0;
    DevBuf<double> momScratch;    // ... and the moment scratch of their wavefronts (TileArgs::momS)
    int momA = 0;
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
    bool ownStreamShared = false;

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
    double* tailMapped = nullptr; // host-mapped (dJMax, idx) of the one-call iteration (a window of hostBlock)
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
    // lwhip_create_like: the structure tables (lwhip_tables.hip) are borrowed from a context of the same structure, which
    // counts its borrowers and cannot be destroyed before them
    lwhip_context* tablesFrom = nullptr;
    std::atomic<int> borrowers{ 0 }; // (contexts are created and destroyed from several host threads: columns, per-GPU workers)
    uint64_t structSig = 0; // fingerprint of everything the structure tables are built from
    bool dJPrdClean = false; // dJ holds zeros outside the wavelengths the PRD rates pass visits (no full sweep since)
    std::vector<int> prdLines;        // PRD lines of the active atoms (global transition indices), reference order
    std::vector<int64_t> prdRowOff;   // first row of each line's grid in the prdJ gather buffer
    int64_t prdRowsTot = 0;
    bool prdPending = false;          // between lwhip_prd_partial and lwhip_prd_finalise
    bool prdFused = false;            // ... and its apply launch reads the stage-1 parts and reduces the lines' changes itself
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
    // pipelined sub-iterations of lwhip_redistribute_prd (one device, 1D, lane sweep): the launches of up to PRD_PIPE_DEPTH
    // sub-iterations are queued without a host round trip in between; the device keeps the loop's stopping rule (prdCtl, see
    // ApplyArgs) and every sub-iteration's results land in its own slot of prdPinnedPipe
    DevBuf<int32_t> prdCtl;
    // every line's profile is the same for the two directions of an angle (static atmosphere): found when the profiles are
    // uploaded (the host's arrays are compared) or generated (all line-of-sight velocities zero); TileDyn::phiSym
    bool phiSym = false, vlosZero = false;
    // LWHIP_PAIR_RAYS=0 / LWHIP_PRD_PIPELINE=0 (experiment knobs, read ONCE per context in lwhip_create and only under
    // LWHIP_DEBUG like the other layout knobs: the first changes the order of the arithmetic)
    bool pairRays = true, prdPipeline = true, prdGeneral = false;
    double* prdPinnedPipe = nullptr;
    size_t prdPinnedPipeN = 0;
    int prdPipeIter = 0;     // > 0: the sub-iteration the calls of lwhip_prd_partial / _finalise belong to
    double prdPipeTol = 0.0;
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
    int* zeroCheckDev = nullptr;  // (windows of hostBlock; null unless the knob is set)
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

    // ONE pinned, host-mapped block per context holds every word the DEVICE stores into host memory (tailMapped, statusHost,
    // changeHost, zeroCheck point into it) and every target of a small device-to-host copy the API used to aim at a stack
    // variable (popStatusHost, prdCtlHost): made once in lwhip_create (host_block_init), released by the destructor AFTER the
    // stream has drained -- so no kernel and no copy engine can hold an address of host memory that is not this context's own
    // for the context's whole life, whatever path a call returns by.  LWHIP_DEBUG: the block is never given back to the
    // runtime (quarantine, lwhip_api.hip) and its canary words are checked when the process ends.
    // peer exchange of the sharded iteration (lwhip_peer_*, lwhip_api.hip): this rank's window [flags 2 x LWHIP_PEER_MAX |
    // slots 2 x world x peerStride doubles] and the device addresses of every rank's window (own included)
    unsigned char* peerWin = nullptr;
    size_t peerWinBytes = 0, peerStride = 0;
    unsigned char* peerPtr[LWHIP_PEER_MAX] = {};
    bool peerIpc[LWHIP_PEER_MAX] = {};   // opened with hipIpcOpenMemHandle (closed in lwhip_peer_detach)
    bool peerOn = false;
    unsigned long long peerSeq = 0;      // exchanges so far (the next one uses buffer peerSeq & 1 with flag value peerSeq + 1)
    DevBuf<int32_t> peerArrive;
    size_t stateBytes = 0;         // what this context's alloc_state asked for (the size of a borrower's arena)
    unsigned char* stateArena = nullptr;
    // lwhip_map_host_J: the caller's J array page-locked and mapped; the sweep stores J there too
    void* JhostReg = nullptr;      // what was registered (prob.J)
    double* JhostDev = nullptr;    // device address of this shard's first row in it
    DevBuf<double> Jsnap;          // lwhip_j_snapshot
    DevBuf<unsigned long long> fpSums; // lwhip_fingerprint_J: the blocks' sums
    unsigned long long* fpPinned = nullptr;
    size_t fpPinnedN = 0;
    bool fpJValid = false;         // fpJValue is the fingerprint of the device's current J as seen at address fpJPtr
    uint64_t fpJValue = 0;
    const void* fpJPtr = nullptr;
    unsigned char* hostBlock = nullptr;
    size_t hostBlockBytes = 0;
    int32_t* turnLateHost = nullptr;  // fixed-order mode: turn waits that timed out (device-written, TileArgs::turnLate)
    int32_t* turnLateDev = nullptr;
    int32_t* popStatusHost = nullptr; // device-to-host target of the status word of time_dep_update / nr_post_update
    int32_t* prdCtlHost = nullptr;    // ... of the stopping-rule words of the pipelined PRD sub-iterations

    ~lwhip_context()
    {
        // (a context that failed half-way through lwhip_create comes here without lwhip_destroy: nothing of it may be
        // released while its stream -- or a clearing fill on the null stream -- still runs)
        if (stream)
            (void)hipStreamSynchronize(stream);
        (void)hipStreamSynchronize(nullptr);
        for (auto& pr : pending)
        {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        if (ownStream && ownStreamShared)
            stream_release_shared(device, ownStream);
        else if (ownStream)
            stream_release(device, ownStream); // (never hipStreamDestroy: see stream_acquire, lwhip_api.hip)
        if (gatherPinned)
            (void)hipHostFree(gatherPinned);
        if (stage)
            pinned_release(device, stage, stageDoubles * sizeof(double));
        if (prdPinned)
            (void)hipHostFree(prdPinned);
        if (prdPinnedPipe)
            (void)hipHostFree(prdPinnedPipe);
        peer_release(this);
        if (stateArena) // (the DevBufs that point into it do not own their pieces)
            (void)hipFree(stateArena);
        if (JhostReg)
            (void)hipHostUnregister(JhostReg);
        if (fpPinned)
            (void)hipHostFree(fpPinned);
        host_block_release(this);
    }
};

#define H2D(dst, src, count)                                                                           \
    HIP_TRY(lwhip::h2d_copy((dst), (src), (size_t)(count) * sizeof(double), c->stream))
#define D2H(dst, src, count)                                                                           \
    HIP_TRY(hipMemcpyAsync((dst), (src), (size_t)(count) * sizeof(double), hipMemcpyDeviceToHost, c->stream))

namespace lwhip
{
inline int env_int(const char* name, int dflt)
{
    const char* v = std::getenv(name);
    if (!v || !*v)
        return dflt;
    return std::atoi(v);
}
// The layout / experiment knobs of tests and tools (tile widths, wavefronts per workgroup, split factors, forced kinds: the list
// is in INTEGRATION.md section 5) are read only when LWHIP_DEBUG is set: a production process cannot change the work
// distribution -- or the summation order -- through a stray environment variable.
inline bool debug_knobs_on()
{
    static const bool on = std::getenv("LWHIP_DEBUG") != nullptr;
    return on;
}
inline int dbg_env_int(const char* name, int dflt) { return debug_knobs_on() ? env_int(name, dflt) : dflt; }

inline double trans_wlambda(const lwhip_transition& t, int lt)
{
    // Transition::wlambda, LwTransition.hpp:71-81
    const int len = t.Nred - t.Nblue;
    if (lt == 0)
        return 0.5 * (t.wavelength[1] - t.wavelength[0]) * t.dopplerWidth;
    if (lt == len - 1)
        return 0.5 * (t.wavelength[len - 1] - t.wavelength[len - 2]) * t.dopplerWidth;
    return 0.5 * (t.wavelength[lt + 1] - t.wavelength[lt - 1]) * t.dopplerWidth;
}

// lwhip_tables.hip
int validate(const lwhip_problem* p, std::string& why);
int build_tables(lwhip_context* c);
uint64_t structure_signature(const lwhip_context* c);
// lwhip_state.hip
int alloc_state(lwhip_context* c);
int build_sweep_args(lwhip_context* c);
int build_tile_args(lwhip_context* c);
int retile_profiles(lwhip_context* c);
bool retile_args(lwhip_context* c, RetileArgs& r);
TileDyn make_dyn(lwhip_context* c, bool upOnly, int lambdaIterate);
hipError_t run_sweep(lwhip_context* c, const TileDyn& dyn, bool rates, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
int flush_prefill(lwhip_context* c);
int verify_zero_check(lwhip_context* c);
int collect_profile(lwhip_context* c);
// lwhip_api.hip
ReduceArgs make_reduce_args(lwhip_context* c);
ApplyArgs make_apply_args(lwhip_context* c);
int ensure_profiles(lwhip_context* c);
int voigt_line_list(lwhip_context* c, std::vector<VoigtLineArgs>& out);
int stat_equil_impl(lwhip_context* c, int atom, bool wait, double* dPops = nullptr, int32_t* dPopsMaxIdx = nullptr);
// lwhip_api2d.hip
int run_2d(lwhip_context* c, int lambdaIterate, int mode = 0);
}

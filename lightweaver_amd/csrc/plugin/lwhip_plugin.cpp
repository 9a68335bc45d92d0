// lwhip_plugin.cpp -- Lightweaver iteration-scheme plugin backed by the MI355X HIP library.
//
// This is the reference-side binding of the C ABI in include/lwhip.h: a shared object exporting
//
//     extern "C" FsIterationFns fs_iteration_fns_provider();      (Source/LwFormalInterface.hpp:110-136)
//
// that Lightweaver loads with FsIterationSchemes.load_fns_from_path(path) and selects with
// Context(..., fsIterScheme="mali_full_precond_hip_gfx950") (Source/LwMiddleLayer.pyx:4172-4184,
// 3077-3087).  It is compiled against the Lightweaver headers exactly like the reference's own
// SIMD plugins (setup.py:255-271: plugin source + LightweaverAmalgamated.cpp), so struct layouts
// agree by construction.  Nothing numerical happens here: fs_iter / simple_fs / stat_eq flatten
// the borrowed numpy-backed views of `Context` into an `lwhip_problem`, move the arrays that may
// have changed to HBM, call the C ABI and copy the results back into the same host arrays, which
// stay the source of truth (pickling, Ng acceleration, the Python layer all keep working).
// Members this build does not accelerate forward to the core's own implementations, as the
// reference plugins do for theirs (Source/SimdImpl_AVX2FMA.cpp:643-658).
#include "Lightweaver.hpp"
#include "lwhip.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <thread>
#include <condition_variable>
#include <chrono>
#include <cstring>
#include <unistd.h>
#include <functional>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace
{
// Exact change detection for the slowly changing inputs: a 64-bit fingerprint of EVERY element of every tracked array
// (plus pointer and length).  A sampled fingerprint would miss an in-place edit of an unsampled depth point -- what
// update_deps() does after a local atmosphere change -- and the iteration would run on stale device inputs without
// any sign of it.  At the bench size that is 90 MB per call (67 MB of line profiles), so it has to run at memory speed:
// arrays are cut into 256 KB blocks handed to a persistent pool of host threads (round 4: spawning 7 threads per call
// was a tenth of a millisecond by itself); a block's fingerprint is a Fletcher-style pair of running sums in 8 independent
// 64-bit lanes -- s1 += w, s2 += s1: additions only, so the loop vectorises and one thread streams ~20 GB/s, against ~8 for
// the multiply-xor chains of rounds 2-3 -- folded through a multiplicative mix with the block's offset; block fingerprints
// are folded in order.  Any change of a single element changes s1 of its lane, a swap of two elements changes s2.
class HashPool
{
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable wake, done;
    std::function<void()> job;
    uint64_t generation = 0;
    int running = 0;
    bool stop = false;

    void loop()
    {
        uint64_t seen = 0;
        for (;;)
        {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lock(m);
                wake.wait(lock, [&] { return stop || generation != seen; });
                if (stop)
                    return;
                seen = generation;
                f = job;
            }
            f();
            {
                std::lock_guard<std::mutex> lock(m);
                if (--running == 0)
                    done.notify_one();
            }
        }
    }

public:
    // (never destroyed: the threads are detached and end with the process; a forked child -- the threads do not exist
    // there -- works through the blocks alone)
    static HashPool& get()
    {
        static HashPool* pool = new HashPool();
        return *pool;
    }
    long owner = 0;
    HashPool()
    {
        owner = (long)getpid();
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        // (round 6: up to 31 -- with J no longer crossing PCIe the fingerprints of the 90 MB are what a call waits for, and
        // one thread streams ~20 GB/s of the host's memory bandwidth)
        unsigned n = std::min(31u, hw > 2 ? hw / 2 - 1 : 0u); // (+ the calling thread)
        // LWHIP_HASH_THREADS=k: k pool threads (0: the calling thread alone) -- a host whose cores are contended may do better
        // with fewer
        if (const char* e = std::getenv("LWHIP_HASH_THREADS"))
            if (*e)
                n = (unsigned)std::min(255, std::max(0, std::atoi(e)));
        for (unsigned t = 0; t < n; ++t)
        {
            threads.emplace_back([this] { loop(); });
            threads.back().detach();
        }
    }
    bool usable() const { return !threads.empty() && (long)getpid() == owner; }
    std::mutex busy; // one round of work at a time: a second caller (another Context on another thread) hashes inline
    // f on every pool thread; returns at once (f must stay valid until wait() returns).  false: the pool is serving
    // somebody else's round (which spans that caller's uploads and kernels) -- the caller works through its blocks alone
    bool start(const std::function<void()>& f)
    {
        if (!busy.try_lock())
            return false;
        {
            std::lock_guard<std::mutex> lock(m);
            job = f;
            running = (int)threads.size();
            ++generation;
        }
        wake.notify_all();
        return true;
    }
    void wait()
    {
        {
            std::unique_lock<std::mutex> lock(m);
            done.wait(lock, [&] { return running == 0; });
        }
        busy.unlock();
    }
    // f on every pool thread and on the caller; returns when all are through
    void run(const std::function<void()>& f)
    {
        if (!usable())
        {
            f();
            return;
        }
        const bool pooled = start(f);
        f();
        if (pooled)
            wait();
    }
};

struct HashSet
{
    enum { BLOCK = 32768 };
    struct Job
    {
        const double* p;
        size_t n;
        uint64_t h;
        bool sampled;
    };
    struct Block
    {
        int job;
        size_t off, len;
        uint64_t h;
    };
    std::vector<Job> jobs;

    int add(const double* p, size_t n)
    {
        jobs.push_back({ p, p ? n : 0, 0, false });
        return (int)jobs.size() - 1;
    }
    // LWHIP_SAMPLED_FINGERPRINTS=1 (opt-in): every 509th element (and the last) of a large array instead of all of them --
    // 0.4 ms per call less at the bench size, at the price the comment above names: an in-place edit that touches none of the
    // sampled elements goes unnoticed (ExtraParams lwhip_force_upload covers a caller that knows it made one)
    int add_sampled(const double* p, size_t n)
    {
        uint64_t h = mix((uint64_t)(uintptr_t)p, n);
        if (p)
        {
            for (size_t i = 0; i < n; i += 509)
            {
                uint64_t w;
                std::memcpy(&w, p + i, 8);
                h = mix(h, w + i);
            }
            if (n)
            {
                uint64_t w;
                std::memcpy(&w, p + n - 1, 8);
                h = mix(h, w);
            }
        }
        jobs.push_back({ p, 0, h, true });
        return (int)jobs.size() - 1;
    }
    static uint64_t mix(uint64_t h, uint64_t bits)
    {
        h = (h ^ bits) * 0x9E3779B97F4A7C15ull;
        return h ^ (h >> 29);
    }
    static uint64_t block_hash(const double* p, size_t n, uint64_t seed)
    {
        uint64_t s1[8] = { 1, 2, 3, 4, 5, 6, 7, 8 }, s2[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        size_t i = 0;
        for (; i + 8 <= n; i += 8)
        {
            uint64_t w[8];
            std::memcpy(w, p + i, 64);
            for (int l = 0; l < 8; ++l)
            {
                s1[l] += w[l];
                s2[l] += s1[l];
            }
        }
        for (int l = 0; i < n; ++i, ++l)
        {
            uint64_t w;
            std::memcpy(&w, p + i, 8);
            s1[l] += w;
            s2[l] += s1[l];
        }
        uint64_t h = mix(1469598103934665603ull ^ seed, (uint64_t)n);
        for (int l = 0; l < 8; ++l)
            h = mix(mix(h, s1[l]), s2[l]);
        return h;
    }
    std::vector<Block> blocks;
    std::atomic<size_t> next{ 0 };
    // the FIRST job's blocks are the first of the list: a caller that needs that one fingerprint before the rest (spect.J, on
    // which the upload decision of a call hangs) takes part in them and waits only for them (wait_first)
    size_t nFirst = 0;
    std::atomic<size_t> doneFirst{ 0 };
    std::function<void()> worker;
    bool inFlight = false;
    size_t prepare()
    {
        blocks.clear();
        next = 0;
        doneFirst = 0;
        nFirst = jobs.empty() ? 0 : (jobs[0].n + BLOCK - 1) / BLOCK;
        size_t total = 0;
        for (size_t j = 0; j < jobs.size(); ++j)
            for (size_t off = 0; off < jobs[j].n; off += BLOCK)
            {
                blocks.push_back({ (int)j, off, std::min<size_t>(BLOCK, jobs[j].n - off), 0 });
                total += blocks.back().len;
            }
        return total;
    }
    void one(size_t b)
    {
        blocks[b].h = block_hash(jobs[blocks[b].job].p + blocks[b].off, blocks[b].len, blocks[b].off);
        if (b < nFirst)
            doneFirst.fetch_add(1, std::memory_order_release);
    }
    void work()
    {
        for (size_t b = next.fetch_add(1); b < blocks.size(); b = next.fetch_add(1))
            one(b);
    }
    // fingerprint of the first job as soon as ITS blocks are through (between start() and finish())
    uint64_t wait_first()
    {
        for (;;)
        {
            size_t b = next.load();
            if (b >= nFirst)
                break;
            if (next.compare_exchange_weak(b, b + 1))
                one(b);
        }
        while (doneFirst.load(std::memory_order_acquire) < nFirst)
            __builtin_ia32_pause();
        uint64_t h = mix((uint64_t)(uintptr_t)jobs[0].p, jobs[0].n);
        for (size_t b = 0; b < nFirst; ++b)
            h = mix(h, blocks[b].h);
        return h;
    }
    void fold()
    {
        for (auto& j : jobs)
            if (!j.sampled)
                j.h = mix((uint64_t)(uintptr_t)j.p, j.n);
        for (auto& b : blocks)
            jobs[b.job].h = mix(jobs[b.job].h, b.h);
    }
    void run()
    {
        start();
        finish();
    }
    // the pool threads begin on the blocks; the caller goes on with something else and joins in at finish()
    void start()
    {
        const size_t total = prepare();
        inFlight = total * sizeof(double) > (1u << 20) && HashPool::get().usable();
        if (inFlight)
        {
            worker = [this] { work(); };
            inFlight = HashPool::get().start(worker); // (false: the pool is busy, finish() does the blocks on this thread)
        }
    }
    void finish()
    {
        work();
        if (inFlight)
            HashPool::get().wait();
        inFlight = false;
        fold();
    }
    ~HashSet()
    {
        if (inFlight)
            HashPool::get().wait(); // (an exception between start and finish: the threads still read this object)
    }
    uint64_t operator[](int j) const { return jobs[j].h; }
};

struct PluginState
{
    lwhip_context* hip = nullptr;
    lwhip_problem prob{};
    std::vector<lwhip_atom> atoms;
    std::vector<std::vector<lwhip_transition>> trans;
    std::vector<uint64_t> shape;       // structural signature: recreate the device context if it changes
    // Ndim = 2: Atmosphere::intersections flattened into the tables of lwhip_grid2d
    lwhip_grid2d grid{};
    lwhip_boundary xLow{}, xUp{};      // fixed (CALLABLE) x boundaries of a 2D atmosphere
    std::vector<lwhip_intersection> g2uw, g2dw, g2sub;
    std::vector<int32_t> g2long, g2subOff;
    std::vector<double> height2d;
    uint64_t gridSig = 0, gridEpoch = 0; // geometry fingerprint; epoch enters the structural signature
    uint64_t fpProfiles = 0, fpBackground = 0, fpAtmos = 0, fpNstar = 0, fpCollisions = 0;
    // hybrid PRD: the tables configure_hprd_coeffs left in the Context (Source/Prd.cpp:697-946), flattened into lwhip_hprd;
    // rebuilt (and the device context with them) when the core re-allocates them (update_hprd_coeffs)
    lwhip_hprd hprd{};
    bool hprdOn = false;
    uint64_t hprdSig = 0, hprdEpoch = 0;
    std::vector<int32_t> hPrdIdxs, hHIdxs, hLineAtom, hLineTrans;
    std::vector<int64_t> hOff;
    std::vector<lwhip_j_coeff> hJc;
    std::vector<std::vector<lwhip_rho_coeff>> hRho;
    std::vector<const lwhip_rho_coeff*> hRhoPtr;
    bool prdDetailed = false;          // the device context was asked to include the detailed atoms' PRD lines
    double* zDown = nullptr;           // ZPlaneDecomposition outputs the device context currently writes
    double* zUp = nullptr;
    bool jDeviceAhead = false;         // LWHIP_RESIDENT_J: the device holds a newer J (JRest) than spect.J
    // Round 6: spect.J is an OUTPUT of the sweep itself (lwhip_map_host_J: page-locked, the kernel stores into it), and its
    // upload is skipped while nobody has rewritten it: fpJ is the fingerprint of what the device's J looks like in spect.J
    // (lwhip_fingerprint_J after every download, formed on the device), compared at the next call with an exact fingerprint
    // of the array as the host holds it then.  The contract stays the reference's: J is in spect.J when a call returns, and
    // whatever the caller writes there between two calls is what the next call starts from.
    bool jMapped = false;
    bool jSynced = false;
    uint64_t fpJ = 0;
    bool fpJHostFresh = false;         // fpJHost was formed in this call (sync_state, before the pool got the slow groups)
    uint64_t fpJHost = 0;
};

bool resident_J();
struct PluginState;
uint64_t host_J_fingerprint(PluginState* st);

std::mutex g_mutex;
std::map<const Atom*, std::pair<Context*, int>> g_atomOwner; // stat_eq receives only the Atom*

void register_atoms(Context* ctx)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    for (auto it = g_atomOwner.begin(); it != g_atomOwner.end();)
        it = (it->second.first == ctx) ? g_atomOwner.erase(it) : std::next(it);
    for (size_t ia = 0; ia < ctx->activeAtoms.size(); ++ia)
        g_atomOwner[ctx->activeAtoms[ia]] = { ctx, (int)ia };
}

// LWHIP_PLUGIN_TRACE=1: one line per hook on stderr (tools/oracle/hook_lifecycle.sh checks that they pair up)
void trace(const char* what, const void* ctx)
{
    static const bool on = std::getenv("LWHIP_PLUGIN_TRACE") != nullptr;
    if (on)
        std::fprintf(stderr, "lwhip-plugin-trace %s ctx=%p\n", what, ctx);
}

[[noreturn]] void fail(const std::string& what)
{
    throw std::runtime_error("lwhip plugin: " + what + ": " + lwhip_last_error());
}

int solver_id(const Context& ctx)
{
    const char* n = ctx.formalSolver.name;
    if (n && !std::strcmp(n, "piecewise_linear_1d"))
        return LWHIP_FS_LINEAR_1D;
    if (n && !std::strcmp(n, "piecewise_besser_1d"))
        return LWHIP_FS_BESSER_1D;
    if (n && !std::strcmp(n, "piecewise_bezier3_1d"))
        return LWHIP_FS_BEZIER3_1D;
    throw std::runtime_error(std::string("lwhip plugin: formal solver '") + (n ? n : "?")
                             + "' is not available on the device (1D linear/besser/bezier3, 2D besser only)");
}

lwhip_intersection flat_hit(const IntersectionResult& q)
{
    lwhip_intersection o{};
    o.axis = (int32_t)q.axis;
    o.fracZ = q.fractionalZ;
    o.fracX = q.fractionalX;
    o.distance = q.distance;
    return o;
}

// Ndim = 2: the x-periodic grid and the core's own intersection table (build_intersection_list,
// Source/FormalScalar2d.cpp:1188-1327; Atmosphere::intersections, Source/LwAtmosphere.hpp:145-173) -> lwhip_grid2d
void flatten_grid2d(Context& ctx, PluginState* st)
{
    Atmosphere& atmos = *ctx.atmos;
    const char* fs = ctx.formalSolver.name;
    const char* ip = ctx.interpFn.name;
    if (!fs || std::strcmp(fs, "piecewise_besser_2d") || !ip || std::strcmp(ip, "interp_linear_2d"))
        throw std::runtime_error("lwhip plugin: 2D needs piecewise_besser_2d with interp_linear_2d");
    const bool periodic = atmos.xLowerBc.type == PERIODIC && atmos.xUpperBc.type == PERIODIC;
    if (!periodic && !(atmos.xLowerBc.type == CALLABLE && atmos.xUpperBc.type == CALLABLE))
        throw std::runtime_error("lwhip plugin: mixed boundary types are not supported on the x axis"); // FormalScalar2d.cpp:755-759
    for (RadiationBc t : { atmos.zLowerBc.type, atmos.zUpperBc.type })
        if (t != ZERO && t != THERMALISED && t != CALLABLE)
            throw std::runtime_error("lwhip plugin: 2D z boundaries must be ZERO, THERMALISED or CALLABLE");
    if (!atmos.intersections)
        throw std::runtime_error("lwhip plugin: the atmosphere has no intersection table (build_intersection_list)");
    const int Nx = atmos.Nx, Nz = atmos.Nz, Nrays = atmos.Nrays;
    // the table is flattened once per geometry: it only changes when the grid or the quadrature does
    auto hashOf = [](const double* p, size_t n) { return HashSet::mix((uint64_t)(uintptr_t)p, HashSet::block_hash(p, n, n)); };
    uint64_t sig = hashOf(atmos.x.data, Nx) ^ (hashOf(atmos.z.data, Nz) * 3)
                   ^ (hashOf(atmos.mux.data, Nrays) * 5) ^ (hashOf(atmos.muz.data, Nrays) * 7)
                   ^ ((uint64_t)atmos.zLowerBc.type << 8) ^ ((uint64_t)atmos.zUpperBc.type << 16)
                   ^ ((uint64_t)atmos.intersections.substeps.size() << 24) ^ (periodic ? 1 : 2);
    // the x boundary data is borrowed like every other array: refreshed on every call, outside the cached table
    auto xbc = [](lwhip_boundary* b, AtmosphericBoundaryCondition& bc) {
        b->type = LWHIP_BC_CALLABLE;
        b->Nmu = (int32_t)bc.bcData.shape(1);
        b->idxs = bc.idxs.data;
        b->bcData = bc.bcData.data();
    };
    st->grid.xLowerBc = st->grid.xUpperBc = nullptr;
    if (!periodic)
    {
        xbc(&st->xLow, atmos.xLowerBc);
        xbc(&st->xUp, atmos.xUpperBc);
        st->grid.xLowerBc = &st->xLow;
        st->grid.xUpperBc = &st->xUp;
    }
    if (sig == st->gridSig && st->grid.Nx == Nx && st->grid.Nz == Nz && st->grid.Nrays == Nrays
        && st->grid.temperature == atmos.temperature.data)
        return;
    st->gridSig = sig;
    ++st->gridEpoch;
    const size_t nSt = (size_t)Nrays * 2 * Nz * Nx;
    st->g2uw.resize(nSt);
    st->g2dw.resize(nSt);
    st->g2long.resize(nSt);
    size_t o = 0;
    for (int mu = 0; mu < Nrays; ++mu)
        for (int d = 0; d < 2; ++d)
            for (int k = 0; k < Nz; ++k)
                for (int j = 0; j < Nx; ++j, ++o)
                {
                    const InterpolationStencil& q = atmos.intersections.intersections(mu, d, k, j);
                    st->g2uw[o] = flat_hit(q.uwIntersection);
                    st->g2dw[o] = flat_hit(q.dwIntersection);
                    st->g2long[o] = q.longCharIdx;
                }
    st->g2sub.clear();
    st->g2subOff.assign(1, 0);
    for (const SubstepIntersections& lc : atmos.intersections.substeps)
    {
        for (const IntersectionResult& q : lc.steps)
            st->g2sub.push_back(flat_hit(q));
        st->g2subOff.push_back((int32_t)st->g2sub.size());
    }
    if (st->g2sub.empty())
        st->g2sub.push_back(lwhip_intersection{});
    st->height2d.resize((size_t)Nx * Nz);
    for (int k = 0; k < Nz; ++k)
        for (int j = 0; j < Nx; ++j)
            st->height2d[(size_t)k * Nx + j] = atmos.z(k);
    lwhip_grid2d& g = st->grid;
    g = lwhip_grid2d{};
    g.Nx = Nx;
    g.Nz = Nz;
    g.Nrays = Nrays;
    g.periodic = periodic ? 1 : 0;
    g.zLowerBc = (int32_t)atmos.zLowerBc.type;
    g.zUpperBc = (int32_t)atmos.zUpperBc.type;
    g.NlongChar = (int32_t)st->g2subOff.size() - 1;
    g.x = atmos.x.data;
    g.z = atmos.z.data;
    g.mux = atmos.mux.data;
    g.muz = atmos.muz.data;
    g.temperature = atmos.temperature.data;
    g.uw = st->g2uw.data();
    g.dw = st->g2dw.data();
    g.longCharIdx = st->g2long.data();
    g.substepOff = st->g2subOff.data();
    g.substeps = st->g2sub.data();
    g.xLowerBc = periodic ? nullptr : &st->xLow;
    g.xUpperBc = periodic ? nullptr : &st->xUp;
}

void flatten_boundary(lwhip_boundary* b, AtmosphericBoundaryCondition& bc)
{
    b->type = (int32_t)bc.type;
    b->Nmu = 0;
    b->idxs = nullptr;
    b->bcData = nullptr;
    if (bc.type == CALLABLE)
    {
        b->Nmu = (int32_t)bc.bcData.shape(1);
        b->idxs = bc.idxs.data;
        b->bcData = bc.bcData.data();
    }
}

// Hybrid PRD: spect.prdIdxs / hPrdIdxs / JCoeffs / JRest and the PRD lines' hPrdCoeffs -> lwhip_hprd, once per
// configuration (the tables are large: 16 bytes per (line wavelength, ray, depth) and per JCoeffs entry).
void flatten_hprd(Context& ctx, PluginState* st)
{
    Spectrum& spect = *ctx.spect;
    st->hprdOn = (bool)spect.JRest;
    if (!st->hprdOn)
        return;
    std::vector<std::pair<int, Transition*>> lines; // (index into the flat atom list, line)
    const int Nactive = (int)ctx.activeAtoms.size();
    for (int ia = 0; ia < Nactive + (int)ctx.detailedAtoms.size(); ++ia)
    {
        Atom& a = ia < Nactive ? *ctx.activeAtoms[ia] : *ctx.detailedAtoms[ia - Nactive];
        for (auto* t : a.trans)
            if (t->rhoPrd && t->hPrdCoeffs)
                lines.push_back({ ia, t });
    }
    if (lines.empty())
        throw std::runtime_error("lwhip plugin: JRest without hybrid-PRD coefficients on any line");
    uint64_t sig = HashSet::mix((uint64_t)(uintptr_t)spect.JRest.data(), (uint64_t)spect.hPrdIdxs.size());
    sig = HashSet::mix(sig, (uint64_t)(uintptr_t)spect.JCoeffs.dataStore.data());
    for (auto& l : lines)
        sig = HashSet::mix(sig, (uint64_t)(uintptr_t)l.second->hPrdCoeffs.data);
    if (sig == st->hprdSig && st->hprd.JRest == spect.JRest.data())
        return;
    st->hprdSig = sig;
    ++st->hprdEpoch;
    const int Ns = ctx.atmos->Nspace, Nrays = ctx.atmos->Nrays;
    st->hPrdIdxs.assign(spect.prdIdxs.begin(), spect.prdIdxs.end());
    st->hHIdxs.assign(spect.hPrdIdxs.begin(), spect.hPrdIdxs.end());
    const int NhPrd = (int)st->hHIdxs.size();
    st->hOff.clear();
    st->hJc.clear();
    st->hOff.reserve((size_t)NhPrd * Nrays * 2 * Ns + 1);
    for (int hq = 0; hq < NhPrd; ++hq)
        for (int mu = 0; mu < Nrays; ++mu)
            for (int toObs = 0; toObs < 2; ++toObs)
                for (int k = 0; k < Ns; ++k)
                {
                    st->hOff.push_back((int64_t)st->hJc.size());
                    for (const auto& c : spect.JCoeffs(hq, mu, toObs, k))
                        st->hJc.push_back(lwhip_j_coeff{ c.frac, c.idx, 0 });
                }
    st->hOff.push_back((int64_t)st->hJc.size());
    st->hLineAtom.clear();
    st->hLineTrans.clear();
    st->hRho.clear();
    for (auto& l : lines)
    {
        Atom& a = l.first < Nactive ? *ctx.activeAtoms[l.first] : *ctx.detailedAtoms[l.first - Nactive];
        int kr = 0;
        for (; kr < a.Ntrans && a.trans[kr] != l.second; ++kr)
            ;
        st->hLineAtom.push_back(l.first);
        st->hLineTrans.push_back(kr);
        Transition& t = *l.second;
        const int nlt = (int)t.wavelength.shape(0);
        std::vector<lwhip_rho_coeff> v((size_t)nlt * Nrays * 2 * Ns);
        for (int lt = 0; lt < nlt; ++lt)
            for (int mu = 0; mu < Nrays; ++mu)
                for (int toObs = 0; toObs < 2; ++toObs)
                    for (int k = 0; k < Ns; ++k)
                    {
                        const auto& c = t.hPrdCoeffs(lt, mu, toObs, k);
                        v[(((size_t)lt * Nrays + mu) * 2 + toObs) * Ns + k] = lwhip_rho_coeff{ c.i0, c.i1, c.frac };
                    }
        st->hRho.push_back(std::move(v));
    }
    st->hRhoPtr.clear();
    for (auto& v : st->hRho)
        st->hRhoPtr.push_back(v.data());
    lwhip_hprd& h = st->hprd;
    h = lwhip_hprd{};
    h.NprdLambda = (int32_t)st->hPrdIdxs.size();
    h.NhPrd = NhPrd;
    h.Nlines = (int32_t)st->hLineAtom.size();
    h.prdIdxs = st->hPrdIdxs.data();
    h.hPrdIdxs = st->hHIdxs.data();
    h.JRest = spect.JRest.data();
    h.jCoeffOff = st->hOff.data();
    h.jCoeffs = st->hJc.data();
    h.lineAtom = st->hLineAtom.data();
    h.lineTrans = st->hLineTrans.data();
    h.rhoCoeffs = st->hRhoPtr.data();
}

// Context -> lwhip_problem (flat restatement, include/lwhip.h); returns the structural signature
// 2D tables live in `keep` (the Context's long-lived state), everything else in `st`
void flatten(Context& ctx, PluginState* st, PluginState* keep)
{
    Atmosphere& atmos = *ctx.atmos;
    Spectrum& spect = *ctx.spect;
    if (atmos.Ndim != 1 && atmos.Ndim != 2)
        throw std::runtime_error("lwhip plugin: only 1D plane-parallel and 2D x-periodic atmospheres are supported");
    const bool is2d = atmos.Ndim == 2;
    flatten_hprd(ctx, keep); // hybrid PRD tables (cached in the Context's long-lived state)
    lwhip_problem& p = st->prob;
    p = lwhip_problem{};
    p.abiVersion = LWHIP_ABI_VERSION;
    p.Nspace = atmos.Nspace;
    p.Nrays = atmos.Nrays;
    p.Nlambda = (int32_t)spect.wavelength.shape(0);
    p.formalSolver = is2d ? LWHIP_FS_BESSER_1D : solver_id(ctx); // 2D: the solver is fixed by grid2d
    p.storeDepthData = (ctx.depthData && ctx.depthData->fill) ? 1 : 0;
    if (is2d)
    {
        flatten_grid2d(ctx, keep);
        p.grid2d = &keep->grid;
    }
    p.height = is2d ? keep->height2d.data() : atmos.height.data;
    p.temperature = atmos.temperature.data;
    p.vlosMu = atmos.vlosMu.data;
    p.muz = atmos.muz.data;
    p.wmu = atmos.wmu.data;
    p.wavelength = spect.wavelength.data;
    flatten_boundary(&p.zLowerBc, atmos.zLowerBc);
    flatten_boundary(&p.zUpperBc, atmos.zUpperBc);
    p.bgChi = ctx.background->chi.data;
    p.bgEta = ctx.background->eta.data;
    p.bgSca = ctx.background->sca.data;
    p.J = spect.J.data;
    p.I = spect.I.data;
    if (p.storeDepthData)
    {
        p.depthChi = ctx.depthData->chi.data;
        p.depthEta = ctx.depthData->eta.data;
        p.depthI = ctx.depthData->I.data;
    }
    std::vector<Atom*> all(ctx.activeAtoms);
    all.insert(all.end(), ctx.detailedAtoms.begin(), ctx.detailedAtoms.end());
    st->atoms.assign(all.size(), lwhip_atom{});
    st->trans.assign(all.size(), {});
    st->shape = { (uint64_t)p.Nspace, (uint64_t)p.Nrays, (uint64_t)p.Nlambda, (uint64_t)p.formalSolver,
                  (uint64_t)p.storeDepthData, (uint64_t)p.zLowerBc.type, (uint64_t)p.zUpperBc.type,
                  (uint64_t)all.size(), (uint64_t)(is2d ? atmos.Nx : 0), (uint64_t)(is2d ? keep->gridEpoch : 0),
                  (uint64_t)keep->prdDetailed, (uint64_t)(keep->hprdOn ? keep->hprdEpoch + 1 : 0) };
    for (size_t ia = 0; ia < all.size(); ++ia)
    {
        Atom& a = *all[ia];
        lwhip_atom& fa = st->atoms[ia];
        fa.Nlevel = a.Nlevel;
        fa.Ntrans = a.Ntrans;
        fa.detailed = ia >= ctx.activeAtoms.size();
        fa.n = a.n.data;
        fa.nStar = a.nStar.data;
        fa.nTotal = a.nTotal.data;
        fa.vBroad = a.vBroad.data;
        fa.Gamma = fa.detailed ? nullptr : a.Gamma.data;
        fa.C = (fa.detailed || !a.C) ? nullptr : a.C.data;
        st->trans[ia].assign(a.Ntrans, lwhip_transition{});
        for (int kr = 0; kr < a.Ntrans; ++kr)
        {
            Transition& t = *a.trans[kr];
            lwhip_transition& ft = st->trans[ia][kr];
            ft.type = (t.type == LINE) ? LWHIP_LINE : LWHIP_CONTINUUM;
            ft.i = t.i;
            ft.j = t.j;
            ft.Nblue = t.Nblue;
            ft.Nred = t.Nred;
            ft.prd = (t.type == LINE && t.rhoPrd) ? 1 : 0;
            ft.Aji = t.Aji;
            ft.Bji = t.Bji;
            ft.Bij = t.Bij;
            ft.lambda0 = t.lambda0;
            ft.dopplerWidth = t.dopplerWidth;
            ft.wavelength = t.wavelength.data;
            ft.alpha = t.alpha.data;
            ft.phi = t.phi.data;
            ft.wphi = t.wphi.data;
            ft.aDamp = t.aDamp.data;
            ft.Qelast = t.Qelast.data;
            ft.rhoPrd = t.rhoPrd.data;
            ft.Rij = t.Rij.data;
            ft.Rji = t.Rji.data;
            for (uint64_t v : { (uint64_t)ft.type, (uint64_t)ft.i, (uint64_t)ft.j, (uint64_t)ft.Nblue,
                                (uint64_t)ft.Nred, (uint64_t)ft.prd })
                st->shape.push_back(v);
        }
        fa.trans = st->trans[ia].data();
        st->shape.push_back((uint64_t)fa.Nlevel);
        st->shape.push_back((uint64_t)fa.detailed);
    }
    p.Natom = (int32_t)all.size();
    p.atoms = st->atoms.data();
}

// fingerprints of the slowly changing inputs -> upload mask.  In two halves, so that fs_iter can let the pool threads work
// through the 90 MB while it uploads the per-call groups and runs the kernels (the common case: nothing changed).
// the input groups tracked by fingerprint (the others go up on every call)
constexpr uint32_t kSlowGroups = LWHIP_PROFILES | LWHIP_BACKGROUND | LWHIP_ATMOS | LWHIP_NSTAR | LWHIP_COLLISIONS;

struct Prints
{
    HashSet hs;
    std::vector<int> jProf, jNstar, jColl, jBg, jAt;
    int jJ = -1;                   // spect.J's own job (first in the list), or -1
    uint32_t groups = kSlowGroups; // the groups looked at: a population update reads only two small ones and leaves the large
                                   // ones (profiles, background: 0.4 ms to read) to the next formal solution, unrecorded
    bool active = false;
};

void prints_begin(PluginState* st, Prints& pr, uint32_t groups = kSlowGroups, bool withJ = false)
{
    pr.groups = groups;
    const lwhip_problem& p = st->prob;
    const size_t Ns = p.Nspace, Nla = p.Nlambda, Nr = p.Nrays;
    HashSet& hs = pr.hs;
    pr.jJ = withJ ? hs.add(p.J, Nla * Ns) : -1; // (the first job: HashSet::wait_first)
    const char* smp = std::getenv("LWHIP_SAMPLED_FINGERPRINTS");
    const bool sampled = smp && smp[0] == '1';
    for (int ia = 0; ia < p.Natom; ++ia)
    {
        const lwhip_atom& a = p.atoms[ia];
        if (groups & LWHIP_NSTAR)
        {
            pr.jNstar.push_back(hs.add(a.nStar, (size_t)a.Nlevel * Ns));
            pr.jNstar.push_back(hs.add(a.nTotal, Ns));
            pr.jNstar.push_back(hs.add(a.vBroad, Ns));
        }
        if (groups & LWHIP_COLLISIONS)
            pr.jColl.push_back(hs.add(a.C, (size_t)a.Nlevel * a.Nlevel * Ns));
        for (int kr = 0; kr < a.Ntrans; ++kr)
        {
            const lwhip_transition& t = a.trans[kr];
            if (t.type != LWHIP_LINE)
                continue;
            if (groups & LWHIP_NSTAR)
            {
                pr.jNstar.push_back(hs.add(t.aDamp, Ns));
                pr.jNstar.push_back(hs.add(t.Qelast, Ns));
            }
            if (groups & LWHIP_PROFILES)
            {
                pr.jProf.push_back(sampled ? hs.add_sampled(t.phi, (size_t)(t.Nred - t.Nblue) * Nr * 2 * Ns)
                                           : hs.add(t.phi, (size_t)(t.Nred - t.Nblue) * Nr * 2 * Ns));
                pr.jProf.push_back(hs.add(t.wphi, Ns));
            }
        }
    }
    if (!(groups & LWHIP_BACKGROUND))
        pr.jBg.clear();
    else if (sampled)
        pr.jBg = { hs.add_sampled(p.bgChi, Nla * Ns), hs.add_sampled(p.bgEta, Nla * Ns), hs.add_sampled(p.bgSca, Nla * Ns) };
    else
        pr.jBg = { hs.add(p.bgChi, Nla * Ns), hs.add(p.bgEta, Nla * Ns), hs.add(p.bgSca, Nla * Ns) };
    if (groups & LWHIP_ATMOS)
        pr.jAt = { hs.add(p.height, Ns), hs.add(p.temperature, Ns), hs.add(p.wavelength, Nla), hs.add(p.muz, Nr), hs.add(p.wmu, Nr),
                   hs.add(p.vlosMu, p.vlosMu ? Nr * Ns : 0) };
    hs.start();
    pr.active = true;
}

uint32_t prints_finish(PluginState* st, Prints& pr, bool force)
{
    HashSet& hs = pr.hs;
    hs.finish();
    pr.active = false;
    auto fold = [&](const std::vector<int>& js) {
        uint64_t h = js.size();
        for (int j : js)
            h = HashSet::mix(h, hs[j]);
        return h;
    };
    const uint64_t fProf = fold(pr.jProf), fNstar = fold(pr.jNstar), fColl = fold(pr.jColl), fBg = fold(pr.jBg), fAt = fold(pr.jAt);
    uint32_t mask = LWHIP_GAMMA | LWHIP_J | LWHIP_POPS | LWHIP_BC | LWHIP_RHOPRD;
    auto look = [&](uint32_t group, uint64_t f, uint64_t& recorded) {
        if (!(pr.groups & group))
            return; // not looked at: whoever looks next compares against what was recorded before
        if (force || f != recorded)
            mask |= group;
        recorded = f;
    };
    look(LWHIP_PROFILES, fProf, st->fpProfiles);
    look(LWHIP_BACKGROUND, fBg, st->fpBackground);
    look(LWHIP_ATMOS, fAt, st->fpAtmos);
    look(LWHIP_NSTAR, fNstar, st->fpNstar);
    look(LWHIP_COLLISIONS, fColl, st->fpCollisions);
    return mask;
}

uint32_t dirty_mask(PluginState* st, bool force, uint32_t groups = kSlowGroups)
{
    Prints pr;
    prints_begin(st, pr, groups);
    return prints_finish(st, pr, force);
}

void destroy_state(Context* ctx)
{
    auto* st = (PluginState*)ctx->methodScratch;
    if (!st)
        return;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        for (auto it = g_atomOwner.begin(); it != g_atomOwner.end();)
            it = (it->second.first == ctx) ? g_atomOwner.erase(it) : std::next(it);
    }
    if (st->hip)
        lwhip_destroy(st->hip);
    delete st;
    ctx->methodScratch = nullptr;
}

// make sure a device context matching the current Context exists; returns the upload mask
// `later`: the fingerprints of the slow groups are only started (prints_begin); the returned mask holds the per-call groups and
// the caller asks prints_finish for the rest when it has nothing better to do (later->active tells whether there is a rest)
uint32_t sync_state(Context& ctx, bool force, Prints* later = nullptr, uint32_t groups = kSlowGroups)
{
    auto* st = (PluginState*)ctx.methodScratch;
    if (!st)
    {
        st = new PluginState();
        ctx.methodScratch = st;
    }
    PluginState fresh;
    flatten(ctx, &fresh, st);
    const bool rebuild = !st->hip || fresh.shape != st->shape;
    // adopt the fresh descriptor (pointers may have moved even when the structure has not)
    const bool moved = !rebuild && std::memcmp(&fresh.prob, &st->prob, offsetof(lwhip_problem, atoms)) != 0;
    lwhip_context* old = st->hip;
    st->prob = fresh.prob;
    st->atoms = std::move(fresh.atoms);
    st->trans = std::move(fresh.trans);
    st->shape = std::move(fresh.shape);
    for (size_t ia = 0; ia < st->atoms.size(); ++ia)
        st->atoms[ia].trans = st->trans[ia].data();
    st->prob.atoms = st->atoms.data();
    if (rebuild || moved)
    {
        if (old)
            lwhip_destroy(old);
        st->hip = nullptr;
        st->zDown = st->zUp = nullptr; // a fresh device context writes no z-plane outputs yet
        st->jDeviceAhead = false;      // (LWHIP_RESIDENT_J: a rebuilt context starts from spect.J again)
        lwhip_options opts{};
        opts.flags = st->prdDetailed ? LWHIP_OPT_PRD_DETAILED : 0;
        // Gamma and the rates are summed in a FIXED order by default (round 5): bit-reproducible iterations, as the reference's
        // schemes are -- the same iteration count from run to run.  The lane sweep's fixed-order mode keeps four wavefronts per
        // workgroup now (they add to the accumulators in turn) and costs ~1.1 x the atomic path (it was 2.7 x).
        // LWHIP_DETERMINISTIC=0 selects the atomic sums; LWHIP_DETERMINISTIC=1 asks for the fixed order also where it is
        // expensive (hybrid PRD: one wavefront per workgroup); a forced march (LWHIP_SWEEP=march) has no such mode.
        {
            const char* det = std::getenv("LWHIP_DETERMINISTIC");
            const char* sw = std::getenv("LWHIP_SWEEP");
            const bool marchForced = sw && std::strcmp(sw, "march") == 0;
            if (det ? det[0] == '1' : (!st->hprdOn && !marchForced))
                opts.flags |= LWHIP_OPT_DETERMINISTIC;
        }
        opts.hprd = st->hprdOn ? &st->hprd : nullptr;
        if (lwhip_create(&st->prob, &opts, &st->hip) != LWHIP_OK)
            fail("lwhip_create");
        {
            // (LWHIP_MAP_HOST_J=0: J comes back by a copy, as before round 6)
            const char* mj = std::getenv("LWHIP_MAP_HOST_J");
            // (hybrid PRD keeps the copies: JRest travels with J and a second run takes both from the host again)
            st->jMapped = !(mj && mj[0] == '0') && !resident_J() && !st->hprdOn && lwhip_map_host_J(st->hip, 1) == LWHIP_OK;
            st->jSynced = false;
        }
        dirty_mask(st, true); // create uploaded everything: just record the fingerprints
        register_atoms(&ctx);
        return LWHIP_GAMMA | LWHIP_J | LWHIP_POPS | LWHIP_BC | LWHIP_RHOPRD;
    }
    // (spect.J's own fingerprint first, with the whole pool: once the pool works on the 90 MB of the slow groups a second
    // round would be hashed by this thread alone, ten times slower)
    st->fpJHostFresh = false;
    const bool withJ = st->jSynced && !force && !st->jDeviceAhead && !resident_J() && !st->hprdOn && groups == kSlowGroups;
    if (later)
    {
        // (spect.J's blocks go first in the pool's round and this thread waits for them alone: the upload decision hangs on
        // that one fingerprint, the 90 MB of the slow groups are read while the uploads and the kernels run)
        prints_begin(st, *later, kSlowGroups, withJ);
        if (withJ)
        {
            st->fpJHost = later->hs.wait_first();
            st->fpJHostFresh = true;
        }
        return LWHIP_GAMMA | LWHIP_J | LWHIP_POPS | LWHIP_BC | LWHIP_RHOPRD;
    }
    return dirty_mask(st, force, force ? kSlowGroups : groups);
}


bool force_upload(ExtraParams& params)
{
    return params.contains("lwhip_force_upload");
}

// ExtraParams "ZPlaneDecomposition" with "ZPlaneDown" / "ZPlaneUp" (Source/SimdFullIterationTemplates.hpp:253-281): the
// caller's arrays become the library's z-plane outputs for this call (filled by the download of I)
void apply_zplane(Context& ctx, PluginState* st, ExtraParams& params)
{
    double* down = nullptr;
    double* up = nullptr;
    if (params.contains("ZPlaneDecomposition"))
    {
        const bool is2d = ctx.atmos->Ndim == 2;
        if (params.contains("ZPlaneDown"))
            down = is2d ? params.get_as<F64View3D>("ZPlaneDown").data : params.get_as<F64View2D>("ZPlaneDown").data;
        if (params.contains("ZPlaneUp"))
            up = is2d ? params.get_as<F64View3D>("ZPlaneUp").data : params.get_as<F64View2D>("ZPlaneUp").data;
    }
    if (down != st->zDown || up != st->zUp)
    {
        if (lwhip_set_zplane_outputs(st->hip, down, up) != LWHIP_OK)
            fail("lwhip_set_zplane_outputs");
        st->zDown = down;
        st->zUp = up;
    }
}

// LWHIP_RESIDENT_J=1 (read at every call): J stays in HBM between the calls -- 6.7 MB each way at the bench size, three
// quarters of what the host-authoritative call adds to the kernels.  While it is set, spect.J (and JRest)
// is neither uploaded nor written back: the caller promises not to rewrite it; the first call made without it brings the
// device's J down before anything is uploaded, so the host is authoritative again from there.  Not the reference's contract
// (every scheme leaves J in spect.J when it returns), hence opt-in.
bool resident_J()
{
    const char* e = std::getenv("LWHIP_RESIDENT_J");
    return e && e[0] == '1';
}

// the upload mask of a call under that policy (before lwhip_upload); `fresh`: the device context was just (re)created
// exact fingerprint of spect.J as the host holds it now (the pool's threads; 6.7 MB at the bench size: ~0.03 ms)
uint64_t host_J_fingerprint(PluginState* st)
{
    HashSet hs;
    const int j = hs.add(st->prob.J, (size_t)st->prob.Nlambda * st->prob.Nspace);
    hs.run();
    return hs[j];
}

uint32_t j_policy_upload(PluginState* st, uint32_t up, bool force = false)
{
    if (st->jMapped && resident_J())
    {
        // (LWHIP_RESIDENT_J promises that spect.J is not written while it is set: the sweep must not store into it)
        if (lwhip_map_host_J(st->hip, 0) != LWHIP_OK)
            fail("lwhip_map_host_J");
        st->jMapped = false;
    }
    // (not with hybrid PRD: JRest travels in the same group and is not covered by J's fingerprint)
    if ((up & LWHIP_J) && st->jSynced && !force && !st->jDeviceAhead && !st->hprdOn
        && (st->fpJHostFresh ? st->fpJHost : host_J_fingerprint(st)) == st->fpJ)
        up &= ~(uint32_t)LWHIP_J; // nobody has rewritten spect.J since the device produced it: the device's copy is the same
    st->fpJHostFresh = false;
    if (st->jDeviceAhead)
    {
        if (!resident_J())
        {
            if (lwhip_download(st->hip, LWHIP_J) != LWHIP_OK) // spect.J catches up first
                fail("lwhip_download");
            st->jDeviceAhead = false;
        }
        up &= ~(uint32_t)LWHIP_J; // the device's J is the current one either way
    }
    return up;
}

// ... and the download mask after it
uint32_t j_policy_download(PluginState* st, uint32_t down)
{
    if ((down & LWHIP_J) && resident_J())
    {
        down &= ~(uint32_t)LWHIP_J;
        st->jDeviceAhead = true;
    }
    return down;
}

// after a download that included J: what spect.J holds now, by the device's account
void j_record_synced(PluginState* st, uint32_t down)
{
    st->jSynced = false;
    if (!(down & LWHIP_J) || st->prob.grid2d)
        return;
    if (lwhip_fingerprint_J(st->hip, st->prob.J, &st->fpJ) == LWHIP_OK)
        st->jSynced = true;
}

// ---- FsIterationFns members ----------------------------------------------------------------------

// fs_iter: replaces formal_sol_iteration_matrices_<simd> (Source/SimdFullIterationTemplates.hpp:588-719)
IterationResult fs_iter_hip(Context& ctx, bool lambdaIterate, ExtraParams params)
{
    trace("fs_iter", &ctx);
    // LWHIP_PLUGIN_TIMING=1: where a call's host time goes (stderr)
    static const bool timing = std::getenv("LWHIP_PLUGIN_TIMING") != nullptr;
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    auto lap = [&](const char* what, clk::time_point& prev) {
        if (!timing)
            return;
        const auto t = clk::now();
        std::fprintf(stderr, "  fs_iter: %-22s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - prev).count());
        prev = t;
    };
    auto tPrev = t0;
    // The slow groups (profiles, background, atmosphere, ...) almost never change between two calls, but finding that out means
    // reading 90 MB of the caller's arrays (0.4 ms at the bench size).  So the pool threads fingerprint them WHILE this thread
    // uploads the per-call groups and runs the kernels as if nothing had changed; if something had, the changed groups and the
    // per-call groups go up (again: J was overwritten) and the kernels run a second time.  Same results either way.  Not with
    // LWHIP_RESIDENT_J (the J the first run started from is gone) and not with a forced upload.
    const bool force = force_upload(params);
    Prints prints;
    const bool speculate = !force && !resident_J() && ((PluginState*)ctx.methodScratch != nullptr)
                           && !((PluginState*)ctx.methodScratch)->jDeviceAhead;
    uint32_t up = sync_state(ctx, force, speculate ? &prints : nullptr);
    lap(speculate ? "flatten" : "flatten + fingerprints", tPrev);
    auto* st = (PluginState*)ctx.methodScratch;
    apply_zplane(ctx, st, params);
    // a Context with one thread runs the reference's single-thread loop, whose dJMaxIdx is NOT the arg-max (max_idx's
    // argument order, SimdFullIterationTemplates.hpp:627): report what that loop reports
    if (lwhip_set_djmax_index_mode(st->hip, ctx.Nthreads <= 1 ? 1 : 0) != LWHIP_OK)
        fail("lwhip_set_djmax_index_mode");
    lwhip_iter_result r{};
    for (int attempt = 0; attempt < 2; ++attempt)
    {
        // (the second attempt: the first run has overwritten the device's J -- and spect.J itself when the array is mapped --,
        // so the J this call started from comes back from the device's snapshot, not from the host)
        const bool restoreJ = attempt == 1 && !st->hprdOn;
        if (restoreJ && lwhip_j_restore(st->hip) != LWHIP_OK)
            fail("lwhip_j_restore");
        if (lwhip_upload(st->hip, restoreJ ? (up & ~(uint32_t)LWHIP_J) : j_policy_upload(st, up, force || attempt == 1)) != LWHIP_OK)
            fail("lwhip_upload");
        if (attempt == 0 && prints.active && !st->hprdOn && lwhip_j_snapshot(st->hip) != LWHIP_OK)
            fail("lwhip_j_snapshot");
        lap("upload", tPrev);
        if (lwhip_formal_sol_gamma_matrices(st->hip, lambdaIterate ? 1 : 0, &r) != LWHIP_OK)
            fail("lwhip_formal_sol_gamma_matrices");
        lap("kernels", tPrev);
        if (!prints.active)
            break;
        const uint32_t all = prints_finish(st, prints, false);
        lap("fingerprints (rest)", tPrev);
        if (!(all & kSlowGroups))
            break;
        up = all; // something had changed: the changed groups and the per-call groups (again), then the kernels again
    }
    uint32_t down = LWHIP_J | LWHIP_I | LWHIP_GAMMA | LWHIP_RATES;
    if (st->prob.storeDepthData)
        down |= LWHIP_DEPTHDATA;
    down = j_policy_download(st, down);
    if (lwhip_download(st->hip, down) != LWHIP_OK)
        fail("lwhip_download");
    j_record_synced(st, down);
    lap("download", tPrev);
    IterationResult result{};
    result.updatedJ = true;
    result.dJMax = r.dJMax;
    result.dJMaxIdx = r.dJMaxIdx;
    return result;
}

// simple_fs: replaces formal_sol_<simd> (Source/SimdFullIterationTemplates.hpp:721-781)
IterationResult simple_fs_hip(Context& ctx, bool upOnly, ExtraParams params)
{
    // (as fs_iter: the slow groups are fingerprinted while the formal solution runs on what the device holds; it changes none of
    // its inputs, so the second run, should something have changed, needs no more than the changed groups)
    const bool force = force_upload(params);
    Prints prints;
    uint32_t up = sync_state(ctx, force, (!force && ctx.methodScratch) ? &prints : nullptr);
    auto* st = (PluginState*)ctx.methodScratch;
    apply_zplane(ctx, st, params);
    for (int attempt = 0; attempt < 2; ++attempt)
    {
        if (lwhip_upload(st->hip, j_policy_upload(st, up & ~(uint32_t)LWHIP_GAMMA, force)) != LWHIP_OK)
            fail("lwhip_upload");
        if (lwhip_formal_sol(st->hip, upOnly ? 1 : 0) != LWHIP_OK)
            fail("lwhip_formal_sol");
        if (!prints.active)
            break;
        up = prints_finish(st, prints, false) & kSlowGroups;
        if (!up)
            break;
    }
    if (lwhip_download(st->hip, LWHIP_I) != LWHIP_OK)
        fail("lwhip_download");
    return IterationResult{};
}

// spaceStart / spaceEnd of the three population updates: set on the device context for the call, back to the whole
// atmosphere when it returns or throws
struct DepthRange
{
    lwhip_context* h;
    DepthRange(lwhip_context* hip, int spaceStart, int spaceEnd) : h(hip)
    {
        if (lwhip_set_depth_range(h, spaceStart, spaceEnd) != LWHIP_OK)
            fail("lwhip_set_depth_range");
    }
    ~DepthRange() { (void)lwhip_set_depth_range(h, -1, -1); }
    DepthRange(const DepthRange&) = delete;
    DepthRange& operator=(const DepthRange&) = delete;
};

// stat_eq: replaces stat_eq_impl (Source/UpdatePopulations.cpp:7-47)
void stat_eq_hip(Atom* atom, ExtraParams params, int spaceStart, int spaceEnd)
{
    Context* ctx = nullptr;
    int idx = -1;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        auto it = g_atomOwner.find(atom);
        if (it != g_atomOwner.end())
        {
            ctx = it->second.first;
            idx = it->second.second;
        }
    }
    if (!ctx)
        throw std::runtime_error("lwhip plugin: stat_eq needs an atom of a live Context");
    // whatever else changed since the last call goes up too: sync_state has just recorded its fingerprint as clean
    const uint32_t up = sync_state(*ctx, force_upload(params), nullptr, LWHIP_NSTAR) & kSlowGroups;
    auto* st = (PluginState*)ctx->methodScratch;
    if (lwhip_upload(st->hip, up | LWHIP_GAMMA | LWHIP_POPS) != LWHIP_OK)
        fail("lwhip_upload");
    const DepthRange range(st->hip, spaceStart, spaceEnd); // [spaceStart, spaceEnd) only (:22)
    const int rc = lwhip_stat_equil(st->hip, idx);
    if (rc == LWHIP_ERR_SINGULAR)
        throw std::runtime_error("Singular Matrix"); // Source/LuSolve.cpp:22-23 -> ExplodingMatrixError
    if (rc != LWHIP_OK)
        fail("lwhip_stat_equil");
    if (lwhip_download(st->hip, LWHIP_POPS) != LWHIP_OK)
        fail("lwhip_download");
}

// time_dep_update: replaces time_dependent_update_impl (Source/UpdatePopulations.cpp:120-151)
void time_dep_update_hip(Atom* atom, F64View2D nOld, f64 dt, ExtraParams params, int spaceStart, int spaceEnd)
{
    Context* ctx = nullptr;
    int idx = -1;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        auto it = g_atomOwner.find(atom);
        if (it != g_atomOwner.end())
        {
            ctx = it->second.first;
            idx = it->second.second;
        }
    }
    if (!ctx)
        throw std::runtime_error("lwhip plugin: time_dep_update needs an atom of a live Context");
    const uint32_t up = sync_state(*ctx, force_upload(params), nullptr, LWHIP_NSTAR) & kSlowGroups;
    auto* st = (PluginState*)ctx->methodScratch;
    if (lwhip_upload(st->hip, up | LWHIP_GAMMA | LWHIP_POPS) != LWHIP_OK)
        fail("lwhip_upload");
    const DepthRange range(st->hip, spaceStart, spaceEnd); // (:135)
    const int rc = lwhip_time_dep_update(st->hip, idx, nOld.data, dt);
    if (rc == LWHIP_ERR_SINGULAR)
        throw std::runtime_error("Singular Matrix");
    if (rc != LWHIP_OK)
        fail("lwhip_time_dep_update");
    if (lwhip_download(st->hip, LWHIP_POPS) != LWHIP_OK)
        fail("lwhip_download");
}

// nr_post_update: replaces nr_post_update_impl (Source/UpdatePopulations.cpp:294-394)
void nr_post_update_hip(Context& ctx, std::vector<Atom*>* atoms, const std::vector<F64View3D>& dC,
                        F64View backgroundNe, const NrTimeDependentData& timeDepData, f64 crswVal,
                        ExtraParams params, int spaceStart, int spaceEnd)
{
    const uint32_t up = sync_state(ctx, force_upload(params), nullptr, LWHIP_NSTAR | LWHIP_COLLISIONS) & kSlowGroups;
    auto* st = (PluginState*)ctx.methodScratch;
    const DepthRange range(st->hip, spaceStart, spaceEnd); // (:316)
    if (lwhip_upload(st->hip, up | LWHIP_GAMMA | LWHIP_POPS | LWHIP_COLLISIONS) != LWHIP_OK)
        fail("lwhip_upload");
    const int n = (int)atoms->size();
    std::vector<int32_t> idx(n);
    std::vector<const double*> stages(n), dCp(n), prev(n);
    for (int q = 0; q < n; ++q)
    {
        auto pos = std::find(ctx.activeAtoms.begin(), ctx.activeAtoms.end(), (*atoms)[q]);
        if (pos == ctx.activeAtoms.end())
            throw std::runtime_error("lwhip plugin: nr_post_update atom is not an active atom of this Context");
        idx[q] = (int32_t)(pos - ctx.activeAtoms.begin());
        stages[q] = (*atoms)[q]->stages.data;
        if (!dC.empty())
            dCp[q] = dC[q].data;
        if (!timeDepData.nPrev.empty())
            prev[q] = timeDepData.nPrev[q].data;
    }
    lwhip_nr_args a{};
    a.Natoms = n;
    a.atoms = idx.data();
    a.stages = stages.data();
    a.dC = dC.empty() ? nullptr : dCp.data();
    a.nPrev = timeDepData.nPrev.empty() ? nullptr : prev.data();
    a.backgroundNe = backgroundNe.data;
    a.ne = ctx.atmos->ne.data;
    a.dt = timeDepData.dt;
    a.crsw = crswVal;
    const int rc = lwhip_nr_post_update(st->hip, &a);
    if (rc == LWHIP_ERR_SINGULAR)
        throw std::runtime_error("Singular Matrix");
    if (rc != LWHIP_OK)
        fail("lwhip_nr_post_update");
    if (lwhip_download(st->hip, LWHIP_POPS) != LWHIP_OK)
        fail("lwhip_download");
}

// redistribute_prd: replaces redistribute_prd_lines_template<SimdType> (Source/PrdTemplates.hpp:175-290)
IterationResult redistribute_prd_hip(Context& ctx, int maxIter, f64 tol, ExtraParams params)
{
    // include_detailed_atoms (the default of LwContext.prd_redistribute, Source/LwMiddleLayer.pyx:3678-3680): a property
    // of the device context -- its PRD line list and the wavelengths of its PRD rates pass -- so a change of the
    // effective value rebuilds the context (it is part of the structural signature)
    bool includeDetailed = params.contains("include_detailed_atoms") && params.get_as<bool>("include_detailed_atoms");
    if (includeDetailed)
    {
        includeDetailed = false;
        for (auto* a : ctx.detailedAtoms)
            for (auto* t : a->trans)
                if (t->rhoPrd)
                    includeDetailed = true;
    }
    if (!ctx.methodScratch)
        ctx.methodScratch = new PluginState();
    ((PluginState*)ctx.methodScratch)->prdDetailed = includeDetailed;
    // (as fs_iter: the fingerprints of the slow groups run beside the sub-iterations.  Everything the sub-iterations change on
    // the device -- rho, J and I of the PRD wavelengths, the PRD lines' rates -- goes up from the host at the start of every
    // call, so a second run after a detected change is a fresh call.  Not for hybrid PRD, whose JRest lives on the device only,
    // and not with LWHIP_RESIDENT_J.)
    const bool force = force_upload(params);
    Prints prints;
    const bool speculate = !force && !resident_J() && !((PluginState*)ctx.methodScratch)->hprdOn
                           && !((PluginState*)ctx.methodScratch)->jDeviceAhead;
    uint32_t up = sync_state(ctx, force, speculate ? &prints : nullptr);
    auto* st = (PluginState*)ctx.methodScratch;
    int Nprd = 0;
    for (auto* a : ctx.activeAtoms)
        for (auto* t : a->trans)
            if (t->rhoPrd)
                ++Nprd;
    if (includeDetailed)
        for (auto* a : ctx.detailedAtoms)
            for (auto* t : a->trans)
                if (t->rhoPrd)
                    ++Nprd;
    IterationResult result{};
    if (Nprd == 0 || maxIter <= 0)
        return result;
    std::vector<f64> dRho((size_t)maxIter * Nprd), dJ(maxIter);
    std::vector<int32_t> dRhoIdx((size_t)maxIter * Nprd), dJIdx(maxIter);
    lwhip_prd_result r{ 0, 0, dRho.data(), dRhoIdx.data(), dJ.data(), dJIdx.data() };
    for (int attempt = 0; attempt < 2; ++attempt)
    {
        const bool restoreJ = attempt == 1 && !st->hprdOn; // (as in fs_iter: the first run has rewritten the device's J -- and spect.J, if mapped)
        if (restoreJ && lwhip_j_restore(st->hip) != LWHIP_OK)
            fail("lwhip_j_restore");
        const uint32_t want = (up & ~(uint32_t)LWHIP_GAMMA) | LWHIP_RATES;
        if (lwhip_upload(st->hip, restoreJ ? (want & ~(uint32_t)LWHIP_J) : j_policy_upload(st, want, force || attempt == 1)) != LWHIP_OK)
            fail("lwhip_upload");
        if (attempt == 0 && prints.active && !st->hprdOn && lwhip_j_snapshot(st->hip) != LWHIP_OK)
            fail("lwhip_j_snapshot");
        if (lwhip_redistribute_prd(st->hip, maxIter, tol, &r) != LWHIP_OK)
            fail("lwhip_redistribute_prd");
        if (!prints.active)
            break;
        const uint32_t all = prints_finish(st, prints, false);
        if (!(all & kSlowGroups))
            break;
        up = all;
    }
    // (the rates pass of every sub-iteration also rewrites spect.I of its wavelengths: intensity_core_opt,
    // SimdFullIterationTemplates.hpp:349)
    {
        const uint32_t down = j_policy_download(st, LWHIP_J | LWHIP_I | LWHIP_RATES | LWHIP_RHOPRD);
        if (lwhip_download(st->hip, down) != LWHIP_OK)
            fail("lwhip_download");
        j_record_synced(st, down);
    }
    const size_t nEnt = (size_t)r.NprdSubIter * r.Nprd;
    result.updatedRho = true;
    result.NprdSubIter = r.NprdSubIter;
    result.dRho.assign(dRho.begin(), dRho.begin() + nEnt);
    result.dRhoMaxIdx.assign(dRhoIdx.begin(), dRhoIdx.begin() + nEnt);
    result.updatedJPrd = true;
    result.dJPrdMax.assign(dJ.begin(), dJ.begin() + r.NprdSubIter);
    result.dJPrdMaxIdx.assign(dJIdx.begin(), dJIdx.begin() + r.NprdSubIter);
    return result;
}

// scratch hooks (Source/LwFormalInterface.hpp:102-108; called from Source/ThreadStorage.cpp:480-493,538-566)
void alloc_global_scratch_hip(Context* ctx)
{
    trace("alloc_global_scratch", ctx);
    ctx->methodScratch = nullptr; // the device context is created on first use, when the arrays are final
    // stat_eq / time_dep_update receive only an Atom*: the core has filled activeAtoms by now (Source/LwMiddleLayer.pyx:
    // 2951-2954 before setup_threads at 3055 -> ThreadStorage.cpp:484-486), so either may be the first call of a run
    register_atoms(ctx);
}

void free_global_scratch_hip(Context* ctx)
{
    trace("free_global_scratch", ctx);
    destroy_state(ctx);
}
}

extern "C"
{
FsIterationFns fs_iteration_fns_provider()
{
    return FsIterationFns{
        -1,     // Ndim: like the built-in scheme, 1D plane-parallel and 2D x-periodic behind one entry
        false,  // dimensionSpecific
        true,   // respectsFormalSolver (the three 1D solvers by name; 2D: piecewise_besser_2d + interp_linear_2d)
        true,   // defaultPerAtomStorage: kept for the forwarded core routines below
        true,   // defaultWlaGijStorage
        "mali_full_precond_hip_gfx950",
        fs_iter_hip,
        simple_fs_hip,
        formal_sol_full_stokes_impl,      // not accelerated: the core's own (final synthesis only)
        redistribute_prd_hip,
        stat_eq_hip,
        time_dep_update_hip,
        nr_post_update_hip,
        nullptr, nullptr, nullptr, nullptr,
        alloc_global_scratch_hip,
        free_global_scratch_hip,
        nullptr
    };
}
}

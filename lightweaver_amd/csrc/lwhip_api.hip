// lwhip_api.hip -- host side of the C ABI declared in include/lwhip.h: create / destroy, the launch sequence of one
// iteration, the population updates, device profiles, measurement helpers.  (The other host-side files: lwhip_host.h.)
//
//     sweep kernel (workgroup sums into the stage-1 buffer, dJ) -> reduce_stage2 (fixed-order sum of the parts; dJ to
//     (dJMax, idx) in the gather tail) [-> caller all-reduces the reduce buffer across GPUs] -> apply_kernel (Gamma
//     pre-fill fused)
//
// There is no CPU fallback: without a HIP device every compute entry point fails with LWHIP_ERR_DEVICE.
#include "lwhip_host.h"

namespace
{
thread_local std::string g_lastError;
}

namespace lwhip
{
int fail(int code, const std::string& msg)
{
    g_lastError = msg;
    return code;
}
}

namespace
{
// ---- the contexts' host blocks (lwhip_host.h: hostBlock) ---------------------------------------------------------------
// layout (bytes): [0,32) tail (dJMax, idx, ticket, -) | [32] status | [40] zeroCheck | [48] popStatus | [56] turnLate | [64,80) prdCtl
// | [96,104) canary | [128, ...) change records
constexpr size_t HB_TAIL = 0, HB_STATUS = 32, HB_ZERO = 40, HB_POPSTATUS = 48, HB_TURNLATE = 56, HB_PRDCTL = 64, HB_CANARY = 96, HB_CHANGE = 128;
constexpr uint64_t HB_CANARY_WORD = 0x6c77686970424c4bull; // "lwhipBLK"
constexpr unsigned char HB_DEAD = 0xC5;
struct Quarantine
{
    std::mutex m;
    std::vector<std::pair<unsigned char*, size_t>> blocks;
    // bytes of released blocks that no longer hold the fill: something stored into host memory of a context that was gone
    long check(bool report)
    {
        std::lock_guard<std::mutex> g(m);
        long bad = 0;
        for (auto& b : blocks)
            for (size_t i = 0; i < b.second; ++i)
                if (b.first[i] != HB_DEAD)
                {
                    if (report && bad < 16)
                        std::fprintf(stderr, "lwhip: QUARANTINE: released host block %p (+%zu of %zu) holds 0x%02x: a late store into "
                                             "host memory of a destroyed context\n", (void*)b.first, i, b.second, b.first[i]);
                    ++bad;
                }
        return bad;
    }
    ~Quarantine() { (void)check(true); }
};
Quarantine& quarantine()
{
    static Quarantine* q = new Quarantine; // (never destroyed before the last context: checked from atexit below)
    static const bool registered = (std::atexit([] { (void)quarantine().check(true); }), true);
    (void)registered;
    return *q;
}
}

namespace lwhip
{
// ---- the library's streams are pooled, never destroyed ------------------------------------------------------------------
// Round 6 located the stray 4-byte host store that rounds 3-5 chased (profiles/r06_stray_write.md): the HIP runtime's HSA
// signal callback (libamdhip64.so of ROCm 7.0 as bundled with torch 2.10, at +0x3d598a / +0x3d5995 of that build:
// `lock subq $1, 0x98(%rdi)` and `xchgl %eax(=0), 0x378(%rdi)` on the queue object of the command whose signal completed) can
// run AFTER hipStreamDestroy has deleted that queue object -- on the runtime's own thread, a hipStreamSynchronize before the
// destroy does not order it.  The two stores then land in whatever malloc() handed the freed 1 KB block to next: in the
// tests, a 920-byte numpy array made right after three contexts were closed -- one double lost its low 32 bits, the oracle's
// continuum rate moved by 1.77e-6.  In drop-in mode the block could be one of the USER's arrays.  A stream that is never
// destroyed has no freed queue object to write into: a context's own stream goes back to a per-device pool when the context
// dies and is handed to the next context made on that device (idle by then: lwhip_destroy has waited for it).  The pool's
// streams live until the process ends.  LWHIP_DEBUG + LWHIP_DESTROY_STREAMS=1 restores the old behaviour (the A/B of
// tools/guard/hunt.sh).
namespace
{
std::mutex g_streamMutex;
std::map<int, std::vector<hipStream_t>> g_streamPool;
std::atomic<long> g_streamsCreated{ 0 };
}
hipError_t stream_acquire(int device, hipStream_t* out)
{
    {
        std::lock_guard<std::mutex> g(g_streamMutex);
        auto& pool = g_streamPool[device];
        if (!pool.empty())
        {
            *out = pool.back();
            pool.pop_back();
            return hipSuccess;
        }
    }
    ++g_streamsCreated;
    return hipStreamCreate(out);
}
void stream_release(int device, hipStream_t s)
{
    static const bool destroy = dbg_env_int("LWHIP_DESTROY_STREAMS", 0) != 0;
    if (destroy)
    {
        (void)hipStreamDestroy(s);
        return;
    }
    std::lock_guard<std::mutex> g(g_streamMutex);
    g_streamPool[device].push_back(s);
}

// ---- a batch column's own stream is shared with the other columns its creating thread makes --------------------------------
// 512 columns took 512 streams out of the pool (hipStreamCreate: 5 ms each under 16 creating threads, 18 % of a batch's set-up)
// although a fused batch runs on ONE stream afterwards.  A borrower made with a batch hint and no stream of the caller's takes
// the stream of its creating thread's slot (16 slots per device, counted references, back to the pool with the last one).
namespace
{
struct SharedStream
{
    hipStream_t s = nullptr;
    int refs = 0;
};
std::mutex g_sharedMutex;
std::map<std::pair<int, int>, SharedStream> g_sharedStreams;
std::atomic<int> g_threadSlots{ 0 };
}
hipError_t stream_acquire_shared(int device, hipStream_t* out)
{
    static thread_local const int slot = g_threadSlots++ % 16;
    std::lock_guard<std::mutex> g(g_sharedMutex);
    SharedStream& e = g_sharedStreams[{ device, slot }];
    if (!e.s)
    {
        const hipError_t err = stream_acquire(device, &e.s);
        if (err != hipSuccess)
        {
            e.s = nullptr;
            return err;
        }
        e.refs = 0;
    }
    ++e.refs;
    *out = e.s;
    return hipSuccess;
}
void stream_release_shared(int device, hipStream_t s)
{
    std::lock_guard<std::mutex> g(g_sharedMutex);
    for (auto& kv : g_sharedStreams)
        if (kv.first.first == device && kv.second.s == s)
        {
            if (--kv.second.refs <= 0)
            {
                stream_release(device, s);
                kv.second = SharedStream{};
            }
            return;
        }
}

// ---- page-locked host memory out of slabs ---------------------------------------------------------------------------------
// hipHostMalloc takes ~3 ms when 16 threads create columns (two per context -- the host block and the upload stage -- were 20 %
// of a batch's set-up).  Blocks up to 512 KB are cut from 4 MB slabs (mapped, coherent: device-visible at the same offset of
// the slab's device pointer); larger ones are slabs of their own.  A released block waits in a list of its rounded size for the
// next request of that size; slabs live until the process ends.
namespace
{
struct PinnedSlab
{
    unsigned char* host = nullptr;
    unsigned char* dev = nullptr;
    size_t cap = 0, used = 0;
};
struct PinnedPool
{
    std::mutex m;
    std::vector<PinnedSlab> slabs;
    std::map<size_t, std::vector<std::pair<void*, void*>>> freeBySize;
};
PinnedPool& pinned_pool(int device)
{
    static std::mutex mm;
    static std::map<int, PinnedPool*>* pools = new std::map<int, PinnedPool*>();
    std::lock_guard<std::mutex> g(mm);
    PinnedPool*& p = (*pools)[device];
    if (!p)
        p = new PinnedPool();
    return *p;
}
constexpr size_t PINNED_SLAB = (size_t)4 << 20, PINNED_SMALL = (size_t)512 << 10;
inline size_t pinned_round(size_t bytes) { return (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255; }
}
void* pinned_acquire(int device, size_t bytes, void** devPtr)
{
    const size_t sz = pinned_round(bytes);
    PinnedPool& pool = pinned_pool(device);
    std::lock_guard<std::mutex> g(pool.m);
    auto it = pool.freeBySize.find(sz);
    if (it != pool.freeBySize.end() && !it->second.empty())
    {
        const auto pr = it->second.back();
        it->second.pop_back();
        if (devPtr)
            *devPtr = pr.second;
        return pr.first;
    }
    PinnedSlab* slab = nullptr;
    if (sz <= PINNED_SMALL)
        for (auto& sl : pool.slabs)
            if (sl.cap == PINNED_SLAB && sl.used + sz <= sl.cap)
            {
                slab = &sl;
                break;
            }
    if (!slab)
    {
        PinnedSlab sl;
        sl.cap = sz <= PINNED_SMALL ? PINNED_SLAB : sz;
        if (hipHostMalloc((void**)&sl.host, sl.cap, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess
            || hipHostGetDevicePointer((void**)&sl.dev, sl.host, 0) != hipSuccess)
        {
            (void)hipGetLastError();
            if (sl.host)
                (void)hipHostFree(sl.host);
            return nullptr;
        }
        pool.slabs.push_back(sl);
        slab = &pool.slabs.back();
    }
    void* h = slab->host + slab->used;
    if (devPtr)
        *devPtr = slab->dev + slab->used;
    slab->used += sz;
    return h;
}
void pinned_release(int device, void* p, size_t bytes)
{
    if (!p)
        return;
    PinnedPool& pool = pinned_pool(device);
    std::lock_guard<std::mutex> g(pool.m);
    void* dev = nullptr;
    for (const auto& sl : pool.slabs)
        if ((unsigned char*)p >= sl.host && (unsigned char*)p < sl.host + sl.cap)
            dev = sl.dev + ((unsigned char*)p - sl.host);
    pool.freeBySize[pinned_round(bytes)].emplace_back(p, dev);
}

// ---- gathered uploads of a batch column's creation (H2DBatch, lwhip_host.h) -----------------------------------------------
namespace
{
constexpr size_t H2D_BATCH_CAP = (size_t)24 << 20;
struct H2DPair
{
    unsigned char* host = nullptr;
    unsigned char* inbox = nullptr;
};
std::mutex g_h2dMutex;
std::map<int, std::vector<H2DPair>>& h2d_pairs()
{
    static std::map<int, std::vector<H2DPair>>* m = new std::map<int, std::vector<H2DPair>>();
    return *m;
}
}
bool H2DBatch::add(void* dst, const void* src, size_t bytes)
{
    if (!host || bytes == 0)
        return bytes == 0;
    const size_t nRec = (bytes + H2D_CHUNK - 1) / H2D_CHUNK;
    auto need = [&](size_t nMore) { return ((used + 15) & ~(size_t)15) + bytes + 16 + (recs.size() + nMore) * sizeof(H2DRec); };
    if (need(nRec) > cap)
    {
        if (flush() != hipSuccess || need(nRec) > cap)
            return false; // (larger than the stage: the ordinary copy)
    }
    if (inFlight)
    {
        // the stage is being read by the copy of the previous flush
        if (hipStreamSynchronize(stream) != hipSuccess)
            return false;
        inFlight = false;
    }
    used = (used + 15) & ~(size_t)15;
    std::memcpy(host + used, src, bytes);
    for (size_t o = 0; o < bytes; o += H2D_CHUNK)
        recs.push_back(H2DRec{ (unsigned long long)(uintptr_t)dst + o, (unsigned long long)(used + o),
                               (unsigned long long)std::min<size_t>(H2D_CHUNK, bytes - o) });
    used += bytes;
    ++copies;
    return true;
}
hipError_t H2DBatch::flush()
{
    if (recs.empty())
        return hipSuccess;
    const size_t recsOff = (used + 15) & ~(size_t)15;
    std::memcpy(host + recsOff, recs.data(), recs.size() * sizeof(H2DRec));
    const size_t total = recsOff + recs.size() * sizeof(H2DRec);
    hipError_t e = hipMemcpyAsync(inbox, host, total, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess)
        e = launch_h2d_scatter(inbox, recsOff, (int)recs.size(), stream);
    recs.clear();
    used = 0;
    inFlight = true;
    return e;
}
namespace
{
// open for the duration of one borrower's lwhip_create: the stage / inbox pair comes out of a per-device pool
struct H2DBatchScope
{
    H2DBatch b;
    int device = 0;
    bool open = false;
    void begin(int dev, hipStream_t stream)
    {
        device = dev;
        H2DPair pr;
        {
            std::lock_guard<std::mutex> g(g_h2dMutex);
            auto& v = h2d_pairs()[dev];
            if (!v.empty())
            {
                pr = v.back();
                v.pop_back();
            }
        }
        if (!pr.host)
        {
            if (hipHostMalloc((void**)&pr.host, H2D_BATCH_CAP, hipHostMallocDefault) != hipSuccess
                || hipMalloc((void**)&pr.inbox, H2D_BATCH_CAP) != hipSuccess)
            {
                (void)hipGetLastError();
                if (pr.host)
                    (void)hipHostFree(pr.host);
                return; // (no batch: the ordinary copies)
            }
        }
        b.host = pr.host;
        b.inbox = pr.inbox;
        b.cap = H2D_BATCH_CAP;
        b.stream = stream;
        h2d_batch() = &b;
        open = true;
    }
    // what was gathered goes out; the stage is free again when the stream has drained
    hipError_t finish()
    {
        if (!open)
            return hipSuccess;
        hipError_t e = b.flush();
        const hipError_t e2 = hipStreamSynchronize(b.stream);
        b.inFlight = false;
        return e != hipSuccess ? e : e2;
    }
    ~H2DBatchScope()
    {
        if (!open)
            return;
        h2d_batch() = nullptr;
        if (b.inFlight || !b.recs.empty())
            (void)hipStreamSynchronize(b.stream); // (an error path: nothing of the stage may still be in flight)
        std::lock_guard<std::mutex> g(g_h2dMutex);
        h2d_pairs()[device].push_back(H2DPair{ b.host, b.inbox });
    }
};
}

int host_block_init(lwhip_context* c)
{
    int maxNl = 1, nActive = 0;
    for (int ia = 0; ia < c->prob.Natom; ++ia)
    {
        maxNl = std::max(maxNl, (int)c->prob.atoms[ia].Nlevel);
        ++nActive;
    }
    // change records: (value, index) per solved atom and solve block of stat_equil's report, per atom of the Ng step
    const size_t nChange = (size_t)2 * std::max(nActive, 1) * (size_t)stat_eq_blocks(std::max((int)c->prob.Nspace, 1), std::min(maxNl, 64)) + 16;
    c->hostBlockBytes = HB_CHANGE + nChange * sizeof(double);
    unsigned char* dev = nullptr;
    c->hostBlock = (unsigned char*)pinned_acquire(c->device, c->hostBlockBytes, (void**)&dev);
    if (!c->hostBlock || !dev)
        return fail(LWHIP_ERR_DEVICE, "lwhip_create: hipHostMalloc of the context's host block failed");
    std::memset(c->hostBlock, 0, c->hostBlockBytes);
    std::memcpy(c->hostBlock + HB_CANARY, &HB_CANARY_WORD, sizeof HB_CANARY_WORD);
    c->tailMapped = (double*)(c->hostBlock + HB_TAIL);
    c->tailMappedDev = (double*)(dev + HB_TAIL);
    c->statusHost = (int32_t*)(c->hostBlock + HB_STATUS);
    c->statusDev = (int32_t*)(dev + HB_STATUS);
    if (dbg_env_int("LWHIP_CHECK_ZERO", 0) == 1)
    {
        c->zeroCheck = (int*)(c->hostBlock + HB_ZERO);
        c->zeroCheckDev = (int*)(dev + HB_ZERO);
    }
    c->popStatusHost = (int32_t*)(c->hostBlock + HB_POPSTATUS);
    c->turnLateHost = (int32_t*)(c->hostBlock + HB_TURNLATE);
    c->turnLateDev = (int32_t*)(dev + HB_TURNLATE);
    c->prdCtlHost = (int32_t*)(c->hostBlock + HB_PRDCTL);
    c->changeHost = (double*)(c->hostBlock + HB_CHANGE);
    c->changeDev = (double*)(dev + HB_CHANGE);
    c->changeCount = nChange;
    return LWHIP_OK;
}

// (the destructor has waited for the stream and the null stream)
void host_block_release(lwhip_context* c)
{
    if (!c->hostBlock)
        return;
    uint64_t canary = 0;
    std::memcpy(&canary, c->hostBlock + HB_CANARY, sizeof canary);
    if (canary != HB_CANARY_WORD)
        std::fprintf(stderr, "lwhip: the canary of a context's host block reads %016llx: something stored past a status word\n",
                     (unsigned long long)canary);
    if (debug_knobs_on())
    {
        std::memset(c->hostBlock, HB_DEAD, c->hostBlockBytes);
        std::lock_guard<std::mutex> g(quarantine().m);
        quarantine().blocks.emplace_back(c->hostBlock, c->hostBlockBytes);
    }
    else
        pinned_release(c->device, c->hostBlock, c->hostBlockBytes);
    c->hostBlock = nullptr;
    c->tailMapped = c->tailMappedDev = c->changeHost = c->changeDev = nullptr;
    c->statusHost = c->statusDev = c->popStatusHost = c->prdCtlHost = c->turnLateHost = c->turnLateDev = nullptr;
    c->zeroCheck = c->zeroCheckDev = nullptr;
}
}

namespace lwhip
{
// ---- peer exchange (include/lwhip.h: lwhip_peer_*) ----------------------------------------------------------------------
constexpr size_t PEER_HDR = 2 * LWHIP_PEER_MAX * sizeof(unsigned long long); // the flags of both buffers

void peer_release(lwhip_context* c)
{
    for (int r = 0; r < LWHIP_PEER_MAX; ++r)
    {
        if (c->peerIpc[r] && c->peerPtr[r])
            (void)hipIpcCloseMemHandle(c->peerPtr[r]);
        c->peerPtr[r] = nullptr;
        c->peerIpc[r] = false;
    }
    c->peerOn = false;
    if (c->peerWin)
        (void)hipFree(c->peerWin);
    c->peerWin = nullptr;
}

// this rank's reduce buffer into every window; the flag value is the exchange's number
int peer_publish(lwhip_context* c)
{
    PeerPublishArgs a{};
    a.red = c->red.p;
    a.n = (int64_t)c->red.n;
    a.world = c->worldSize;
    a.rank = c->worldRank;
    a.seq = c->peerSeq + 1;
    a.arrive = c->peerArrive.p;
    const size_t buf = (size_t)(c->peerSeq & 1);
    for (int r = 0; r < c->worldSize; ++r)
    {
        a.slot[r] = (double*)(c->peerPtr[r] + PEER_HDR) + (buf * c->worldSize + c->worldRank) * c->peerStride;
        a.flag[r] = (unsigned long long*)c->peerPtr[r] + buf * LWHIP_PEER_MAX + c->worldRank;
    }
    HIP_TRY(launch_peer_publish(a, c->stream));
    c->peerSeq += 1;
    return LWHIP_OK;
}

// the consumer side of the exchange published last
void peer_apply_args(lwhip_context* c, ApplyArgs& a)
{
    const size_t buf = (size_t)((c->peerSeq - 1) & 1);
    a.peerWorld = c->worldSize;
    a.peerSlots = (const double*)(c->peerWin + PEER_HDR) + buf * c->worldSize * c->peerStride;
    a.peerStride = (int64_t)c->peerStride;
    a.peerFlags = (const unsigned long long*)c->peerWin + buf * LWHIP_PEER_MAX;
    a.peerSeq = c->peerSeq;
}
}

namespace lwhip
{
// the blocks' sums of the device's J into pinned memory, queued on the stream (lwhip_fingerprint_J, and every download of J
// into a mapped array: one wait serves both)
int fingerprint_J_enqueue(lwhip_context* c)
{
    const size_t n = (size_t)c->Nla * c->Ns;
    const size_t nb = (n + LWHIP_FP_BLOCK - 1) / LWHIP_FP_BLOCK;
    if (c->fpSums.n < nb * 16)
        HIP_TRY(c->fpSums.alloc(nb * 16));
    if (c->fpPinnedN < nb * 16)
    {
        if (c->fpPinned)
        {
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipHostFree(c->fpPinned));
        }
        c->fpPinned = nullptr;
        HIP_TRY(hipHostMalloc((void**)&c->fpPinned, nb * 16 * sizeof(unsigned long long), hipHostMallocDefault));
        c->fpPinnedN = nb * 16;
    }
    HIP_TRY(launch_fp_blocks(c->J.p, n, c->fpSums.p, c->stream));
    HIP_TRY(hipMemcpyAsync(c->fpPinned, c->fpSums.p, nb * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    return LWHIP_OK;
}
// (after the stream has been waited for)
void fingerprint_J_fold(lwhip_context* c, const void* p)
{
    const size_t n = (size_t)c->Nla * c->Ns;
    const size_t nb = (n + LWHIP_FP_BLOCK - 1) / LWHIP_FP_BLOCK;
    std::vector<uint64_t> hashes(nb);
    for (size_t b = 0; b < nb; ++b)
        hashes[b] = lwhip_fp_block_fold((const uint64_t*)c->fpPinned + b * 16, std::min<size_t>(LWHIP_FP_BLOCK, n - b * LWHIP_FP_BLOCK),
                                        (uint64_t)(b * LWHIP_FP_BLOCK));
    c->fpJValue = lwhip_fp_array_fold(hashes.data(), nb, p, n);
    c->fpJPtr = p;
    c->fpJValid = true;
}
}

extern "C"
{
int lwhip_peer_window(lwhip_context* c, void** devPtr, size_t* bytes)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    if (c->worldSize < 1 || c->worldSize > LWHIP_PEER_MAX)
        return fail(LWHIP_ERR_UNSUPPORTED, "peer exchange: at most 16 ranks");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->peerWin)
    {
        c->peerStride = (c->red.n + 31) & ~(size_t)31;
        c->peerWinBytes = PEER_HDR + (size_t)2 * c->worldSize * c->peerStride * sizeof(double);
        // Device memory the OTHER devices store into while this device's kernels poll it: uncached (else fine-grained) where
        // the runtime offers it -- a coarse-grained allocation's lines may sit in this device's L2 and never show a peer's
        // store within a kernel; plain hipMalloc as the last resort (ranks sharing one device).  The consumers' accesses are
        // system-scope either way, and lwhip_peer_selftest checks the exchange on the actual topology before it is used.
        if (hipExtMallocWithFlags((void**)&c->peerWin, c->peerWinBytes, hipDeviceMallocUncached) != hipSuccess)
        {
            (void)hipGetLastError();
            c->peerWin = nullptr;
            if (hipExtMallocWithFlags((void**)&c->peerWin, c->peerWinBytes, hipDeviceMallocFinegrained) != hipSuccess)
            {
                (void)hipGetLastError();
                c->peerWin = nullptr;
                HIP_TRY(hipMalloc((void**)&c->peerWin, c->peerWinBytes));
            }
        }
        HIP_TRY(hipMemset(c->peerWin, 0, c->peerWinBytes));
        HIP_TRY(hipStreamSynchronize(nullptr));
        HIP_TRY(c->peerArrive.alloc_zero(1));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    if (devPtr)
        *devPtr = c->peerWin;
    if (bytes)
        *bytes = c->peerWinBytes;
    return LWHIP_OK;
}

int lwhip_peer_export(lwhip_context* c, void* handle64)
{
    if (!c || !handle64)
        return fail(LWHIP_ERR_INVALID, "null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI passes IPC handles as 64 opaque bytes");
    const int st = lwhip_peer_window(c, nullptr, nullptr);
    if (st != LWHIP_OK)
        return st;
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, c->peerWin));
    std::memcpy(handle64, &h, sizeof h);
    return LWHIP_OK;
}

static int peer_attach_impl(lwhip_context* c, void* const* windows, const unsigned char* handles)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    if (c->is2d && false)
        return fail(LWHIP_ERR_UNSUPPORTED, "peer exchange");
    const int st = lwhip_peer_window(c, nullptr, nullptr);
    if (st != LWHIP_OK)
        return st;
    if (c->peerOn)
        return fail(LWHIP_ERR_INVALID, "lwhip_peer_attach: already attached (lwhip_peer_detach first)");
    for (int r = 0; r < c->worldSize; ++r)
    {
        if (r == c->worldRank)
        {
            c->peerPtr[r] = c->peerWin;
            continue;
        }
        if (windows)
        {
            if (!windows[r])
                return fail(LWHIP_ERR_INVALID, "lwhip_peer_attach_pointers: null window");
            c->peerPtr[r] = (unsigned char*)windows[r];
        }
        else
        {
            hipIpcMemHandle_t h;
            std::memcpy(&h, handles + (size_t)r * 64, sizeof h);
            void* p = nullptr;
            const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess)
            {
                for (int q = 0; q < r; ++q)
                    if (c->peerIpc[q])
                    {
                        (void)hipIpcCloseMemHandle(c->peerPtr[q]);
                        c->peerIpc[q] = false;
                        c->peerPtr[q] = nullptr;
                    }
                return fail(LWHIP_ERR_DEVICE, std::string("lwhip_peer_attach: hipIpcOpenMemHandle of rank ") + std::to_string(r)
                                                  + "'s window failed: " + hipGetErrorString(e));
            }
            c->peerPtr[r] = (unsigned char*)p;
            c->peerIpc[r] = true;
        }
    }
    c->peerOn = true;
    // (peerSeq is NOT reset: the flags in the windows are monotonic exchange numbers, and ranks that detach and attach again --
    // all of them, having made the same number of exchanges -- must not find the old, larger values satisfying a new wait)
    return LWHIP_OK;
}

int lwhip_peer_attach(lwhip_context* c, const void* handles)
{
    if (!handles)
        return fail(LWHIP_ERR_INVALID, "null argument");
    return peer_attach_impl(c, nullptr, (const unsigned char*)handles);
}

int lwhip_peer_attach_pointers(lwhip_context* c, void* const* windows)
{
    if (!windows)
        return fail(LWHIP_ERR_INVALID, "null argument");
    return peer_attach_impl(c, windows, nullptr);
}

int lwhip_peer_selftest(lwhip_context* c, int timeoutMs, int32_t* result)
{
    if (!c || !result)
        return fail(LWHIP_ERR_INVALID, "null argument");
    if (!c->peerOn)
        return fail(LWHIP_ERR_INVALID, "lwhip_peer_selftest: not attached");
    HIP_TRY(hipSetDevice(c->device));
    // one exchange with a known pattern in place of the partial sums (the reduce buffer is scratch between iterations)
    HIP_TRY(launch_peer_pattern(c->red.p, (int64_t)c->red.n, c->worldRank, c->stream));
    const int st = peer_publish(c);
    if (st != LWHIP_OK)
        return st;
    ApplyArgs a{};
    peer_apply_args(c, a);
    *c->popStatusHost = -1;
    int32_t* resDev = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void**)&resDev, c->popStatusHost, 0));
    const long spins = std::max(1L, (long)timeoutMs) * 1500L; // (a poll with its sleep is ~0.6 us)
    HIP_TRY(launch_peer_check(a.peerSlots, a.peerStride, a.peerFlags, c->worldSize, a.peerSeq, (int64_t)c->red.n, spins, resDev, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *result = *(volatile int32_t*)c->popStatusHost;
    return LWHIP_OK;
}

int lwhip_peer_detach(lwhip_context* c)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int r = 0; r < LWHIP_PEER_MAX; ++r)
    {
        if (c->peerIpc[r] && c->peerPtr[r])
            (void)hipIpcCloseMemHandle(c->peerPtr[r]);
        c->peerPtr[r] = nullptr;
        c->peerIpc[r] = false;
    }
    c->peerOn = false;
    return LWHIP_OK;
}

const char* lwhip_last_error(void)
{
    return g_lastError.c_str();
}

long lwhip_debug_check_quarantine(void)
{
    return quarantine().check(true);
}

long lwhip_debug_streams_created(void)
{
    return g_streamsCreated.load();
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

static int create_impl(const lwhip_problem* prob, const lwhip_options* opts, lwhip_context* like, lwhip_context** out);

int lwhip_create(const lwhip_problem* prob, const lwhip_options* opts, lwhip_context** out)
{
    return create_impl(prob, opts, nullptr, out);
}

int lwhip_create_like(const lwhip_problem* prob, const lwhip_options* opts, lwhip_context* like, lwhip_context** out)
{
    if (!like)
        return fail(LWHIP_ERR_INVALID, "lwhip_create_like: null context to share the tables of");
    return create_impl(prob, opts, like, out);
}

static int create_impl(const lwhip_problem* prob, const lwhip_options* opts, lwhip_context* like, lwhip_context** out)
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

    DevBuf<double>::alloc_seq() = 0; // (diagnosis: allocations are numbered per context)
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
        if (c->tablesFrom)
            c->tablesFrom->borrowers--;
        delete c;
        return code;
    };
    if (hipSetDevice(c->device) != hipSuccess)
        return bail(fail(LWHIP_ERR_DEVICE, "hipSetDevice failed"));
    if (opts && opts->stream)
        c->stream = (hipStream_t)opts->stream;
    else
    {
        c->ownStreamShared = like != nullptr && c->batchHint > 1;
        if ((c->ownStreamShared ? stream_acquire_shared(c->device, &c->ownStream) : stream_acquire(c->device, &c->ownStream)) != hipSuccess)
            return bail(fail(LWHIP_ERR_DEVICE, "hipStreamCreate failed"));
        c->stream = c->ownStream;
    }
    if (host_block_init(c) != LWHIP_OK)
        return bail(LWHIP_ERR_DEVICE);
    c->pairRays = dbg_env_int("LWHIP_PAIR_RAYS", 1) != 0;
    c->prdPipeline = dbg_env_int("LWHIP_PRD_PIPELINE", 1) != 0;
    c->prdGeneral = dbg_env_int("LWHIP_PRD_GENERAL", 0) != 0;
    const bool timing = std::getenv("LWHIP_CREATE_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    c->structSig = structure_signature(c);
    if (like)
    {
        // a borrower of a borrower borrows from the owner
        lwhip_context* owner = like->tablesFrom ? like->tablesFrom : like;
        if (owner->structSig != c->structSig)
            return bail(fail(LWHIP_ERR_INVALID, "lwhip_create_like: the problems differ in structure (sizes, solver, shard, wavelength "
                                                "grid, or a transition's levels / range / coefficients / grid / cross-sections)"));
        c->tablesFrom = owner;
    }
    // One arena for everything a borrower allocates between here and the end of its first upload (DevArena, lwhip_host.h); the
    // owner counts what the same path takes in its own case.  (The structure tables that borrowers share are outside: see
    // DevBuf::upload_or_borrow.)
    DevArena arena;
    const bool plainAllocs = DevBuf<double>::poison_on() || DevBuf<double>::sentinel_on() || std::getenv("LWHIP_TRACE_ALLOC") != nullptr;
    if (!plainAllocs)
    {
        if (c->tablesFrom && c->tablesFrom->stateBytes > 0)
        {
            arena.size = c->tablesFrom->stateBytes + (64u << 10);
            // (a batch column's uploads all run on its own stream -- the gathered upload below --, so its clearing fill does too;
            // otherwise the null stream, which the synchronous table uploads of DevBuf::upload follow)
            if (hipMalloc((void**)&arena.base, arena.size) == hipSuccess
                && hipMemsetAsync(arena.base, 0, arena.size, c->batchHint > 1 ? c->stream : nullptr) == hipSuccess)
                c->stateArena = arena.base;
            else
            {
                (void)hipGetLastError();
                if (arena.base)
                    (void)hipFree(arena.base);
                arena = DevArena{};
            }
        }
        else if (!c->tablesFrom)
            arena.counting = true;
    }
    struct ArenaScope
    {
        explicit ArenaScope(DevArena* a) { devbuf_arena() = a; }
        ~ArenaScope() { devbuf_arena() = nullptr; }
    } arenaScope((arena.base || arena.counting) ? &arena : nullptr);
    // ... and, for a column of a batch, one gathered upload (H2DBatch, lwhip_host.h) on the column's stream
    H2DBatchScope h2d;
    if (arena.base && c->batchHint > 1 && dbg_env_int("LWHIP_H2D_BATCH", 1) != 0)
        h2d.begin(c->device, c->stream);
    st = build_tables(c);
    if (st != LWHIP_OK)
    {
        c->tablesFrom = nullptr; // (not counted yet)
        return bail(st);
    }
    if (c->tablesFrom)
        c->tablesFrom->borrowers++;
    const auto t1 = now();
    {
        // (the state allocations' clearing fills queue up on the null stream; one wait for all of them)
        struct Defer
        {
            Defer() { ++devbuf_defer_fill_sync(); }
            ~Defer() { --devbuf_defer_fill_sync(); }
        } defer;
        struct Skip
        {
            const bool on;
            explicit Skip(bool o) : on(o) { devbuf_skip_safety_clear() += on ? 1 : 0; }
            ~Skip() { devbuf_skip_safety_clear() -= on ? 1 : 0; }
        } skip(c->tablesFrom != nullptr);
        st = alloc_state(c);
        // (a batch column whose allocations all came out of its arena has queued nothing on the null stream -- and waiting for the
        // null stream means waiting for every other creating thread's stream)
        if (!(h2d_batch() && !arena.overflowed) && hipStreamSynchronize(nullptr) != hipSuccess && st == LWHIP_OK)
            st = fail(LWHIP_ERR_DEVICE, "lwhip_create: clearing the state allocations failed");
    }
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
    if (arena.counting)
        c->stateBytes = arena.counted;
    devbuf_arena() = nullptr;
    if (h2d.finish() != hipSuccess)
        return bail(fail(LWHIP_ERR_DEVICE, "lwhip_create: the gathered upload of a batch column failed"));
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
                                       + (size_t)c->tileWaves * 3 * 64 * c->laneD + (size_t)c->tileWaves * std::min(c->maxC, (int)LWHIP_POST_MAXC) * c->laneR * 3)
                         + (size_t)c->depArena.n * sizeof(double));
    *out = c;
    return LWHIP_OK;
}

int lwhip_destroy(lwhip_context* c)
{
    if (!c)
        return LWHIP_OK;
    if (c->borrowers.load() > 0)
        return fail(LWHIP_ERR_BUSY, "lwhip_destroy: " + std::to_string(c->borrowers.load())
                                           + " context(s) made with lwhip_create_like still use this context's tables");
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->tablesFrom)
    {
        c->tablesFrom->borrowers--;
        c->tablesFrom = nullptr;
    }
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
            double w[4] = { 0, 0, 0, 0 }, wmax = 0.0;
            int nw = 0;
            for (int b = 0; b < 4 * c->nTiles; ++b)
            {
                const double* o = c->lsDbgHost + (size_t)c->nTiles * 8 + (size_t)b * 4;
                if (o[1] > 0.0)
                {
                    ++nw;
                    for (int q = 0; q < 4; ++q)
                        w[q] += o[q];
                    wmax = std::max(wmax, o[3]);
                }
            }
            if (nw)
                std::fprintf(stderr, "lane sweep workgroup clocks (mean over %d workgroups): prologue %.0f tasks %.0f epilogue %.0f; fixed-order mode: "
                                     "clocks between a wavefront's leaving its first turn wait and its handing the turn on (line / mixed slots), summed over the workgroup, mean %.0f max %.0f\n",
                             nw, w[0] / nw, w[1] / nw, w[2] / nw, w[3] / nw, wmax);
        }
        std::fprintf(stderr, "lane sweep phase clocks per tile (mean): setup %.0f gather %.0f stencils %.0f coefficients %.0f scan %.0f rates %.0f finish %.0f total %.0f (max %.0f) over %d tiles\n",
                     sum[0] / c->nTiles, sum[1] / c->nTiles, sum[2] / c->nTiles, sum[3] / c->nTiles, sum[4] / c->nTiles, sum[5] / c->nTiles,
                     sum[6] / c->nTiles, sum[7] / c->nTiles, mx, c->nTiles);
        (void)hipHostFree(c->lsDbgHost);
    }
    delete c;
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
}

namespace lwhip
{
// argument blocks of the reduce / apply kernels of one context (also the per-column blocks of a batch)
ReduceArgs make_reduce_args(lwhip_context* c)
{
    ReduceArgs r{};
    r.Ns = c->Ns;
    r.Ntrans = c->Ntrans;
    // (fixed-order mode: the "parts" are the segment sums of the slab reduce, in segment order)
    const bool detParts = c->laneSweep && c->deterministic && !c->is2d;
    r.red8 = detParts ? c->detPart.p : c->red8.p;
    r.nParts = detParts ? (int)LWHIP_DET_SEG : (int)LWHIP_REDUCE_SPLIT;
    r.red = c->red.p;
    r.dJ = c->dJ.p;
    r.Nla = c->Nla;
    r.idxOffset = c->laStart;
    r.world = c->worldSize;
    r.rank = c->worldRank;
    r.batchTail = nullptr;
    return r;
}

ApplyArgs make_apply_args(lwhip_context* c)
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
}

extern "C"
{

static int fs_partial_impl(lwhip_context* c, int lambdaIterate, bool partsOnly)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    c->lastSweepUpOnly = false;
    c->fpJValid = false; // (the sweep rewrites J)
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
        if (c->peerOn && c->worldSize > 1)
        {
            const int stq = peer_publish(c);
            if (stq != LWHIP_OK)
                return stq;
        }
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
    if (c->peerOn && c->worldSize > 1 && !partsOnly)
    {
        const int stq = peer_publish(c);
        if (stq != LWHIP_OK)
            return stq;
    }
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
            a.red8 = (c->laneSweep && c->deterministic) ? c->detPart.p : c->red8.p;
            a.nParts = (c->laneSweep && c->deterministic) ? (int)LWHIP_DET_SEG : (int)LWHIP_REDUCE_SPLIT;
            if (c->laneSweep && c->deterministic)
                a.zeroParts = 0; // (the segment sums are rewritten by every slab reduce)
            a.dJ = c->dJ.p;
            a.NlaDj = c->Nla;
            a.idxOffset = c->laStart;
            a.scalarIdx = c->djIdxMode;
            a.tail = c->red.p + (size_t)c->Ntrans * 4 * c->Ns;
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
            if (c->peerOn && c->worldSize > 1)
                peer_apply_args(c, a);
            a.tailHost = c->tailMappedDev;
            c->tailTicket += 1.0;
            a.tailTicket = c->tailTicket;
            mapped = true;
        }
        if (!c->partsOnly && c->peerOn && c->worldSize > 1 && a.peerWorld == 0)
            peer_apply_args(c, a); // (no result wanted: the sums still come from the exchange, not from `red`)
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
        if (c->deterministic && c->turnLateHost && *(volatile int32_t*)c->turnLateHost != 0)
        {
            // (the apply launch that published the result ran after the sweep: the count is final)
            const int late = *(volatile int32_t*)c->turnLateHost;
            *(volatile int32_t*)c->turnLateHost = 0;
            return fail(LWHIP_ERR_DEVICE, "fixed-order mode: " + std::to_string(late) + " turn wait(s) of the sweep timed out -- the sums of this "
                                          "iteration were formed out of order (results are valid to rounding, not bit-reproducible)");
        }
        if (c->peerOn && c->tailMapped[3] != 0.0)
        {
            c->tailMapped[3] = 0.0;
            return fail(LWHIP_ERR_DEVICE, "peer exchange: a rank's partial sums never arrived (its flag was not raised within the "
                                          "wait of the apply launch): the results of this iteration are incomplete");
        }
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

// the device-resident iteration in one call: lwhip_gamma_prefill_from_C + lwhip_formal_sol_gamma_matrices (a caller that
// iterates from Python pays a microsecond or two per ABI call, and the device idles while it does)
int lwhip_iterate_from_C(lwhip_context* c, int lambdaIterate, double crsw, lwhip_iter_result* res)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    c->prefillPending = true;
    c->prefillCrsw = crsw;
    return lwhip_formal_sol_gamma_matrices(c, lambdaIterate, res);
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

}

namespace lwhip
{
int stat_equil_impl(lwhip_context* c, int atom, bool wait, double* dPops, int32_t* dPopsMaxIdx)
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
        if (c->changeCount < need) // (sized for every active atom at lwhip_create: host_block_init)
            return fail(LWHIP_ERR_INVALID, "stat_equil report: the context's host block is too small for the change records");
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
}

extern "C"
{
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
    if (c->changeCount < (size_t)2 * nAt)
        return fail(LWHIP_ERR_INVALID, "Ng: the context's host block is too small for the change records");
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
    // (the copy's target is the context's own pinned word, not a stack variable: an early return cannot leave a copy in flight
    // that aims at a dead frame)
    *c->popStatusHost = 0;
    HIP_TRY(hipMemcpyAsync(c->popStatusHost, c->status.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int32_t st = *(volatile int32_t*)c->popStatusHost;
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
    *c->popStatusHost = 0;
    HIP_TRY(hipMemcpyAsync(c->popStatusHost, c->status.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    {
        // (the caller's array is the target of this copy: wait for the stream on EVERY path out)
        const hipError_t ec = hipMemcpyAsync(args->ne, dNe, Ns * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        const hipError_t es = hipStreamSynchronize(c->stream);
        HIP_TRY(ec);
        HIP_TRY(es);
    }
    const int32_t st = *(volatile int32_t*)c->popStatusHost;
    if (st == LWHIP_ERR_SINGULAR)
        return fail(LWHIP_ERR_SINGULAR, "Singular Matrix");
    return LWHIP_OK;
}
}

namespace lwhip
{
// the argument blocks of the context's lines (wavelength slices and tickets of their own: the lines run side by side)
int voigt_line_list(lwhip_context* c, std::vector<VoigtLineArgs>& out)
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
    c->phiSym = c->vlosZero; // (the Voigt arguments of the two directions differ by the sign of the line-of-sight velocity alone)
    return retile_profiles(c);
}

// before a sweep: device-generated profiles whose inputs were uploaded again are brought up to date
int ensure_profiles(lwhip_context* c)
{
    return c->profilesStale ? compute_profiles_impl(c) : LWHIP_OK;
}
}

extern "C"
{
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

int lwhip_map_host_J(lwhip_context* c, int enable)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->JhostReg)
    {
        (void)hipHostUnregister(c->JhostReg);
        c->JhostReg = nullptr;
        c->JhostDev = nullptr;
        const int st = build_tile_args(c);
        if (st != LWHIP_OK)
            return st;
    }
    if (!enable)
        return LWHIP_OK;
    if (c->is2d || !c->laneSweep || c->worldSize > 1 || !c->prob.J)
        return fail(LWHIP_ERR_UNSUPPORTED, "lwhip_map_host_J: a 1D problem on the depth-across-lanes sweep, unsharded");
    const size_t bytes = (size_t)c->prob.Nlambda * c->Ns * sizeof(double);
    hipError_t e = hipHostRegister(c->prob.J, bytes, hipHostRegisterMapped);
    void* dev = nullptr;
    if (e == hipSuccess)
    {
        e = hipHostGetDevicePointer(&dev, c->prob.J, 0);
        if (e != hipSuccess)
            (void)hipHostUnregister(c->prob.J);
    }
    if (e != hipSuccess)
    {
        (void)hipGetLastError();
        return fail(LWHIP_ERR_UNSUPPORTED, std::string("lwhip_map_host_J: the caller's J array cannot be page-locked: ") + hipGetErrorString(e));
    }
    c->JhostReg = c->prob.J;
    c->JhostDev = (double*)dev + (size_t)c->laStart * c->Ns;
    return build_tile_args(c); // the sweep's argument block carries the pointer
}

int lwhip_j_snapshot(lwhip_context* c)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (c->Jsnap.n < c->J.n)
        HIP_TRY(c->Jsnap.alloc(c->J.n, false));
    HIP_TRY(hipMemcpyAsync(c->Jsnap.p, c->J.p, c->J.n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return LWHIP_OK;
}

int lwhip_j_restore(lwhip_context* c)
{
    if (!c)
        return fail(LWHIP_ERR_INVALID, "null context");
    if (c->Jsnap.n < c->J.n || !c->Jsnap.p)
        return fail(LWHIP_ERR_INVALID, "lwhip_j_restore without lwhip_j_snapshot");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->J.p, c->Jsnap.p, c->J.n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    c->fpJValid = false;
    return LWHIP_OK;
}

uint64_t lwhip_host_fingerprint(const double* p, size_t n)
{
    std::vector<uint64_t> hashes;
    for (size_t off = 0; off < n; off += LWHIP_FP_BLOCK)
    {
        const size_t len = std::min<size_t>(LWHIP_FP_BLOCK, n - off);
        uint64_t s[16] = { 1, 2, 3, 4, 5, 6, 7, 8, 0, 0, 0, 0, 0, 0, 0, 0 };
        size_t i = 0;
        for (; i + 8 <= len; i += 8)
            for (int l = 0; l < 8; ++l)
            {
                uint64_t w;
                std::memcpy(&w, p + off + i + l, 8);
                s[l] += w;
                s[8 + l] += s[l];
            }
        for (int l = 0; i < len; ++i, ++l)
        {
            uint64_t w;
            std::memcpy(&w, p + off + i, 8);
            s[l] += w;
            s[8 + l] += s[l];
        }
        hashes.push_back(lwhip_fp_block_fold(s, len, (uint64_t)off));
    }
    return lwhip_fp_array_fold(hashes.data(), hashes.size(), p, p ? n : 0);
}

int lwhip_fingerprint_J(lwhip_context* c, const void* p, uint64_t* out)
{
    if (!c || !out)
        return fail(LWHIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (c->fpJValid && c->fpJPtr == p) // (the last download of J left it: nothing has written the device's J since)
    {
        *out = c->fpJValue;
        return LWHIP_OK;
    }
    const int st = fingerprint_J_enqueue(c);
    if (st != LWHIP_OK)
        return st;
    HIP_TRY(hipStreamSynchronize(c->stream));
    fingerprint_J_fold(c, p);
    *out = c->fpJValue;
    return LWHIP_OK;
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
            HIP_TRY(stream_acquire(c->device, &c->ownStream));
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

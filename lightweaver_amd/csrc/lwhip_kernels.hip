// lwhip_kernels.hip -- the small kernels around the sweep of the formal_sol_gamma_matrices iteration on gfx950 (MI355X):
// the second reduce stage (sum of the stage-1 parts the sweep's workgroups add into, dJMax), apply (Gamma += radiative
// terms, finalise_Gamma, Rij / Rji out) and the Gamma pre-fill.  The sweep itself is lwhip_raymarch.hip.
//
// Reference restated (file:line under Lightweaver's Source/): see each kernel.
#include "lwhip_device.h"
#include <type_traits>
#include "../../include/lwhip.h"

#include <algorithm>

namespace lwhip
{
// ---- reduction of the chunk sums -----------------------------------------------------------------------------------
// replaces AtomStorageFactory::accumulate_Gamma / TransitionStorageFactory::accumulate_rates
// (ThreadStorage.cpp:73-88,150-166): the sweep's workgroups add their sums into red8[transition][part][4][Ns] (part =
// workgroup % 8 spreads the adders); stage 2 sums the parts in a fixed order into the reduce buffer (what a sharded
// iteration all-reduces) and zeroes them for the next iteration.
// (BATCH: column batches -- blockIdx.z picks the column's argument block from a table read through the constant
// address space; the single-problem instantiation keeps its kernel-argument block untouched)
__device__ void djmax_block(const double* dJ, int laEnd, int idxOffset, double* tail, int world, int rank,
                            double* mirror = nullptr /* host-mapped copy of the tail: no read-back copy needed */,
                            double ticket = 0.0 /* stored behind the mirror once it is complete */, int scalarIdx = 0)
{
    // dJMax and the index of its first occurrence (max_idx, Constants.hpp:114-125) into this
    // shard's slot of the gather tail; every other slot is zeroed (the all-reduce sums them)
    __shared__ double sV[1024];
    __shared__ int sI[1024];
    double best = -1.0;
    int bi = 0;
    for (int la = threadIdx.x; la < laEnd; la += blockDim.x)
    {
        const double v = dJ[la];
        if (v > best)
        {
            best = v;
            bi = la;
        }
    }
    sV[threadIdx.x] = best;
    sI[threadIdx.x] = bi;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1)
    {
        if (threadIdx.x < s)
        {
            const double v = sV[threadIdx.x + s];
            const int i = sI[threadIdx.x + s];
            if (v > sV[threadIdx.x] || (v == sV[threadIdx.x] && i < sI[threadIdx.x]))
            {
                sV[threadIdx.x] = v;
                sI[threadIdx.x] = i;
            }
        }
        __syncthreads();
    }
    if (scalarIdx)
    {
        // The index of the reference's single-thread loop (SimdFullIterationTemplates.hpp:620-628): `dJMax = max_idx(dJ,
        // dJMax, maxIdx, la)` with max_idx(a, b, aIdx, bIdx) = (a < b ? (aIdx = bIdx, b) : a) records la whenever dJ(la) is
        // BELOW the running maximum: the last wavelength whose dJ is smaller than the largest before it, or 0.  Every thread
        // walks a contiguous segment behind the exclusive prefix maximum of the segments before it.
        const double dJMaxAll = sV[0];
        __syncthreads();
        const int T = blockDim.x, seg = (laEnd + T - 1) / T;
        const int a0 = min((int)threadIdx.x * seg, laEnd), a1 = min(a0 + seg, laEnd);
        double segMax = 0.0; // (the loop starts from dJMax = 0.0)
        for (int la = a0; la < a1; ++la)
            segMax = fmax(segMax, dJ[la]);
        sV[threadIdx.x] = segMax;
        __syncthreads();
        for (int s = 1; s < T; s <<= 1) // inclusive prefix maximum over the threads
        {
            const double o = (int)threadIdx.x >= s ? sV[threadIdx.x - s] : 0.0;
            __syncthreads();
            sV[threadIdx.x] = fmax(sV[threadIdx.x], o);
            __syncthreads();
        }
        double run = threadIdx.x > 0 ? sV[threadIdx.x - 1] : 0.0;
        int last = -1;
        for (int la = a0; la < a1; ++la)
        {
            const double v = dJ[la];
            if (v < run)
                last = la;
            else
                run = v;
        }
        sI[threadIdx.x] = last;
        __syncthreads();
        for (int s = T / 2; s > 0; s >>= 1)
        {
            if ((int)threadIdx.x < s)
                sI[threadIdx.x] = max(sI[threadIdx.x], sI[threadIdx.x + s]);
            __syncthreads();
        }
        if (threadIdx.x == 0)
        {
            sI[0] = max(sI[0], 0);
            sV[0] = dJMaxAll;
        }
        __syncthreads();
    }
    for (int q = threadIdx.x; q < 2 * world; q += blockDim.x)
    {
        double v = 0.0;
        if (q == 2 * rank)
            v = sV[0] < 0.0 ? 0.0 : sV[0];
        else if (q == 2 * rank + 1)
            v = (double)(sI[0] + idxOffset);
        tail[q] = v;
    }
    if (mirror && threadIdx.x == 0)
    {
        // the host polls mirror[2] for this launch's ticket (lwhip_fs_finalise): value and index first, then the ticket
        // with system-scope release, so the result can be picked up while the other blocks of the launch still run
        __hip_atomic_store(&mirror[0], sV[0] < 0.0 ? 0.0 : sV[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mirror[1], (double)(sI[0] + idxOffset), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mirror[2], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <class A> DEVINL void reduce_stage2_kernel_body(const A& a)
{
    const int tr = blockIdx.x;
    if (tr == a.Ntrans) // the extra block: dJMax
    {
        djmax_block(a.dJ, a.Nla, a.idxOffset, a.red + (size_t)a.Ntrans * 4 * a.Ns, a.world, a.rank);
        if (a.batchTail) // column batches: every column's (dJMax, idx) in one buffer
        {
            __syncthreads();
            if (threadIdx.x < 2)
                a.batchTail[threadIdx.x] = a.red[(size_t)a.Ntrans * 4 * a.Ns + 2 * a.rank + threadIdx.x];
        }
        return;
    }
    const int n4 = 4 * a.Ns;
    for (int idx = threadIdx.x; idx < n4; idx += blockDim.x)
    {
        double s = 0.0;
        const int nParts = a.nParts > 0 ? a.nParts : (int)LWHIP_REDUCE_SPLIT;
        for (int part = 0; part < nParts; ++part)
        {
            const size_t o = ((size_t)tr * nParts + part) * n4 + idx;
            s += a.red8[o];
            if (a.zeroParts)
                a.red8[o] = 0.0;
        }
        a.red[(size_t)tr * n4 + idx] = s;
    }
}

template <bool BATCH>
__global__ void __launch_bounds__(512) reduce_stage2_kernel(const ReduceArgs a0, const ReduceArgs* __restrict__ list)
{
    dbg_poison_lds();
    if constexpr (BATCH)
        reduce_stage2_kernel_body(CTAB(ReduceArgs, list)[blockIdx.z]);
    else
        reduce_stage2_kernel_body(a0);
}

hipError_t launch_reduce_sum(const ReduceArgs& a, hipStream_t stream, const ReduceArgs* list, int nBatch)
{
    if (list)
        LWHIP_LAUNCH(reduce_stage2_kernel<true>, dim3(a.Ntrans + 1, 1, std::max(nBatch, 1)), dim3(512), 0, stream, a, list);
    else
        LWHIP_LAUNCH(reduce_stage2_kernel<false>, dim3(a.Ntrans + 1), dim3(512), 0, stream, a, list);
    return hipGetLastError();
}

// ---- peer exchange of the shards' partial sums (one node; replaces the all-reduce of the reduce buffer) --------------------
// What the reference does across threads (ThreadStorage.cpp:150-166, 334-396: every thread's private Gamma / rates added into
// the main copies in thread order) across GPUs: rank r stores its [Ntrans, 4, Ns] sums + gather tail into slot r of EVERY
// rank's window (peer-mapped device memory: xGMI stores, or plain stores when the ranks share a device), then raises flag r
// there; every rank's apply launch waits for the `world` flags of its own window and adds the slots in rank order.  No
// collective call, no host in the loop; the windows have two buffers used alternately (a rank can be at most one exchange
// ahead of the slowest: its next publish needs its own apply, which needed everybody's previous publish).
__global__ void __launch_bounds__(256) peer_publish_kernel(const PeerPublishArgs a)
{
    dbg_poison_lds();
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i0 < a.n)
    {
        const double v0 = a.red[i0], v1 = i0 + 1 < a.n ? a.red[i0 + 1] : 0.0;
        for (int r = 0; r < a.world; ++r)
        {
            a.slot[r][i0] = v0;
            if (i0 + 1 < a.n)
                a.slot[r][i0 + 1] = v1;
        }
    }
    // every storing wave waits for its stores, then one lane per workgroup releases them to the system and counts the
    // workgroup in; the last one to arrive raises this rank's flag in every window
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(a.arrive, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (int)gridDim.x - 1)
        {
            __hip_atomic_store(a.arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
            for (int r = 0; r < a.world; ++r)
                __hip_atomic_store(a.flag[r], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

hipError_t launch_peer_publish(const PeerPublishArgs& a, hipStream_t stream)
{
    const int64_t pairs = (a.n + 1) / 2;
    LWHIP_LAUNCH(peer_publish_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---- self-test of the exchange on the actual topology (lwhip_peer_selftest) ---------------------------------------------------
DEVINL double peer_pattern(int rank, int64_t i) { return (double)(rank + 1) * 1000.0 + (double)(i % 97); }
__global__ void __launch_bounds__(256) peer_pattern_kernel(double* buf, int64_t n, int rank)
{
    dbg_poison_lds();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        buf[i] = peer_pattern(rank, i);
}
__global__ void __launch_bounds__(256) peer_check_kernel(const double* slots, int64_t stride, const unsigned long long* flags, int world,
                                                         unsigned long long seq, int64_t n, long spinLimit, int32_t* result)
{
    dbg_poison_lds();
    __shared__ int sBad;
    if (threadIdx.x == 0)
        sBad = 0;
    __syncthreads();
    if ((int)threadIdx.x < world)
    {
        bool seen = false;
        for (long spin = 0; spin < spinLimit; ++spin)
        {
            if (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= seq)
            {
                seen = true;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!seen)
            atomicMax(&sBad, 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    if (sBad == 0)
        for (int r = 0; r < world; ++r)
            for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
                if (__hip_atomic_load(slots + (int64_t)r * stride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != peer_pattern(r, i))
                    atomicMax(&sBad, 2);
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(result, (int32_t)sBad, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_peer_pattern(double* buf, int64_t n, int rank, hipStream_t stream)
{
    LWHIP_LAUNCH(peer_pattern_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, buf, n, rank);
    return hipGetLastError();
}
hipError_t launch_peer_check(const double* slots, int64_t stride, const unsigned long long* flags, int world, unsigned long long seq, int64_t n,
                             long spinLimit, int32_t* result, hipStream_t stream)
{
    LWHIP_LAUNCH(peer_check_kernel, dim3(1), dim3(256), 0, stream, slots, stride, flags, world, seq, n, spinLimit, result);
    return hipGetLastError();
}

// the wait of a consumer workgroup: lanes 0 .. world-1 poll one flag each; bounded (a peer that died must not hang the device:
// ~2 s, then the sums are whatever the slots hold and the launch reports it through the mirror's fourth word)
DEVINL bool peer_wait(const unsigned long long* flags, int world, unsigned long long seq)
{
    __shared__ int sLate;
    if (threadIdx.x == 0)
        sLate = 0;
    __syncthreads();
    if ((int)threadIdx.x < world)
    {
        bool seen = false;
        for (long spin = 0; spin < (1l << 24); ++spin)
        {
            if (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= seq)
            {
                seen = true;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!seen)
            sLate = 1;
    }
    if (threadIdx.x < 64)
    {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return sLate == 0;
}

// ---- apply: Gamma += radiative terms (Gamma keeps its collisional pre-fill), finalise_Gamma
// (SimdFullIterationTemplates.hpp:491-508), Rij/Rji out ----------------------------------------------------
// One block per (atom, slab of APPLY_K depth points): thread (e, kk) owns Gamma entry e = to*Nl+from
// at depth k; the extra block column x == Natom copies Rij/Rji.
#ifndef APPLY_K_1D
#define APPLY_K_1D 8 // (at the bench size, tools/apply_abl.sh: 4 -> 15.8 us, 8 -> 11.0, 16 -> 11.8, 32 -> 14.6; 512 / 320 / 256 threads: no better)
#endif
#ifndef APPLY_T
#define APPLY_T 1024
#endif
enum { APPLY_K = APPLY_K_1D, APPLY_K_WIDE = 64 }; // depth points per workgroup: 8 (a 1D column's 82 points spread over 11 workgroups per
                                           // atom), 64 for the tens of thousands of points of a 2D domain (whole 512-byte rows per wavefront)
template <int APPLY_K, class A> DEVINL void apply_kernel_body(const A& a)
{
    const int Ns = a.Ns;
    const int k0 = blockIdx.y * APPLY_K;
    if constexpr (std::is_same<A, ApplyArgs>::value)
    {
        // pipelined PRD sub-iterations: an earlier one met the tolerance -- this launch does nothing
        if (a.prdCtl)
        {
            const int s = *(const volatile int32_t*)a.prdCtl;
            if (s != 0 && s < a.prdIter)
                return;
        }
    }
    bool peersSeen = true;
    if constexpr (std::is_same<A, ApplyArgs>::value)
    {
        if (a.peerWorld > 0)
            peersSeen = peer_wait(a.peerFlags, a.peerWorld, a.peerSeq);
    }
    // element i of the exchanged buffer: the slots of this rank's window in rank order (loads that bypass the caches: the
    // slots were written by other devices)
    auto peerSum = [&](size_t i) -> double {
        double s = 0.0;
        if constexpr (std::is_same<A, ApplyArgs>::value)
        {
            for (int r = 0; r < a.peerWorld; ++r)
                s += __hip_atomic_load(a.peerSlots + (size_t)r * a.peerStride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return s;
    };
    if ((int)blockIdx.x >= a.Natom + 2) // PRD sub-iteration on one device: the lines' max |d rho / rho| (one wavefront each)
    {
        if (blockIdx.y == 0 && threadIdx.x < 64 && a.prdList && (int)blockIdx.x - (a.Natom + 2) < a.nPrdChange)
        {
            const double best = prd_change_reduce(ld_c(CTAB(PrdLineArgs, a.prdList) + ((int)blockIdx.x - (a.Natom + 2))), (int)threadIdx.x);
            if constexpr (std::is_same<A, ApplyArgs>::value)
            {
                if (a.prdCtl && threadIdx.x == 0)
                {
                    // the loop of redistribute_prd (PrdTemplates.hpp:250-275) stops after the sub-iteration whose largest change
                    // over all lines is below the tolerance: the last line's workgroup to arrive records it for the launches
                    // of the following sub-iterations, which are queued already
                    if (best < a.prdTol)
                        atomicAdd(a.prdCtl + 1, 1);
                    __threadfence();
                    const int t = atomicAdd(a.prdCtl + 2, 1);
                    if (t == a.nPrdChange - 1)
                    {
                        const int below = atomicAdd(a.prdCtl + 1, 0);
                        a.prdCtl[1] = 0;
                        a.prdCtl[2] = 0;
                        a.prdCtl[3] = a.prdIter;
                        __threadfence();
                        if (below == a.nPrdChange)
                            atomicExch(a.prdCtl, a.prdIter);
                    }
                }
            }
        }
        return;
    }
    if ((int)blockIdx.x == a.Natom + 1) // dJMax (only launched when a.dJ or a.gatherWorld is set)
    {
        if (blockIdx.y == 0 && a.dJ)
            djmax_block(a.dJ, a.NlaDj, a.idxOffset, a.tail, 1, 0, a.tailHost, a.tailTicket, a.scalarIdx);
        else if (blockIdx.y == 0 && threadIdx.x == 0 && a.gatherWorld > 0)
        {
            // sharded iteration: max over the shards' slots, first (lowest) wavelength index on ties (max_idx,
            // Constants.hpp:114-125), published like the single-device result
            double best = -1.0, bestIdx = 0.0;
            bool peer = false;
            size_t tailOff = 0;
            if constexpr (std::is_same<A, ApplyArgs>::value)
            {
                peer = a.peerWorld > 0;
                tailOff = (size_t)(a.tail - a.red);
            }
            for (int r = 0; r < a.gatherWorld; ++r)
            {
                const double v = peer ? peerSum(tailOff + 2 * r) : a.tail[2 * r], i = peer ? peerSum(tailOff + 2 * r + 1) : a.tail[2 * r + 1];
                if (v > best || (v == best && i < bestIdx))
                {
                    best = v;
                    bestIdx = i;
                }
            }
            __hip_atomic_store(&a.tailHost[0], best < 0.0 ? 0.0 : best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&a.tailHost[1], bestIdx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&a.tailHost[3], peersSeen ? 0.0 : 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // (a peer's flag never came)
            __hip_atomic_store(&a.tailHost[2], a.tailTicket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    // the reduced sum of transition tr, component q at depth k: from the reduce buffer, or from the stage-1 parts
    auto redAt = [&](int tr, int q, int k) -> double {
        if constexpr (std::is_same<A, ApplyArgs>::value)
        {
            if (a.peerWorld > 0)
                return peerSum(((size_t)tr * 4 + q) * Ns + k);
        }
        if (!a.fromParts)
            return a.red[((size_t)tr * 4 + q) * Ns + k];
        double s = 0.0;
        double* r8 = const_cast<double*>(a.red8);
        if (a.nParts > 0 && a.nParts != (int)LWHIP_REDUCE_SPLIT)
        {
            // (fixed-order mode: the segment sums of the slab reduce, in segment order; eight loads in flight)
            for (int p0 = 0; p0 < a.nParts; p0 += 8)
            {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = p0 + u < a.nParts ? r8[(((size_t)tr * a.nParts + p0 + u) * 4 + q) * Ns + k] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    s += v[u];
            }
            return s;
        }
#pragma unroll
        for (int part = 0; part < LWHIP_REDUCE_SPLIT; ++part)
        {
            const size_t o = (((size_t)tr * LWHIP_REDUCE_SPLIT + part) * 4 + q) * Ns + k;
            s += r8[o];
            if (a.zeroParts)
                r8[o] = 0.0; // every element has exactly one reader: ready for the next iteration's atomic adds
        }
        return s;
    };
    if ((int)blockIdx.x == a.Natom) // rates out
    {
#ifdef AP_ABL_NO_RATES
        if (a.Ns != 77)
            return;
#endif
        const int Ntr = a.atomTrOff[a.Natom];
        for (int idx = threadIdx.x; idx < Ntr * APPLY_K; idx += blockDim.x)
        {
            const int tr = idx / APPLY_K;
            const int k = k0 + (idx - tr * APPLY_K);
            if (k < Ns)
            {
                if (a.prdOnly && a.fromParts && a.zeroParts)
                {
                    // the PRD rates pass on one device, read straight from the stage-1 parts: nobody else reads them, so every
                    // part of every transition the pass added to is summed -- and zeroed -- here, used or not
                    const double r2 = redAt(tr, 2, k), r3 = redAt(tr, 3, k);
                    (void)redAt(tr, 0, k);
                    (void)redAt(tr, 1, k);
                    if (!(a.trans[tr].prd && a.trans[tr].rhoOff >= 0))
                        continue;
                    if (!a.trans[tr].op && !a.prdDetailed)
                    {
                        a.Rij[(size_t)tr * Ns + k] += r2;
                        a.Rji[(size_t)tr * Ns + k] += r3;
                        continue;
                    }
                    a.Rij[(size_t)tr * Ns + k] = r2;
                    a.Rji[(size_t)tr * Ns + k] = r3;
                    continue;
                }
                if (a.prdOnly && !(a.trans[tr].prd && a.trans[tr].rhoOff >= 0))
                    continue;
                if (a.prdOnly && !a.trans[tr].op && !a.prdDetailed)
                {
                    // a detailed atom's PRD line that is not part of the PRD calls: integrated over the visited
                    // wavelengths on top of what it holds (see ApplyArgs::prdDetailed)
                    a.Rij[(size_t)tr * Ns + k] += redAt(tr, 2, k);
                    a.Rji[(size_t)tr * Ns + k] += redAt(tr, 3, k);
                    continue;
                }
                a.Rij[(size_t)tr * Ns + k] = redAt(tr, 2, k);
                a.Rji[(size_t)tr * Ns + k] = redAt(tr, 3, k);
                if (a.zeroParts && !a.trans[tr].op) // the Gamma terms of a detailed atom have no reader: drop them here
                {
                    (void)redAt(tr, 0, k);
                    (void)redAt(tr, 1, k);
                }
            }
        }
        return;
    }
    const int at = blockIdx.x;
    if (a.atomDetailed[at] || a.prdOnly)
        return;
#ifdef AP_ABL_NO_GAMMA
    if (a.Ns != 77)
        return;
#endif
    const int Nl = a.atomNlevel[at];
    double* G = a.Gamma + a.atomGammaOff[at];
    const double* Cm = a.Cmat + a.atomGammaOff[at];
    const int tr0 = a.atomTrOff[at], tr1 = a.atomTrOff[at + 1];
    // the atom's (i, j) pairs once into LDS: read from global memory inside the loop below they were a chain of ~2 Ntrans
    // dependent memory round trips per thread (7 of the launch's 11 microseconds)
    enum { APPLY_TR = 512, APPLY_TILE = 4096 };
    __shared__ short sLi[APPLY_TR], sLj[APPLY_TR];
    // (the workgroup is a chain of memory round trips -- 12.9 us against the 7 us of the dJMax block beside it: the entry's own
    // start value is requested before the wait for the staged pairs, and the diagonal sums read the entries from an LDS tile
    // instead of from memory again, where the atom's slab fits it)
    __shared__ double sG[APPLY_TILE];
    const bool tileFits = Nl * Nl * APPLY_K <= (int)APPLY_TILE;
    double v0 = 0.0;
    if ((int)threadIdx.x < Nl * Nl * APPLY_K)
    {
        const int e = threadIdx.x / APPLY_K;
        const int k = k0 + ((int)threadIdx.x - e * APPLY_K);
        if (k < Ns)
            v0 = a.prefill ? a.crsw * Cm[(size_t)e * Ns + k] : G[(size_t)e * Ns + k];
    }
    for (int q = threadIdx.x; q < min(tr1 - tr0, (int)APPLY_TR); q += blockDim.x)
    {
        sLi[q] = (short)a.transLi[tr0 + q];
        sLj[q] = (short)a.transLj[tr0 + q];
    }
    __syncthreads();
    // off-diagonal entries: pre-fill (optional) + radiative terms of the transitions that feed them
    for (int idx = threadIdx.x; idx < Nl * Nl * APPLY_K; idx += blockDim.x)
    {
        const int e = idx / APPLY_K;
        const int k = k0 + (idx - e * APPLY_K);
        if (k >= Ns)
            continue;
        const int to = e / Nl, from = e - to * Nl;
        double v = (idx == (int)threadIdx.x) ? v0 : (a.prefill ? a.crsw * Cm[(size_t)e * Ns + k] : G[(size_t)e * Ns + k]);
        for (int tr = tr0; tr < tr1; ++tr) // transitions in reference order
        {
            const int i = tr - tr0 < APPLY_TR ? (int)sLi[tr - tr0] : a.transLi[tr], j = tr - tr0 < APPLY_TR ? (int)sLj[tr - tr0] : a.transLj[tr];
            if (to == i && from == j)
                v += redAt(tr, 0, k);
            if (to == j && from == i)
                v += redAt(tr, 1, k);
        }
        G[(size_t)e * Ns + k] = v;
        if (tileFits)
            sG[idx] = v;
    }
#ifdef AP_ABL_NO_DIAG
    if (a.Ns != 77)
        return;
#endif
    __syncthreads();
    // finalise_Gamma (SimdFullIterationTemplates.hpp:491-508): Gamma(i,i) = -sum_{j != i} Gamma(j,i)
    for (int idx = threadIdx.x; idx < Nl * APPLY_K; idx += blockDim.x)
    {
        const int i = idx / APPLY_K;
        const int k = k0 + (idx - i * APPLY_K);
        if (k >= Ns)
            continue;
        double gd = 0.0;
        for (int j = 0; j < Nl; ++j)
            if (j != i)
                gd += tileFits ? sG[(j * Nl + i) * APPLY_K + (k - k0)] : G[((size_t)j * Nl + i) * Ns + k];
        G[((size_t)i * Nl + i) * Ns + k] = -gd;
    }
}

template <bool BATCH, int K = APPLY_K>
__global__ void __launch_bounds__(APPLY_T) apply_kernel(const ApplyArgs a0, const ApplyArgs* __restrict__ list)
{
    dbg_poison_lds();
    if constexpr (BATCH)
        apply_kernel_body<K>(CTAB(ApplyArgs, list)[blockIdx.z]);
    else
        apply_kernel_body<K>(a0);
}

hipError_t launch_apply(const ApplyArgs& a, hipStream_t stream, const ApplyArgs* list, int nBatch)
{
    if (list)
        LWHIP_LAUNCH(apply_kernel<true>, dim3(a.Natom + 1, (a.Ns + APPLY_K - 1) / APPLY_K, std::max(nBatch, 1)), dim3(APPLY_T), 0,
                           stream, a, list);
    else
    {
        // (+ the dJMax workgroup, + one workgroup per PRD line whose change is reduced here)
        const int nx = a.nPrdChange > 0 ? a.Natom + 2 + a.nPrdChange : a.Natom + 1 + ((a.dJ || a.gatherWorld) ? 1 : 0);
        if (a.Ns >= 2048)
            LWHIP_LAUNCH((apply_kernel<false, APPLY_K_WIDE>), dim3(nx, (a.Ns + APPLY_K_WIDE - 1) / APPLY_K_WIDE), dim3(APPLY_T), 0, stream, a, list);
        else
            LWHIP_LAUNCH(apply_kernel<false>, dim3(nx, (a.Ns + APPLY_K - 1) / APPLY_K), dim3(APPLY_T), 0, stream, a, list);
    }
    return hipGetLastError();
}

// ---- the fingerprint of include/lwhip.h (lwhip_fingerprint_J) on the device ---------------------------------------------------
// One workgroup per block of LWHIP_FP_BLOCK doubles.  The host's loop is, per lane l of 8 and row m (word 8 m + l):
// s1 += w; s2 += s1, from s1 = l + 1, s2 = 0 -- so after R_l rows s1 = l + 1 + sum w_m and s2 = R_l (l + 1) + sum (R_l - m) w_m
// (mod 2^64): two weighted sums, which threads form over rows m = t, t + T, ... and a tree adds up.
__global__ void __launch_bounds__(256) fp_blocks_kernel(const double* __restrict__ p, size_t n, unsigned long long* __restrict__ sums)
{
    dbg_poison_lds();
    __shared__ unsigned long long sh[256][16];
    const size_t off = (size_t)blockIdx.x * LWHIP_FP_BLOCK;
    const size_t len = min((size_t)LWHIP_FP_BLOCK, n - off);
    const size_t rowsFull = len / 8, rem = len - rowsFull * 8;
    unsigned long long a1[8], a2[8];
#pragma unroll
    for (int l = 0; l < 8; ++l)
        a1[l] = a2[l] = 0ull;
    const unsigned long long* w = (const unsigned long long*)(p + off);
    const size_t rows = rowsFull + (rem ? 1 : 0);
    for (size_t m = threadIdx.x; m < rows; m += blockDim.x)
    {
#pragma unroll
        for (int l = 0; l < 8; ++l)
        {
            const bool have = m < rowsFull || (size_t)l < rem;
            const unsigned long long R = rowsFull + ((size_t)l < rem ? 1 : 0);
            const unsigned long long x = have ? w[m * 8 + l] : 0ull;
            a1[l] += x;
            a2[l] += (R - m) * x;
        }
    }
#pragma unroll
    for (int l = 0; l < 8; ++l)
    {
        sh[threadIdx.x][l] = a1[l];
        sh[threadIdx.x][8 + l] = a2[l];
    }
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1)
    {
        if ((int)threadIdx.x < s)
            for (int q = 0; q < 16; ++q)
                sh[threadIdx.x][q] += sh[threadIdx.x + s][q];
        __syncthreads();
    }
    if (threadIdx.x < 16)
    {
        const int l = threadIdx.x & 7;
        const unsigned long long R = rowsFull + ((size_t)l < rem ? 1 : 0);
        const unsigned long long init = (unsigned long long)(l + 1);
        sums[(size_t)blockIdx.x * 16 + threadIdx.x] = threadIdx.x < 8 ? init + sh[0][threadIdx.x] : R * init + sh[0][threadIdx.x];
    }
}

hipError_t launch_fp_blocks(const double* p, size_t n, unsigned long long* sums, hipStream_t stream)
{
    if (!n)
        return hipSuccess;
    const unsigned nb = (unsigned)((n + LWHIP_FP_BLOCK - 1) / LWHIP_FP_BLOCK);
    LWHIP_LAUNCH(fp_blocks_kernel, dim3(nb), dim3(256), 0, stream, p, n, sums);
    return hipGetLastError();
}

// ---- diagnosis (LWHIP_CHECK_ZERO=1): the invariant "the stage-1 buffer holds zeros when a sweep starts adding to it" checked
// on the device: every non-zero (or NaN) word counts into a host-mapped counter that the library reads at its next
// synchronising call and reports as LWHIP_ERR_DEVICE ------------------------------------------------------------------
__global__ void check_zero_kernel(const double* __restrict__ p, size_t n, int* __restrict__ counter)
{
    dbg_poison_lds();
    int bad = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        bad += (__double_as_longlong(p[i]) & 0x7fffffffffffffffLL) != 0 ? 1 : 0;
    if (bad)
        atomicAdd_system(counter, bad);
}

hipError_t launch_check_zero(const double* p, size_t n, int* counter, hipStream_t stream)
{
    if (!n)
        return hipSuccess;
    const int blocks = (int)((n + 255) / 256);
    LWHIP_LAUNCH(check_zero_kernel, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, stream, p, n, counter);
    return hipGetLastError();
}

// ---- Gamma <- crsw * C (LwMiddleLayer.pyx:3198-3203) ------------------------------------------------------------
__global__ void prefill_kernel(double* G, const double* C, double crsw, size_t n)
{
    dbg_poison_lds();
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        G[i] = crsw * C[i];
}

hipError_t launch_prefill(double* Gamma, const double* C, double crsw, size_t n, hipStream_t stream)
{
    const int blocks = (int)((n + 255) / 256);
    LWHIP_LAUNCH(prefill_kernel, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, stream, Gamma, C, crsw, n);
    return hipGetLastError();
}

}

namespace lwhip
{
// The pieces of a gathered upload from the inbox to their buffers: one workgroup per record (<= H2D_CHUNK bytes), the widest
// access the three alignments allow.
__global__ void __launch_bounds__(256) h2d_scatter_kernel(const unsigned char* __restrict__ inbox, const H2DRec* __restrict__ recs)
{
    const H2DRec r = recs[blockIdx.x];
    const unsigned char* s = inbox + r.off;
    unsigned char* d = (unsigned char*)r.dst;
    const unsigned long long al = r.dst | r.off | r.bytes;
    if ((al & 15) == 0)
    {
        const uint4* s4 = (const uint4*)s;
        uint4* d4 = (uint4*)d;
        for (size_t i = threadIdx.x; i < r.bytes / 16; i += blockDim.x)
            d4[i] = s4[i];
    }
    else if ((al & 7) == 0)
    {
        const unsigned long long* s8 = (const unsigned long long*)s;
        unsigned long long* d8 = (unsigned long long*)d;
        for (size_t i = threadIdx.x; i < r.bytes / 8; i += blockDim.x)
            d8[i] = s8[i];
    }
    else if ((al & 3) == 0)
    {
        const unsigned int* s1 = (const unsigned int*)s;
        unsigned int* d1 = (unsigned int*)d;
        for (size_t i = threadIdx.x; i < r.bytes / 4; i += blockDim.x)
            d1[i] = s1[i];
    }
    else
        for (size_t i = threadIdx.x; i < r.bytes; i += blockDim.x)
            d[i] = s[i];
}
hipError_t launch_h2d_scatter(const unsigned char* inbox, size_t recsOff, int n, hipStream_t stream)
{
    if (n <= 0)
        return hipSuccess;
    LWHIP_LAUNCH(h2d_scatter_kernel, dim3(n), dim3(256), 0, stream, inbox, (const H2DRec*)(inbox + recsOff));
    return hipGetLastError();
}
}

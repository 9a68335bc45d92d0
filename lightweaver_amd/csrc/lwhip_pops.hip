// lwhip_pops.hip -- population updates: statistical equilibrium, the time-dependent (backward Euler)
// update and the Newton-Raphson charge-conservation step.  One thread per depth point; every
// variant assembles a small dense system and solves it with the reference's solver:
// Crout LU with implicit row scaling and partial pivoting + one residual-correction pass
// (solve_lin_eq / lu_decompose / lu_backsub, Source/LuSolve.cpp:8-132).  The systems are tiny
// (N = Nlevel = 6, or sum Nlevel + 1 = 13), so this is latency-, not throughput-bound (< 1 % of an
// iteration); coalescing comes from the depth index being the fastest axis of every array.
#include "lwhip_device.h"
#include "../../include/lwhip.h"

// These systems are ill conditioned at large dt / strong rates (kappa up to ~1e8), so rounding-level
// differences are amplified to 1e-9 .. 1e-7 in the populations.  No fused multiply-adds here: the
// operations then match the reference's (and the oracle's) one for one, and so do the results.
#pragma clang fp contract(off)

namespace lwhip
{
// Per-thread workspace in LDS: element e of thread t lives at base[e * stride + t] (stride = block
// size), so the threads of a wavefront never collide on a bank and the dynamically indexed rows of
// the pivoting LU cost an LDS access instead of a scratch (HBM-backed) one.
struct LdsVec
{
    double* p;
    int stride;
    DEVINL double& operator[](int e) const { return p[(size_t)e * stride]; }
};
// doubles of workspace per thread for an N x N system: A, A0 (N^2 each), b, b0, res, vv, index (N each)
__host__ __device__ inline size_t solve_ws_doubles(int N) { return (size_t)2 * N * N + 5 * N; }
struct SolveWs
{
    LdsVec A, A0, b, b0, res, vv, index;
    DEVINL SolveWs(double* base, int N, int stride, int tid)
    {
        double* q = base + tid;
        A = { q, stride };
        q += (size_t)N * N * stride;
        A0 = { q, stride };
        q += (size_t)N * N * stride;
        b = { q, stride };
        q += (size_t)N * stride;
        b0 = { q, stride };
        q += (size_t)N * stride;
        res = { q, stride };
        q += (size_t)N * stride;
        vv = { q, stride };
        q += (size_t)N * stride;
        index = { q, stride };
    }
};

// solve A x = b in place (x returned in b); A is destroyed.  Returns false for a singular matrix
// (an all-zero row, where the reference throws "Singular Matrix", LuSolve.cpp:22-23).
DEVINL bool d_solve_lin_eq(int N, const SolveWs& w)
{
    const LdsVec& A = w.A;
    for (int i = 0; i < N; ++i)
    {
        w.b0[i] = w.b[i];
        for (int j = 0; j < N; ++j)
            w.A0[i * N + j] = A[i * N + j];
    }
    // lu_decompose :8-70
    bool singular = false;
    for (int i = 0; i < N; ++i)
    {
        double big = 0.0;
        for (int j = 0; j < N; ++j)
            big = fmax(big, fabs(A[i * N + j]));
        if (big == 0.0)
            singular = true;
        w.vv[i] = 1.0 / big;
    }
    if (singular)
        return false;
    for (int j = 0; j < N; ++j)
    {
        for (int i = 0; i < j; ++i)
        {
            double sum = A[i * N + j];
            for (int q = 0; q < i; ++q)
                sum -= A[i * N + q] * A[q * N + j];
            A[i * N + j] = sum;
        }
        int iMax = 0;
        double big = 0.0;
        for (int i = j; i < N; ++i)
        {
            double sum = A[i * N + j];
            for (int q = 0; q < j; ++q)
                sum -= A[i * N + q] * A[q * N + j];
            A[i * N + j] = sum;
            const double cand = w.vv[i] * fabs(sum);
            if (big < cand)
            {
                iMax = i;
                big = cand;
            }
        }
        if (j != iMax)
        {
            for (int q = 0; q < N; ++q)
            {
                const double tmp = A[iMax * N + q];
                A[iMax * N + q] = A[j * N + q];
                A[j * N + q] = tmp;
            }
            w.vv[iMax] = w.vv[j];
        }
        w.index[j] = (double)iMax;
        if (A[j * N + j] == 0.0)
            A[j * N + j] = 1e-20;
        const double tmp = 1.0 / A[j * N + j];
        for (int i = j + 1; i < N; ++i)
            A[i * N + j] *= tmp;
    }
    // lu_backsub :72-101
    auto backsub = [&](const LdsVec& x) {
        int ii = -1;
        for (int i = 0; i < N; ++i)
        {
            const int ip = (int)w.index[i];
            double sum = x[ip];
            x[ip] = x[i];
            if (ii >= 0)
            {
                for (int j = ii; j < i; ++j)
                    sum -= A[i * N + j] * x[j];
            }
            else if (sum != 0.0)
            {
                ii = i;
            }
            x[i] = sum;
        }
        for (int i = N - 1; i >= 0; --i)
        {
            double sum = x[i];
            for (int j = i + 1; j < N; ++j)
                sum -= A[i * N + j] * x[j];
            x[i] = sum / A[i * N + i];
        }
    };
    backsub(w.b);
    // one pass of iterative improvement :114-131
    for (int i = 0; i < N; ++i)
    {
        double r = w.b0[i];
        for (int j = 0; j < N; ++j)
            r -= w.A0[i * N + j] * w.b[j];
        w.res[i] = r;
    }
    backsub(w.res);
    for (int i = 0; i < N; ++i)
        w.b[i] += w.res[i];
    return true;
}


// The same solver with the system in REGISTERS, for the small level counts (N <= 6: H and Ca II of the benchmark): every
// loop is unrolled, the dynamically indexed accesses of the pivoting (row iMax, x[index[i]]) become selects over the
// rows.  Operation for operation the routine above (each element's sums are formed in the same order), so the results
// are the same bits; what changes is the latency: an LDS round trip per matrix element made the 6 x 6 solve of one
// depth point ~25 us, here it is a straight-line stream of ~1 500 instructions.
template <int N> DEVINL bool d_solve_lin_eq_reg(double (&A)[N][N], double (&b)[N])
{
    double A0[N][N], b0[N], vv[N], res[N];
    int index[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
        b0[i] = b[i];
#pragma unroll
        for (int j = 0; j < N; ++j)
            A0[i][j] = A[i][j];
    }
    // lu_decompose :8-70
    bool singular = false;
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
        double big = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j)
            big = fmax(big, fabs(A[i][j]));
        if (big == 0.0)
            singular = true;
        vv[i] = 1.0 / big;
    }
    if (singular)
        return false;
#pragma unroll
    for (int j = 0; j < N; ++j)
    {
#pragma unroll
        for (int i = 0; i < j; ++i)
        {
            double sum = A[i][j];
#pragma unroll
            for (int q = 0; q < i; ++q)
                sum -= A[i][q] * A[q][j];
            A[i][j] = sum;
        }
        int iMax = 0;
        double big = 0.0;
#pragma unroll
        for (int i = j; i < N; ++i)
        {
            double sum = A[i][j];
#pragma unroll
            for (int q = 0; q < j; ++q)
                sum -= A[i][q] * A[q][j];
            A[i][j] = sum;
            const double cand = vv[i] * fabs(sum);
            if (big < cand)
            {
                iMax = i;
                big = cand;
            }
        }
        // rows j and iMax change places (iMax stays 0 when no candidate is positive: the reference then swaps with row 0)
#pragma unroll
        for (int r = 0; r < N; ++r)
        {
            if (r == j)
                continue;
            const bool sw = iMax == r;
#pragma unroll
            for (int q = 0; q < N; ++q)
            {
                const double ar = A[r][q], aj = A[j][q];
                A[r][q] = sw ? aj : ar;
                A[j][q] = sw ? ar : aj;
            }
            vv[r] = sw ? vv[j] : vv[r];
        }
        index[j] = iMax;
        if (A[j][j] == 0.0)
            A[j][j] = 1e-20;
        const double tmp = 1.0 / A[j][j];
#pragma unroll
        for (int i = j + 1; i < N; ++i)
            A[i][j] *= tmp;
    }
    // lu_backsub :72-101
    auto backsub = [&](double (&x)[N]) {
        int ii = -1;
#pragma unroll
        for (int i = 0; i < N; ++i)
        {
            const int ip = index[i];
            double sum = x[i];
#pragma unroll
            for (int r = 0; r < N; ++r)
                sum = (ip == r) ? x[r] : sum;
            const double xi = x[i];
#pragma unroll
            for (int r = 0; r < N; ++r)
                x[r] = (ip == r) ? xi : x[r];
            if (ii >= 0)
            {
#pragma unroll
                for (int j = 0; j < i; ++j)
                    if (j >= ii)
                        sum -= A[i][j] * x[j];
            }
            else if (sum != 0.0)
                ii = i;
            x[i] = sum;
        }
#pragma unroll
        for (int i = N - 1; i >= 0; --i)
        {
            double sum = x[i];
#pragma unroll
            for (int j = i + 1; j < N; ++j)
                sum -= A[i][j] * x[j];
            x[i] = sum / A[i][i];
        }
    };
    backsub(b);
    // one pass of iterative improvement :114-131
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
        double r = b0[i];
#pragma unroll
        for (int j = 0; j < N; ++j)
            r -= A0[i][j] * b[j];
        res[i] = r;
    }
    backsub(res);
#pragma unroll
    for (int i = 0; i < N; ++i)
        b[i] += res[i];
    return true;
}
enum { SOLVE_REG_MAXN = 6 };

// threads per block such that the workspace fits the LDS (a power of two between 1 and 64)
static int solve_block_threads(int N)
{
    const size_t budget = 150 * 1024;
    int tb = 64;
    while (tb > 1 && solve_ws_doubles(N) * sizeof(double) * tb > budget)
        tb >>= 1;
    return tb;
}
static hipError_t solve_set_lds(const void* fn, size_t lds)
{
    if (lds > 48 * 1024)
        return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return hipSuccess;
}

// max |(new - old) / new| over the populations a block has just updated, with the flattened [level, depth]
// index of its first occurrence (Ng::max_change, Source/Ng.hpp:138-156, which LwContext.stat_equil /
// time_dep_update report as dPops): one (value, index) pair per block, combined on the host.
DEVINL void report_change(double best, int bestIdx, double* out /* [gridDim.x][2] of this atom, or null */)
{
    if (!out)
        return;
    __shared__ double sV[64];
    __shared__ int sI[64];
    sV[threadIdx.x] = best;
    sI[threadIdx.x] = bestIdx;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double v = 0.0;
        int idx = 0x7fffffff;
        for (int q = 0; q < (int)blockDim.x; ++q)
            if (sV[q] > v || (sV[q] == v && sV[q] > 0.0 && sI[q] < idx))
            {
                v = sV[q];
                idx = sI[q];
            }
        out[2 * blockIdx.x] = v;
        out[2 * blockIdx.x + 1] = (v > 0.0) ? (double)idx : 0.0;
    }
}

// ---- stat_eq_impl: Source/UpdatePopulations.cpp:7-47 -------------------------------------------------------
// one depth point of an atom with N <= 6 levels, the system in registers
template <int N>
DEVINL bool stat_eq_point_reg(double* n, const double* Gamma, const double nTot, const int Ns, const int k, double& best, int& bestIdx)
{
    double A[N][N], b[N], old[N];
    // Gamma_k and the elimination row: the level with the largest population
    int iElim = 0;
    double nMax = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
        old[i] = n[(size_t)i * Ns + k];
        if (nMax < old[i])
        {
            iElim = i;
            nMax = old[i];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
#pragma unroll
        for (int j = 0; j < N; ++j)
        {
            const double g = Gamma[((size_t)i * N + j) * Ns + k];
            A[i][j] = (i == iElim) ? 1.0 : g;
        }
        b[i] = (i == iElim) ? nTot : 0.0;
    }
    if (!d_solve_lin_eq_reg<N>(A, b))
        return false;
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
        const double cur = b[i];
        n[(size_t)i * Ns + k] = cur;
        if (cur != 0.0)
        {
            const double change = fabs((cur - old[i]) / cur);
            if (change > best)
            {
                best = change;
                bestIdx = i * Ns + k;
            }
        }
    }
    return true;
}

// blockIdx.y selects the atom, so every active atom is solved by one launch
template <class A> DEVINL void stat_eq_body(const A& a)
{
    extern __shared__ double lds[];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int Ns = a.Ns;
    const NrAtom at = a.atoms[blockIdx.y];
    const int N = at.Nlevel;
    double best = 0.0;
    int bestIdx = 0x7fffffff;
    if (k >= a.k0 && k < a.k1 && N >= 2 && N <= SOLVE_REG_MAXN)
    {
        double* n = a.n + (size_t)at.levelOff * Ns;
        const double* Gamma = a.Gamma + at.gammaOff;
        const double nTot = a.nTotal[(size_t)at.atom * Ns + k];
        bool ok = true;
        switch (N)
        {
        case 2: ok = stat_eq_point_reg<2>(n, Gamma, nTot, Ns, k, best, bestIdx); break;
        case 3: ok = stat_eq_point_reg<3>(n, Gamma, nTot, Ns, k, best, bestIdx); break;
        case 4: ok = stat_eq_point_reg<4>(n, Gamma, nTot, Ns, k, best, bestIdx); break;
        case 5: ok = stat_eq_point_reg<5>(n, Gamma, nTot, Ns, k, best, bestIdx); break;
        default: ok = stat_eq_point_reg<6>(n, Gamma, nTot, Ns, k, best, bestIdx); break;
        }
        if (!ok)
            atomicExch(a.status, LWHIP_ERR_SINGULAR);
    }
    else if (k >= a.k0 && k < a.k1)
    {
        double* n = a.n + (size_t)at.levelOff * Ns;
        const double* Gamma = a.Gamma + at.gammaOff;
        const SolveWs w(lds, N, blockDim.x, threadIdx.x);
        // Gamma_k and the elimination row: the level with the largest population
        int iElim = 0;
        double nMax = 0.0;
        for (int i = 0; i < N; ++i)
        {
            const double ni = n[(size_t)i * Ns + k];
            if (nMax < ni)
            {
                iElim = i;
                nMax = ni;
            }
            for (int j = 0; j < N; ++j)
                w.A[i * N + j] = Gamma[((size_t)i * N + j) * Ns + k];
        }
        for (int i = 0; i < N; ++i)
        {
            w.A[iElim * N + i] = 1.0;
            w.b[i] = 0.0;
        }
        w.b[iElim] = a.nTotal[(size_t)at.atom * Ns + k];
        if (!d_solve_lin_eq(N, w))
            atomicExch(a.status, LWHIP_ERR_SINGULAR);
        else
        {
            for (int i = 0; i < N; ++i)
            {
                const double cur = w.b[i];
                const double old = n[(size_t)i * Ns + k];
                n[(size_t)i * Ns + k] = cur;
                if (cur != 0.0)
                {
                    const double change = fabs((cur - old) / cur);
                    if (change > best)
                    {
                        best = change;
                        bestIdx = i * Ns + k;
                    }
                }
            }
        }
    }
    report_change(best, bestIdx, a.change ? a.change + (size_t)blockIdx.y * 2 * gridDim.x : nullptr);
}

// BATCH: column batches -- blockIdx.z picks the column's argument block (read through the constant address space)
// (blocks of at most 64 threads -- solve_block_threads --: the register form of the 6 x 6 solve may use the whole file)
template <bool BATCH> __global__ void __launch_bounds__(64) stat_eq_kernel(const StatEqArgs a0, const StatEqArgs* __restrict__ list)
{
    dbg_poison_lds();
    if constexpr (BATCH)
        stat_eq_body(CTAB(StatEqArgs, list)[blockIdx.z]);
    else
        stat_eq_body(a0);
}

int stat_eq_blocks(int Ns, int maxNlevel)
{
    const int tb = solve_block_threads(maxNlevel);
    return (Ns + tb - 1) / tb;
}

hipError_t launch_stat_eq(const StatEqArgs& a, int maxNlevel, hipStream_t stream, const StatEqArgs* list, int nBatch)
{
    if (maxNlevel > 64 || a.Natoms <= 0)
        return hipErrorInvalidValue;
    const int tb = solve_block_threads(maxNlevel);
    const size_t lds = solve_ws_doubles(maxNlevel) * sizeof(double) * tb;
    hipError_t e = solve_set_lds(list ? (const void*)stat_eq_kernel<true> : (const void*)stat_eq_kernel<false>, lds);
    if (e != hipSuccess)
        return e;
    if (list)
        LWHIP_LAUNCH(stat_eq_kernel<true>, dim3((a.Ns + tb - 1) / tb, a.Natoms, nBatch > 0 ? nBatch : 1), dim3(tb), lds, stream, a,
                           list);
    else
        LWHIP_LAUNCH(stat_eq_kernel<false>, dim3((a.Ns + tb - 1) / tb, a.Natoms), dim3(tb), lds, stream, a, list);
    return hipGetLastError();
}

// ---- time_dependent_update_impl: Source/UpdatePopulations.cpp:120-151 --------------------------------------
template <int N>
DEVINL bool time_dep_point_reg(double* n, const double* nOld, const double* Gamma, const double dt, const int Ns, const int k)
{
    double A[N][N], b[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
        b[i] = nOld[(size_t)i * Ns + k];
#pragma unroll
        for (int j = 0; j < N; ++j)
        {
            const double g = Gamma[((size_t)i * N + j) * Ns + k];
            A[i][j] = (i == j) ? 1.0 - g * dt : -g * dt;
        }
    }
    if (!d_solve_lin_eq_reg<N>(A, b))
        return false;
#pragma unroll
    for (int i = 0; i < N; ++i)
        n[(size_t)i * Ns + k] = b[i];
    return true;
}

__global__ void __launch_bounds__(64) time_dep_kernel(int N, int Ns, int k0, int k1, double* n, const double* nOld, const double* Gamma,
                                double dt, int* status)
{
    dbg_poison_lds();
    extern __shared__ double lds[];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < k0 || k >= k1)
        return;
    if (N >= 2 && N <= SOLVE_REG_MAXN)
    {
        bool ok = true;
        switch (N)
        {
        case 2: ok = time_dep_point_reg<2>(n, nOld, Gamma, dt, Ns, k); break;
        case 3: ok = time_dep_point_reg<3>(n, nOld, Gamma, dt, Ns, k); break;
        case 4: ok = time_dep_point_reg<4>(n, nOld, Gamma, dt, Ns, k); break;
        case 5: ok = time_dep_point_reg<5>(n, nOld, Gamma, dt, Ns, k); break;
        default: ok = time_dep_point_reg<6>(n, nOld, Gamma, dt, Ns, k); break;
        }
        if (!ok)
            atomicExch(status, LWHIP_ERR_SINGULAR);
        return;
    }
    const SolveWs w(lds, N, blockDim.x, threadIdx.x);
    for (int i = 0; i < N; ++i)
    {
        w.b[i] = nOld[(size_t)i * Ns + k];
        for (int j = 0; j < N; ++j)
            w.A[i * N + j] = -Gamma[((size_t)i * N + j) * Ns + k] * dt;
        w.A[i * N + i] = 1.0 - Gamma[((size_t)i * N + i) * Ns + k] * dt;
    }
    if (!d_solve_lin_eq(N, w))
    {
        atomicExch(status, LWHIP_ERR_SINGULAR);
        return;
    }
    for (int i = 0; i < N; ++i)
        n[(size_t)i * Ns + k] = w.b[i];
}

hipError_t launch_time_dep(int Nlevel, int Ns, int k0, int k1, double* n, const double* nOld, const double* Gamma,
                           double dt, int* status, hipStream_t stream)
{
    if (Nlevel > 64)
        return hipErrorInvalidValue;
    const int tb = solve_block_threads(Nlevel);
    const size_t lds = solve_ws_doubles(Nlevel) * sizeof(double) * tb;
    hipError_t e = solve_set_lds((const void*)time_dep_kernel, lds);
    if (e != hipSuccess)
        return e;
    LWHIP_LAUNCH(time_dep_kernel, dim3((Ns + tb - 1) / tb), dim3(tb), lds, stream, Nlevel, Ns, k0, k1, n, nOld, Gamma,
                       dt, status);
    return hipGetLastError();
}

// ---- nr_post_update_impl with F / Ftd: Source/UpdatePopulations.cpp:230-394 --------------------------------
__global__ void nr_post_kernel(const NrArgs a)
{
    dbg_poison_lds();
    extern __shared__ double lds[];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int Ns = a.Ns;
    if (k < a.k0 || k >= a.k1)
        return;
    const int Neqn = a.Neqn;
    const SolveWs w(lds, Neqn, blockDim.x, threadIdx.x);
    const LdsVec& dF = w.A;
    const LdsVec& Fg = w.b;
    for (int i = 0; i < Neqn; ++i)
    {
        Fg[i] = 0.0;
        for (int j = 0; j < Neqn; ++j)
            dF[i * Neqn + j] = 0.0;
    }
    const double theta = 1.0;
    const double ne = a.ne[k];
    Fg[Neqn - 1] = ne;
    int start = 0;
    for (int q = 0; q < a.Natoms; ++q)
    {
        const NrAtom at = a.atoms[q];
        const int Nl = at.Nlevel;
        const double* G = a.Gamma + at.gammaOff;
        const double* n = a.n + (size_t)at.levelOff * Ns;
        // F :230-257 / Ftd :259-292
        for (int l = 0; l < Nl; ++l)
        {
            double v = 0.0;
            if (a.timeDep)
            {
                for (int ll = 0; ll < Nl; ++ll)
                    v += G[((size_t)l * Nl + ll) * Ns + k] * n[(size_t)ll * Ns + k];
                v *= theta * a.dt;
                v -= n[(size_t)l * Ns + k] - a.nPrev[(size_t)(at.eqOff + l) * Ns + k];
            }
            else
            {
                for (int ll = 0; ll < Nl; ++ll)
                    v -= G[((size_t)l * Nl + ll) * Ns + k] * n[(size_t)ll * Ns + k];
            }
            Fg[start + l] = v;
        }
        double nTotCur = 0.0;
        for (int ll = 0; ll < Nl; ++ll)
            nTotCur += n[(size_t)ll * Ns + k];
        Fg[start + Nl - 1] = nTotCur - a.nTotal[(size_t)at.atom * Ns + k];
        double eleContrib = 0.0;
        for (int ll = 0; ll < Nl; ++ll)
            eleContrib += a.stages[at.eqOff + ll] * n[(size_t)ll * Ns + k];
        Fg[Neqn - 1] -= eleContrib;
        // Jacobian :322-372
        for (int l = 0; l < Nl; ++l)
            for (int ll = 0; ll < Nl; ++ll)
                dF[(start + l) * Neqn + start + ll] = -G[((size_t)l * Nl + ll) * Ns + k];
        if (a.timeDep)
        {
            for (int l = 0; l < Nl; ++l)
                for (int ll = 0; ll < Nl; ++ll)
                    dF[(start + l) * Neqn + start + ll] *= -theta * a.dt;
            for (int l = 0; l < Nl; ++l)
                dF[(start + l) * Neqn + start + l] -= 1.0;
        }
        const double* Cm = a.Cmat + at.gammaOff;
        for (int tr = at.trBegin; tr < at.trEnd; ++tr)
        {
            if (a.transType[tr] != LWHIP_CONTINUUM)
                continue;
            const int ti = a.transLi[tr], tj = a.transLj[tr];
            const double preconRji = G[((size_t)ti * Nl + tj) * Ns + k] - a.crsw * Cm[((size_t)ti * Nl + tj) * Ns + k];
            double entry = -(preconRji / ne) * n[(size_t)tj * Ns + k];
            if (a.timeDep)
                entry *= -theta * a.dt;
            dF[(start + ti) * Neqn + Neqn - 1] += entry;
        }
        if (a.dC)
        {
            const double* dC = a.dC + (size_t)at.dcOff * Ns;
            for (int i = 0; i < Nl; ++i)
            {
                double entry = 0.0;
                for (int ll = 0; ll < Nl; ++ll)
                    entry -= dC[((size_t)i * Nl + ll) * Ns + k] * n[(size_t)ll * Ns + k];
                if (a.timeDep)
                    entry *= -theta * a.dt;
                dF[(start + i) * Neqn + Neqn - 1] += entry;
            }
        }
        for (int c = 0; c < Neqn; ++c)
            dF[(start + Nl - 1) * Neqn + c] = 0.0;
        for (int ll = 0; ll < Nl; ++ll)
        {
            dF[(start + Nl - 1) * Neqn + start + ll] = 1.0;
            dF[(Neqn - 1) * Neqn + start + ll] = -a.stages[at.eqOff + ll];
        }
        start += Nl;
    }
    Fg[Neqn - 1] -= a.backgroundNe[k];
    dF[(Neqn - 1) * Neqn + Neqn - 1] = 1.0;
    for (int i = 0; i < Neqn; ++i)
        Fg[i] *= -1.0;
    if (!d_solve_lin_eq(Neqn, w))
    {
        atomicExch(a.status, LWHIP_ERR_SINGULAR);
        return;
    }
    start = 0;
    for (int q = 0; q < a.Natoms; ++q)
    {
        const NrAtom at = a.atoms[q];
        double* n = a.n + (size_t)at.levelOff * Ns;
        for (int ll = 0; ll < at.Nlevel; ++ll)
            n[(size_t)ll * Ns + k] += Fg[start + ll];
        start += at.Nlevel;
    }
    a.ne[k] = ne + Fg[Neqn - 1];
}

hipError_t launch_nr_post(const NrArgs& a, hipStream_t stream)
{
    if (a.Neqn > 64)
        return hipErrorInvalidValue;
    const int tb = solve_block_threads(a.Neqn);
    const size_t lds = solve_ws_doubles(a.Neqn) * sizeof(double) * tb;
    hipError_t e = solve_set_lds((const void*)nr_post_kernel, lds);
    if (e != hipSuccess)
        return e;
    LWHIP_LAUNCH(nr_post_kernel, dim3((a.Ns + tb - 1) / tb), dim3(tb), lds, stream, a);
    return hipGetLastError();
}

// ---- Ng acceleration of one atom's populations: Ng::accelerate + Ng::max_change (Source/Ng.hpp:52-156) ------
// One workgroup per atom (the vectors are Nlevel x Nspace ~ 500 long).  The weighted dot products that form
// the Norder x Norder system are accumulated by ONE thread in the reference's order: the system can be
// ill conditioned, and a few microseconds every Nperiod iterations buy bit-identical coefficients.
__global__ void __launch_bounds__(256) ng_kernel(const NgArgs a)
{
    dbg_poison_lds();
    extern __shared__ double lds[];
    const NgAtom at = a.atoms[blockIdx.x];
    const int L = at.len, No = a.Norder;
    double* n = a.n + (size_t)at.nOff;
    double* prev = a.history + (size_t)at.histOff;
    auto slot = [&](int s) { return prev + (size_t)s * L; };
    // store the new solution :62-65
    for (int k = threadIdx.x; k < L; k += blockDim.x)
        slot(a.storeSlot)[k] = n[k];
    __syncthreads();
    if (a.doAccel)
    {
        // slots of count-1-i (ip) and count-2-i (ipp), i = 0 .. Norder, precomputed on the host in a.slots
        double* bco = lds;                       // [No]
        double* ws = lds + 64;                   // solver workspace of thread 0
        if (threadIdx.x == 0)
        {
            const SolveWs w(ws, No, 1, 0);
            auto Delta = [&](int i, int k) { return slot(a.slots[i])[k] - slot(a.slots[i + 1])[k]; };
            for (int j = 0; j < No; ++j)
            {
                double bj = 0.0;
                for (int k = 0; k < L; ++k)
                    bj += (1.0 / fabs(n[k])) * Delta(0, k) * (Delta(0, k) - Delta(j + 1, k));
                w.b[j] = bj;
                for (int i = 0; i < No; ++i)
                {
                    double aij = 0.0;
                    for (int k = 0; k < L; ++k)
                        aij += (1.0 / fabs(n[k])) * (Delta(j + 1, k) - Delta(0, k)) * (Delta(i + 1, k) - Delta(0, k));
                    w.A[i * No + j] = aij;
                }
            }
            const bool ok = d_solve_lin_eq(No, w);
            for (int i = 0; i < No; ++i)
                bco[i] = ok ? (double)w.b[i] : 0.0;
            if (!ok)
                atomicExch(a.status, LWHIP_ERR_SINGULAR);
        }
        __syncthreads();
        // sol += sum_i b_i (previous(count-i-2) - previous(count-1)), then previous(count-1) = sol  :104-113
        double* p0 = slot(a.slots[0]);
        for (int k = threadIdx.x; k < L; k += blockDim.x)
        {
            double v = n[k];
            const double p0k = p0[k];
            for (int i = 0; i < No; ++i)
                v += bco[i] * (slot(a.slots[i + 1])[k] - p0k);
            n[k] = v;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < L; k += blockDim.x)
            p0[k] = n[k];
        __syncthreads();
    }
    // max_change :138-156 between the last two stored solutions
    double best = 0.0;
    int bestIdx = 0x7fffffff;
    if (a.haveTwo)
    {
        const double* old = slot(a.oldSlot);
        const double* cur = slot(a.storeSlot);
        for (int k = threadIdx.x; k < L; k += blockDim.x)
        {
            const double c = cur[k];
            if (c != 0.0)
            {
                const double change = fabs((c - old[k]) / c);
                if (change > best)
                {
                    best = change;
                    bestIdx = k;
                }
            }
        }
    }
    __shared__ double sV[256];
    __shared__ int sI[256];
    sV[threadIdx.x] = best;
    sI[threadIdx.x] = bestIdx;
    __syncthreads();
    for (unsigned s2 = blockDim.x / 2; s2 > 0; s2 >>= 1)
    {
        if (threadIdx.x < s2)
        {
            const double v = sV[threadIdx.x + s2];
            const int i = sI[threadIdx.x + s2];
            if (v > sV[threadIdx.x] || (v == sV[threadIdx.x] && i < sI[threadIdx.x]))
            {
                sV[threadIdx.x] = v;
                sI[threadIdx.x] = i;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        a.change[2 * blockIdx.x] = sV[0];
        a.change[2 * blockIdx.x + 1] = sV[0] > 0.0 ? (double)sI[0] : 0.0;
    }
}

hipError_t launch_ng(const NgArgs& a, int nAtoms, hipStream_t stream)
{
    const size_t lds = (64 + solve_ws_doubles(a.Norder > 0 ? a.Norder : 1)) * sizeof(double);
    LWHIP_LAUNCH(ng_kernel, dim3(nAtoms), dim3(256), lds, stream, a);
    return hipGetLastError();
}
}

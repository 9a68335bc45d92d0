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
// solve A x = b in place (x returned in b); A is destroyed.  Returns false for a singular matrix
// (an all-zero row, where the reference throws "Singular Matrix", LuSolve.cpp:22-23).
template <int MAXN> DEVINL bool d_solve_lin_eq(int N, double* A, double* b)
{
    double A0[MAXN * MAXN], b0[MAXN], res[MAXN], vv[MAXN];
    int index[MAXN];
    for (int i = 0; i < N; ++i)
    {
        b0[i] = b[i];
        for (int j = 0; j < N; ++j)
            A0[i * MAXN + j] = A[i * MAXN + j];
    }
    // lu_decompose :8-70
    bool singular = false;
    for (int i = 0; i < N; ++i)
    {
        double big = 0.0;
        for (int j = 0; j < N; ++j)
            big = fmax(big, fabs(A[i * MAXN + j]));
        if (big == 0.0)
            singular = true;
        vv[i] = 1.0 / big;
    }
    if (singular)
        return false;
    for (int j = 0; j < N; ++j)
    {
        for (int i = 0; i < j; ++i)
        {
            double sum = A[i * MAXN + j];
            for (int q = 0; q < i; ++q)
                sum -= A[i * MAXN + q] * A[q * MAXN + j];
            A[i * MAXN + j] = sum;
        }
        int iMax = 0;
        double big = 0.0;
        for (int i = j; i < N; ++i)
        {
            double sum = A[i * MAXN + j];
            for (int q = 0; q < j; ++q)
                sum -= A[i * MAXN + q] * A[q * MAXN + j];
            A[i * MAXN + j] = sum;
            const double cand = vv[i] * fabs(sum);
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
                const double tmp = A[iMax * MAXN + q];
                A[iMax * MAXN + q] = A[j * MAXN + q];
                A[j * MAXN + q] = tmp;
            }
            vv[iMax] = vv[j];
        }
        index[j] = iMax;
        if (A[j * MAXN + j] == 0.0)
            A[j * MAXN + j] = 1e-20;
        const double tmp = 1.0 / A[j * MAXN + j];
        for (int i = j + 1; i < N; ++i)
            A[i * MAXN + j] *= tmp;
    }
    // lu_backsub :72-101
    auto backsub = [&](double* x) {
        int ii = -1;
        for (int i = 0; i < N; ++i)
        {
            const int ip = index[i];
            double sum = x[ip];
            x[ip] = x[i];
            if (ii >= 0)
            {
                for (int j = ii; j < i; ++j)
                    sum -= A[i * MAXN + j] * x[j];
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
                sum -= A[i * MAXN + j] * x[j];
            x[i] = sum / A[i * MAXN + i];
        }
    };
    backsub(b);
    // one pass of iterative improvement :114-131
    for (int i = 0; i < N; ++i)
    {
        double r = b0[i];
        for (int j = 0; j < N; ++j)
            r -= A0[i * MAXN + j] * b[j];
        res[i] = r;
    }
    backsub(res);
    for (int i = 0; i < N; ++i)
        b[i] += res[i];
    return true;
}

// ---- stat_eq_impl: Source/UpdatePopulations.cpp:7-47 -------------------------------------------------------
template <int MAXN>
__global__ void stat_eq_kernel(int N, int Ns, double* n, const double* nTotal, const double* Gamma, int* status)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Ns)
        return;
    double A[MAXN * MAXN], bvec[MAXN];
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
            A[i * MAXN + j] = Gamma[((size_t)i * N + j) * Ns + k];
    }
    for (int i = 0; i < N; ++i)
    {
        A[iElim * MAXN + i] = 1.0;
        bvec[i] = 0.0;
    }
    bvec[iElim] = nTotal[k];
    if (!d_solve_lin_eq<MAXN>(N, A, bvec))
    {
        atomicExch(status, LWHIP_ERR_SINGULAR);
        return;
    }
    for (int i = 0; i < N; ++i)
        n[(size_t)i * Ns + k] = bvec[i];
}

hipError_t launch_stat_eq(int Nlevel, int Ns, double* n, const double* nTotal, const double* Gamma,
                          int* status, hipStream_t stream)
{
    const int blocks = (Ns + 63) / 64;
    if (Nlevel <= 8)
        hipLaunchKernelGGL(stat_eq_kernel<8>, dim3(blocks), dim3(64), 0, stream, Nlevel, Ns, n, nTotal, Gamma, status);
    else if (Nlevel <= 16)
        hipLaunchKernelGGL(stat_eq_kernel<16>, dim3(blocks), dim3(64), 0, stream, Nlevel, Ns, n, nTotal, Gamma, status);
    else if (Nlevel <= 32)
        hipLaunchKernelGGL(stat_eq_kernel<32>, dim3(blocks), dim3(64), 0, stream, Nlevel, Ns, n, nTotal, Gamma, status);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- time_dependent_update_impl: Source/UpdatePopulations.cpp:120-151 --------------------------------------
template <int MAXN>
__global__ void time_dep_kernel(int N, int Ns, double* n, const double* nOld, const double* Gamma, double dt,
                                int* status)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Ns)
        return;
    double A[MAXN * MAXN], nk[MAXN];
    for (int i = 0; i < N; ++i)
    {
        nk[i] = nOld[(size_t)i * Ns + k];
        for (int j = 0; j < N; ++j)
            A[i * MAXN + j] = -Gamma[((size_t)i * N + j) * Ns + k] * dt;
        A[i * MAXN + i] = 1.0 - Gamma[((size_t)i * N + i) * Ns + k] * dt;
    }
    if (!d_solve_lin_eq<MAXN>(N, A, nk))
    {
        atomicExch(status, LWHIP_ERR_SINGULAR);
        return;
    }
    for (int i = 0; i < N; ++i)
        n[(size_t)i * Ns + k] = nk[i];
}

hipError_t launch_time_dep(int Nlevel, int Ns, double* n, const double* nOld, const double* Gamma, double dt,
                           int* status, hipStream_t stream)
{
    const int blocks = (Ns + 63) / 64;
    if (Nlevel <= 8)
        hipLaunchKernelGGL(time_dep_kernel<8>, dim3(blocks), dim3(64), 0, stream, Nlevel, Ns, n, nOld, Gamma, dt, status);
    else if (Nlevel <= 16)
        hipLaunchKernelGGL(time_dep_kernel<16>, dim3(blocks), dim3(64), 0, stream, Nlevel, Ns, n, nOld, Gamma, dt, status);
    else if (Nlevel <= 32)
        hipLaunchKernelGGL(time_dep_kernel<32>, dim3(blocks), dim3(64), 0, stream, Nlevel, Ns, n, nOld, Gamma, dt, status);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- nr_post_update_impl with F / Ftd: Source/UpdatePopulations.cpp:230-394 --------------------------------
template <int MAXN> __global__ void nr_post_kernel(const NrArgs a)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int Ns = a.Ns;
    if (k >= Ns)
        return;
    const int Neqn = a.Neqn;
    double dF[MAXN * MAXN], Fg[MAXN];
    for (int i = 0; i < Neqn; ++i)
    {
        Fg[i] = 0.0;
        for (int j = 0; j < Neqn; ++j)
            dF[i * MAXN + j] = 0.0;
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
                dF[(start + l) * MAXN + start + ll] = -G[((size_t)l * Nl + ll) * Ns + k];
        if (a.timeDep)
        {
            for (int l = 0; l < Nl; ++l)
                for (int ll = 0; ll < Nl; ++ll)
                    dF[(start + l) * MAXN + start + ll] *= -theta * a.dt;
            for (int l = 0; l < Nl; ++l)
                dF[(start + l) * MAXN + start + l] -= 1.0;
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
            dF[(start + ti) * MAXN + Neqn - 1] += entry;
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
                dF[(start + i) * MAXN + Neqn - 1] += entry;
            }
        }
        for (int c = 0; c < Neqn; ++c)
            dF[(start + Nl - 1) * MAXN + c] = 0.0;
        for (int ll = 0; ll < Nl; ++ll)
        {
            dF[(start + Nl - 1) * MAXN + start + ll] = 1.0;
            dF[(Neqn - 1) * MAXN + start + ll] = -a.stages[at.eqOff + ll];
        }
        start += Nl;
    }
    Fg[Neqn - 1] -= a.backgroundNe[k];
    dF[(Neqn - 1) * MAXN + Neqn - 1] = 1.0;
    for (int i = 0; i < Neqn; ++i)
        Fg[i] *= -1.0;
    if (!d_solve_lin_eq<MAXN>(Neqn, dF, Fg))
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
    const int blocks = (a.Ns + 63) / 64;
    if (a.Neqn <= 16)
        hipLaunchKernelGGL(nr_post_kernel<16>, dim3(blocks), dim3(64), 0, stream, a);
    else if (a.Neqn <= 32)
        hipLaunchKernelGGL(nr_post_kernel<32>, dim3(blocks), dim3(64), 0, stream, a);
    else if (a.Neqn <= 64)
        hipLaunchKernelGGL(nr_post_kernel<64>, dim3(blocks), dim3(64), 0, stream, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
}

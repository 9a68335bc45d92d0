// lwhip_prd.hip -- partial frequency redistribution on the device: the angle-averaged scattering
// integral of prd_scatter / scattering_int (Source/Prd.cpp:468-645) with Gouttebroze's GII
// (Source/Prd.cpp:46-124), the total depopulation + elastic rate (Source/Prd.cpp:9-30) and the
// max relative change of rho that Ng(0,0,0)::max_change reports (Source/Ng.hpp:136-156).
//
// One workgroup per (PRD line, depth point, slice of the line's wavelengths): J(:, k) and the line's wavelengths in Doppler units
// are staged in LDS, then every thread owns one emission wavelength and walks its <= 87-point
// fine grid (spacing 0.15 Doppler widths) exactly as the reference does -- same interpolation,
// same sequentially accumulated q', same end-corrected trapezoid weights -- evaluating GII on the
// fly instead of caching it per (depth, wavelength) as the reference does (the cached values
// depend only on aDamp, vBroad and the grid, so the numbers are the same).
#include "lwhip_device.h"
#include "../../include/lwhip.h"

#include <hip/hip_runtime.h>

namespace lwhip
{
namespace
{
constexpr double PrdQWing = 4.0;
constexpr double PrdQCore = 2.0;
constexpr double PrdQSpread = 5.0;
constexpr double PrdDQ = 0.15;
constexpr int PRD_MAX_FINE = 88; // >= max_fine_grid_size() = 87, Prd.cpp:126-129

__device__ __forceinline__ double p_G_zero(double x)
{
    return 1.0 / (fabs(x) + sqrt(x * x + 1.273239545));
}

// GII: Prd.cpp:51-124 (waveratio = 1)
__device__ double p_GII(double aDamp, double qEmit, double qAbs)
{
    if (qEmit < 0.0)
    {
        qEmit = -qEmit;
        qAbs = -qAbs;
    }
    double giiCore = 0.0;
    double coreFactor = 0.0;
    if (qEmit < PrdQWing)
    {
        if ((qAbs < -PrdQWing) || (qAbs > qEmit + PrdQSpread))
            return 0.0;
        if (fabs(qAbs) <= qEmit)
            giiCore = p_G_zero(qEmit);
        else
            giiCore = exp(qEmit * qEmit - qAbs * qAbs) * p_G_zero(qAbs);
        if (qEmit >= PrdQCore && qEmit <= PrdQWing)
        {
            const double phiCore = exp(-(qEmit * qEmit));
            const double phiWing = aDamp / (sqrt(Pi) * (aDamp * aDamp + qEmit * qEmit));
            coreFactor = phiCore / (phiCore + phiWing);
        }
        else
            return giiCore;
    }
    double gii = 0.0;
    if (qEmit >= PrdQCore)
    {
        if ((qEmit >= PrdQWing) && (fabs(qAbs - qEmit) > PrdQSpread))
            return 0.0;
        const double uMin = fabs((qAbs - qEmit) / 2.0);
        double giiWing = 2.0 * (1.0 - 2.0 * uMin * p_G_zero(uMin)) * exp(-(uMin * uMin)) / (2.0 * sqrt(Pi));
        const double ratio = qAbs / qEmit;
        giiWing *= (2.75 - (2.5 - 0.75 * ratio) * ratio);
        gii = coreFactor * giiCore + (1.0 - coreFactor) * giiWing;
    }
    return gii;
}

// The line's grid in Doppler units at depth k (Prd.cpp:480-483)
DEVINL double prd_q(const double wave, const double lambda0, const double vb)
{
    return (wave - lambda0) * CLight / (lambda0 * vb);
}

// Linear interpolation of J between the line's grid points as ONE multiply-add per fine point: bracket `it` (between grid
// points it - 1 and it) carries (A, B) with J(x) = A + B x.  Both kernels below form (A, B) and evaluate J(x) through these two
// functions with explicit fma, so rho does not depend on which of them ran (filling the cache or reading it).
DEVINL double2 prd_bracket(const double q0, const double q1, const double J0, const double J1)
{
    const double B = (J1 - J0) * (1.0 / (q1 - q0));
    return make_double2(fma(-q0, B, J0), B);
}
DEVINL double prd_interp(const double2 ab, const double x, const double qFirst, const double qLast)
{
    // beyond the table ends the reference takes the end value (Prd.cpp:196-204): the clamped abscissa gives the same
    return fma(ab.y, fmin(fmax(x, qFirst), qLast), ab.x);
}

// scattering_int_range (Prd.cpp:233-263): the fine grid [q0, qN] of an emission wavelength
DEVINL void prd_fine_range(const double qEmit, double& q0, double& qN)
{
    if (fabs(qEmit) < PrdQCore)
    {
        q0 = -PrdQWing;
        qN = PrdQWing;
    }
    else if (fabs(qEmit) < PrdQWing)
    {
        if (qEmit > 0.0)
        {
            q0 = -PrdQWing;
            qN = qEmit + PrdQSpread;
        }
        else
        {
            q0 = qEmit - PrdQSpread;
            qN = PrdQWing;
        }
    }
    else
    {
        q0 = qEmit - PrdQSpread;
        qN = qEmit + PrdQSpread;
    }
}

// total_depop_elastic_scattering_rate (Prd.cpp:9-30) of the line's upper level at depth k, by the whole workgroup (<= 128
// threads fetch the terms in one memory round trip, one thread adds them in the reference's order); returns after a barrier
DEVINL double prd_PjQj(const PrdLineArgs& a, const int k, double* sTerm, double* sOut)
{
    const int Ns = a.Ns;
    const int nTr = a.trEnd - a.trBegin;
    const int nTerm = a.Nlevel + nTr;
    for (int q = threadIdx.x; q < nTerm && q < 128; q += blockDim.x)
    {
        double v = 0.0;
        if (q < a.Nlevel)
            v = a.Cmat ? a.Cmat[((size_t)q * a.Nlevel + a.lj) * Ns + k] : 0.0; // a detailed atom carries no C
        else
        {
            const int tr = a.trBegin + (q - a.Nlevel);
            if (a.transLj[tr] == a.lj)
                v += a.Rji[(size_t)tr * Ns + k];
            if (a.transLi[tr] == a.lj)
                v += a.Rij[(size_t)tr * Ns + k];
        }
        sTerm[q] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double pq = a.Qelast[k];
        if (nTerm <= 128)
        {
            for (int q = 0; q < nTerm; ++q)
                pq += sTerm[q];
        }
        else
        {
            for (int i = 0; i < a.Nlevel && a.Cmat; ++i)
                pq += a.Cmat[((size_t)i * a.Nlevel + a.lj) * Ns + k];
            for (int tr = a.trBegin; tr < a.trEnd; ++tr)
            {
                if (a.transLj[tr] == a.lj)
                    pq += a.Rji[(size_t)tr * Ns + k];
                if (a.transLi[tr] == a.lj)
                    pq += a.Rij[(size_t)tr * Ns + k];
            }
        }
        *sOut = pq;
    }
    __syncthreads();
    return *sOut;
}

// rho of one emission wavelength from its integrals, and the workgroup's largest relative change (first index on ties)
DEVINL void prd_store_rho(const PrdLineArgs& a, const int la, const int k, const double scatInt, const double gNorm,
                          const double gammaPrefactor, const double Jbar, double& best, int& bestIdx)
{
    const size_t o = (size_t)(la - a.laLo) * a.Ns + k;
    const double rhoNew = 1.0 + gammaPrefactor * (scatInt / gNorm - Jbar);
    const double rhoOld = a.rho[o];
    a.rho[o] = rhoNew;
    if (rhoNew != 0.0)
    {
        const double change = fabs((rhoNew - rhoOld) / rhoNew);
        const int flat = la * a.Ns + k;
        if (change > best || (change == best && flat < bestIdx))
        {
            best = change;
            bestIdx = flat;
        }
    }
}
DEVINL void prd_store_change(const PrdLineArgs& a, const int k, double best, int bestIdx, double* sBest, int* sBestIdx)
{
    sBest[threadIdx.x] = best;
    sBestIdx[threadIdx.x] = bestIdx;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1)
    {
        if (threadIdx.x < s)
        {
            const double v = sBest[threadIdx.x + s];
            const int i = sBestIdx[threadIdx.x + s];
            if (v > sBest[threadIdx.x] || (v == sBest[threadIdx.x] && i < sBestIdx[threadIdx.x]))
            {
                sBest[threadIdx.x] = v;
                sBestIdx[threadIdx.x] = i;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        a.changeVal[k * gridDim.y + blockIdx.y] = sBest[0];
        a.changeIdx[k * gridDim.y + blockIdx.y] = sBestIdx[0];
    }
}

}

// All PRD lines of a sub-iteration go through ONE launch of each kernel: blockIdx.z picks the line's argument block from
// a device table read through the constant address space (scalar loads).  One line alone leaves the chip half empty
// (Ns x slices workgroups of two wavefronts); two lines side by side take barely longer than one.

// J(Nblue : Nred, :) -> Jt[k][la]: the scattering integral reads J along wavelength at fixed depth
// (stopCtl / stopIter: the pipelined sub-iterations of lwhip_redistribute_prd -- a launch of a sub-iteration that comes after
// the one that met the tolerance does nothing)
DEVINL bool prd_stopped(const int32_t* stopCtl, const int stopIter)
{
    if (!stopCtl)
        return false;
    const int s = *(const volatile int32_t*)stopCtl;
    return s != 0 && s < stopIter;
}

__global__ void prd_transpose_kernel(const PrdLineArgs* __restrict__ list, const int32_t* stopCtl, const int stopIter)
{
    dbg_poison_lds();
    if (prd_stopped(stopCtl, stopIter))
        return;
    const PrdLineArgs a = ld_c(CTAB(PrdLineArgs, list) + blockIdx.z);
    const double* J = a.J;
    double* Jt = a.Jt;
    const int Nblue = a.Nblue, Nl = a.Nl, Ns = a.Ns;
    if (a.laHi <= a.laLo || (int)blockIdx.x * 32 >= Nl)
        return;
    __shared__ double tile[32][33];
    const int la0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y)
    {
        const int la = la0 + r, k = k0 + threadIdx.x;
        if (la < Nl && k < Ns)
            tile[r][threadIdx.x] = J[(size_t)(la + Nblue) * Ns + k];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y)
    {
        const int k = k0 + r, la = la0 + threadIdx.x;
        if (la < Nl && k < Ns)
            Jt[(size_t)k * Nl + la] = tile[threadIdx.x][r];
    }
}

// The general kernel: computes gII (and, with a cache buffer, stores the weights, the bracket indices and the slice's index
// window); also reads a filled cache, so a launch with filled and unfilled lines side by side needs nothing else.
__global__ void __launch_bounds__(128) prd_scatter_kernel(const PrdLineArgs* __restrict__ list, const int32_t* stopCtl, const int stopIter)
{
    dbg_poison_lds();
    extern __shared__ double sm[];
    if (prd_stopped(stopCtl, stopIter))
        return;
    const PrdLineArgs a = ld_c(CTAB(PrdLineArgs, list) + blockIdx.z);
    if (a.laHi <= a.laLo || (int)blockIdx.y >= a.slices)
        return;
    const int k = blockIdx.x;
    const int Ns = a.Ns;
    const int Nl = a.Nl;
    double* qW = sm;                       // [Nl]
    double2* AB = (double2*)(sm + Nl);     // [Nl]: bracket it = (it - 1, it); entry 0 unused
    __shared__ double sPjQj;
    __shared__ double sBest[128];
    __shared__ int sBestIdx[128];
    __shared__ double sTerm[128];
    __shared__ int sWin[2];
    if (threadIdx.x == 0)
    {
        sWin[0] = Nl - 1;
        sWin[1] = 1;
    }
    const double vb = a.vBroad[k];
    for (int la = threadIdx.x; la < Nl; la += blockDim.x)
    {
        const double q1 = prd_q(a.wave[la], a.lambda0, vb);
        qW[la] = q1;
        if (la > 0)
            AB[la] = prd_bracket(prd_q(a.wave[la - 1], a.lambda0, vb), q1, a.Jt[(size_t)k * Nl + la - 1], a.Jt[(size_t)k * Nl + la]);
    }
    const double pjqj = prd_PjQj(a, k, sTerm, &sPjQj); // (two barriers: the tables above are complete behind them)
    const double qFirst = qW[0], qLast = qW[Nl - 1];
    const double gammaPrefactor = a.n[(size_t)a.gi * Ns + k] / a.n[(size_t)a.gj * Ns + k] * a.Bij / pjqj;
    const double Jbar = a.Rij[(size_t)a.tr * Ns + k] / a.Bij;
    const double aDamp = a.aDamp[k];
    double best = 0.0;
    int bestIdx = 0x7fffffff;
    int itMin = Nl - 1, itMax = 1;
    // this workgroup's share of the emission wavelengths (blockIdx.y of gridDim.y slices)
    const int laPer = (a.laHi - a.laLo + gridDim.y - 1) / gridDim.y;
    const int laBeg = a.laLo + blockIdx.y * laPer;
    const int laEnd = min(a.laHi, laBeg + laPer);
    for (int la = laBeg + threadIdx.x; la < laEnd; la += blockDim.x)
    {
        const double qEmit = qW[la];
        double q0, qN;
        prd_fine_range(qEmit, q0, qN);
        const int Np = (int)((double)(qN - q0) / PrdDQ) + 1;
        // What of the fine grid of this (depth, emission wavelength) does not depend on J is cached, laid out [k][fine point][la]
        // so that a wavefront's reads coalesce: the end-corrected gII weights (the reference caches these too, t->gII,
        // Prd.cpp:499-553) and the bracketing index of the line's grid as 16 bits -- 10 bytes per fine point.
        const size_t cOff = ((size_t)k * PRD_MAX_FINE) * Nl + la;
        double gNorm = 0.0, scatInt = 0.0;
        if (!a.gIIFill)
        {
            const double* gc = a.gII + cOff;
            const int16_t* ic = (const int16_t*)a.gIT + cOff;
            for (int i = 0; i < Np; ++i)
            {
                const double gii = gc[(size_t)i * Nl];
                const int it = (int)ic[(size_t)i * Nl];
                const double Jf = prd_interp(AB[it], q0 + i * PrdDQ, qFirst, qLast);
                gNorm += gii;
                scatInt = fma(Jf, gii, scatInt);
            }
        }
        else
        {
            // optimised_fine_linear_fixed_spacing: Prd.cpp:180-230 (upper bound, then walk upwards)
            int iter;
            if (q0 <= qW[0])
                iter = 0;
            else if (q0 >= qW[Nl - 1])
                iter = Nl - 1;
            else
            {
                int lo = 0, hi = Nl;
                while (lo < hi)
                {
                    const int mid = (lo + hi) / 2;
                    if (qW[mid] <= q0)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                iter = lo;
            }
            double qPrime = q0;
            for (int i = 0; i < Np; ++i)
            {
                const double x = q0 + i * PrdDQ;
                while (iter < Nl && qW[iter] <= x)
                    ++iter;
                // beyond the table ends the reference takes the end value: the last (first) bracket at the clamped abscissa
                const int it = min(max(iter, 1), Nl - 1);
                const double Jf = prd_interp(AB[it], x, qFirst, qLast);
                // end-corrected trapezoid weights, q' accumulated step by step: Prd.cpp:526-551
                if (i > 0)
                    qPrime += PrdDQ;
                double gii;
                if (i == 0 || i == Np - 1)
                    gii = p_GII(aDamp, qEmit, qPrime) * 5.0 / 12.0 * PrdDQ;
                else if (i == 1 || i == Np - 2)
                    gii = p_GII(aDamp, qEmit, qPrime) * 13.0 / 12.0 * PrdDQ;
                else
                    gii = p_GII(aDamp, qEmit, qPrime) * PrdDQ;
                if (a.gII)
                {
                    a.gII[cOff + (size_t)i * Nl] = gii;
                    ((int16_t*)a.gIT)[cOff + (size_t)i * Nl] = (int16_t)it;
                    itMin = min(itMin, it);
                    itMax = max(itMax, it);
                }
                gNorm += gii;
                scatInt = fma(Jf, gii, scatInt);
            }
        }
        prd_store_rho(a, la, k, scatInt, gNorm, gammaPrefactor, Jbar, best, bestIdx);
    }
    if (a.gIIFill && a.gWin)
    {
        // the brackets this slice's fine grids touch: what prd_scatter_cached_kernel stages
        atomicMin(&sWin[0], itMin);
        atomicMax(&sWin[1], itMax);
    }
    prd_store_change(a, k, best, bestIdx, sBest, sBestIdx);
    if (a.gIIFill && a.gWin && threadIdx.x == 0)
    {
        a.gWin[((size_t)k * PRD_MAX_SLICES + blockIdx.y) * 2] = min(sWin[0], sWin[1]);
        a.gWin[((size_t)k * PRD_MAX_SLICES + blockIdx.y) * 2 + 1] = sWin[1];
    }
}

// The kernel of every sub-iteration but the first: all lines' caches are filled.  A workgroup stages only the brackets its
// slice's fine grids touch (the window recorded by the filling pass: ~1/8 of the line's grid) straight from J -- no transposed
// copy of J, no J / q tables in LDS, one 16-byte LDS read and one multiply-add per fine point -- and requests PRD_UNROLL fine
// points' weights and indices together (the pass streams its cache, 10 B per fine point; 1 ... 16 measured within 3 %, 4 best).
#ifndef PRD_UNROLL
#define PRD_UNROLL 4
#endif
#ifndef PRD_BLOCK
#define PRD_BLOCK 128
#endif
__global__ void __launch_bounds__(128) prd_scatter_cached_kernel(const PrdLineArgs* __restrict__ list, const int32_t* stopCtl, const int stopIter)
{
    dbg_poison_lds();
    extern __shared__ double sm[];
    if (prd_stopped(stopCtl, stopIter))
        return;
    const PrdLineArgs a = ld_c(CTAB(PrdLineArgs, list) + blockIdx.z);
    if (a.laHi <= a.laLo || (int)blockIdx.y >= a.slices)
        return;
    const int k = blockIdx.x;
    const int Ns = a.Ns;
    const int Nl = a.Nl;
    __shared__ double sPjQj;
    __shared__ double sBest[128];
    __shared__ int sBestIdx[128];
    __shared__ double sTerm[128];
    const int laPer = (a.laHi - a.laLo + gridDim.y - 1) / gridDim.y;
    const int laBeg = a.laLo + blockIdx.y * laPer;
    const int laEnd = min(a.laHi, laBeg + laPer);
    const int la = laBeg + threadIdx.x; // (laPer <= 128 is NOT guaranteed: the loop below strides)
    const double vb = a.vBroad[k];
    const int itLo = a.gWin[((size_t)k * PRD_MAX_SLICES + blockIdx.y) * 2];
    const int itHi = a.gWin[((size_t)k * PRD_MAX_SLICES + blockIdx.y) * 2 + 1];
    double2* AB = (double2*)sm; // [itHi - itLo + 1]: AB[it - itLo]
    const double* Jcol = a.J + (size_t)a.Nblue * Ns + k; // J(la, k) = Jcol[la Ns]
    for (int it = itLo + threadIdx.x; it <= itHi; it += blockDim.x)
        AB[it - itLo] = prd_bracket(prd_q(a.wave[it - 1], a.lambda0, vb), prd_q(a.wave[it], a.lambda0, vb),
                                    Jcol[(size_t)(it - 1) * Ns], Jcol[(size_t)it * Ns]);
    const double pjqj = prd_PjQj(a, k, sTerm, &sPjQj);
    const double qFirst = prd_q(a.wave[0], a.lambda0, vb), qLast = prd_q(a.wave[Nl - 1], a.lambda0, vb);
    const double gammaPrefactor = a.n[(size_t)a.gi * Ns + k] / a.n[(size_t)a.gj * Ns + k] * a.Bij / pjqj;
    const double Jbar = a.Rij[(size_t)a.tr * Ns + k] / a.Bij;
    double best = 0.0;
    int bestIdx = 0x7fffffff;
    for (int l = la; l < laEnd; l += blockDim.x)
    {
        const double qEmit = prd_q(a.wave[l], a.lambda0, vb);
        double q0, qN;
        prd_fine_range(qEmit, q0, qN);
        const int Np = (int)((double)(qN - q0) / PrdDQ) + 1;
        const size_t cOff = ((size_t)k * PRD_MAX_FINE) * Nl + l;
        const double* gc = a.gII + cOff;
        const int16_t* ic = (const int16_t*)a.gIT + cOff;
        double gNorm = 0.0, scatInt = 0.0;
        for (int i0 = 0; i0 < Np; i0 += PRD_UNROLL)
        {
            double gv[PRD_UNROLL];
            int iv[PRD_UNROLL];
#pragma unroll
            for (int u = 0; u < PRD_UNROLL; ++u)
            {
                // (past the end of the fine grid: weight 0 on a valid bracket adds +0 to both sums)
                const bool in = i0 + u < Np;
                gv[u] = in ? gc[(size_t)u * Nl] : 0.0;
                iv[u] = in ? (int)ic[(size_t)u * Nl] : itLo;
            }
            gc += (size_t)PRD_UNROLL * Nl;
            ic += (size_t)PRD_UNROLL * Nl;
#pragma unroll
            for (int u = 0; u < PRD_UNROLL; ++u)
            {
                const double Jf = prd_interp(AB[iv[u] - itLo], q0 + (i0 + u) * PrdDQ, qFirst, qLast);
                gNorm += gv[u];
                scatInt = fma(Jf, gv[u], scatInt);
            }
        }
        prd_store_rho(a, l, k, scatInt, gNorm, gammaPrefactor, Jbar, best, bestIdx);
    }
    prd_store_change(a, k, best, bestIdx, sBest, sBestIdx);
}

// per line: max over depth of the per-depth maxima (first flattened index on ties)
__global__ void prd_change_kernel(const PrdLineArgs* __restrict__ list)
{
    dbg_poison_lds();
    const PrdLineArgs a = ld_c(CTAB(PrdLineArgs, list) + blockIdx.x);
    prd_change_reduce(a, (int)threadIdx.x);
}

hipError_t launch_prd_scatter(const PrdLineArgs* devList, const PrdLineArgs* hostList, int nLines, hipStream_t stream, bool reduceChange,
                              const int32_t* stopCtl, int stopIter)
{
    if (nLines <= 0)
        return hipSuccess;
    int maxNl = 1, maxSlices = 1, Ns = hostList[0].Ns;
    bool any = false, allCached = true;
    for (int q = 0; q < nLines; ++q)
    {
        maxNl = std::max(maxNl, hostList[q].Nl);
        maxSlices = std::max(maxSlices, hostList[q].slices);
        const bool has = hostList[q].laHi > hostList[q].laLo;
        any = any || has;
        if (has && (hostList[q].gIIFill || !hostList[q].gII || !hostList[q].gWin))
            allCached = false;
    }
    // general kernel: q table + brackets of the whole grid; cached kernel: the brackets of a window (at most the whole grid)
    const size_t lds = (size_t)(allCached ? 2 : 3) * maxNl * sizeof(double);
    const void* fn = allCached ? (const void*)prd_scatter_cached_kernel : (const void*)prd_scatter_kernel;
    if (lds > 48 * 1024)
    {
        hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess)
            return err;
    }
    if (any && allCached)
        LWHIP_LAUNCH(prd_scatter_cached_kernel, dim3(Ns, maxSlices, nLines), dim3(PRD_BLOCK), lds, stream, devList, stopCtl, stopIter);
    else if (any)
    {
        LWHIP_LAUNCH(prd_transpose_kernel, dim3((maxNl + 31) / 32, (Ns + 31) / 32, nLines), dim3(32, 8), 0, stream, devList, stopCtl, stopIter);
        LWHIP_LAUNCH(prd_scatter_kernel, dim3(Ns, maxSlices, nLines), dim3(128), lds, stream, devList, stopCtl, stopIter);
    }
    if (reduceChange)
        LWHIP_LAUNCH(prd_change_kernel, dim3(nLines), dim3(64), 0, stream, devList);
    return hipGetLastError();
}
}

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
    double* Jk = sm;
    double* qW = sm + Nl;
    double* rq = sm + 2 * Nl; // 1 / (q[i] - q[i - 1]): the interpolation fraction of a fine point without a division
    __shared__ double sPjQj;
    __shared__ double sBest[128];
    __shared__ int sBestIdx[128];

    for (int la = threadIdx.x; la < Nl; la += blockDim.x)
    {
        Jk[la] = a.Jt[(size_t)k * Nl + la];
        qW[la] = (a.wave[la] - a.lambda0) * CLight / (a.lambda0 * a.vBroad[k]);
        if (la > 0)
            rq[la] = 1.0 / ((a.wave[la] - a.lambda0) * CLight / (a.lambda0 * a.vBroad[k])
                            - (a.wave[la - 1] - a.lambda0) * CLight / (a.lambda0 * a.vBroad[k]));
        else
            rq[la] = 0.0;
    }
    // total_depop_elastic_scattering_rate: Prd.cpp:9-30.  The terms are fetched by separate lanes
    // (one memory round trip), then added by one thread in the reference's order.
    __shared__ double sTerm[128];
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
        sPjQj = pq;
    }
    __syncthreads();

    const double gammaPrefactor = a.n[(size_t)a.gi * Ns + k] / a.n[(size_t)a.gj * Ns + k] * a.Bij / sPjQj;
    const double Jbar = a.Rij[(size_t)a.tr * Ns + k] / a.Bij;
    const double aDamp = a.aDamp[k];
    double best = 0.0;
    int bestIdx = 0x7fffffff;
    // this workgroup's share of the emission wavelengths (blockIdx.y of gridDim.y slices)
    const int laPer = (a.laHi - a.laLo + gridDim.y - 1) / gridDim.y;
    const int laBeg = a.laLo + blockIdx.y * laPer;
    const int laEnd = min(a.laHi, laBeg + laPer);
    for (int la = laBeg + threadIdx.x; la < laEnd; la += blockDim.x)
    {
        const double qEmit = qW[la];
        double q0, qN; // scattering_int_range: Prd.cpp:233-263
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
        const int Np = (int)((double)(qN - q0) / PrdDQ) + 1;
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
        // What of the fine grid of this (depth, emission wavelength) does not depend on J is cached, laid out [k][fine point][la]
        // so that a wavefront's reads coalesce: the end-corrected gII weights (the reference caches these too, t->gII,
        // Prd.cpp:499-553) and the bracketing index of the line's grid as 16 bits -- 10 bytes per fine point.  (Round 3 also
        // cached the interpolation fraction, 20 bytes per point, and was bound by streaming them; the fraction is re-derived
        // from the index with a tabulated reciprocal.  Re-deriving the index too -- the reference's walk -- was measured:
        // 87 dependent LDS round trips per thread, twice the time of streaming it.)
        const size_t cOff = ((size_t)k * PRD_MAX_FINE) * Nl + la;
        double gNorm = 0.0, scatInt = 0.0;
        if (!a.gIIFill)
        {
            const double* gc = a.gII + cOff;
            const int16_t* ic = (const int16_t*)a.gIT + cOff;
            // (eight fine points' weights and indices requested together: the pass streams its cache, and one dependent pair
            // of loads per iteration left the memory pipe mostly idle; the sums are formed in the same order)
            constexpr int PRD_UNROLL = 8;
            for (int i0 = 0; i0 < Np; i0 += PRD_UNROLL)
            {
                double gv[PRD_UNROLL];
                int iv[PRD_UNROLL];
#pragma unroll
                for (int u = 0; u < PRD_UNROLL; ++u)
                {
                    const bool in = i0 + u < Np;
                    gv[u] = in ? gc[(size_t)(i0 + u) * Nl] : 0.0;
                    iv[u] = in ? (int)ic[(size_t)(i0 + u) * Nl] : 1;
                }
#pragma unroll
                for (int u = 0; u < PRD_UNROLL; ++u)
                {
                    if (i0 + u >= Np)
                        break;
                    const double gii = gv[u];
                    const int it = iv[u];
                    const double x = q0 + (i0 + u) * PrdDQ;
                    // (beyond the table ends the fraction leaves [0, 1]: clamped, it is the 1 / 0 of the filling pass)
                    const double tt = fmin(fmax((x - qW[it - 1]) * rq[it], 0.0), 1.0);
                    const double Jf = (1.0 - tt) * Jk[it - 1] + tt * Jk[it];
                    gNorm += gii;
                    scatInt += Jf * gii;
                }
            }
        }
        else
        {
            double qPrime = q0;
            for (int i = 0; i < Np; ++i)
            {
                const double x = q0 + i * PrdDQ;
                while (iter < Nl && qW[iter] <= x)
                    ++iter;
                // beyond the table ends the reference takes the end value: the same number comes out
                // of the interpolation formula with fraction 1 (0) on the last (first) bracket
                int it = iter;
                double tt;
                if (iter == Nl)
                {
                    it = Nl - 1;
                    tt = 1.0;
                }
                else if (iter == 0)
                {
                    it = 1;
                    tt = 0.0;
                }
                else
                    tt = (x - qW[iter - 1]) * rq[iter]; // (the same expression as the cached pass: one rho from either)
                const double Jf = (1.0 - tt) * Jk[it - 1] + tt * Jk[it];
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
                }
                gNorm += gii;
                scatInt += Jf * gii;
            }
        }
        const size_t o = (size_t)(la - a.laLo) * Ns + k;
        const double rhoNew = 1.0 + gammaPrefactor * (scatInt / gNorm - Jbar);
        const double rhoOld = a.rho[o];
        a.rho[o] = rhoNew;
        if (rhoNew != 0.0)
        {
            const double change = fabs((rhoNew - rhoOld) / rhoNew);
            const int flat = la * Ns + k;
            if (change > best || (change == best && flat < bestIdx))
            {
                best = change;
                bestIdx = flat;
            }
        }
    }
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
    bool any = false;
    for (int q = 0; q < nLines; ++q)
    {
        maxNl = std::max(maxNl, hostList[q].Nl);
        maxSlices = std::max(maxSlices, hostList[q].slices);
        any = any || hostList[q].laHi > hostList[q].laLo;
    }
    const size_t lds = (size_t)3 * maxNl * sizeof(double);
    if (lds > 48 * 1024)
    {
        hipError_t err = hipFuncSetAttribute((const void*)prd_scatter_kernel,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess)
            return err;
    }
    if (any)
    {
        LWHIP_LAUNCH(prd_transpose_kernel, dim3((maxNl + 31) / 32, (Ns + 31) / 32, nLines), dim3(32, 8), 0, stream, devList, stopCtl, stopIter);
        LWHIP_LAUNCH(prd_scatter_kernel, dim3(Ns, maxSlices, nLines), dim3(128), lds, stream, devList, stopCtl, stopIter);
    }
    if (reduceChange)
        LWHIP_LAUNCH(prd_change_kernel, dim3(nLines), dim3(64), 0, stream, devList);
    return hipGetLastError();
}
}

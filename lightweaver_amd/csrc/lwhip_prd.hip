// lwhip_prd.hip -- partial frequency redistribution on the device: the angle-averaged scattering
// integral of prd_scatter / scattering_int (Source/Prd.cpp:468-645) with Gouttebroze's GII
// (Source/Prd.cpp:46-124), the total depopulation + elastic rate (Source/Prd.cpp:9-30) and the
// max relative change of rho that Ng(0,0,0)::max_change reports (Source/Ng.hpp:136-156).
//
// One workgroup per (PRD line, depth point): J(:, k) and the line's wavelengths in Doppler units
// are staged in LDS, then every thread owns one emission wavelength and walks its <= 87-point
// fine grid (spacing 0.15 Doppler widths) exactly as the reference does -- same interpolation,
// same sequentially accumulated q', same end-corrected trapezoid weights -- evaluating GII on the
// fly instead of caching it per (depth, wavelength) as the reference does (the cached values
// depend only on aDamp, vBroad and the grid, so the numbers are the same).
#include "lwhip_internal.h"
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

__global__ void __launch_bounds__(256) prd_scatter_kernel(const PrdLineArgs a)
{
    extern __shared__ double sm[];
    const int k = blockIdx.x;
    const int Ns = a.Ns;
    const int Nl = a.Nl;
    double* Jk = sm;
    double* qW = sm + Nl;
    __shared__ double sPjQj;
    __shared__ double sBest[256];
    __shared__ int sBestIdx[256];

    for (int la = threadIdx.x; la < Nl; la += blockDim.x)
    {
        Jk[la] = a.J[(size_t)(la + a.Nblue) * Ns + k];
        qW[la] = (a.wave[la] - a.lambda0) * CLight / (a.lambda0 * a.vBroad[k]);
    }
    if (threadIdx.x == 0)
    {
        // total_depop_elastic_scattering_rate: Prd.cpp:9-30
        double pq = a.Qelast[k];
        for (int i = 0; i < a.Nlevel; ++i)
            pq += a.Cmat[((size_t)i * a.Nlevel + a.lj) * Ns + k];
        for (int tr = a.trBegin; tr < a.trEnd; ++tr)
        {
            if (a.transLj[tr] == a.lj)
                pq += a.Rji[(size_t)tr * Ns + k];
            if (a.transLi[tr] == a.lj)
                pq += a.Rij[(size_t)tr * Ns + k];
        }
        sPjQj = pq;
    }
    __syncthreads();

    const double gammaPrefactor = a.n[(size_t)a.gi * Ns + k] / a.n[(size_t)a.gj * Ns + k] * a.Bij / sPjQj;
    const double Jbar = a.Rij[(size_t)a.tr * Ns + k] / a.Bij;
    const double aDamp = a.aDamp[k];
    double best = 0.0;
    int bestIdx = 0x7fffffff;
    for (int la = threadIdx.x; la < Nl; la += blockDim.x)
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
        double gNorm = 0.0, scatInt = 0.0;
        double qPrime = q0;
        for (int i = 0; i < Np; ++i)
        {
            const double x = q0 + i * PrdDQ;
            while (iter < Nl && qW[iter] <= x)
                ++iter;
            double Jf;
            if (iter == Nl)
                Jf = Jk[Nl - 1];
            else if (iter == 0)
                Jf = Jk[0];
            else
            {
                const double xp = qW[iter - 1], xn = qW[iter];
                const double tt = (x - xp) / (xn - xp);
                Jf = (1.0 - tt) * Jk[iter - 1] + tt * Jk[iter];
            }
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
            gNorm += gii;
            scatInt += Jf * gii;
        }
        const size_t o = (size_t)la * Ns + k;
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
        a.changeVal[k] = sBest[0];
        a.changeIdx[k] = sBestIdx[0];
    }
}

// per line: max over depth of the per-depth maxima (first flattened index on ties)
__global__ void prd_change_kernel(const double* changeVal, const int* changeIdx, int Ns, double* out2)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    double best = 0.0;
    int idx = 0;
    bool any = false;
    for (int k = 0; k < Ns; ++k)
    {
        const double v = changeVal[k];
        const int i = changeIdx[k];
        if (i == 0x7fffffff)
            continue;
        if (!any || v > best || (v == best && i < idx))
        {
            if (v > 0.0 || !any)
            {
                best = v;
                idx = i;
                any = true;
            }
        }
    }
    // max_idx semantics (strictly greater updates, initial dMax = 0, idx = 0): a zero maximum keeps idx 0
    if (!(best > 0.0))
    {
        best = 0.0;
        idx = 0;
    }
    out2[0] = best;
    out2[1] = (double)idx;
}

hipError_t launch_prd_scatter(const PrdLineArgs& a, double* out2, hipStream_t stream)
{
    const size_t lds = (size_t)2 * a.Nl * sizeof(double);
    static size_t configured = 0;
    if (lds > configured && lds > 48 * 1024)
    {
        hipError_t err = hipFuncSetAttribute((const void*)prd_scatter_kernel,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess)
            return err;
        configured = lds;
    }
    hipLaunchKernelGGL(prd_scatter_kernel, dim3(a.Ns), dim3(256), lds, stream, a);
    hipLaunchKernelGGL(prd_change_kernel, dim3(1), dim3(64), 0, stream, a.changeVal, a.changeIdx, a.Ns, out2);
    return hipGetLastError();
}
}

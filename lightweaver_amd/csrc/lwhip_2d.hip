// lwhip_2d.hip -- the pointwise halves of the 2D iteration: intensity_core_opt
// (Source/SimdFullIterationTemplates.hpp:238-487) around the 2D formal solver of lwhip_fs2d.hip.
//
// A 2D domain (Nz x Nx = 20 992 points per ray at BASELINE config 5) does not fit the fused 1D sweep, so the
// iteration runs batch by batch over the wavelengths with its per-ray intermediates in HBM:
//   cont_kernel (lwhip_march.hip)   continuum rows of the batch, once per (wavelength, point)
//   gather2d_kernel                 chi, S per (wavelength, ray, point)         :113-179
//   fs2d_coef / fs2d_scan (lwhip_fs2d.hip)   I, Psi* per (wavelength, ray)
//   rates2d_kernel                  J, Gamma / rate integrals, dJ               :181-234, 477-486
// rates2d gives every spatial point to one thread, which walks the batch's wavelengths and rays and adds
// into its own column of the running sums red[Ntrans, 4, Nspace]: no atomics, no slabs; apply_kernel then
// finishes Gamma exactly as in 1D.  All of these are HBM-streaming kernels: per (wavelength, ray, point) the
// traffic is phi once per line and pass (gather + rates), chi / S / I / Psi* once each way.
#include "lwhip_device.h"
#include "../../include/lwhip.h"

#include <algorithm>

namespace lwhip
{
namespace
{
// ---- gather: Transition::uv + chi_eta_aux_accum + compute_source_fn for one (wavelength, ray, point) ----------
__global__ void __launch_bounds__(256) gather2d_kernel(const Batch2dArgs a)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.Ns)
        return;
    const int r = blockIdx.y;
    const int b = blockIdx.z;
    const int la = a.la0 + b;
    const int Ns = a.Ns;
    const int ray = a.rayList[r];
    const DevLaHeader h = a.laHdr[la];
    const DevSlot* slots = a.slots + h.off;
    const size_t laNs = (size_t)la * Ns;
    double chi, eta;
    if (h.rowBase >= 0)
    {
        chi = a.rows[(size_t)h.rowBase * Ns + k];
        eta = a.rows[((size_t)h.rowBase + 1) * Ns + k];
    }
    else
    {
        chi = a.bgChi[laNs + k];
        eta = a.bgEta[laNs + k];
    }
    for (int l = 0; l < h.nLine; ++l)
    {
        const DevSlot& sl = slots[l];
        const double vij = sl.p[0] * a.phi[sl.phiOff + (size_t)ray * Ns + k];
        double g = sl.p[2];
        if (sl.flags & 2)
            g *= a.rho[sl.rhoOff + k];
        const double vji = g * vij;
        const double nj = a.n[(size_t)sl.gj * Ns + k];
        chi += a.n[(size_t)sl.gi * Ns + k] * vij - nj * vji;
        eta += nj * (sl.p[3] * vji);
    }
    const size_t o = ((size_t)b * a.nRaysActive + r) * Ns + k;
    a.chi[o] = chi;
    a.S[o] = (eta + a.bgSca[laNs + k] * a.J[laNs + k]) / chi;
    if (a.storeDepth)
    {
        const size_t d = ((size_t)la * a.Nrays * 2 + ray) * Ns + k;
        a.depthChi[d] = chi;
        a.depthEta[d] = eta;
    }
}

// ---- rates: one thread per point walks the wavelengths of the batch and the rays ----------------------------
// Per wavelength, everything that does not depend on the ray -- populations, rho, the line weight, the continuum
// rows the level sums start from -- is fetched once into registers / the thread's LDS column; the ray loop then
// reads only I, Psi* and one phi per line from memory.
template <int MAXL>
__global__ void __launch_bounds__(128) rates2d_kernel(const Batch2dArgs a)
{
    constexpr int MAXM = 4; // mixed continua integrated ray by ray at one wavelength
    constexpr int NROW = 5; // rEtaA, rChiI, rChiJ, rUI, rUJ
    __shared__ double sRow[(MAXL + MAXM) * NROW + MAXM][128];
    const int tid = threadIdx.x;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int Ns = a.Ns;
    const bool on = k < Ns;
    const int kk = on ? k : 0;
    double Wsum = 0.0;
    for (int r = 0; r < a.nRaysActive; ++r)
        Wsum += 0.5 * a.wmu[a.rayList[r] >> 1];
    // blockIdx.y: a group of the batch's wavelengths with its own copy of the running sums (added up afterwards)
    double* redG = a.red + (size_t)blockIdx.y * a.redStride;
    for (int b = blockIdx.y; b < a.nLa; b += gridDim.y)
    {
        const int la = a.la0 + b;
        const DevLaHeader h = a.laHdr[la];
        const DevSlot* slots = a.slots + h.off;
        const int nLine = h.nLine, nMixed = h.nMixed, nPure = h.nPure;
        const bool hasRows = h.rowBase >= 0;
        const size_t laNs = (size_t)la * Ns;
        auto rowAt = [&](int ri) -> double { return (hasRows && ri >= 0) ? a.rows[((size_t)h.rowBase + ri) * Ns + kk] : 0.0; };
        double accL[MAXL][4], accM[MAXM][4];
#pragma unroll
        for (int l = 0; l < MAXL; ++l)
            accL[l][0] = accL[l][1] = accL[l][2] = accL[l][3] = 0.0;
#pragma unroll
        for (int c = 0; c < MAXM; ++c)
            accM[c][0] = accM[c][1] = accM[c][2] = accM[c][3] = 0.0;
        double mJ = 0.0, mM1 = 0.0, mE0 = 0.0, mE1 = 0.0;
        // ---- ray-independent part of the wavelength ---------------------------------------------------------
        double lG[MAXL], lNi[MAXL], lNj[MAXL], lW[MAXL];
        int lgi[MAXL], lgj[MAXL], lat[MAXL], lop[MAXL];
#pragma unroll
        for (int l = 0; l < MAXL; ++l)
        {
            lG[l] = lNi[l] = lNj[l] = lW[l] = 0.0;
            lgi[l] = lgj[l] = lat[l] = -1;
            lop[l] = 0;
            if (l < nLine)
            {
                const DevSlot& sl = slots[l];
                double g = sl.p[2];
                if (sl.flags & 2)
                    g *= a.rho[sl.rhoOff + kk];
                lG[l] = g;
                lNi[l] = a.n[(size_t)sl.gi * Ns + kk];
                lNj[l] = a.n[(size_t)sl.gj * Ns + kk];
                lW[l] = (sl.p[1] * a.wphi[(size_t)sl.row * Ns + kk]) * PI4_HC;
                lgi[l] = sl.gi;
                lgj[l] = sl.gj;
                lat[l] = sl.atom;
                lop[l] = sl.flags & 1;
                if (lop[l])
                {
                    sRow[l * NROW + 0][tid] = rowAt(sl.rEtaA);
                    sRow[l * NROW + 1][tid] = rowAt(sl.rChiI);
                    sRow[l * NROW + 2][tid] = rowAt(sl.rChiJ);
                    sRow[l * NROW + 3][tid] = rowAt(sl.rUI);
                    sRow[l * NROW + 4][tid] = rowAt(sl.rUJ);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < MAXM; ++c)
            if (c < nMixed)
            {
                const DevSlot& sl = slots[nLine + c];
                sRow[(MAXL + c) * NROW + 0][tid] = rowAt(sl.rEtaA);
                sRow[(MAXL + c) * NROW + 1][tid] = rowAt(sl.rChiI);
                sRow[(MAXL + c) * NROW + 2][tid] = rowAt(sl.rChiJ);
                sRow[(MAXL + c) * NROW + 3][tid] = rowAt(sl.rUI);
                sRow[(MAXL + c) * NROW + 4][tid] = rowAt(sl.rUJ);
                sRow[(MAXL + MAXM) * NROW + c][tid] = rowAt(sl.rVji);
            }
        for (int r = 0; r < a.nRaysActive; ++r)
        {
            const int ray = a.rayList[r];
            const int mu = ray >> 1;
            const double halfwmu = 0.5 * a.wmu[mu];
            const size_t o = ((size_t)b * a.nRaysActive + r) * Ns + kk;
            const double I = a.I[o];
            const double PsiS = a.lambdaIterate ? 0.0 : a.PsiStar[o];
            const double wPsi = halfwmu * PsiS;
            mJ += halfwmu * I; // accumulate_J :181-190
            mM1 += wPsi;
            if ((ray & 1) && kk < a.Nx && on) // Spectrum::I: the top row, last written by the toObs ray (:365-370)
                a.Iout[((size_t)la * a.Nrays + mu) * a.Nx + kk] = I;
            if (a.storeDepth && on)
                a.depthI[((size_t)la * a.Nrays * 2 + ray) * Ns + kk] = I;
            if (nLine + nMixed == 0)
                continue;
            double Vij[MAXL], Vji[MAXL], xl[MAXL], el[MAXL], luji[MAXL];
#pragma unroll
            for (int l = 0; l < MAXL; ++l)
            {
                Vij[l] = Vji[l] = xl[l] = el[l] = luji[l] = 0.0;
                if (l < nLine)
                {
                    const DevSlot& sl = slots[l];
                    const double vij = sl.p[0] * a.phi[sl.phiOff + (size_t)ray * Ns + kk];
                    const double vji = lG[l] * vij;
                    const double uji = sl.p[3] * vji;
                    Vij[l] = vij;
                    Vji[l] = vji;
                    luji[l] = uji;
                    xl[l] = lNi[l] * vij - lNj[l] * vji;
                    el[l] = lNj[l] * uji;
                    if (lop[l])
                    {
                        if (sl.ms == 0)
                            mE0 += wPsi * el[l];
                        else if (sl.ms == 1)
                            mE1 += wPsi * el[l];
                    }
                }
            }
            // level sums: continuum rows + this ray's lines (chi_eta_aux_accum bookkeeping)
            auto levChi = [&](int g, double v) {
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                {
                    if (lop[l] && lgi[l] == g)
                        v += xl[l];
                    if (lop[l] && lgj[l] == g)
                        v -= xl[l];
                }
                return v;
            };
            auto levU = [&](int g, double v) {
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                    if (lop[l] && lgj[l] == g)
                        v += luji[l];
                return v;
            };
            auto atomEta = [&](int at, double v) {
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                    if (lop[l] && lat[l] == at)
                        v += el[l];
                return v;
            };
#pragma unroll
            for (int l = 0; l < MAXL; ++l)
            {
                if (l < nLine)
                {
                    const double wlamu = lW[l] * halfwmu;
                    if (lop[l])
                    {
                        // compute_full_Ieff :192-204, compute_full_operator_rates :206-234
                        const double* q = &sRow[l * NROW][tid];
                        const double Ieff = I - PsiS * atomEta(lat[l], q[0]);
                        accL[l][0] += ((luji[l] + Vji[l] * Ieff) - (PsiS * levChi(lgi[l], q[128]) * levU(lgj[l], q[4 * 128]))) * wlamu;
                        accL[l][1] += ((Vij[l] * Ieff) - (PsiS * levChi(lgj[l], q[2 * 128]) * levU(lgi[l], q[3 * 128]))) * wlamu;
                    }
                    accL[l][2] += I * Vij[l] * wlamu;
                    accL[l][3] += (luji[l] + I * Vji[l]) * wlamu;
                }
            }
#pragma unroll
            for (int c = 0; c < MAXM; ++c)
            {
                if (c < nMixed) // mixed continua (always of active atoms)
                {
                    const DevSlot& sl = slots[nLine + c];
                    const double* q = &sRow[(MAXL + c) * NROW][tid];
                    const double vij = sl.p[0];
                    const double vji = sRow[(MAXL + MAXM) * NROW + c][tid];
                    const double uji = sl.p[2] * vji;
                    const double wlamu = sl.p[1] * halfwmu;
                    const double Ieff = I - PsiS * atomEta(sl.atom, q[0]);
                    accM[c][0] += ((uji + vji * Ieff) - (PsiS * levChi(sl.gi, q[128]) * levU(sl.gj, q[4 * 128]))) * wlamu;
                    accM[c][1] += ((vij * Ieff) - (PsiS * levChi(sl.gj, q[2 * 128]) * levU(sl.gi, q[3 * 128]))) * wlamu;
                    accM[c][2] += I * vij * wlamu;
                    accM[c][3] += (uji + I * vji) * wlamu;
                }
            }
        } // rays
        if (on)
        {
        // ---- this wavelength's contributions into the point's column of the running sums ------------------
#pragma unroll
        for (int l = 0; l < MAXL; ++l)
            if (l < nLine)
            {
                double* red = redG + ((size_t)a.slotTr[h.off + l] * 4) * Ns + k;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    red[(size_t)q * Ns] += accL[l][q];
            }
#pragma unroll
        for (int c = 0; c < MAXM; ++c)
            if (c < nMixed)
            {
                double* red = redG + ((size_t)a.slotTr[h.off + nLine + c] * 4) * Ns + k;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    red[(size_t)q * Ns] += accM[c][q];
            }
        // pure continua from the angle moments
        for (int cq = 0; cq < nPure; ++cq)
        {
            const DevSlot& sl = slots[nLine + nMixed + cq];
            const double vij = sl.p[0];
            const double vji = rowAt(sl.rVji);
            const double uji = sl.p[2] * vji;
            const double wla = sl.p[1];
            double* red = redG + ((size_t)a.slotTr[h.off + nLine + nMixed + cq] * 4) * Ns + k;
            if (sl.flags & 1)
            {
                const double Meta = (sl.ms == 0) ? mE0 : ((sl.ms == 1) ? mE1 : 0.0);
                const double Jeff = mJ - mM1 * rowAt(sl.rEtaA) - Meta;
                red[0] += ((uji * Wsum + vji * Jeff) - (mM1 * rowAt(sl.rChiI) * rowAt(sl.rUJ))) * wla;
                red[(size_t)Ns] += ((vij * Jeff) - (mM1 * rowAt(sl.rChiJ) * rowAt(sl.rUI))) * wla;
            }
            red[(size_t)2 * Ns] += mJ * vij * wla;
            red[(size_t)3 * Ns] += (uji * Wsum + mJ * vji) * wla;
        }
        }
        // J out, dJ = max_k |1 - Jdag/J|  :477-486 -- the wavefront's maximum first: one atomic per wavefront, not
        // one per point on the same address
        double dj = 0.0;
        if (on)
        {
            const double Jd = a.J[laNs + k];
            a.J[laNs + k] = mJ;
            dj = fabs(1.0 - Jd / mJ);
        }
        for (int sh = 32; sh > 0; sh >>= 1)
            dj = fmax(dj, __shfl_xor(dj, sh));
        if ((threadIdx.x & 63) == 0)
            atomicMax(a.dJbits + la, (unsigned long long)__double_as_longlong(dj)); // non-negative doubles order like their bits
    }
}
}

// ---- formal_sol (intensity only, formal_sol_impl :722-784): the emergent intensity of the up-going rays and, when
// asked for, I at depth; J, Gamma and the rates stay untouched ---------------------------------------------------
namespace
{
__global__ void __launch_bounds__(256) iout2d_kernel(const Batch2dArgs a)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.Ns || (!a.storeDepth && k >= a.Nx))
        return;
    const int r = blockIdx.y, b = blockIdx.z;
    const int la = a.la0 + b;
    const int ray = a.rayList[r];
    const double I = a.I[((size_t)b * a.nRaysActive + r) * a.Ns + k];
    if ((ray & 1) && k < a.Nx) // Spectrum::I: the top row, written by the toObs ray (:365-370)
        a.Iout[((size_t)la * a.Nrays + (ray >> 1)) * a.Nx + k] = I;
    if (a.storeDepth)
        a.depthI[((size_t)la * a.Nrays * 2 + ray) * a.Ns + k] = I;
}
}

hipError_t launch_iout2d(const Batch2dArgs& a, hipStream_t stream)
{
    const int n = a.storeDepth ? a.Ns : a.Nx;
    hipLaunchKernelGGL(iout2d_kernel, dim3((n + 255) / 256, a.nRaysActive, a.nLa), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// red[0] += red[1] + ... + red[G-1] (fixed order)
__global__ void sum_groups_kernel(double* red, size_t n, size_t stride, int G)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        double s = red[i];
        for (int g = 1; g < G; ++g)
            s += red[i + g * stride];
        red[i] = s;
    }
}

hipError_t launch_sum_groups(double* red, size_t n, size_t stride, int G, hipStream_t stream)
{
    if (G <= 1 || n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sum_groups_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, stream, red,
                       n, stride, G);
    return hipGetLastError();
}

hipError_t launch_gather2d(const Batch2dArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(gather2d_kernel, dim3((a.Ns + 255) / 256, a.nRaysActive, a.nLa), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_rates2d(const Batch2dArgs& a, int maxL, hipStream_t stream)
{
    const dim3 grid((a.Ns + 127) / 128, a.nGroups);
    if (maxL <= 2)
        hipLaunchKernelGGL(rates2d_kernel<2>, grid, dim3(128), 0, stream, a);
    else if (maxL <= 4)
        hipLaunchKernelGGL(rates2d_kernel<4>, grid, dim3(128), 0, stream, a);
    else
        hipLaunchKernelGGL(rates2d_kernel<8>, grid, dim3(128), 0, stream, a);
    return hipGetLastError();
}
}

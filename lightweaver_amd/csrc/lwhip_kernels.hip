// lwhip_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the formal_sol_gamma_matrices iteration.
//
// Mapping: one workgroup owns a contiguous chunk of wavelengths and walks
// it; inside a wavelength the 2*Nmu rays x Nspace depth points are flattened over the threads of
// the workgroup (10 x 82 = 820 points -> 13 wavefronts, 98.6 % lane use), so every load of
// phi[lt, mu, dir, k] is one fully coalesced 6.5 KB block.  The ray columns live in LDS for the
// short-characteristic stencils; the only true recurrence, I_k = e_k I_{k-1} + b_k, is a chunked
// scan.  Continua are ray-independent, so their opacities and per-level sums are evaluated once
// per (lambda, k) and shared by all rays through LDS.  Gamma / rate integrals are summed over
// rays and wavelengths in LDS accumulators and leave the workgroup once per chunk as a partial
// slab; a second kernel adds the slabs in a fixed order.  fp64 throughout, no MFMA: the sweep is
// a streaming stencil + scan, not a contraction.
//
// Reference restated (file:line under Lightweaver's Source/): see each device function.
#include "lwhip_internal.h"
#include "../../include/lwhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>

namespace lwhip
{
#define DEVINL __device__ __forceinline__

// ---- leaf numerics -------------------------------------------------------------------------------

// w2: LwInternal.hpp:90-110
DEVINL void d_w2(double dtau, double& w0, double& w1)
{
    if (dtau < 5.0E-4)
    {
        w0 = dtau * (1.0 - 0.5 * dtau);
        w1 = (dtau * dtau) * (0.5 - dtau * (1.0 / 3.0));
    }
    else if (dtau > 50.0)
    {
        w0 = 1.0;
        w1 = 1.0;
    }
    else
    {
        const double expdt = exp(-dtau);
        w0 = 1.0 - expdt;
        w1 = w0 - dtau * expdt;
    }
}

// cent_deriv (Steffen 1990): Bezier.hpp:58-65
DEVINL double d_cent_deriv(double dsuw, double dsdw, double yuw, double y0, double ydw)
{
    const double S0 = (ydw - y0) / dsdw;
    const double Suw = (y0 - yuw) / dsuw;
    const double P0 = fabs((Suw * dsdw + S0 * dsuw) / (dsdw + dsuw));
    return (copysign(1.0, S0) + copysign(1.0, Suw)) * fmin(fabs(Suw), fmin(fabs(S0), 0.5 * P0));
}

// Bezier3_coeffs: Bezier.hpp:81-127
DEVINL void d_bezier3_coeffs(double dt, double& alpha, double& beta, double& gamma, double& delta,
                             double& edt)
{
    const double dt2 = dt * dt;
    const double dt3 = dt2 * dt;
    if (dt < 5e-2)
    {
        edt = 1.0 - dt + 0.5 * dt2 - dt3 / 6.0;
        alpha = 0.25 * dt - 0.2 * dt2 + dt3 / 12.0;
        beta = 0.25 * dt - 0.05 * dt2 + dt3 / 120.0;
        gamma = 0.25 * dt - 0.15 * dt2 + 0.05 * dt3;
        delta = 0.25 * dt - 0.1 * dt2 + 0.025 * dt3;
    }
    else if (dt > 30.0)
    {
        edt = 0.0;
        alpha = 6.0 / dt3;
        beta = (-6.0 + 6.0 * dt - 3.0 * dt2 + dt3) / dt3;
        gamma = 3.0 * (2.0 * dt - 6.0) / dt3;
        delta = 3.0 * (6.0 - 4.0 * dt + dt2) / dt3;
    }
    else
    {
        edt = exp(-dt);
        alpha = (6.0 - edt * (6.0 + 6.0 * dt + 3 * dt2 + dt3)) / dt3;
        beta = (6.0 * edt - 6.0 + 6.0 * dt - 3.0 * dt2 + dt3) / dt3;
        gamma = 3.0 * (2.0 * dt - 6.0 + edt * (6.0 + 4.0 * dt + dt2)) / dt3;
        delta = 3.0 * (6.0 - 4.0 * dt + dt2 - 2.0 * edt * (3.0 + dt)) / dt3;
    }
}

// planck_nu at one temperature: LwMisc.hpp:29-46
DEVINL double d_planck(double T, double lambda)
{
    const double hc_kla = HC_K / lambda;
    const double twohnu3_c2 = TWO_HC_NM3 / (lambda * lambda * lambda);
    const double x = hc_kla / T;
    return (x <= 150.0) ? twohnu3_c2 / (exp(x) - 1.0) : 0.0;
}

// besser_control_point_1d: FormalScalar.cpp:327-363
DEVINL double d_besser_control_point(double hM, double hP, double yM, double yO, double yP)
{
    const double dM = (yO - yM) / hM;
    const double dP = (yP - yO) / hP;
    if (dM * dP <= 0.0)
        return yO;
    double yOp = (hM * dP + hP * dM) / (hM + hP);
    double cM = yO - 0.5 * hM * yOp;
    double cP = yO + 0.5 * hP * yOp;
    double minYMO = yM, maxYMO = yO, minYOP = yO, maxYOP = yP;
    if (dM < 0.0)
    {
        minYMO = yO;
        maxYMO = yM;
        minYOP = yP;
        maxYOP = yO;
    }
    if (cM < minYMO || cM > maxYMO)
        return yM;
    if (cP < minYOP || cP > maxYOP)
    {
        cP = yP;
        yOp = (cP - yO) / (0.5 * hP);
        cM = yO - 0.5 * hM * yOp;
    }
    return cM;
}

// besser_coeffs_1d: FormalScalar.cpp:373-393
DEVINL void d_besser_coeffs(double t, double& M, double& O, double& Cc, double& edt)
{
    if (t < 0.14)
    {
        M = (t * (t * (t * (t * (t * (t * ((140.0 - 18.0 * t) * t - 945.0) + 5400.0) - 25200.0) + 90720.0) - 226800.0) + 302400.0)) / 907200.0;
        O = (t * (t * (t * (t * (t * (t * ((10.0 - t) * t - 90.0) + 720.0) - 5040.0) + 30240.0) - 151200.0) + 604800.0)) / 1814400.0;
        Cc = (t * (t * (t * (t * (t * (t * ((35.0 - 4.0 * t) * t - 270.0) + 1800.0) - 10080.0) + 45360.0) - 151200.0) + 302400.0)) / 907200.0;
        const double t2 = t * t, t3 = t * t * t;
        edt = 1.0 - t + 0.5 * t2 - t3 / 6.0 + t * t3 / 24.0 - t2 * t3 / 120.0 + t3 * t3 / 720.0 - t3 * t3 * t / 5040.0;
    }
    else
    {
        const double t2 = t * t;
        edt = exp(-t);
        M = (2.0 - edt * (t2 + 2.0 * t + 2.0)) / t2;
        O = 1.0 - 2.0 * (edt + t - 1.0) / t2;
        Cc = 2.0 * (t - 2.0 + edt * (t + 2.0)) / t2;
    }
}

DEVINL void lds_add(double* addr, double v)
{
    // ds_add_f64 on gfx950
    unsafeAtomicAdd(addr, v);
}

// ---- LDS layout ------------------------------------------------------------------------------------
// [GNs] x5 : sChi(->b->I)  sS  sA(dchi->e->wI)  sDtau  sDS
// [Ns]  x7 : height temperature Jdag chiC etaC sca J
// zeroed   : etaA[Natom][Ns] levChi[NlevTot][Ns] levU[NlevTot][Ns]
// [maxC][Ns] Vji of the active continua
// [maxCT][4][Ns] Gamma/rate accumulators of the chunk
// [G*P*2] scan carries
__host__ __device__ inline size_t lds_doubles(int Ns, int G, int P, int Natom, int NlevTot, int maxC,
                                              int maxCT)
{
    return (size_t)5 * G * Ns + (size_t)7 * Ns + (size_t)(Natom + 2 * NlevTot) * Ns
           + (size_t)maxC * Ns + (size_t)maxCT * 4 * Ns + (size_t)G * P * 2 + 16;
}

size_t sweep_lds_bytes(const SweepArgs& a)
{
    return sizeof(double) * lds_doubles(a.Ns, a.G, a.P, a.Natom, a.NlevTot, a.maxC, a.maxCT);
}

// ---- the sweep ---------------------------------------------------------------------------------------
// intensity_core_opt (SimdFullIterationTemplates.hpp:238-487) for every wavelength of the chunk,
// with UpdateRates = ComputeOperator = RATES, PrdRatesOnly = false.
template <int SOLVER, int MAXL, bool RATES>
__global__ void __launch_bounds__(1024) sweep_kernel(const SweepArgs a)
{
    extern __shared__ double lds[];
    const int Ns = a.Ns;
    const int GNs = a.G * Ns;
    const int T = blockDim.x;
    const int t = threadIdx.x;
    const int Nrays = a.Nrays;

    double* sChi = lds;
    double* sS = sChi + GNs;
    double* sA = sS + GNs;
    double* sDtau = sA + GNs;
    double* sDS = sDtau + GNs;
    double* sK = sDS + GNs;
    double* sHgt = sK;
    double* sTemp = sK + Ns;
    double* sJdag = sK + 2 * Ns;
    double* sChiC = sK + 3 * Ns;
    double* sEtaC = sK + 4 * Ns;
    double* sSca = sK + 5 * Ns;
    double* sJ = sK + 6 * Ns;
    double* sEtaA = sK + 7 * Ns;
    double* sLevChi = sEtaA + a.Natom * Ns;
    double* sLevU = sLevChi + a.NlevTot * Ns;
    double* sVji = sLevU + a.NlevTot * Ns;
    double* sAcc = sVji + a.maxC * Ns;
    double* sScan = sAcc + a.maxCT * 4 * Ns;
    double* sB = sChi; // alias: chi column is dead once the optical depths exist
    double* sE = sA;   // alias: dchi is dead once the optical depths exist

    const bool isPoint = t < GNs;
    const int rl = t / Ns;
    const int k = t - rl * Ns;

    const int chunk = blockIdx.x;
    const int la0 = a.chunkLa[chunk];
    const int la1 = a.chunkLa[chunk + 1];

    if (t < Ns)
    {
        sHgt[t] = a.height[t];
        sTemp[t] = a.temperature[t];
    }
    if (RATES)
    {
        for (int idx = t; idx < a.maxCT * 4 * Ns; idx += T)
            sAcc[idx] = 0.0;
    }
    __syncthreads();

    for (int la = la0; la < la1; ++la)
    {
        const DevLaHeader h = a.laHdr[la];
        const DevLaEntry* __restrict__ ent = a.laEnt + h.off;
        const int nLine = h.nLine;
        const int nCont = h.nCont;
        const double wav = a.wavelength[la];
        const double hc_kl = HC_K / wav;

        // ---- per-wavelength, ray-independent part -----------------------------------------------
        if (RATES)
        {
            const int zeroCount = (a.Natom + 2 * a.NlevTot) * Ns;
            for (int idx = t; idx < zeroCount; idx += T)
                sEtaA[idx] = 0.0;
        }
        if (t < Ns)
        {
            const size_t o = (size_t)la * Ns + t;
            sJdag[t] = a.J[o];
            sChiC[t] = a.bgChi[o];
            sEtaC[t] = a.bgEta[o];
            sSca[t] = a.bgSca[o];
        }
        // continuum Vji = gij * alpha,  gij = nStar_i/nStar_j * exp(-hc/(k lambda T))
        // (Atom::setup_wavelength LwAtom.hpp:107-115, Transition::uv LwTransition.hpp:132-142)
        for (int idx = t; idx < nCont * Ns; idx += T)
        {
            const int c = idx / Ns;
            const int kk = idx - c * Ns;
            const DevTrans& tr = a.trans[ent[nLine + c].tr];
            const double* par = a.par + tr.parOff + 4 * (la - tr.Nblue);
            const double gij = a.ratio[(size_t)tr.row * Ns + kk] * exp(-hc_kl / sTemp[kk]);
            sVji[c * Ns + kk] = gij * par[0];
        }
        __syncthreads();
        // chi_eta_aux_accum for the continua (SimdFullIterationTemplates.hpp:59-109): totals and
        // per-level / per-atom sums, once per depth point
        if (t < Ns)
        {
            double chi = sChiC[t], eta = sEtaC[t];
            for (int c = 0; c < nCont; ++c)
            {
                const DevTrans& tr = a.trans[ent[nLine + c].tr];
                const double* par = a.par + tr.parOff + 4 * (la - tr.Nblue);
                const double Vij = par[0];
                const double Vji = sVji[c * Ns + t];
                const double Uji = par[2] * Vji;
                const double ni = a.n[(size_t)tr.gi * Ns + t];
                const double nj = a.n[(size_t)tr.gj * Ns + t];
                const double x = ni * Vij - nj * Vji;
                const double em = nj * Uji;
                if (RATES && tr.op)
                {
                    sLevChi[tr.gi * Ns + t] += x;
                    sLevChi[tr.gj * Ns + t] -= x;
                    sLevU[tr.gj * Ns + t] += Uji;
                    sEtaA[tr.atom * Ns + t] += em;
                }
                chi += x;
                eta += em;
            }
            sChiC[t] = chi;
            sEtaC[t] = eta;
        }
        __syncthreads();

        // ---- passes over groups of G rays -----------------------------------------------------------
        for (int g0 = 0; g0 < a.nRaysActive; g0 += a.G)
        {
            const int nr = min(a.G, a.nRaysActive - g0);
            const bool active = isPoint && rl < nr;
            int mu = 0, toObs = 0;
            if (active)
            {
                const int r = a.rayList[g0 + rl];
                mu = r >> 1;
                toObs = r & 1;
            }
            double chi = 1.0, S = 0.0, eta = 0.0;
            double Vij[MAXL], Vji[MAXL], xl[MAXL], el[MAXL];
#pragma unroll
            for (int l = 0; l < MAXL; ++l)
            {
                Vij[l] = 0.0;
                Vji[l] = 0.0;
                xl[l] = 0.0;
                el[l] = 0.0;
            }
            // gather: Transition::uv + chi_eta_aux_accum for the (ray-dependent) lines
            if (active)
            {
                chi = sChiC[k];
                eta = sEtaC[k];
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                {
                    if (l < nLine)
                    {
                        const DevTrans& tr = a.trans[ent[l].tr];
                        const int lt = la - tr.Nblue;
                        const double* par = a.par + tr.parOff + 4 * lt;
                        const double ph = a.phi[tr.phiOff + ((size_t)(lt * Nrays + mu) * 2 + toObs) * Ns + k];
                        const double vij = par[0] * ph;
                        double g = par[2];
                        if (tr.prd)
                            g *= a.rho[tr.rhoOff + (size_t)lt * Ns + k];
                        const double vji = g * vij;
                        const double uji = par[3] * vji;
                        const double ni = a.n[(size_t)tr.gi * Ns + k];
                        const double nj = a.n[(size_t)tr.gj * Ns + k];
                        const double x = ni * vij - nj * vji;
                        const double em = nj * uji;
                        Vij[l] = vij;
                        Vji[l] = vji;
                        xl[l] = x;
                        el[l] = em;
                        chi += x;
                        eta += em;
                    }
                }
                // compute_source_fn :169-179
                S = (eta + sSca[k] * sJdag[k]) / chi;
                sChi[t] = chi;
                sS[t] = S;
                if (RATES && a.storeDepth)
                {
                    const size_t o = ((size_t)(la * Nrays + mu) * 2 + toObs) * Ns + k;
                    a.depthChi[o] = chi;
                    a.depthEta[o] = eta;
                }
            }
            __syncthreads();

            // ---- formal solver: per-point coefficients e, b, Psi of I_k = e I_upwind + b ------------
            const int dk = toObs ? -1 : 1;
            const int kS = toObs ? Ns - 1 : 0;
            const int kE = toObs ? 0 : Ns - 1;
            const int cb = rl * Ns; // column base
            double e = 0.0, b = 0.0, Psi = 0.0;

            // boundary intensity (wrappers FormalScalar.cpp:471-666); needs chi before it is aliased
            double Istart = 0.0;
            if (active && k == kS)
            {
                const double zb = (SOLVER == LWHIP_FS_LINEAR_1D ? 0.5 : 1.0) / a.muz[mu];
                double dtau_b;
                if (SOLVER == LWHIP_FS_LINEAR_1D)
                    dtau_b = zb * (sChi[cb + kS] + sChi[cb + kS + dk]) * fabs(sHgt[kS] - sHgt[kS + dk]);
                else
                    dtau_b = 0.5 * zb * (sChi[cb + kS] + sChi[cb + kS + dk]) * fabs(sHgt[kS] - sHgt[kS + dk]);
                if (toObs)
                {
                    if (a.lowerBc == LWHIP_BC_THERMALISED)
                    {
                        const double B0 = d_planck(sTemp[Ns - 2], wav);
                        const double B1 = d_planck(sTemp[Ns - 1], wav);
                        Istart = B1 - (B0 - B1) / dtau_b;
                    }
                    else if (a.lowerBc == LWHIP_BC_CALLABLE)
                    {
                        const int mi = a.lowerIdx[mu * 2 + 1];
                        if (mi >= 0)
                            Istart = a.lowerBcData[(size_t)la * a.lowerNmu + mi];
                    }
                }
                else
                {
                    if (a.upperBc == LWHIP_BC_THERMALISED)
                    {
                        const double B0 = d_planck(sTemp[0], wav);
                        const double B1 = d_planck(sTemp[1], wav);
                        Istart = B0 - (B1 - B0) / dtau_b;
                    }
                    else if (a.upperBc == LWHIP_BC_CALLABLE)
                    {
                        const int mi = a.upperIdx[mu * 2 + 0];
                        if (mi >= 0)
                            Istart = a.upperBcData[(size_t)la * a.upperNmu + mi];
                    }
                }
            }

            if (SOLVER == LWHIP_FS_BEZIER3_1D)
            {
                // piecewise_bezier3_1d_impl FormalScalar.cpp:209-325, restated per point:
                // every carried quantity except I_upw is a local stencil.
                const double zmu = active ? 1.0 / a.muz[mu] : 1.0;
                // A: derivative of chi along the ray at each point
                if (active)
                {
                    double d;
                    if (k == kS)
                    {
                        const double ds = fabs(sHgt[k + dk] - sHgt[k]) * zmu;
                        d = (sChi[cb + k + dk] - sChi[cb + k]) / ds; // dx_uw :239
                    }
                    else if (k == kE)
                    {
                        const double ds = fabs(sHgt[k] - sHgt[k - dk]) * zmu;
                        d = (sChi[cb + k] - sChi[cb + k - dk]) / ds; // dx_dw :288
                    }
                    else
                    {
                        const double dsuw = fabs(sHgt[k] - sHgt[k - dk]) * zmu;
                        const double dsdw = fabs(sHgt[k + dk] - sHgt[k]) * zmu;
                        d = d_cent_deriv(dsuw, dsdw, sChi[cb + k - dk], chi, sChi[cb + k + dk]); // :240,251
                    }
                    sA[t] = d;
                }
                __syncthreads();
                // B: optical depth of the upwind interval (k-dk -> k)  :242-246, 261-263, 289-292
                double dtau = 0.0, dtauLin = 0.0;
                if (active && k != kS)
                {
                    const double ds = fabs(sHgt[k] - sHgt[k - dk]) * zmu;
                    const double cuw = sChi[cb + k - dk] + (ds / 3.0) * sA[cb + k - dk];
                    const double c0 = chi - (ds / 3.0) * sA[t];
                    dtau = ds * (sChi[cb + k - dk] + chi + cuw + c0) * 0.25;
                    if (k == kE) // piecewise linear on end :309
                        dtauLin = 0.5 * zmu * (chi + sChi[cb + k - dk]) * fabs(sHgt[k] - sHgt[k - dk]);
                }
                if (active)
                    sDtau[t] = dtau;
                __syncthreads();
                // C: derivative of S with respect to tau at each point  :247, 254-255, 268
                if (active)
                {
                    double dS = 0.0;
                    if (k == kS)
                        dS = (sS[cb + k + dk] - S) / sDtau[cb + k + dk];
                    else if (k != kE)
                        dS = d_cent_deriv(dtau, sDtau[cb + k + dk], sS[cb + k - dk], S, sS[cb + k + dk]);
                    sDS[t] = dS;
                }
                __syncthreads();
                // D: coefficients  :265-275, 294-304, 314-321
                if (active)
                {
                    if (k == kS)
                    {
                        e = 0.0;
                        b = Istart;
                        Psi = 0.0;
                    }
                    else if (k == kE)
                    {
                        const double dSuw = (S - sS[cb + k - dk]) / dtauLin;
                        double w0, w1;
                        d_w2(dtauLin, w0, w1);
                        e = 1.0 - w0;
                        b = w0 * S - w1 * dSuw;
                        Psi = w0 - w1 / dtauLin;
                    }
                    else
                    {
                        double al, be, ga, de, edt;
                        d_bezier3_coeffs(dtau, al, be, ga, de, edt);
                        const double Suw = sS[cb + k - dk];
                        const double cuw = Suw + (dtau / 3.0) * sDS[cb + k - dk];
                        const double c0 = S - (dtau / 3.0) * sDS[t];
                        e = edt;
                        b = al * Suw + be * S + ga * cuw + de * c0;
                        Psi = be + de;
                    }
                }
            }
            else if (SOLVER == LWHIP_FS_LINEAR_1D)
            {
                // piecewise_linear_1d_impl FormalScalar.cpp:136-207
                if (active)
                {
                    if (k == kS)
                    {
                        e = 0.0;
                        b = Istart;
                    }
                    else
                    {
                        const double zmu = 0.5 / a.muz[mu];
                        const double dtau = zmu * (sChi[cb + k - dk] + chi) * fabs(sHgt[k - dk] - sHgt[k]);
                        const double rcp = 1.0 / dtau;
                        const double dSuw = (sS[cb + k - dk] - S) * rcp;
                        double w0, w1;
                        d_w2(dtau, w0, w1);
                        e = 1.0 - w0;
                        b = w0 * S + w1 * dSuw;
                        Psi = w0 - w1 * rcp;
                    }
                }
            }
            else
            {
                // piecewise_besser_1d_impl FormalScalar.cpp:395-467
                if (active)
                {
                    const double zmu = 1.0 / a.muz[mu];
                    if (k == kS)
                    {
                        e = 0.0;
                        b = Istart;
                    }
                    else if (k == kE)
                    {
                        const double dtau = 0.5 * zmu * (chi + sChi[cb + k - dk]) * fabs(sHgt[k] - sHgt[k - dk]);
                        const double dSuw = (S - sS[cb + k - dk]) / dtau;
                        double w0, w1;
                        d_w2(dtau, w0, w1);
                        e = 1.0 - w0;
                        b = w0 * S - w1 * dSuw;
                        Psi = w0 - w1 / dtau;
                    }
                    else
                    {
                        const double ds_uw = fabs(sHgt[k] - sHgt[k - dk]) * zmu;
                        const double ds_dw = fabs(sHgt[k + dk] - sHgt[k]) * zmu;
                        const double chi_uw = sChi[cb + k - dk], chi_dw = sChi[cb + k + dk];
                        const double chiC = d_besser_control_point(ds_uw, ds_dw, chi_uw, chi, chi_dw);
                        const double dtauUw = (1.0 / 3.0) * (chi_uw + chiC + chi) * ds_uw;
                        const double dtauDw = 0.5 * (chi + chi_dw) * ds_dw;
                        const double Suw = sS[cb + k - dk], Sdw = sS[cb + k + dk];
                        const double SC = d_besser_control_point(dtauUw, dtauDw, Suw, S, Sdw);
                        double cM, cO, cC, edt;
                        d_besser_coeffs(dtauUw, cM, cO, cC, edt);
                        e = edt;
                        b = cM * Suw + cO * S + cC * SC;
                        Psi = cO + cC;
                    }
                }
            }
            if (SOLVER != LWHIP_FS_BEZIER3_1D)
                __syncthreads(); // all stencil reads of sChi done before it is overwritten by b
            if (active)
            {
                sE[t] = e;
                sB[t] = b;
            }
            __syncthreads();

            // ---- the recurrence: chunked scan, P lanes per ray ----------------------------------------
            {
                const int P = a.P;
                const bool scanLane = t < nr * P;
                const int rr = t / P;
                const int c = t - rr * P;
                const int Lc = (Ns + P - 1) / P;
                const int q0 = c * Lc;
                const int q1 = min(q0 + Lc, Ns);
                int up = 0;
                if (scanLane)
                    up = a.rayList[g0 + rr] & 1;
                if (scanLane)
                {
                    double E = 1.0, B = 0.0;
                    for (int q = q0; q < q1; ++q)
                    {
                        const int kk = up ? Ns - 1 - q : q;
                        const double ee = sE[rr * Ns + kk];
                        const double bb = sB[rr * Ns + kk];
                        B = ee * B + bb;
                        E = ee * E;
                    }
                    sScan[(rr * P + c) * 2] = E;
                    sScan[(rr * P + c) * 2 + 1] = B;
                }
                __syncthreads();
                if (scanLane)
                {
                    double Iin = 0.0;
                    for (int cc = 0; cc < c; ++cc)
                        Iin = sScan[(rr * P + cc) * 2] * Iin + sScan[(rr * P + cc) * 2 + 1];
                    for (int q = q0; q < q1; ++q)
                    {
                        const int kk = up ? Ns - 1 - q : q;
                        Iin = sE[rr * Ns + kk] * Iin + sB[rr * Ns + kk];
                        sB[rr * Ns + kk] = Iin;
                    }
                }
                __syncthreads();
            }

            double I = 0.0;
            if (active)
            {
                I = sB[t];
                Psi = Psi / chi; // PsiStar :322-323
                sA[t] = (0.5 * a.wmu[mu]) * I;
                if (toObs && k == 0)
                    a.I[(size_t)la * Nrays + mu] = I; // spect.I(la, mu, 0) :349
                if (RATES && a.storeDepth)
                    a.depthI[((size_t)(la * Nrays + mu) * 2 + toObs) * Ns + k] = I;
            }
            __syncthreads();
            // accumulate_J :181-190, rays in reference order
            if (RATES && t < Ns)
            {
                double j = (g0 == 0) ? 0.0 : sJ[t];
                for (int r2 = 0; r2 < nr; ++r2)
                    j += sA[r2 * Ns + t];
                sJ[t] = j;
            }

            // ---- compute_full_Ieff + compute_full_operator_rates :192-234 ------------------------------
            if (RATES && active)
            {
                const double halfwmu = 0.5 * a.wmu[mu];
                const double PsiS = a.lambdaIterate ? 0.0 : Psi;
                int lgi[MAXL], lgj[MAXL], lat[MAXL], lop[MAXL];
                double luji[MAXL];
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                {
                    lgi[l] = -1;
                    lgj[l] = -1;
                    lat[l] = -1;
                    lop[l] = 0;
                    luji[l] = 0.0;
                    if (l < nLine)
                    {
                        const DevTrans& tr = a.trans[ent[l].tr];
                        lgi[l] = tr.gi;
                        lgj[l] = tr.gj;
                        lat[l] = tr.atom;
                        lop[l] = tr.op;
                        luji[l] = a.par[tr.parOff + 4 * (la - tr.Nblue) + 3] * Vji[l];
                    }
                }
                auto levChi = [&](int g) {
                    double v = sLevChi[g * Ns + k];
#pragma unroll
                    for (int l = 0; l < MAXL; ++l)
                    {
                        if (lop[l])
                        {
                            v += (lgi[l] == g) ? xl[l] : 0.0;
                            v -= (lgj[l] == g) ? xl[l] : 0.0;
                        }
                    }
                    return v;
                };
                auto levU = [&](int g) {
                    double v = sLevU[g * Ns + k];
#pragma unroll
                    for (int l = 0; l < MAXL; ++l)
                    {
                        if (lop[l])
                            v += (lgj[l] == g) ? luji[l] : 0.0;
                    }
                    return v;
                };
                auto atomEta = [&](int at) {
                    double v = sEtaA[at * Ns + k];
#pragma unroll
                    for (int l = 0; l < MAXL; ++l)
                    {
                        if (lop[l])
                            v += (lat[l] == at) ? el[l] : 0.0;
                    }
                    return v;
                };
                // lines
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                {
                    if (l < nLine)
                    {
                        const DevLaEntry en = ent[l];
                        const DevTrans& tr = a.trans[en.tr];
                        const double* par = a.par + tr.parOff + 4 * (la - tr.Nblue);
                        const double wla = (par[1] * a.wphi[(size_t)tr.row * Ns + k]) * PI4_HC;
                        const double wlamu = wla * halfwmu;
                        double* acc = sAcc + (size_t)en.cs * 4 * Ns + k;
                        if (tr.op)
                        {
                            const double Ieff = I - PsiS * atomEta(tr.atom);
                            const double g0v = ((luji[l] + Vji[l] * Ieff) - (PsiS * levChi(tr.gi) * levU(tr.gj))) * wlamu;
                            const double g1v = ((Vij[l] * Ieff) - (PsiS * levChi(tr.gj) * levU(tr.gi))) * wlamu;
                            lds_add(acc, g0v);
                            lds_add(acc + Ns, g1v);
                        }
                        lds_add(acc + 2 * Ns, I * Vij[l] * wlamu);
                        lds_add(acc + 3 * Ns, (luji[l] + I * Vji[l]) * wlamu);
                    }
                }
                // continua
                for (int c = 0; c < nCont; ++c)
                {
                    const DevLaEntry en = ent[nLine + c];
                    const DevTrans& tr = a.trans[en.tr];
                    const double* par = a.par + tr.parOff + 4 * (la - tr.Nblue);
                    const double vij = par[0];
                    const double vji = sVji[c * Ns + k];
                    const double uji = par[2] * vji;
                    const double wlamu = par[1] * halfwmu;
                    double* acc = sAcc + (size_t)en.cs * 4 * Ns + k;
                    if (tr.op)
                    {
                        const double Ieff = I - PsiS * atomEta(tr.atom);
                        const double g0v = ((uji + vji * Ieff) - (PsiS * levChi(tr.gi) * levU(tr.gj))) * wlamu;
                        const double g1v = ((vij * Ieff) - (PsiS * levChi(tr.gj) * levU(tr.gi))) * wlamu;
                        lds_add(acc, g0v);
                        lds_add(acc + Ns, g1v);
                    }
                    lds_add(acc + 2 * Ns, I * vij * wlamu);
                    lds_add(acc + 3 * Ns, (uji + I * vji) * wlamu);
                }
            }
            __syncthreads();
        } // ray groups

        // ---- J out, dJ = max_k |1 - Jdag/J|  :477-486 ---------------------------------------------------
        if (RATES && (t & ~63) < Ns) // whole wavefronts, so the shuffles see every lane
        {
            double dj = 0.0;
            if (t < Ns)
            {
                const double Jn = sJ[t];
                a.J[(size_t)la * Ns + t] = Jn;
                dj = fabs(1.0 - sJdag[t] / Jn);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                dj = fmax(dj, __shfl_xor(dj, off));
            if ((t & 63) == 0)
                atomicMax((unsigned long long*)(a.dJ + la), (unsigned long long)__double_as_longlong(dj));
        }
        __syncthreads();
    } // wavelengths

    if (RATES)
    {
        double* out = a.partial + (size_t)chunk * a.maxCT * 4 * Ns;
        for (int idx = t; idx < a.maxCT * 4 * Ns; idx += T)
            out[idx] = sAcc[idx];
    }
}

template <int SOLVER, int MAXL, bool RATES>
static hipError_t launch_sweep_t(const SweepArgs& a, int nChunks, hipStream_t stream)
{
    const size_t lds = sweep_lds_bytes(a);
    static size_t configured = 0;
    if (lds > configured)
    {
        hipError_t err = hipFuncSetAttribute((const void*)sweep_kernel<SOLVER, MAXL, RATES>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess)
            return err;
        configured = lds;
    }
    int T = ((std::max(a.G * a.Ns, a.G * a.P) + 63) / 64) * 64;
    if (T < 64)
        T = 64;
    if (T > 1024)
        return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL((sweep_kernel<SOLVER, MAXL, RATES>), dim3(nChunks), dim3(T), lds, stream, a);
    return hipGetLastError();
}

template <int SOLVER, bool RATES>
static hipError_t launch_sweep_l(const SweepArgs& a, int maxL, int nChunks, hipStream_t stream)
{
    if (maxL <= 2)
        return launch_sweep_t<SOLVER, 2, RATES>(a, nChunks, stream);
    if (maxL <= 4)
        return launch_sweep_t<SOLVER, 4, RATES>(a, nChunks, stream);
    return launch_sweep_t<SOLVER, 8, RATES>(a, nChunks, stream);
}

hipError_t launch_sweep(const SweepArgs& a, int solver, int maxL, bool rates, int nChunks,
                        hipStream_t stream)
{
    switch (solver)
    {
        case LWHIP_FS_LINEAR_1D:
            return rates ? launch_sweep_l<LWHIP_FS_LINEAR_1D, true>(a, maxL, nChunks, stream)
                         : launch_sweep_l<LWHIP_FS_LINEAR_1D, false>(a, maxL, nChunks, stream);
        case LWHIP_FS_BESSER_1D:
            return rates ? launch_sweep_l<LWHIP_FS_BESSER_1D, true>(a, maxL, nChunks, stream)
                         : launch_sweep_l<LWHIP_FS_BESSER_1D, false>(a, maxL, nChunks, stream);
        default:
            return rates ? launch_sweep_l<LWHIP_FS_BEZIER3_1D, true>(a, maxL, nChunks, stream)
                         : launch_sweep_l<LWHIP_FS_BEZIER3_1D, false>(a, maxL, nChunks, stream);
    }
}

// ---- slab reduction: fixed-order sum of the per-chunk partials of each transition ----------------------
// replaces AtomStorageFactory::accumulate_Gamma / TransitionStorageFactory::accumulate_rates
// (ThreadStorage.cpp:73-88,150-166)
__global__ void reduce_kernel(const ReduceArgs a)
{
    const int tr = blockIdx.x;
    const int n4 = 4 * a.Ns;
    for (int idx = threadIdx.x; idx < n4; idx += blockDim.x)
    {
        double s = 0.0;
        for (int e = a.trOff[tr]; e < a.trOff[tr + 1]; ++e)
            s += a.partial[(size_t)a.trRow[e] * n4 + idx];
        a.red[(size_t)tr * n4 + idx] = s;
    }
}

hipError_t launch_reduce(const ReduceArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(reduce_kernel, dim3(a.Ntrans), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---- apply: Gamma += radiative terms (Gamma keeps its collisional pre-fill), finalise_Gamma
// (SimdFullIterationTemplates.hpp:491-508), Rij/Rji out ----------------------------------------------------
__global__ void apply_kernel(const ApplyArgs a)
{
    const int at = blockIdx.x;
    const int Ns = a.Ns;
    const int Nl = a.atomNlevel[at];
    for (int k = threadIdx.x; k < Ns; k += blockDim.x)
    {
        double* G = a.Gamma + a.atomGammaOff[at];
        for (int tr = a.atomTrOff[at]; tr < a.atomTrOff[at + 1]; ++tr)
        {
            const double* r = a.red + (size_t)tr * 4 * Ns;
            if (!a.atomDetailed[at])
            {
                const int i = a.transLi[tr], j = a.transLj[tr];
                G[((size_t)i * Nl + j) * Ns + k] += r[k];
                G[((size_t)j * Nl + i) * Ns + k] += r[Ns + k];
            }
            a.Rij[(size_t)tr * Ns + k] = r[2 * Ns + k];
            a.Rji[(size_t)tr * Ns + k] = r[3 * Ns + k];
        }
        if (!a.atomDetailed[at])
        {
            for (int i = 0; i < Nl; ++i)
            {
                G[((size_t)i * Nl + i) * Ns + k] = 0.0;
                double gd = 0.0;
                for (int j = 0; j < Nl; ++j)
                    gd += G[((size_t)j * Nl + i) * Ns + k];
                G[((size_t)i * Nl + i) * Ns + k] = -gd;
            }
        }
    }
}

hipError_t launch_apply(const ApplyArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(apply_kernel, dim3(a.Natom), dim3(128), 0, stream, a);
    return hipGetLastError();
}

// ---- dJMax and the index of its first occurrence (max_idx, Constants.hpp:114-125) ---------------------------
__global__ void djmax_kernel(const double* dJ, int laStart, int laEnd, int idxOffset, double* out2)
{
    __shared__ double sV[1024];
    __shared__ int sI[1024];
    double best = -1.0;
    int bi = 0;
    for (int la = laStart + threadIdx.x; la < laEnd; la += blockDim.x)
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
    if (threadIdx.x == 0)
    {
        out2[0] = sV[0] < 0.0 ? 0.0 : sV[0];
        out2[1] = (double)(sI[0] + idxOffset);
    }
}

hipError_t launch_djmax(const double* dJ, int laStart, int laEnd, int idxOffset, double* out2,
                        hipStream_t stream)
{
    hipLaunchKernelGGL(djmax_kernel, dim3(1), dim3(1024), 0, stream, dJ, laStart, laEnd, idxOffset, out2);
    return hipGetLastError();
}

// ---- Gamma <- crsw * C (LwMiddleLayer.pyx:3198-3203) ------------------------------------------------------------
__global__ void prefill_kernel(double* G, const double* C, double crsw, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        G[i] = crsw * C[i];
}

hipError_t launch_prefill(double* Gamma, const double* C, double crsw, size_t n, hipStream_t stream)
{
    const int blocks = (int)((n + 255) / 256);
    hipLaunchKernelGGL(prefill_kernel, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, stream, Gamma, C, crsw, n);
    return hipGetLastError();
}

// ---- stat_eq: one thread per depth point; Crout LU with implicit scaling + one refinement pass
// (UpdatePopulations.cpp:7-47, LuSolve.cpp:8-132).  Nlevel <= MAXN, matrix in registers/scratch. -----------
template <int MAXN>
__global__ void stat_eq_kernel(int N, int Ns, double* n, const double* nTotal, const double* Gamma,
                               int* status)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Ns)
        return;
    double A[MAXN * MAXN], A0[MAXN * MAXN], bvec[MAXN], b0[MAXN], res[MAXN], vv[MAXN];
    int index[MAXN];
    // Gamma_k and the elimination row
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
    for (int i = 0; i < N; ++i)
    {
        b0[i] = bvec[i];
        for (int j = 0; j < N; ++j)
            A0[i * MAXN + j] = A[i * MAXN + j];
    }
    // lu_decompose
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
    {
        atomicExch(status, LWHIP_ERR_SINGULAR);
        return;
    }
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
    auto backsub = [&](double* b) {
        int ii = -1;
        for (int i = 0; i < N; ++i)
        {
            const int ip = index[i];
            double sum = b[ip];
            b[ip] = b[i];
            if (ii >= 0)
            {
                for (int j = ii; j < i; ++j)
                    sum -= A[i * MAXN + j] * b[j];
            }
            else if (sum != 0.0)
            {
                ii = i;
            }
            b[i] = sum;
        }
        for (int i = N - 1; i >= 0; --i)
        {
            double sum = b[i];
            for (int j = i + 1; j < N; ++j)
                sum -= A[i * MAXN + j] * b[j];
            b[i] = sum / A[i * MAXN + i];
        }
    };
    backsub(bvec);
    for (int i = 0; i < N; ++i)
    {
        double r = b0[i];
        for (int j = 0; j < N; ++j)
            r -= A0[i * MAXN + j] * bvec[j];
        res[i] = r;
    }
    backsub(res);
    for (int i = 0; i < N; ++i)
        n[(size_t)i * Ns + k] = bvec[i] + res[i];
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
}

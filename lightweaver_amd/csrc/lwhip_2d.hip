// lwhip_2d.hip -- the pointwise halves of the 2D iteration: intensity_core_opt
// (Source/SimdFullIterationTemplates.hpp:238-487) around the 2D formal solver of lwhip_fs2d.hip.
//
// A 2D domain (Nz x Nx = 20 992 points per ray at BASELINE config 5) does not fit the fused 1D sweep, so the
// iteration runs batch by batch over the wavelengths with its per-ray intermediates in HBM:
//   cont_kernel                     continuum rows of the batch, once per (wavelength, point)
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
#define ARGS (*launder_args(ap))

// ---- continuum rows: everything about the continua that does not depend on the ray ------------------
// Atom::setup_wavelength (LwAtom.hpp:107-115) + Transition::uv (LwTransition.hpp:132-142) +
// chi_eta_aux_accum (SimdFullIterationTemplates.hpp:59-109) for the continua of one wavelength:
// Vji_c, the totals chiC = bgChi + sum chi_c, etaC = bgEta + sum eta_c, and the per-atom / per-level
// sums the preconditioned rates need, written once per (wavelength, depth) for all rays to read.
__global__ void __launch_bounds__(128) cont_kernel(const ContArgs* __restrict__ ap, int contOff, double* rowsBase)
{
    dbg_poison_lds();
    extern __shared__ double lds[]; // [2 * maxC][blockDim.x]: thread-private columns
    const int Ns = ARGS.Ns;
    const int T = blockDim.x;
    const int la = CTAB(int32_t, ARGS.contLa)[contOff + blockIdx.x];
    const DevLaHeader h = ld_c(CTAB(DevLaHeader, ARGS.laHdr) + la);
    const CONST_AS DevSlot* slots = CTAB(DevSlot, ARGS.slots) + h.off + h.nLine;
    const int nCont = h.nMixed + h.nPure;
    const DevProgram pg = ld_c(CTAB(DevProgram, ARGS.progs) + h.prog);
    const CONST_AS DevProgRow* prow = CTAB(DevProgRow, ARGS.progRows) + pg.rowOff;
    const CONST_AS DevProgEnt* pent = CTAB(DevProgEnt, ARGS.progEnts) + pg.entOff;
    const double hc_kl = HC_K / CTAB(double, ARGS.wavelength)[la];
    double* colV = lds + threadIdx.x;
    double* colX = lds + (size_t)ARGS.maxC * T + threadIdx.x;
    const GLOBAL_AS double* gN = GCD(ARGS.n);
    // rowsBase: a window of the row numbering (2D iteration: only a batch of wavelengths is resident)
    GLOBAL_AS double* out = (rowsBase ? (GLOBAL_AS double*)rowsBase : GD(ARGS.rows)) + (size_t)h.rowBase * Ns;
    // blockIdx.y strides over the points (2D domains: tens of thousands of points per wavelength)
    for (int k = blockIdx.y * T + threadIdx.x; k < Ns; k += gridDim.y * T)
    {
        const double boltz = d_exp_neg(-hc_kl * (1.0 / GCD(ARGS.temperature)[k]));
        for (int c = 0; c < nCont; ++c)
        {
            const CONST_AS DevSlot& sl = slots[c];
            const double gij = GCD(ARGS.ratio)[(size_t)sl.row * Ns + k] * boltz;
            const double vji = gij * sl.p[0];
            colV[(size_t)c * T] = vji;
            colX[(size_t)c * T] = gN[sl.gi * Ns + k] * sl.p[0] - gN[sl.gj * Ns + k] * vji;
            out[(size_t)(h.nRows + c) * Ns + k] = vji;
        }
        // (measured: requesting the next row's header and a row's entries four at a time -- one wait per row instead of
        // 1 + nEnt -- changes nothing, 0.197 ms either way: the kernel is not bound by this chain of scalar loads)
        for (int ri = 0; ri < pg.nRows; ++ri)
        {
            const DevProgRow pr = ld_c(prow + ri);
            double v = 0.0;
            if (pr.row == 0)
                v = GCD(ARGS.bgChi)[(size_t)la * Ns + k];
            else if (pr.row == 1)
                v = GCD(ARGS.bgEta)[(size_t)la * Ns + k];
            for (int e = 0; e < pr.nEnt; ++e) // continua in reference order
            {
                const DevProgEnt en = ld_c(pent + pr.entOff + e);
                const double x = colX[(size_t)en.c * T];
                const double uji = slots[en.c].p[2] * colV[(size_t)en.c * T];
                if (en.kind == 0)
                    v += x;
                else if (en.kind == 1)
                    v -= x;
                else if (en.kind == 2)
                    v += uji;
                else
                    v += gN[slots[en.c].gj * Ns + k] * uji;
            }
            out[(size_t)ri * Ns + k] = v;
        }
    }
}

hipError_t launch_cont_range(const ContArgs* devArgs, const ContArgs& a, int contOff, int nCont, double* rowsBase,
                             hipStream_t stream)
{
    if (nCont <= 0)
        return hipSuccess;
    const int T = 128;
    const size_t lds = (size_t)2 * a.maxC * T * sizeof(double);
    if (lds > 48 * 1024)
    {
        hipError_t err = hipFuncSetAttribute((const void*)cont_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess)
            return err;
    }
    const int ky = std::max(1, std::min(256, (a.Ns + T - 1) / T));
    LWHIP_LAUNCH(cont_kernel, dim3(nCont, ky), dim3(T), lds, stream, devArgs, contOff, rowsBase);
    return hipGetLastError();
}
#undef ARGS

namespace
{
// ---- gather: Transition::uv + chi_eta_aux_accum + compute_source_fn for one (wavelength, point), ray by ray --------
// Everything but the line profile is the same for all rays of the wavelength -- the continuum rows, the populations,
// rho, the scattering term -- and is fetched once; the rays go four at a time so that their profile loads are in
// flight together (the stores of a ray keep the compiler from hoisting the next ray's loads).
template <int MAXL>
__global__ void __launch_bounds__(256) gather2d_kernel(const Batch2dArgs a)
{
    dbg_poison_lds();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.Ns)
        return;
    const int b = blockIdx.y;
    const int la = a.la0 + b;
    const int Ns = a.Ns;
    // the activity tables are read through the constant address space: scalar loads (a generic pointer next to the
    // kernel's stores would be read with vector loads even at uniform addresses)
    const DevLaHeader h = ld_c(CTAB(DevLaHeader, a.laHdr) + la);
    const CONST_AS DevSlot* slots = CTAB(DevSlot, a.slots) + h.off;
    const size_t laNs = (size_t)la * Ns;
    double chi0, eta0;
    if (h.rowBase >= 0)
    {
        chi0 = a.rows[(size_t)h.rowBase * Ns + k];
        eta0 = a.rows[((size_t)h.rowBase + 1) * Ns + k];
    }
    else
    {
        chi0 = a.bgChi[laNs + k];
        eta0 = a.bgEta[laNs + k];
    }
    const double sca = a.bgSca[laNs + k] * a.J[laNs + k];
    double lV[MAXL], lG[MAXL], lNi[MAXL], lNj[MAXL], lU[MAXL];
    size_t lOff[MAXL];
#pragma unroll
    for (int l = 0; l < MAXL; ++l)
    {
        lV[l] = lG[l] = lNi[l] = lNj[l] = lU[l] = 0.0;
        lOff[l] = 0;
        if (l < h.nLine)
        {
            const CONST_AS DevSlot& sl = slots[l];
            double g = sl.p[2];
            if (sl.flags & 2)
                g *= a.rho[sl.rhoOff + k];
            lV[l] = sl.p[0];
            lG[l] = g;
            lU[l] = sl.p[3];
            lNi[l] = a.n[(size_t)sl.gi * Ns + k];
            lNj[l] = a.n[(size_t)sl.gj * Ns + k];
            lOff[l] = sl.phiOff + k;
        }
    }
    constexpr int RB = 4;
    const int nR = a.nRaysActive;
    for (int r0 = 0; r0 < nR; r0 += RB)
    {
        double ph[RB][MAXL];
        int rayOf[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q)
        {
            rayOf[q] = CTAB(int32_t, a.rayList)[min(r0 + q, nR - 1)];
#pragma unroll
            for (int l = 0; l < MAXL; ++l)
                ph[q][l] = (l < h.nLine) ? a.phi[lOff[l] + (size_t)rayOf[q] * Ns] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < RB; ++q)
        {
            if (r0 + q >= nR)
                break;
            double chi = chi0, eta = eta0;
#pragma unroll
            for (int l = 0; l < MAXL; ++l)
                if (l < h.nLine)
                {
                    const double vij = lV[l] * ph[q][l];
                    const double vji = lG[l] * vij;
                    chi += lNi[l] * vij - lNj[l] * vji;
                    eta += lNj[l] * (lU[l] * vji);
                }
            const size_t o = ((size_t)b * nR + (r0 + q)) * Ns + k;
            a.cs[o] = make_double2(chi, (eta + sca) / chi);
            if (a.storeDepth)
            {
                const size_t d = ((size_t)la * a.Nrays * 2 + rayOf[q]) * Ns + k;
                a.depthChi[d] = chi;
                a.depthEta[d] = eta;
            }
        }
    }
}

// ---- rates: one thread per point walks the wavelengths of the batch and the rays ----------------------------
// Per wavelength, everything that does not depend on the ray -- populations, rho, the line weight, the continuum
// rows the level sums start from -- is fetched once into registers / the thread's LDS column; the ray loop then
// reads only I, Psi* and one phi per line from memory.
// MAXM: mixed continua integrated ray by ray at one wavelength (the host refuses a 2D problem with more than 4); MAXP: pure
// continua with register accumulators, any beyond go straight to memory.  The accumulators are most of the kernel's registers
// (4 doubles per slot): the host picks the smallest instance that covers the problem -- <2, 2, 5> (hydrogen-like problems:
// 36 doubles of sums) instead of <2, 4, 8> (56 doubles, 161 spilled registers at the 168 the occupancy allows).
#ifndef R2D_WAVES
#define R2D_WAVES 3 // wavefronts per SIMD the register budget is cut for
#endif
#ifndef R2D_RB
#define R2D_RB 4 // rays whose loads are requested together
#endif
template <int MAXL, int MAXM, int MAXP>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(R2D_WAVES, R2D_WAVES))) rates2d_kernel(const Batch2dArgs a)
{
    dbg_poison_lds();
    extern __shared__ double sRowAll[]; // [maxRowsLa][128]: every continuum row of the wavelength at the thread's point
    const int tid = threadIdx.x;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int Ns = a.Ns;
    const bool on = k < Ns;
    const int kk = on ? k : 0;
    // the rays of the pass and their half weights, once, behind the row block (uniform LDS reads in the ray loop:
    // no scalar-memory round trips there)
    double* sHalfW = sRowAll + (size_t)a.maxRowsLa * 128;
    int* sRayId = (int*)(sHalfW + 64);
    constexpr int MAXSLOT = MAXL + MAXM + MAXP;
    DevSlot* sSlot = (DevSlot*)(sHalfW + 64 + 32);
    if (tid < a.nRaysActive)
    {
        const int ray = a.rayList[tid];
        sRayId[tid] = ray;
        sHalfW[tid] = 0.5 * a.wmu[ray >> 1];
    }
    __syncthreads();
    double Wsum = 0.0;
    for (int r = 0; r < a.nRaysActive; ++r)
        Wsum += sHalfW[r];
    // blockIdx.y: a contiguous group of the batch's wavelengths with its own copy of the running sums (added up
    // afterwards).  Contiguous, because neighbouring wavelengths belong to the same transitions: the thread keeps a
    // transition's four sums in registers for as long as it stays in the same slot and touches its column of the
    // running sums only when the slot changes hands (a line's ~100 wavelengths -> one read-modify-write, not 100).
    double* redG = a.red + (size_t)blockIdx.y * a.redStride;
    double accL[MAXL][4], accM[MAXM][4], accP[MAXP][4];
    int trL[MAXL], trM[MAXM], trP[MAXP];
#pragma unroll
    for (int l = 0; l < MAXL; ++l)
    {
        accL[l][0] = accL[l][1] = accL[l][2] = accL[l][3] = 0.0;
        trL[l] = -1;
    }
#pragma unroll
    for (int c = 0; c < MAXM; ++c)
    {
        accM[c][0] = accM[c][1] = accM[c][2] = accM[c][3] = 0.0;
        trM[c] = -1;
    }
#pragma unroll
    for (int c = 0; c < MAXP; ++c)
    {
        accP[c][0] = accP[c][1] = accP[c][2] = accP[c][3] = 0.0;
        trP[c] = -1;
    }
    auto hand_over = [&](double (&acc)[4], int& trCur, int trNew) {
        if (trNew == trCur)
            return;
        if (trCur >= 0 && on)
        {
            double* red = redG + ((size_t)trCur * 4) * Ns + k;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                red[(size_t)q * Ns] += acc[q];
        }
        acc[0] = acc[1] = acc[2] = acc[3] = 0.0;
        trCur = trNew;
    };
    const int bLo = (int)(((long long)blockIdx.y * a.nLa) / gridDim.y);
    const int bHi = (int)(((long long)(blockIdx.y + 1) * a.nLa) / gridDim.y);
    for (int b = bLo; b < bHi; ++b)
    {
        const int la = a.la0 + b;
        const DevLaHeader h = ld_c(CTAB(DevLaHeader, a.laHdr) + la);
        const int nLine = h.nLine, nMixed = h.nMixed, nPure = h.nPure;
        // the wavelength's slot records into LDS: LDS reads are not ordered against the global stores of the ray
        // loop, so the compiler keeps or re-reads them cheaply (from global memory every field access would be a
        // scalar load re-issued after each store, and waited for)
        const int nSlot = min(nLine + nMixed + nPure, MAXSLOT);
        __syncthreads();
        for (int i = tid; i < nSlot * (int)(sizeof(DevSlot) / 4); i += blockDim.x)
            ((int*)sSlot)[i] = ((const int*)(a.slots + h.off))[i];
        __syncthreads();
        const DevSlot* slots = sSlot;
        const bool hasRows = h.rowBase >= 0;
        const size_t laNs = (size_t)la * Ns;
        // the wavelength's block of continuum rows (deduplicated on the host: chi_c, eta_c, per-atom eta, per-level
        // chi / U, one Vji per continuum), once from memory into the thread's LDS column
        const int nRowsLa = hasRows ? h.nRows + nMixed + nPure : 0;
        for (int ri = 0; ri < nRowsLa; ++ri)
            sRowAll[ri * 128 + tid] = a.rows[((size_t)h.rowBase + ri) * Ns + kk];
        auto rowAt = [&](int ri) -> double { return (hasRows && ri >= 0) ? sRowAll[ri * 128 + tid] : 0.0; };
        // slots that changed hands since the previous wavelength give up their sums
#pragma unroll
        for (int l = 0; l < MAXL; ++l)
            hand_over(accL[l], trL[l], (l < nLine) ? CTAB(int32_t, a.slotTr)[h.off + l] : -1);
#pragma unroll
        for (int c = 0; c < MAXM; ++c)
            hand_over(accM[c], trM[c], (c < nMixed) ? CTAB(int32_t, a.slotTr)[h.off + nLine + c] : -1);
#pragma unroll
        for (int c = 0; c < MAXP; ++c)
            hand_over(accP[c], trP[c], (c < nPure) ? CTAB(int32_t, a.slotTr)[h.off + nLine + nMixed + c] : -1);
        double mJ = 0.0, mM1 = 0.0, mE0 = 0.0, mE1 = 0.0;
        double mP0[MAXL], mP1[MAXL], mQ1[MAXL], mQ2[MAXL][MAXL];
#pragma unroll
        for (int l = 0; l < MAXL; ++l)
        {
            mP0[l] = mP1[l] = mQ1[l] = 0.0;
#pragma unroll
            for (int m = 0; m < MAXL; ++m)
                mQ2[l][m] = 0.0;
        }
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
            }
        }
        const bool topRow = kk < a.Nx && on;
        // the rays' I, Psi* and profiles are requested RB at a time and then worked through: with the moments the ray loop
        // holds no per-line temporaries any more, so four rays of loads in flight fit the registers (one ray of look-ahead
        // -- 8 VGPRs -- was all the ray-by-ray integrands left room for), and a wavelength costs two load round trips, not eight
        const int nR = a.nRaysActive;
        constexpr int RB = R2D_RB; // (measured again with the <2, 2, 5> instance, 2.11 ms per iteration: RB 2 -> 2.14, 4 wavefronts
                                   // per SIMD with RB 4 / 2 -> 2.19 / 2.17, 2 wavefronts with RB 8 -> 2.15)
        for (int r0 = 0; r0 < nR; r0 += RB)
        {
            double bI[RB], bP[RB], bPhi[RB][MAXL];
#pragma unroll
            for (int q = 0; q < RB; ++q)
            {
                const int r = min(r0 + q, nR - 1);
                const size_t o = ((size_t)b * nR + r) * Ns + kk;
                bI[q] = a.I[o];
                bP[q] = a.lambdaIterate ? 0.0 : a.PsiStar[o];
                const size_t rayOff = (size_t)sRayId[r] * Ns + kk;
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                    bPhi[q][l] = (l < nLine) ? a.phi[slots[l].phiOff + rayOff] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < RB; ++q)
            {
            const int r = r0 + q;
            if (r >= nR)
                break;
            const int ray = sRayId[r];
            const int mu = ray >> 1;
            const double halfwmu = sHalfW[r];
            const double I = bI[q];
            const double PsiS = bP[q];
            double phiR[MAXL];
#pragma unroll
            for (int l = 0; l < MAXL; ++l)
                phiR[l] = bPhi[q][l];
            const double wPsi = halfwmu * PsiS;
            mJ += halfwmu * I; // accumulate_J :181-190
            mM1 += wPsi;
            if ((ray & 1) && topRow) // Spectrum::I: the top row, last written by the toObs ray (:365-370)
                a.Iout[((size_t)la * a.Nrays + mu) * a.Nx + kk] = I;
            if (a.storeDepth && on)
                a.depthI[((size_t)la * a.Nrays * 2 + ray) * Ns + kk] = I;
            if (on) // ZPlaneDecomposition :372-384: plane 1 of the up rays, plane Nz - 2 of the down rays
            {
                if ((ray & 1) && a.zUp && kk >= a.Nx && kk < 2 * a.Nx)
                    a.zUp[((size_t)la * a.Nrays + mu) * a.Nx + (kk - a.Nx)] = I;
                else if (!(ray & 1) && a.zDown && kk >= Ns - 2 * a.Nx && kk < Ns - a.Nx)
                    a.zDown[((size_t)la * a.Nrays + mu) * a.Nx + (kk - (Ns - 2 * a.Nx))] = I;
            }
            // the angle moments of the line and mixed-continuum rates (lwhip_lanesweep.hip: every integrand is a polynomial in
            // the rays' profiles with ray-independent coefficients, so the sums over the rays need only P0_l = sum w phi_l,
            // P1_l = sum w phi_l I, Q_l = sum w Psi* phi_l, Q_ll' = sum w Psi* phi_l phi_l' next to J and M1): six flops per
            // ray and line here, the integrals once per wavelength below -- no continuum row, no level matching in the ray loop
#pragma unroll
            for (int l = 0; l < MAXL; ++l)
            {
                if (l < nLine)
                {
                    const double wa = halfwmu * phiR[l], wb = wPsi * phiR[l];
                    mP0[l] += wa;
                    mP1[l] = fma(wa, I, mP1[l]);
                    mQ1[l] += wb;
#pragma unroll
                    for (int m = l; m < MAXL; ++m)
                        if (m < nLine)
                            mQ2[l][m] = fma(wb, phiR[m], mQ2[l][m]);
                }
            }
            } // rays of the round
        } // rounds
        if (nLine + nMixed > 0)
        {
            // per unit profile: V_ij = c1 phi, V_ji = c2 phi, U_ji = cu phi, chi_line = cx phi, eta_line = ce phi
            double c1[MAXL], c2[MAXL], cu[MAXL], cx[MAXL], ce[MAXL];
#pragma unroll
            for (int l = 0; l < MAXL; ++l)
            {
                c1[l] = c2[l] = cu[l] = cx[l] = ce[l] = 0.0;
                if (l < nLine)
                {
                    const DevSlot& sl = slots[l];
                    c1[l] = sl.p[0];
                    c2[l] = lG[l] * sl.p[0];
                    cu[l] = sl.p[3] * c2[l];
                    cx[l] = lNi[l] * c1[l] - lNj[l] * c2[l];
                    ce[l] = lNj[l] * cu[l];
                    if (lop[l])
                    {
                        if (sl.ms == 0)
                            mE0 = fma(ce[l], mQ1[l], mE0);
                        else if (sl.ms == 1)
                            mE1 = fma(ce[l], mQ1[l], mE1);
                    }
                }
            }
            auto Q2of = [&](int l, int m) { return l <= m ? mQ2[l][m] : mQ2[m][l]; };
            // a level sum X = X0 + sum_l X_l phi_l (X0: the continuum row; X_l from the active lines that share the level /
            // the atom: chi_eta_aux_accum's bookkeeping);  L(X) = sum w Psi* X,  L_s(X) = sum w Psi* phi_s X,
            // B(X, Y) = sum w Psi* X Y in terms of the moments
            struct Aff
            {
                double x0, x[MAXL];
            };
            auto levChi = [&](int g, double row) {
                Aff f{ row, {} };
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                    f.x[l] = (lop[l] && lgi[l] == g) ? cx[l] : (lop[l] && lgj[l] == g) ? -cx[l] : 0.0;
                return f;
            };
            auto levU = [&](int g, double row) {
                Aff f{ row, {} };
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                    f.x[l] = (lop[l] && lgj[l] == g) ? cu[l] : 0.0;
                return f;
            };
            auto atomEta = [&](int at, double row) {
                Aff f{ row, {} };
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                    f.x[l] = (lop[l] && lat[l] == at) ? ce[l] : 0.0;
                return f;
            };
            auto Lin = [&](const Aff& X) {
                double r = X.x0 * mM1;
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                    r = fma(X.x[l], mQ1[l], r);
                return r;
            };
            auto LinS = [&](int s, const Aff& X) {
                double r = X.x0 * mQ1[s];
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                    r = fma(X.x[l], Q2of(s, l), r);
                return r;
            };
            auto Bil = [&](const Aff& X, const Aff& Y) {
                double r = (X.x0 * Y.x0) * mM1;
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                {
                    r = fma(fma(X.x0, Y.x[l], X.x[l] * Y.x0), mQ1[l], r);
#pragma unroll
                    for (int m = 0; m < MAXL; ++m)
                        r = fma(X.x[l] * Y.x[m], Q2of(l, m), r);
                }
                return r;
            };
#pragma unroll
            for (int l = 0; l < MAXL; ++l)
            {
                if (l < nLine)
                {
                    const DevSlot& sl = slots[l];
                    const double up = fma(cu[l], mP0[l], c2[l] * mP1[l]); // sum w (U_ji + V_ji I)
                    if (lop[l])
                    {
                        // compute_full_Ieff :192-204, compute_full_operator_rates :206-234, summed over the rays
                        const double LsE = LinS(l, atomEta(lat[l], rowAt(sl.rEtaA)));
                        accL[l][0] += ((up - c2[l] * LsE) - Bil(levChi(lgi[l], rowAt(sl.rChiI)), levU(lgj[l], rowAt(sl.rUJ)))) * lW[l];
                        accL[l][1] += ((c1[l] * (mP1[l] - LsE)) - Bil(levChi(lgj[l], rowAt(sl.rChiJ)), levU(lgi[l], rowAt(sl.rUI)))) * lW[l];
                    }
                    accL[l][2] += (c1[l] * mP1[l]) * lW[l];
                    accL[l][3] += up * lW[l];
                }
            }
#pragma unroll
            for (int c = 0; c < MAXM; ++c)
            {
                if (c < nMixed) // mixed continua (always of active atoms)
                {
                    const DevSlot& sl = slots[nLine + c];
                    const double vij = sl.p[0], vji = rowAt(sl.rVji);
                    const double uji = sl.p[2] * vji;
                    const double Jeff = mJ - Lin(atomEta(sl.atom, rowAt(sl.rEtaA)));
                    accM[c][0] += ((uji * Wsum + vji * Jeff) - Bil(levChi(sl.gi, rowAt(sl.rChiI)), levU(sl.gj, rowAt(sl.rUJ)))) * sl.p[1];
                    accM[c][1] += ((vij * Jeff) - Bil(levChi(sl.gj, rowAt(sl.rChiJ)), levU(sl.gi, rowAt(sl.rUI)))) * sl.p[1];
                    accM[c][2] += (mJ * vij) * sl.p[1];
                    accM[c][3] += (uji * Wsum + mJ * vji) * sl.p[1];
                }
            }
        }
        // pure continua from the angle moments
        auto pure = [&](int cq, double (&out)[4]) {
            const int si = nLine + nMixed + cq;
            const DevSlot& sl = (si < MAXSLOT) ? slots[si] : a.slots[h.off + si];
            const double vij = sl.p[0];
            const double vji = rowAt(sl.rVji);
            const double uji = sl.p[2] * vji;
            const double wla = sl.p[1];
            out[0] = out[1] = 0.0;
            if (sl.flags & 1)
            {
                const double Meta = (sl.ms == 0) ? mE0 : ((sl.ms == 1) ? mE1 : 0.0);
                const double Jeff = mJ - mM1 * rowAt(sl.rEtaA) - Meta;
                out[0] = ((uji * Wsum + vji * Jeff) - (mM1 * rowAt(sl.rChiI) * rowAt(sl.rUJ))) * wla;
                out[1] = ((vij * Jeff) - (mM1 * rowAt(sl.rChiJ) * rowAt(sl.rUI))) * wla;
            }
            out[2] = mJ * vij * wla;
            out[3] = (uji * Wsum + mJ * vji) * wla;
        };
#pragma unroll
        for (int cq = 0; cq < MAXP; ++cq)
            if (cq < nPure)
            {
                double v[4];
                pure(cq, v);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    accP[cq][q] += v[q];
            }
        for (int cq = MAXP; cq < nPure && on; ++cq)
        {
            double v[4];
            pure(cq, v);
            double* red = redG + ((size_t)CTAB(int32_t, a.slotTr)[h.off + nLine + nMixed + cq] * 4) * Ns + k;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                red[(size_t)q * Ns] += v[q];
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
#pragma unroll
    for (int l = 0; l < MAXL; ++l)
        hand_over(accL[l], trL[l], -1);
#pragma unroll
    for (int c = 0; c < MAXM; ++c)
        hand_over(accM[c], trM[c], -1);
#pragma unroll
    for (int c = 0; c < MAXP; ++c)
        hand_over(accP[c], trP[c], -1);
}
}

// ---- formal_sol (intensity only, formal_sol_impl :722-784): the emergent intensity of the up-going rays and, when
// asked for, I at depth; J, Gamma and the rates stay untouched ---------------------------------------------------
namespace
{
__global__ void __launch_bounds__(256) iout2d_kernel(const Batch2dArgs a)
{
    dbg_poison_lds();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const bool planes = a.zUp || a.zDown;
    if (k >= a.Ns || (!a.storeDepth && !planes && k >= a.Nx))
        return;
    const int r = blockIdx.y, b = blockIdx.z;
    const int la = a.la0 + b;
    const int ray = a.rayList[r];
    const double I = a.I[((size_t)b * a.nRaysActive + r) * a.Ns + k];
    if ((ray & 1) && a.zUp && k >= a.Nx && k < 2 * a.Nx) // ZPlaneDecomposition :372-384
        a.zUp[((size_t)la * a.Nrays + (ray >> 1)) * a.Nx + (k - a.Nx)] = I;
    else if (!(ray & 1) && a.zDown && k >= a.Ns - 2 * a.Nx && k < a.Ns - a.Nx)
        a.zDown[((size_t)la * a.Nrays + (ray >> 1)) * a.Nx + (k - (a.Ns - 2 * a.Nx))] = I;
    if ((ray & 1) && k < a.Nx) // Spectrum::I: the top row, written by the toObs ray (:365-370)
        a.Iout[((size_t)la * a.Nrays + (ray >> 1)) * a.Nx + k] = I;
    if (a.storeDepth)
        a.depthI[((size_t)la * a.Nrays * 2 + ray) * a.Ns + k] = I;
}
}

hipError_t launch_iout2d(const Batch2dArgs& a, hipStream_t stream)
{
    const int n = (a.storeDepth || a.zUp || a.zDown) ? a.Ns : a.Nx;
    LWHIP_LAUNCH(iout2d_kernel, dim3((n + 255) / 256, a.nRaysActive, a.nLa), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// red[0] += red[1] + ... + red[G-1] (fixed order)
__global__ void sum_groups_kernel(double* red, size_t n, size_t stride, int G)
{
    dbg_poison_lds();
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
    LWHIP_LAUNCH(sum_groups_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, stream, red,
                       n, stride, G);
    return hipGetLastError();
}

hipError_t launch_gather2d(const Batch2dArgs& a, int maxL, hipStream_t stream)
{
    const dim3 grid((a.Ns + 255) / 256, a.nLa);
    if (maxL <= 2)
        LWHIP_LAUNCH(gather2d_kernel<2>, grid, dim3(256), 0, stream, a);
    else if (maxL <= 4)
        LWHIP_LAUNCH(gather2d_kernel<4>, grid, dim3(256), 0, stream, a);
    else
        LWHIP_LAUNCH(gather2d_kernel<8>, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_rates2d(const Batch2dArgs& a, int maxL, int maxM, int maxP, hipStream_t stream)
{
    if (maxM > 4)
        return hipErrorInvalidValue; // (refused at lwhip_create)
    const dim3 grid((a.Ns + 127) / 128, a.nGroups);
    const size_t lds = (size_t)std::max(a.maxRowsLa, 1) * 128 * sizeof(double) + 64 * (sizeof(double) + sizeof(int))
                       + (size_t)((maxL <= 2 ? 2 : (maxL <= 4 ? 4 : 8)) + 4 + 8) * sizeof(DevSlot);
    if (maxL <= 2 && maxM <= 1 && maxP <= 5)
        LWHIP_LAUNCH((rates2d_kernel<2, 1, 5>), grid, dim3(128), lds, stream, a);
    else if (maxL <= 2 && maxM <= 2 && maxP <= 5)
        LWHIP_LAUNCH((rates2d_kernel<2, 2, 5>), grid, dim3(128), lds, stream, a);
    else if (maxL <= 2)
        LWHIP_LAUNCH((rates2d_kernel<2, 4, 8>), grid, dim3(128), lds, stream, a);
    else if (maxL <= 4 && maxM <= 2 && maxP <= 5)
        LWHIP_LAUNCH((rates2d_kernel<4, 2, 5>), grid, dim3(128), lds, stream, a);
    else if (maxL <= 4)
        LWHIP_LAUNCH((rates2d_kernel<4, 4, 8>), grid, dim3(128), lds, stream, a);
    else
        LWHIP_LAUNCH((rates2d_kernel<8, 4, 8>), grid, dim3(128), lds, stream, a);
    return hipGetLastError();
}
}

// lwhip_march.hip -- the wavefront-autonomous sweep of the formal_sol_gamma_matrices iteration.
//
// Mapping: a workgroup owns a chunk of wavelengths; each of its wavefronts
// takes whole wavelengths from the chunk and needs no workgroup barrier while it works.  Inside
// a wavelength a pass covers R rays at once: lane = (ray r, depth segment s), and every lane
// marches over D consecutive depth points of its ray with all stencil values in registers
// (D = 7, S = 12 segments, R = 5 rays for the 82-point FAL-C atmosphere: 60 of 64 lanes busy).
//   1. gather chi, S at the lane's D points (continuum rows come from cont_kernel, once per
//      (wavelength, depth), the lines are added per ray);
//   2. seven lane shuffles hand the segment-boundary values to the neighbours, after which the
//      derivative / optical-depth / source-derivative stencils of piecewise_bezier3_1d are local;
//   3. the recurrence I_m = e_m I_(m-1) + b_m is composed over the lane's D points, scanned over
//      the S lanes of the ray (log2 S shuffle steps for D points each), and replayed;
//   4. Gamma / rate integrals go to the workgroup's LDS accumulators with ds_add_f64 (lanes of one
//      ray sit at distinct depths), J and the Psi* moments to a wavefront-private LDS row.
// Compared with one thread per (ray, depth) this evaluates every stencil once instead of three
// times, replaces ~20 workgroup barriers per wavelength by none, and amortises the scan over D
// points.  fp64 throughout; no MFMA (streaming stencil + scan, not a contraction).
//
// Reference restated: intensity_core_opt (Source/SimdFullIterationTemplates.hpp:238-487) and the 1D
// formal solvers (Source/FormalScalar.cpp:136-666); see the comments at each step.
#include "lwhip_device.h"
#include "../../include/lwhip.h"

#include <algorithm>

#ifndef LWHIP_MARCH_WPE
#define LWHIP_MARCH_WPE 2
#endif
#ifndef LWHIP_MARCH_B
#define LWHIP_MARCH_B 2 // points whose rate-phase loads are in flight together
#endif

namespace lwhip
{
#define ARGS (*launder_args(ap))

// ---- continuum rows: everything about the continua that does not depend on the ray ------------------
// Atom::setup_wavelength (LwAtom.hpp:107-115) + Transition::uv (LwTransition.hpp:132-142) +
// chi_eta_aux_accum (SimdFullIterationTemplates.hpp:59-109) for the continua of one wavelength:
// Vji_c, the totals chiC = bgChi + sum chi_c, etaC = bgEta + sum eta_c, and the per-atom / per-level
// sums the preconditioned rates need, written once per (wavelength, depth) for all rays to read.
__global__ void __launch_bounds__(128) cont_kernel(const SweepArgs* __restrict__ ap, int contOff, double* rowsBase)
{
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

hipError_t launch_cont(const SweepArgs* devArgs, const SweepArgs& a, hipStream_t stream)
{
    if (a.nContLa <= 0)
        return hipSuccess;
    const int T = 128;
    const size_t lds = (size_t)2 * a.maxC * T * sizeof(double);
    static size_t configured = 0;
    if (lds > configured && lds > 48 * 1024)
    {
        hipError_t err = hipFuncSetAttribute((const void*)cont_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess)
            return err;
        configured = lds;
    }
    hipLaunchKernelGGL(cont_kernel, dim3(a.nContLa), dim3(T), lds, stream, devArgs, 0, (double*)nullptr);
    return hipGetLastError();
}

hipError_t launch_cont_range(const SweepArgs* devArgs, const SweepArgs& a, int contOff, int nCont, double* rowsBase,
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
    const int ky = std::max(1, std::min(64, (a.Ns + T - 1) / T));
    hipLaunchKernelGGL(cont_kernel, dim3(nCont, ky), dim3(T), lds, stream, devArgs, contOff, rowsBase);
    return hipGetLastError();
}

// ---- the march -------------------------------------------------------------------------------------------
DEVINL void wave_lds_fence()
{
    // orders this wavefront's own LDS traffic (ds ops of one wavefront complete in issue order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

size_t march_lds_bytes(const SweepArgs& a, int wavesPerGroup)
{
    size_t d = (size_t)3 * a.Ns + (size_t)wavesPerGroup * 4 * a.Ns + 8 + (size_t)(a.NlevTot + a.Nline) * a.Ns;
    if (!a.accGlobal)
        d += (size_t)a.maxCT * 4 * a.Ns;
    return d * sizeof(double);
}

// loads with a uniform (SGPR) base and a 32-bit per-lane byte offset: global_load ... v_off, s[base]
DEVINL double ldg(const GLOBAL_AS double* base, unsigned byteOff)
{
    return *(const GLOBAL_AS double*)((const GLOBAL_AS char*)base + byteOff);
}
DEVINL void stg(GLOBAL_AS double* base, unsigned byteOff, double v)
{
    *(GLOBAL_AS double*)((GLOBAL_AS char*)base + byteOff) = v;
}

// scalar copy of one line's slot record for the wavelength in flight
struct LineS
{
    double p0, p1, p2, p3;
    const GLOBAL_AS double* phi; // this wavelength's [Nrays, 2, Ns] block
    const GLOBAL_AS double* rho; // this wavelength's [Ns] row (PRD) or null
    int gi, gj, atom, op, row, ms, cs;
    int rEtaA, rChiI, rChiJ, rUI, rUJ;
};

template <int SOLVER, int D, int MAXL, bool RATES>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(LWHIP_MARCH_WPE)))
march_kernel(const SweepArgs* __restrict__ ap, const SweepDyn dyn)
{
    extern __shared__ double lds[];
    constexpr int GL = (MAXL < 2) ? MAXL : 2; // lines whose phi is kept in registers for the rates
    const int Ns = ARGS.Ns;
    const int Nrays = ARGS.Nrays;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int nWaves = blockDim.x >> 6;
    const int S = dyn.S;
    const int R = dyn.R;
    const int chunk = blockIdx.x;
    const int la0 = CTAB(int32_t, ARGS.chunkLa)[chunk];
    const int la1 = CTAB(int32_t, ARGS.chunkLa)[chunk + 1];
    const int maxCT = ARGS.maxCT;
    const bool accGlobal = ARGS.accGlobal != 0;

    double* sDh = lds;            // |h[k] - h[k+1]|
    double* sRdh = sDh + Ns;      // 1 / dh
    double* sRsum = sRdh + Ns;    // 1 / (dh[k-1] + dh[k])
    double* sMomAll = sRsum + Ns; // [nWaves][4][Ns]: J, M1, Meta0, Meta1 of the wavelength in flight
    double* sN = sMomAll + (size_t)nWaves * 4 * Ns + 8; // populations [NlevTot][Ns]: constant during the sweep
    double* sWphi = sN + (size_t)ARGS.NlevTot * Ns;      // [Nline][Ns]
    double* sAccL = sWphi + (size_t)ARGS.Nline * Ns;     // [maxCT][4][Ns] unless accGlobal
    double* sMom = sMomAll + (size_t)wave * 4 * Ns;
    // Gamma / rate accumulators: ds_add_f64 on the LDS copy, or global atomics on the chunk's slab
    GLOBAL_AS double* gAcc = GD(ARGS.partial) + (size_t)chunk * maxCT * 4 * Ns;
    auto accAdd = [&](int off, double v) {
        if (accGlobal)
            unsafeAtomicAdd((double*)(gAcc + off), v);
        else
            lds_add(sAccL + off, v);
    };

    for (int k = t; k < Ns; k += blockDim.x)
    {
        const GLOBAL_AS double* gh = GCD(ARGS.height);
        double dh = 0.0, rdh = 0.0, rsum = 0.0;
        if (k < Ns - 1)
        {
            dh = fabs(gh[k] - gh[k + 1]);
            rdh = 1.0 / dh;
        }
        if (k >= 1 && k < Ns - 1)
            rsum = 1.0 / (fabs(gh[k - 1] - gh[k]) + dh);
        sDh[k] = dh;
        sRdh[k] = rdh;
        sRsum[k] = rsum;
    }
    if (RATES)
    {
        // accumulators of the chunk (the slab itself in accGlobal mode: only ever touched atomically
        // after this zeroing, and the workgroup barrier orders the two)
        for (int idx = t; idx < maxCT * 4 * Ns; idx += blockDim.x)
        {
            if (accGlobal)
                gAcc[idx] = 0.0;
            else
                sAccL[idx] = 0.0;
        }
    }
    for (int idx = t; idx < ARGS.NlevTot * Ns; idx += blockDim.x)
        sN[idx] = GCD(ARGS.n)[idx];
    for (int idx = t; idx < ARGS.Nline * Ns; idx += blockDim.x)
        sWphi[idx] = GCD(ARGS.wphi)[idx];
    __syncthreads();

    // lane geometry: ray slot r, segment s, marching indices m0 .. m0 + D - 1
    const int r = lane / S;
    const int s = lane - r * S;
    const int m0 = s * D;
    const bool laneOk = (r < R) && (m0 < Ns);
    // total quadrature weight of the active rays, in ray order (pure continua)
    double Wsum = 0.0;
    for (int q = 0; q < dyn.nRaysActive; ++q)
        Wsum += 0.5 * CTAB(double, ARGS.wmu)[CTAB(int32_t, dyn.rayList)[q] >> 1];

    // pointers are read from the argument block once: a wavefront works alone here, so every
    // scalar reload would sit on its critical path
    const GLOBAL_AS double* gRows = GCD(ARGS.rows);
    const GLOBAL_AS double* gPhi = GCD(ARGS.phi);
    const GLOBAL_AS double* gRho = GCD(ARGS.rho);
    const GLOBAL_AS double* gBgChi = GCD(ARGS.bgChi);
    const GLOBAL_AS double* gBgEta = GCD(ARGS.bgEta);
    const GLOBAL_AS double* gBgSca = GCD(ARGS.bgSca);
    GLOBAL_AS double* gJ = GD(ARGS.J);
    const CONST_AS DevLaHeader* cHdr = CTAB(DevLaHeader, ARGS.laHdr);
    const CONST_AS DevSlot* cSlots = CTAB(DevSlot, ARGS.slots);
    const bool storeDepth = RATES && ARGS.storeDepth && !dyn.prdOnly;
    const int lowerBc = ARGS.lowerBc, upperBc = ARGS.upperBc;
#ifdef LWHIP_STAMPS
    long long stampAcc[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    long long stampLast = clock64();
#define STAMP(i)                                                                                       \
    do                                                                                                 \
    {                                                                                                  \
        const long long now__ = clock64();                                                             \
        stampAcc[i] += now__ - stampLast;                                                              \
        stampLast = now__;                                                                             \
    } while (0)
#else
#define STAMP(i)
#endif

    // wavelengths of the chunk are dealt to the wavefronts round-robin (neighbours cost the same)
    const CONST_AS int32_t* cLaList = CTAB(int32_t, ARGS.laList); // PRD rates pass: explicit wavelength list
    for (int li = la0 + wave; li < la1; li += nWaves)
    {
        const int la = cLaList ? cLaList[li] : li;
        const DevLaHeader h = ld_c(cHdr + la);
        if (dyn.prdOnly && !h.hasPrd) // formal_sol_prd_update_rates visits prdIdxs only (PrdTemplates.hpp:226-239)
        {
            if (RATES && lane == 0)
                GD(ARGS.dJ)[la] = 0.0;
            continue;
        }
        const CONST_AS DevSlot* slots = cSlots + h.off;
        const int nLine = h.nLine;
        const int nMixed = h.nMixed;
        const bool hasRows = h.rowBase >= 0;
        const GLOBAL_AS double* pRows = gRows + (hasRows ? (size_t)h.rowBase * Ns : 0);
        const double wav = CTAB(double, ARGS.wavelength)[la];
        const size_t laNs = (size_t)la * Ns;
        // continuum totals: row 0 / 1 of the block, or the background itself where no continuum is active
        const GLOBAL_AS double* pChiC = hasRows ? pRows : gBgChi + laNs;
        const GLOBAL_AS double* pEtaC = hasRows ? pRows + Ns : gBgEta + laNs;
        const GLOBAL_AS double* pSca = gBgSca + laNs;
        GLOBAL_AS double* pJ = gJ + laNs;
        auto rowAt = [&](int ri, unsigned kb) -> double { return (ri >= 0) ? ldg(pRows + (size_t)ri * Ns, kb) : 0.0; };

        LineS L[MAXL];
#pragma unroll
        for (int l = 0; l < MAXL; ++l)
        {
            L[l].p0 = L[l].p1 = L[l].p2 = L[l].p3 = 0.0;
            L[l].phi = gPhi;
            L[l].rho = nullptr;
            L[l].gi = L[l].gj = L[l].atom = L[l].row = L[l].cs = 0;
            L[l].op = 0;
            L[l].ms = -1;
            L[l].rEtaA = L[l].rChiI = L[l].rChiJ = L[l].rUI = L[l].rUJ = -1;
            if (l < nLine)
            {
                const DevSlot sl = ld_c(slots + l);
                L[l].p0 = sl.p[0];
                L[l].p1 = sl.p[1];
                L[l].p2 = sl.p[2];
                L[l].p3 = sl.p[3];
                L[l].phi = gPhi + sl.phiOff;
                L[l].rho = (sl.flags & 2) ? gRho + sl.rhoOff : nullptr;
                L[l].gi = sl.gi;
                L[l].gj = sl.gj;
                L[l].atom = sl.atom;
                L[l].op = sl.flags & 1;
                L[l].row = sl.row;
                L[l].ms = sl.ms;
                L[l].cs = sl.cs;
                if (hasRows && (sl.flags & 1))
                {
                    L[l].rEtaA = sl.rEtaA;
                    L[l].rChiI = sl.rChiI;
                    L[l].rChiJ = sl.rChiJ;
                    L[l].rUI = sl.rUI;
                    L[l].rUJ = sl.rUJ;
                }
            }
        }

        if (RATES)
        {
            for (int idx = lane; idx < 4 * Ns; idx += 64)
                sMom[idx] = 0.0;
            wave_lds_fence();
        }

        STAMP(0);
        for (int p0 = 0; p0 < dyn.nRaysActive; p0 += R)
        {
            const int rayIdx = p0 + r;
            const bool active = laneOk && rayIdx < dyn.nRaysActive;
            int mu = 0, toObs = 0;
            if (active)
            {
                const int ray = CTAB(int32_t, dyn.rayList)[rayIdx];
                mu = ray >> 1;
                toObs = ray & 1;
            }
            const double muz = active ? CTAB(double, ARGS.muz)[mu] : 1.0;
            const double halfwmu = active ? 0.5 * CTAB(double, ARGS.wmu)[mu] : 0.0;
            const double zmu = 1.0 / muz;
            const int nValid = active ? min(D, Ns - m0) : 0;
            auto kOf = [&](int m) { return toObs ? Ns - 1 - m : m; };
            // interval (m-1 -> m): the geometry row it lives in
            auto kIv = [&](int m) { return toObs ? Ns - 1 - m : m - 1; };
            const unsigned rayB = (unsigned)((mu * 2 + toObs) * Ns) * 8u;
            // depth index of each point, clamped into the ray: points past the end repeat the last one
            // (finite values everywhere, e = 1 / b = 0 and no accumulation keep them inert)
            int kk_[D];
            unsigned kb_[D];
#pragma unroll
            for (int j = 0; j < D; ++j)
            {
                kk_[j] = kOf(min(m0 + j, Ns - 1));
                kb_[j] = (unsigned)kk_[j] * 8u;
            }

            // ---- 1. gather: chi, eta, S at the lane's points (Transition::uv + chi_eta_aux_accum for
            // the lines, continuum totals from cont_kernel); compute_source_fn :169-179.
            // All global loads of the pass are issued first, so the wavefront waits for one memory
            // round trip instead of one per point. -------------------------------------------------------
            double c0_[D], e0_[D], sca_[D], jd_[D];
#pragma unroll
            for (int j = 0; j < D; ++j)
            {
                c0_[j] = ldg(pChiC, kb_[j]);
                e0_[j] = ldg(pEtaC, kb_[j]);
                sca_[j] = ldg(pSca, kb_[j]);
                jd_[j] = ldg(pJ, kb_[j]);
            }
            double ph_[GL][D], rh_[GL][D];
#pragma unroll
            for (int l = 0; l < GL; ++l)
            {
#pragma unroll
                for (int j = 0; j < D; ++j)
                {
                    ph_[l][j] = 0.0;
                    rh_[l][j] = 1.0;
                }
                if (l < nLine)
                {
#pragma unroll
                    for (int j = 0; j < D; ++j)
                        ph_[l][j] = ldg(L[l].phi, rayB + kb_[j]);
                    if (L[l].rho)
                    {
#pragma unroll
                        for (int j = 0; j < D; ++j)
                            rh_[l][j] = ldg(L[l].rho, kb_[j]);
                    }
                }
            }
            double chi_[D], S_[D], rchi_[D], eta_[D];
#pragma unroll
            for (int j = 0; j < D; ++j)
            {
                double chi = c0_[j];
                double eta = e0_[j];
#pragma unroll
                for (int l = 0; l < GL; ++l)
                {
                    if (l < nLine)
                    {
                        const double vij = L[l].p0 * ph_[l][j];
                        const double vji = (L[l].p2 * rh_[l][j]) * vij;
                        const double nj = sN[L[l].gj * Ns + kk_[j]];
                        chi += sN[L[l].gi * Ns + kk_[j]] * vij - nj * vji;
                        eta += nj * (L[l].p3 * vji);
                    }
                }
                chi_[j] = chi;
                eta_[j] = eta;
            }
            if (MAXL > GL)
            {
                for (int l = GL; l < nLine; ++l) // more overlapping lines than the unrolled pair
                {
                    const CONST_AS DevSlot& sl = slots[l];
#pragma unroll
                    for (int j = 0; j < D; ++j)
                    {
                        const int k = kk_[j];
                        const double vij = sl.p[0] * ldg(gPhi + sl.phiOff, rayB + kb_[j]);
                        double g = sl.p[2];
                        if (sl.flags & 2)
                            g *= ldg(gRho + sl.rhoOff, kb_[j]);
                        const double vji = g * vij;
                        const double nj = sN[sl.gj * Ns + k];
                        chi_[j] += sN[sl.gi * Ns + k] * vij - nj * vji;
                        eta_[j] += nj * (sl.p[3] * vji);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < D; ++j)
            {
                const double rchi = d_rcp(chi_[j]);
                if (storeDepth && j < nValid)
                {
                    const size_t o = ((size_t)(la * Nrays + mu) * 2 + toObs) * Ns + kk_[j];
                    GD(ARGS.depthChi)[o] = chi_[j];
                    GD(ARGS.depthEta)[o] = eta_[j];
                }
                S_[j] = (eta_[j] + sca_[j] * jd_[j]) * rchi;
                rchi_[j] = rchi;
            }
            STAMP(1);

            // ---- 2. segment boundaries: the neighbours' edge values ---------------------------------------
            // (lane - 1 always holds a full segment; lane + 1's first point exists iff m0 + D < Ns)
            const double chiPrev = __shfl_up(chi_[D - 1], 1);
            const double SPrev = __shfl_up(S_[D - 1], 1);
            const double chiNext = __shfl_down(chi_[0], 1);
            const double SNext = __shfl_down(S_[0], 1);
            auto chiAt = [&](int j) { return j < 0 ? chiPrev : (j >= D ? chiNext : chi_[j < 0 ? 0 : (j >= D ? D - 1 : j)]); };
            auto SAt = [&](int j) { return j < 0 ? SPrev : (j >= D ? SNext : S_[j < 0 ? 0 : (j >= D ? D - 1 : j)]); };

            double e_[D], b_[D], Psi_[D];

            // boundary intensity of the ray (wrappers FormalScalar.cpp:471-666); lane s == 0 only
            double Istart = 0.0;
            if (active && s == 0)
            {
                const double dtau_b = 0.5 * zmu * (chi_[0] + chi_[1]) * sDh[kIv(1)];
                const GLOBAL_AS double* gT = GCD(ARGS.temperature);
                if (toObs)
                {
                    if (lowerBc == LWHIP_BC_THERMALISED)
                    {
                        const double B0 = d_planck(gT[Ns - 2], wav);
                        const double B1 = d_planck(gT[Ns - 1], wav);
                        Istart = B1 - (B0 - B1) / dtau_b;
                    }
                    else if (lowerBc == LWHIP_BC_CALLABLE)
                    {
                        const int mi = ARGS.lowerIdx[mu * 2 + 1];
                        if (mi >= 0)
                            Istart = ARGS.lowerBcData[(size_t)la * ARGS.lowerNmu + mi];
                    }
                }
                else
                {
                    if (upperBc == LWHIP_BC_THERMALISED)
                    {
                        const double B0 = d_planck(gT[0], wav);
                        const double B1 = d_planck(gT[1], wav);
                        Istart = B0 - (B1 - B0) / dtau_b;
                    }
                    else if (upperBc == LWHIP_BC_CALLABLE)
                    {
                        const int mi = ARGS.upperIdx[mu * 2 + 0];
                        if (mi >= 0)
                            Istart = ARGS.upperBcData[(size_t)la * ARGS.upperNmu + mi];
                    }
                }
            }

            // ---- 3. per-point coefficients e, b, Psi of I_m = e I_(m-1) + b ------------------------------
            if (SOLVER == LWHIP_FS_BEZIER3_1D)
            {
                // piecewise_bezier3_1d_impl FormalScalar.cpp:209-325
                // path length (and its reciprocal) of the interval ending at point j of this lane;
                // j = D: the one ending at the next lane's first point
                double ds_[D + 1], rds_[D + 1];
#pragma unroll
                for (int j = 0; j <= D; ++j)
                {
                    const int m = min(max(m0 + j, 1), Ns - 1);
                    const int kv = kIv(m);
                    ds_[j] = sDh[kv] * zmu;
                    rds_[j] = sRdh[kv] * muz;
                }
                double dchi_[D];
#pragma unroll
                for (int j = 0; j < D; ++j)
                {
                    const int m = m0 + j;
                    const double dC = d_cent_deriv_r(ds_[j], ds_[j + 1], rds_[j], rds_[j + 1], sRsum[kk_[j]] * muz,
                                                     chiAt(j - 1), chi_[j], chiAt(j + 1)); // :240,251
                    const double dU = (chiAt(j + 1) - chi_[j]) * rds_[j + 1];               // dx_uw :239
                    const double dD = (chi_[j] - chiAt(j - 1)) * rds_[j];                   // dx_dw :288
                    dchi_[j] = (m == 0) ? dU : ((m >= Ns - 1) ? dD : dC);
                }
                const double dchiPrev = __shfl_up(dchi_[D - 1], 1);
                // optical depth of the interval ending at point j  :242-246, 261-263, 289-292
                double dtau_[D + 1], rdt_[D + 1];
#pragma unroll
                for (int j = 0; j < D; ++j)
                {
                    const double ds3 = ds_[j] * (1.0 / 3.0);
                    const double cPrev = chiAt(j - 1);
                    const double cuw = cPrev + ds3 * (j == 0 ? dchiPrev : dchi_[j == 0 ? 0 : j - 1]);
                    const double c0 = chi_[j] - ds3 * dchi_[j];
                    const double dt = ds_[j] * (cPrev + chi_[j] + cuw + c0) * 0.25;
                    dtau_[j] = (m0 + j >= 1 && j < nValid) ? dt : 1.0;
                    rdt_[j] = d_rcp(dtau_[j]);
                }
                dtau_[D] = __shfl_down(dtau_[0], 1);
                rdt_[D] = __shfl_down(rdt_[0], 1);
                // derivative of S with respect to tau  :247, 254-255, 268
                double dS_[D];
#pragma unroll
                for (int j = 0; j < D; ++j)
                {
                    const int m = m0 + j;
                    const double dC = d_cent_deriv_r(dtau_[j], dtau_[j + 1], rdt_[j], rdt_[j + 1],
                                                     d_rcp(dtau_[j] + dtau_[j + 1]), SAt(j - 1), S_[j], SAt(j + 1));
                    const double dU = (SAt(j + 1) - S_[j]) * rdt_[j + 1];
                    dS_[j] = (m == 0) ? dU : ((m >= Ns - 1) ? 0.0 : dC);
                }
                const double dSPrev = __shfl_up(dS_[D - 1], 1);
                // coefficients  :265-275, 294-304, 314-321
#pragma unroll
                for (int j = 0; j < D; ++j)
                {
                    const int m = m0 + j;
                    const double S = S_[j];
                    const double Suw = SAt(j - 1);
                    const double x = dtau_[j];
                    double al, be, ga, de, edt;
                    const double dt2 = x * x;
                    const double dt3c = dt2 * x;
                    if (x < 5e-2) // Bezier3_coeffs, Bezier.hpp:81-127
                    {
                        edt = 1.0 - x + 0.5 * dt2 - dt3c * (1.0 / 6.0);
                        al = 0.25 * x - 0.2 * dt2 + dt3c * (1.0 / 12.0);
                        be = 0.25 * x - 0.05 * dt2 + dt3c * (1.0 / 120.0);
                        ga = 0.25 * x - 0.15 * dt2 + 0.05 * dt3c;
                        de = 0.25 * x - 0.1 * dt2 + 0.025 * dt3c;
                    }
                    else
                    {
                        edt = (x > 30.0) ? 0.0 : d_exp_neg(-x); // beyond 30 the asymptotic branch is this one with edt = 0
                        const double r1 = rdt_[j];
                        const double r3 = r1 * r1 * r1;
                        al = (6.0 - edt * (6.0 + 6.0 * x + 3 * dt2 + dt3c)) * r3;
                        be = (6.0 * edt - 6.0 + 6.0 * x - 3.0 * dt2 + dt3c) * r3;
                        ga = 3.0 * (2.0 * x - 6.0 + edt * (6.0 + 4.0 * x + dt2)) * r3;
                        de = 3.0 * (6.0 - 4.0 * x + dt2 - 2.0 * edt * (3.0 + x)) * r3;
                    }
                    const double dt3 = x * (1.0 / 3.0);
                    const double cuw = Suw + dt3 * (j == 0 ? dSPrev : dS_[j == 0 ? 0 : j - 1]);
                    const double c0 = S - dt3 * dS_[j];
                    e_[j] = edt;
                    b_[j] = al * Suw + be * S + ga * cuw + de * c0;
                    Psi_[j] = be + de;
                    if (m == Ns - 1 && j < nValid)
                    {
                        // piecewise linear on the last point :307-317
                        const double xl_ = 0.5 * zmu * (chi_[j] + chiAt(j - 1)) * sDh[kIv(m)];
                        const double rl_ = d_rcp(xl_);
                        const double dSuw = (S - Suw) * rl_;
                        double w0, w1;
                        if (xl_ < 5.0E-4) // w2, LwInternal.hpp:90-110
                        {
                            w0 = xl_ * (1.0 - 0.5 * xl_);
                            w1 = (xl_ * xl_) * (0.5 - xl_ * (1.0 / 3.0));
                        }
                        else if (xl_ > 50.0)
                        {
                            w0 = 1.0;
                            w1 = 1.0;
                        }
                        else
                        {
                            const double ex = d_exp_neg(-xl_);
                            w0 = 1.0 - ex;
                            w1 = w0 - xl_ * ex;
                        }
                        e_[j] = 1.0 - w0;
                        b_[j] = w0 * S - w1 * dSuw;
                        Psi_[j] = w0 - w1 * rl_;
                    }
                    if (m == 0)
                    {
                        e_[j] = 0.0;
                        b_[j] = Istart;
                        Psi_[j] = 0.0;
                    }
                    if (j >= nValid)
                    {
                        e_[j] = 1.0;
                        b_[j] = 0.0;
                    }
                }
            }
            else if (SOLVER == LWHIP_FS_LINEAR_1D)
            {
                // piecewise_linear_1d_impl FormalScalar.cpp:136-207
#pragma unroll
                for (int j = 0; j < D; ++j)
                {
                    const int m = m0 + j;
                    e_[j] = 1.0;
                    b_[j] = 0.0;
                    Psi_[j] = 0.0;
                    if (j < nValid)
                    {
                        if (m == 0)
                        {
                            e_[j] = 0.0;
                            b_[j] = Istart;
                        }
                        else
                        {
                            const double dtau = (0.5 * zmu) * (chiAt(j - 1) + chi_[j]) * sDh[kIv(m)];
                            const double rcp = d_rcp(dtau);
                            const double dSuw = (SAt(j - 1) - S_[j]) * rcp;
                            double w0, w1;
                            d_w2(dtau, w0, w1);
                            e_[j] = 1.0 - w0;
                            b_[j] = w0 * S_[j] + w1 * dSuw;
                            Psi_[j] = w0 - w1 * rcp;
                        }
                    }
                }
            }
            else
            {
                // piecewise_besser_1d_impl FormalScalar.cpp:395-467
#pragma unroll
                for (int j = 0; j < D; ++j)
                {
                    const int m = m0 + j;
                    e_[j] = 1.0;
                    b_[j] = 0.0;
                    Psi_[j] = 0.0;
                    if (j < nValid)
                    {
                        if (m == 0)
                        {
                            e_[j] = 0.0;
                            b_[j] = Istart;
                        }
                        else if (m == Ns - 1)
                        {
                            const double dtau = 0.5 * zmu * (chi_[j] + chiAt(j - 1)) * sDh[kIv(m)];
                            const double rcp = d_rcp(dtau);
                            const double dSuw = (S_[j] - SAt(j - 1)) * rcp;
                            double w0, w1;
                            d_w2(dtau, w0, w1);
                            e_[j] = 1.0 - w0;
                            b_[j] = w0 * S_[j] - w1 * dSuw;
                            Psi_[j] = w0 - w1 * rcp;
                        }
                        else
                        {
                            const double ds_uw = sDh[kIv(m)] * zmu;
                            const double ds_dw = sDh[kIv(m + 1)] * zmu;
                            const double chi_uw = chiAt(j - 1), chi_dw = chiAt(j + 1);
                            const double chiC = d_besser_control_point(ds_uw, ds_dw, chi_uw, chi_[j], chi_dw);
                            const double dtauUw = (1.0 / 3.0) * (chi_uw + chiC + chi_[j]) * ds_uw;
                            const double dtauDw = 0.5 * (chi_[j] + chi_dw) * ds_dw;
                            const double Suw = SAt(j - 1), Sdw = SAt(j + 1);
                            const double SC = d_besser_control_point(dtauUw, dtauDw, Suw, S_[j], Sdw);
                            double cM, cO, cC, edt;
                            d_besser_coeffs(dtauUw, cM, cO, cC, edt);
                            e_[j] = edt;
                            b_[j] = cM * Suw + cO * S_[j] + cC * SC;
                            Psi_[j] = cO + cC;
                        }
                    }
                }
            }
            STAMP(2);

            // ---- 4. the recurrence: compose the lane's points, scan the S lanes of the ray, replay ------
            double E = 1.0, B = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j)
            {
                B = e_[j] * B + b_[j];
                E = e_[j] * E;
            }
            for (int off = 1; off < S; off <<= 1)
            {
                const double Ep = __shfl_up(E, off);
                const double Bp = __shfl_up(B, off);
                if (s >= off)
                {
                    B = E * Bp + B;
                    E = E * Ep;
                }
            }
            double I = __shfl_up(B, 1);
            if (s == 0)
                I = 0.0;
            STAMP(3);

            // ---- 5. intensities, J, rates :181-234 -----------------------------------------------------------
            double I_[D];
#pragma unroll
            for (int j = 0; j < D; ++j)
            {
                I = e_[j] * I + b_[j];
                I_[j] = I;
                if (j < nValid)
                {
                    if (toObs && kk_[j] == 0)
                        GD(ARGS.I)[(size_t)la * Nrays + mu] = I; // spect.I(la, mu, 0) :349
                    if (storeDepth)
                        GD(ARGS.depthI)[((size_t)(la * Nrays + mu) * 2 + toObs) * Ns + kk_[j]] = I;
                }
            }
            if (RATES)
            {
                const bool anyPerRay = (nLine + nMixed) > 0;
                if (nLine == 1 && nMixed == 0)
                {
                    // the common case, one active line and no mixed continuum: its level sums are the
                    // continuum rows plus its own chi / Uji, no searching.  phi is still in registers.
                    const LineS& l0 = L[0];
                    const double wl = (l0.p1 * PI4_HC) * halfwmu;
#pragma unroll
                    for (int j = 0; j < D; ++j)
                    {
                        if (j < nValid)
                        {
                            const int k = kk_[j];
                            const double Ij = I_[j];
                            const double PsiS = dyn.lambdaIterate ? 0.0 : Psi_[j] * rchi_[j]; // PsiStar :322-323, :418-419
                            const double wPsi = halfwmu * PsiS;
                            lds_add(sMom + k, halfwmu * Ij); // accumulate_J :181-190
                            if (h.nPure > 0)
                                lds_add(sMom + Ns + k, wPsi);
                            const double vij = l0.p0 * ph_[0][j];
                            const double vji = (l0.p2 * rh_[0][j]) * vij;
                            const double uji = l0.p3 * vji;
                            const double nj = sN[l0.gj * Ns + k];
                            const double x = sN[l0.gi * Ns + k] * vij - nj * vji;
                            const double el = nj * uji;
                            const double wlamu = wl * sWphi[l0.row * Ns + k];
                            const int acc = l0.cs * 4 * Ns + k;
                            if (l0.op)
                            {
                                if (h.nPure > 0 && l0.ms >= 0)
                                    lds_add(sMom + (2 + l0.ms) * Ns + k, wPsi * el);
                                // compute_full_Ieff :192-204, compute_full_operator_rates :206-234
                                const double Ieff = Ij - PsiS * (rowAt(l0.rEtaA, kb_[j]) + el);
                                const double chi_i = rowAt(l0.rChiI, kb_[j]) + x;
                                const double chi_j = rowAt(l0.rChiJ, kb_[j]) - x;
                                const double U_j = rowAt(l0.rUJ, kb_[j]) + uji;
                                const double U_i = rowAt(l0.rUI, kb_[j]);
                                accAdd(acc, ((uji + vji * Ieff) - (PsiS * chi_i * U_j)) * wlamu);
                                accAdd(acc + Ns, ((vij * Ieff) - (PsiS * chi_j * U_i)) * wlamu);
                            }
                            accAdd(acc + 2 * Ns, Ij * vij * wlamu);
                            accAdd(acc + 3 * Ns, (uji + Ij * vji) * wlamu);
                        }
                    }
                }
                else
                {
#pragma unroll
                    for (int j = 0; j < D; ++j)
                    {
                        if (j < nValid)
                        {
                            const int k = kk_[j];
                            const unsigned kb = kb_[j];
                            const double Ij = I_[j];
                            const double PsiS = dyn.lambdaIterate ? 0.0 : Psi_[j] * rchi_[j]; // PsiStar :322-323, :418-419
                            const double wPsi = halfwmu * PsiS;
                            lds_add(sMom + k, halfwmu * Ij); // accumulate_J :181-190
                            if (h.nPure > 0)
                                lds_add(sMom + Ns + k, wPsi);
                            if (anyPerRay)
                            {
                                // this ray's line quantities at the point (second evaluation of uv, as the
                                // reference does in its rates loop :425-437)
                                double Vij[MAXL], Vji[MAXL], xl[MAXL], el[MAXL], luji[MAXL];
#pragma unroll
                                for (int l = 0; l < MAXL; ++l)
                                {
                                    Vij[l] = Vji[l] = xl[l] = el[l] = luji[l] = 0.0;
                                    if (l < nLine)
                                    {
                                        const double phv = (l < GL) ? ph_[l < GL ? l : 0][j] : ldg(L[l].phi, rayB + kb);
                                        const double rhv = (l < GL) ? rh_[l < GL ? l : 0][j] : (L[l].rho ? ldg(L[l].rho, kb) : 1.0);
                                        const double vij = L[l].p0 * phv;
                                        const double vji = (L[l].p2 * rhv) * vij;
                                        const double uji = L[l].p3 * vji;
                                        const double njv = sN[L[l].gj * Ns + k];
                                        Vij[l] = vij;
                                        Vji[l] = vji;
                                        luji[l] = uji;
                                        xl[l] = sN[L[l].gi * Ns + k] * vij - njv * vji;
                                        el[l] = njv * uji;
                                        if (L[l].op && h.nPure > 0 && L[l].ms >= 0)
                                            lds_add(sMom + (2 + L[l].ms) * Ns + k, wPsi * el[l]);
                                    }
                                }
                                // level sums: continuum rows + this ray's lines (chi_eta_aux_accum bookkeeping)
                                auto levChi = [&](int g, int ri) {
                                    double v = rowAt(ri, kb);
#pragma unroll
                                    for (int l = 0; l < MAXL; ++l)
                                    {
                                        if (L[l].op && L[l].gi == g)
                                            v += xl[l];
                                        if (L[l].op && L[l].gj == g)
                                            v -= xl[l];
                                    }
                                    return v;
                                };
                                auto levU = [&](int g, int ri) {
                                    double v = rowAt(ri, kb);
#pragma unroll
                                    for (int l = 0; l < MAXL; ++l)
                                        if (L[l].op && L[l].gj == g)
                                            v += luji[l];
                                    return v;
                                };
                                auto atomEta = [&](int at, int ri) {
                                    double v = rowAt(ri, kb);
#pragma unroll
                                    for (int l = 0; l < MAXL; ++l)
                                        if (L[l].op && L[l].atom == at)
                                            v += el[l];
                                    return v;
                                };
#pragma unroll
                                for (int l = 0; l < MAXL; ++l)
                                {
                                    if (l < nLine)
                                    {
                                        const double wlamu = ((L[l].p1 * sWphi[L[l].row * Ns + k]) * PI4_HC) * halfwmu;
                                        const int acc = L[l].cs * 4 * Ns + k;
                                        if (L[l].op)
                                        {
                                            // compute_full_Ieff :192-204, compute_full_operator_rates :206-234
                                            const double Ieff = Ij - PsiS * atomEta(L[l].atom, L[l].rEtaA);
                                            const double g0v = ((luji[l] + Vji[l] * Ieff) - (PsiS * levChi(L[l].gi, L[l].rChiI) * levU(L[l].gj, L[l].rUJ))) * wlamu;
                                            const double g1v = ((Vij[l] * Ieff) - (PsiS * levChi(L[l].gj, L[l].rChiJ) * levU(L[l].gi, L[l].rUI))) * wlamu;
                                            accAdd(acc, g0v);
                                            accAdd(acc + Ns, g1v);
                                        }
                                        accAdd(acc + 2 * Ns, Ij * Vij[l] * wlamu);
                                        accAdd(acc + 3 * Ns, (luji[l] + Ij * Vji[l]) * wlamu);
                                    }
                                }
                                for (int c = 0; c < nMixed; ++c) // mixed continua (always of active atoms)
                                {
                                    const CONST_AS DevSlot& sl = slots[nLine + c];
                                    const double vij = sl.p[0];
                                    const double vji = rowAt(sl.rVji, kb);
                                    const double uji = sl.p[2] * vji;
                                    const double wlamu = sl.p[1] * halfwmu;
                                    const int acc = sl.cs * 4 * Ns + k;
                                    const double Ieff = Ij - PsiS * atomEta(sl.atom, sl.rEtaA);
                                    const double g0v = ((uji + vji * Ieff) - (PsiS * levChi(sl.gi, sl.rChiI) * levU(sl.gj, sl.rUJ))) * wlamu;
                                    const double g1v = ((vij * Ieff) - (PsiS * levChi(sl.gj, sl.rChiJ) * levU(sl.gi, sl.rUI))) * wlamu;
                                    accAdd(acc, g0v);
                                    accAdd(acc + Ns, g1v);
                                    accAdd(acc + 2 * Ns, Ij * vij * wlamu);
                                    accAdd(acc + 3 * Ns, (uji + Ij * vji) * wlamu);
                                }
                            }
                        }
                    }
                }
            }
            STAMP(4);
        } // passes over the rays

        if (RATES)
        {
            wave_lds_fence();
            // ---- J out, dJ = max_k |1 - Jdag/J|  :477-486; pure continua from the angle moments:
            // sum_r w_r [Uji + Vji (I_r - Psi_r eta_r) - Psi_r chi_i U_j] with chi, U, Uji, Vji ray-
            // independent and eta_r = etaC_atom + etaLine_r  ->  Uji W + Vji (J - M1 etaA - Meta) - M1 chi_i U_j
            double dj = 0.0;
            for (int k = lane; k < Ns; k += 64)
            {
                const unsigned kb = (unsigned)k * 8u;
                const double Jn = sMom[k];
                const double Jd = ldg(pJ, kb);
                stg(pJ, kb, Jn);
                dj = fmax(dj, fabs(1.0 - Jd / Jn));
                const double M1 = sMom[Ns + k];
                for (int cq = 0; cq < h.nPure; ++cq)
                {
                    const CONST_AS DevSlot& sl = slots[nLine + nMixed + cq];
                    const double vij = sl.p[0];
                    const double vji = rowAt(sl.rVji, kb);
                    const double uji = sl.p[2] * vji;
                    const double wla = sl.p[1];
                    const int acc = sl.cs * 4 * Ns + k;
                    if (sl.flags & 1)
                    {
                        const double Meta = (sl.ms >= 0) ? sMom[(2 + sl.ms) * Ns + k] : 0.0;
                        const double Jeff = Jn - M1 * rowAt(sl.rEtaA, kb) - Meta;
                        const double g0v = ((uji * Wsum + vji * Jeff) - (M1 * rowAt(sl.rChiI, kb) * rowAt(sl.rUJ, kb))) * wla;
                        const double g1v = ((vij * Jeff) - (M1 * rowAt(sl.rChiJ, kb) * rowAt(sl.rUI, kb))) * wla;
                        accAdd(acc, g0v);
                        accAdd(acc + Ns, g1v);
                    }
                    accAdd(acc + 2 * Ns, Jn * vij * wla);
                    accAdd(acc + 3 * Ns, (uji * Wsum + Jn * vji) * wla);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                dj = fmax(dj, __shfl_xor(dj, off));
            if (lane == 0)
                GD(ARGS.dJ)[la] = dj;
            wave_lds_fence(); // sMom is reused by the next wavelength
        }
        STAMP(5);
    } // wavelengths

#ifdef LWHIP_STAMPS
    if (t == 0 && ARGS.stamps)
        for (int i = 0; i < 12; ++i)
            ARGS.stamps[(size_t)chunk * 12 + i] = stampAcc[i];
#endif
    if (RATES && !accGlobal)
    {
        __syncthreads();
        GLOBAL_AS double* out = GD(ARGS.partial) + (size_t)chunk * maxCT * 4 * Ns;
        for (int idx = t; idx < maxCT * 4 * Ns; idx += blockDim.x)
            out[idx] = sAccL[idx];
    }
}

template <int SOLVER, int D, int MAXL, bool RATES>
static hipError_t launch_march_t(const SweepArgs* devArgs, const SweepArgs& a, const SweepDyn& dyn, int nChunks,
                                 int wavesPerGroup, hipStream_t stream)
{
    const size_t lds = march_lds_bytes(a, wavesPerGroup);
    static size_t configured = 0;
    if (lds > configured && lds > 48 * 1024)
    {
        hipError_t err = hipFuncSetAttribute((const void*)march_kernel<SOLVER, D, MAXL, RATES>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess)
            return err;
        configured = lds;
    }
    hipLaunchKernelGGL((march_kernel<SOLVER, D, MAXL, RATES>), dim3(nChunks), dim3(64 * wavesPerGroup), lds, stream,
                       devArgs, dyn);
    return hipGetLastError();
}

template <int SOLVER, int D>
static hipError_t launch_march_d(const SweepArgs* devArgs, const SweepArgs& a, const SweepDyn& dyn, int maxL, bool rates,
                                 int nChunks, int wavesPerGroup, hipStream_t stream)
{
    if (!rates)
        return launch_march_t<SOLVER, D, 8, false>(devArgs, a, dyn, nChunks, wavesPerGroup, stream);
    if (maxL <= 2)
        return launch_march_t<SOLVER, D, 2, true>(devArgs, a, dyn, nChunks, wavesPerGroup, stream);
    return launch_march_t<SOLVER, D, 8, true>(devArgs, a, dyn, nChunks, wavesPerGroup, stream);
}

template <int SOLVER>
static hipError_t launch_march_s(const SweepArgs* devArgs, const SweepArgs& a, const SweepDyn& dyn, int D, int maxL,
                                 bool rates, int nChunks, int wavesPerGroup, hipStream_t stream)
{
    switch (D)
    {
    case 2:
        return launch_march_d<SOLVER, 2>(devArgs, a, dyn, maxL, rates, nChunks, wavesPerGroup, stream);
    case 3:
        return launch_march_d<SOLVER, 3>(devArgs, a, dyn, maxL, rates, nChunks, wavesPerGroup, stream);
    case 4:
        return launch_march_d<SOLVER, 4>(devArgs, a, dyn, maxL, rates, nChunks, wavesPerGroup, stream);
    case 7:
        return launch_march_d<SOLVER, 7>(devArgs, a, dyn, maxL, rates, nChunks, wavesPerGroup, stream);
    case 8:
        return launch_march_d<SOLVER, 8>(devArgs, a, dyn, maxL, rates, nChunks, wavesPerGroup, stream);
    default:
        return hipErrorInvalidValue;
    }
}

hipError_t launch_march(const SweepArgs* devArgs, const SweepArgs& a, const SweepDyn& dyn, int solver, int D, int maxL,
                        bool rates, int nChunks, int wavesPerGroup, hipStream_t stream)
{
    switch (solver)
    {
    case LWHIP_FS_LINEAR_1D:
        return launch_march_s<LWHIP_FS_LINEAR_1D>(devArgs, a, dyn, D, maxL, rates, nChunks, wavesPerGroup, stream);
    case LWHIP_FS_BESSER_1D:
        return launch_march_s<LWHIP_FS_BESSER_1D>(devArgs, a, dyn, D, maxL, rates, nChunks, wavesPerGroup, stream);
    default:
        return launch_march_s<LWHIP_FS_BEZIER3_1D>(devArgs, a, dyn, D, maxL, rates, nChunks, wavesPerGroup, stream);
    }
}

#undef ARGS
}

// lwhip_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the formal_sol_gamma_matrices iteration.
//
// Mapping: one workgroup owns a contiguous chunk of wavelengths and walks
// it; inside a wavelength the rays x Nspace depth points of a pass of G rays are flattened over the
// threads of the workgroup (G = 5: 5 x 82 = 410 points -> 7 wavefronts, two workgroups per CU), so every
// load of phi[lt, mu, dir, k] is a fully coalesced block.  The ray columns live in LDS for the
// short-characteristic stencils; the only true recurrence, I_k = e_k I_{k-1} + b_k, is a chunked
// scan.  Continua are ray-independent, so their opacities and per-level sums are evaluated once
// per (lambda, k) and shared by all rays through LDS.  Gamma / rate integrals are summed over
// rays and wavelengths in LDS accumulators and leave the workgroup once per chunk as a partial
// slab; a second kernel adds the slabs in a fixed order.  fp64 throughout, no MFMA: the sweep is
// a streaming stencil + scan, not a contraction.
//
// Reference restated (file:line under Lightweaver's Source/): see each device function.
#include "lwhip_device.h"
#include "../../include/lwhip.h"

#include <algorithm>

namespace lwhip
{
// ---- LDS layout (doubles) ------------------------------------------------------------------------
// [G*Ns] x4 : sChi(->e ->w I)  sS(->w Psi)  sDS(->w Psi eta_0)  sB(->I ->w Psi eta_1)
// [Ns]      : dh rdh rsumh temp J M1 Meta0 Meta1 | Jdag sca bgChi bgEta (inputs, prefetched)
// rows      : chiC etaC etaA[Natom] levChi[NlevTot] levU[NlevTot]      (row-program targets)
// [maxC][Ns] Vji   [maxC][Ns] chi_c   [maxCT][4][Ns] accumulators   (populations n: global, L1)
// program   : rows + entries of the current row program
// BIG mode (deep atmospheres): the row-program rows, the continuum tables and the accumulators live
// in an L2-resident global scratch of the chunk instead of LDS.
__host__ __device__ inline size_t lds_doubles(int Ns, int G, int Natom, int NlevTot, int maxC, int maxCT, bool big)
{
    return (size_t)4 * G * Ns + (size_t)13 * Ns
           + (big ? 0 : (size_t)(2 + Natom + 2 * NlevTot) * Ns + (size_t)2 * maxC * Ns + (size_t)maxCT * 4 * Ns)
           + (LWHIP_MAX_PROG_ROWS * sizeof(DevProgRow) + LWHIP_MAX_PROG_ENTS * sizeof(DevProgEnt)) / 8 + 8
           + 64 + 64 + 32 + 2 * maxC + 8 + 16;
}

size_t sweep_lds_bytes(const SweepArgs& a, int G)
{
    return sizeof(double) * lds_doubles(a.Ns, G, a.Natom, a.NlevTot, a.maxC, a.maxCT, a.big != 0);
}

size_t sweep_big_scratch_doubles(const SweepArgs& a)
{
    return (size_t)(2 + a.Natom + 2 * a.NlevTot + 2 * a.maxC) * a.Ns;
}

template <bool BIG> DEVINL void wg_barrier()
{
    if (BIG)
        __syncthreads(); // also orders the global scratch traffic inside the workgroup
    else
        lds_barrier();
}

// ---- the sweep ---------------------------------------------------------------------------------------
// intensity_core_opt (SimdFullIterationTemplates.hpp:238-487) for every wavelength of the chunk,
// with UpdateRates = ComputeOperator = RATES, PrdRatesOnly = false.
#ifndef LWHIP_ABLATE
#define LWHIP_ABLATE 0
#endif

#ifdef LWHIP_STAMPS
#define STAMP(i)                                                                                       \
    do                                                                                                 \
    {                                                                                                  \
        if (t == 0)                                                                                    \
        {                                                                                              \
            const long long now__ = clock64();                                                         \
            stampAcc[i] += now__ - stampLast;                                                          \
            stampLast = now__;                                                                         \
        }                                                                                              \
    } while (0)
#else
#define STAMP(i)
#endif

// The body is a function so that the single-problem kernel and the column-batch kernel share it; `ap` stays a
// __restrict__ *parameter* in both (the no-alias guarantee is what lets the compiler keep argument-block fields
// across the kernel's stores: a pointer merely loaded from a table costs 40 % of the sweep, measured).
template <int SOLVER, int MAXL, bool RATES, bool BIG, bool ATOMIC = false>
DEVINL void sweep_body(const SweepArgs* __restrict__ ap, const SweepDyn& dyn)
{
    extern __shared__ double lds[];
#ifdef LWHIP_STAMPS
    long long stampAcc[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    long long stampLast = clock64();
    const long long wallStart = wall_clock64(); // device-wide constant-rate clock: when this workgroup started
#endif
    // the argument block is read through the constant address space: always scalar loads, also when `ap` itself
    // came from the batch's pointer table (a generic pointer the compiler cannot prove unclobbered would be read
    // with vector loads: 40 % of the sweep)
    const CONST_AS SweepArgs& a = *(const CONST_AS SweepArgs*)ap;
    const int Ns = a.Ns;
    const int G = dyn.G;
    const int GNs = G * Ns;
    const int T = blockDim.x;
    const int t = threadIdx.x;
    const int Nrays = a.Nrays;
    const int Natom = a.Natom;
    const int NlevTot = a.NlevTot;
    const int maxC = a.maxC;
    const int maxCT = a.maxCT;
    // Pointers are re-read from the (laundered) argument block at each use site instead of being
    // kept in SGPRs for the whole kernel: the block is tiny and scalar-cache resident, whereas
    // ~30 long-lived pointers forced hundreds of SGPR spill moves per wavelength.
#define ARGS (*launder_args(ap))
#define gPhi GCD(ARGS.phi)
#define gRho GCD(ARGS.rho)
#define gRatio GCD(ARGS.ratio)
#define gWphi GCD(ARGS.wphi)
#define gBgChi GCD(ARGS.bgChi)
#define gBgEta GCD(ARGS.bgEta)
#define gBgSca GCD(ARGS.bgSca)
#define gN GCD(ARGS.n)
#define gHeight GCD(ARGS.height)
#define gTemperature GCD(ARGS.temperature)
#define gJ GD(ARGS.J)
#define gI GD(ARGS.I)
#define gDepthChi GD(ARGS.depthChi)
#define gDepthEta GD(ARGS.depthEta)
#define gDepthI GD(ARGS.depthI)
#define gPartial GD(ARGS.partial)
#define cMuz CTAB(double, ARGS.muz)
#define cWmu CTAB(double, ARGS.wmu)
#define cWavelength CTAB(double, ARGS.wavelength)
#define cRayList CTAB(int32_t, dyn.rayList)
#define cChunkLa CTAB(int32_t, ARGS.chunkLa)
#define cLaHdr CTAB(DevLaHeader, ARGS.laHdr)
#define cSlots CTAB(DevSlot, ARGS.slots)
#define cProgs CTAB(DevProgram, ARGS.progs)

    double* sChi = lds;
    double* sS = sChi + GNs;
    double* sDS = sS + GNs;
    double* sB = sDS + GNs;
    double* sE = sChi; // alias: the chi column is dead once the optical depths exist
    double* sK = sB + GNs;
    double* sDh = sK;
    double* sRdh = sK + Ns;
    double* sRsum = sK + 2 * Ns;
    double* sTemp = sK + 3 * Ns;
    double* sJ = sK + 4 * Ns;   // sJ, sM1, sMeta0, sMeta1 are contiguous: moment m at sJ + m*Ns
    double* sIn = sK + 8 * Ns;  // Jdag, sca, bgChi, bgEta are contiguous: input m at sIn + m*Ns
    double* sJdag = sIn;
    double* sSca = sIn + Ns;
    double* sBgChi = sIn + 2 * Ns;
    double* sBgEta = sIn + 3 * Ns;
    double* sRT = sK + 12 * Ns; // 1 / T
    double* carve = sK + 13 * Ns;
    double* sRows;
    double* sVji;
    double* sChiCo;
    double* sAcc;
    const int nRowsTot = 2 + Natom + 2 * NlevTot;
    if (BIG)
    {
        double* gs = ARGS.bigScratch + (size_t)blockIdx.x * (size_t)(nRowsTot + 2 * maxC) * Ns;
        sRows = gs;
        sVji = gs + (size_t)nRowsTot * Ns;
        sChiCo = sVji + (size_t)maxC * Ns;
        sAcc = ARGS.partial + (size_t)blockIdx.x * maxCT * 4 * Ns;
    }
    else
    {
        sRows = carve;
        sVji = carve + nRowsTot * Ns;
        sChiCo = sVji + maxC * Ns;
        sAcc = sChiCo + maxC * Ns;
        carve = sAcc + maxCT * 4 * Ns;
    }
    double* sChiC = sRows;
    double* sEtaC = sRows + Ns;
    double* sEtaA = sRows + 2 * Ns;
    double* sLevChi = sEtaA + Natom * Ns;
    double* sLevU = sLevChi + NlevTot * Ns;
    DevProgRow* sProgRows = (DevProgRow*)carve;
    DevProgEnt* sProgEnts = (DevProgEnt*)(sProgRows + LWHIP_MAX_PROG_ENTS * 0 + LWHIP_MAX_PROG_ROWS);
    double* sRayMuz = (double*)(sProgEnts + LWHIP_MAX_PROG_ENTS); // [64] muz of each active ray
    double* sRayW = sRayMuz + 64;                                  // [64] 0.5 * wmu
    int* sRayId = (int*)(sRayW + 64);                               // [64] r = 2 mu + toObs
    double* sCp2 = (double*)(sRayId + 64);                          // [maxC] 2hc/lambda^3 of continuum c
    int* sCgj = (int*)(sCp2 + maxC);                                // [maxC] upper level row of continuum c
    double* sPlanck = (double*)(sCgj + maxC + (maxC & 1));          // [4] B_nu at T[Ns-2], T[Ns-1], T[0], T[1]
    double* sDjWave = sPlanck + 4;                                   // [16] per-wavefront max of dJ

    // idx / Ns for idx * Ns < 2^32 as one mul_hi (magic = ceil(2^32 / Ns))
    const unsigned divMagic = (unsigned)((0x100000000ull + (unsigned)Ns - 1) / (unsigned)Ns);
    auto divNs = [&](int idx) { return (int)__umulhi((unsigned)idx, divMagic); };
    const bool isPoint = t < GNs;
    const int rl = t / Ns;
    const int k = t - rl * Ns;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int nWaves = T >> 6;

    const int chunk = blockIdx.x;
    const int la0 = cChunkLa[chunk];
    const int la1 = cChunkLa[chunk + 1];

    // ---- once per chunk: geometry, temperature, populations, accumulators ---------------------------
    if (t < Ns)
    {
        sTemp[t] = gTemperature[t];
        sRT[t] = 1.0 / gTemperature[t];
        double dh = 0.0, rdh = 0.0, rsum = 0.0;
        if (t < Ns - 1)
        {
            dh = fabs(gHeight[t] - gHeight[t + 1]);
            rdh = 1.0 / dh;
        }
        if (t >= 1 && t < Ns - 1)
            rsum = 1.0 / (fabs(gHeight[t - 1] - gHeight[t]) + dh);
        sDh[t] = dh;       // |h[k] - h[k+1]|: interval between k and k+1
        sRdh[t] = rdh;
        sRsum[t] = rsum;   // 1 / (dh[k-1] + dh[k])
    }
    if (RATES)
    {
        for (int idx = t; idx < maxCT * 4 * Ns; idx += T)
            sAcc[idx] = 0.0;
    }
    if (t < dyn.nRaysActive && t < 64)
    {
        const int r = cRayList[t];
        sRayId[t] = r;
        sRayMuz[t] = cMuz[r >> 1];
        sRayW[t] = 0.5 * cWmu[r >> 1];
    }
    // total quadrature weight of the active rays, in ray order
    double Wsum = 0.0;
    for (int r = 0; r < dyn.nRaysActive; ++r)
        Wsum += 0.5 * cWmu[cRayList[r] >> 1];
    int curProg = -1;
    int progRows = 0;
    // per-depth inputs of a wavelength: input m (Jdag, sca, bgChi, bgEta) row kk is owned by thread
    // m*Ns + kk, which loads it one wavelength ahead
    const int inM = t / Ns;
    const int inK = t - inM * Ns;
    const bool inOwner = t < 4 * Ns;
    auto load_input = [&](int la_) -> double {
        const size_t o = (size_t)la_ * Ns + inK;
        return (inM == 0) ? gJ[o] : (inM == 1) ? gBgSca[o] : (inM == 2) ? gBgChi[o] : gBgEta[o];
    };
    // the chunk is a range of wavelengths, or (PRD rates pass) a range of an explicit wavelength list
    const CONST_AS int32_t* cLaList = CTAB(int32_t, ARGS.laList);
    const int laFirst = cLaList ? cLaList[la0] : la0;
    for (int task = t; task < 4 * Ns; task += T) // prologue (also when T < 4 Ns)
    {
        const int m = divNs(task), kk = task - m * Ns;
        const size_t o = (size_t)laFirst * Ns + kk;
        sIn[task] = (m == 0) ? gJ[o] : (m == 1) ? gBgSca[o] : (m == 2) ? gBgChi[o] : gBgEta[o];
    }
    // otherwise (and in the PRD rates pass, which skips wavelengths) every wavelength reloads synchronously
    const bool pfInputs = (T >= 4 * Ns) && !dyn.prdOnly;
    double pfK = 0.0, pfRatio = 0.0;
    bool havePfRatio = false;
    double pfPhi[MAXL];
    bool havePfPhi = false;
#pragma unroll
    for (int l = 0; l < MAXL; ++l)
        pfPhi[l] = 0.0;
    __syncthreads();
    STAMP(0);

    DevLaHeader hNext = ld_c(cLaHdr + laFirst);
    for (int li = la0; li < la1; ++li)
    {
        const int la = cLaList ? cLaList[li] : li;
        const DevLaHeader h = hNext;
        const bool lastLa = (li + 1 >= la1);
        if (!lastLa)
            hNext = ld_c(cLaHdr + (cLaList ? cLaList[li + 1] : li + 1));
        if (dyn.prdOnly && !h.hasPrd) // formal_sol_prd_update_rates visits prdIdxs only (PrdTemplates.hpp:226-239)
        {
            if (RATES && t == 0)
                GD(ARGS.dJ)[la] = 0.0;
            continue;
        }
        const CONST_AS DevSlot* slots = cSlots + h.off;
        const CONST_AS DevSlot* slotsNext = cSlots + hNext.off;
        const int nLine = h.nLine;
        const int nPerRayC = h.nMixed;
        const int nCont = h.nMixed + h.nPure;
        // wavelengths without an active continuum (most of a line-dominated grid) need no row program:
        // the totals are the background rows themselves and every per-level continuum sum is zero
        const bool hasCont = nCont > 0;
        const double wav = cWavelength[la];
        const double hc_kl = HC_K / wav;

        // ---- ray-independent part ---------------------------------------------------------------------
        if (hasCont && h.prog != curProg) // uniform: the active continuum set changed -> stage its row program
        {
            const DevProgram pg = ld_c(cProgs + h.prog);
            for (int idx = t; idx < pg.nRows; idx += T)
                sProgRows[idx] = a.progRows[pg.rowOff + idx];
            for (int idx = t; idx < pg.nEnt; idx += T)
                sProgEnts[idx] = a.progEnts[pg.entOff + idx];
            curProg = h.prog;
            progRows = pg.nRows;
        }
        if (!pfInputs && li != la0)
        {
            for (int task = t; task < 4 * Ns; task += T)
            {
                const int m = divNs(task), kk = task - m * Ns;
                const size_t o = (size_t)la * Ns + kk;
                sIn[task] = (m == 0) ? gJ[o] : (m == 1) ? gBgSca[o] : (m == 2) ? gBgChi[o] : gBgEta[o];
            }
        }
        // software prefetch of the next wavelength's per-depth inputs (consumed after this one)
        if (pfInputs && inOwner && !lastLa)
            pfK = load_input(la + 1);
        // continua: Vji = gij alpha with gij = nStar_i/nStar_j exp(-hc/(k lambda T))
        // (Atom::setup_wavelength LwAtom.hpp:107-115, Transition::uv LwTransition.hpp:132-142)
        // and chi_c = n_i Vij - n_j Vji (chi_eta_aux_accum :70)
        if (t < nCont)
        {
            sCp2[t] = slots[nLine + t].p[2];
            sCgj[t] = slots[nLine + t].gj;
        }
        if (t >= T - 4) // boundary Planck functions, once per wavelength (the last wavefront has idle lanes)
        {
            const int q = t - (T - 4);
            const int kq = (q == 0) ? Ns - 2 : (q == 1) ? Ns - 1 : (q == 2) ? 0 : 1;
            sPlanck[q] = d_planck(sTemp[kq], wav);
        }
        for (int idx = t; idx < nCont * Ns && (!(LWHIP_ABLATE & 8) || li == la0); idx += T)
        {
            const int c = divNs(idx);
            const int kk = idx - c * Ns;
            const CONST_AS DevSlot& sl = slots[nLine + c];
            const double ratio = (havePfRatio && idx == t) ? pfRatio : gRatio[(size_t)sl.row * Ns + kk];
            const double gij = ratio * d_exp_neg(-hc_kl * sRT[kk]);
            const double vji = gij * sl.p[0];
            sVji[c * Ns + kk] = vji;
            sChiCo[c * Ns + kk] = gN[sl.gi * Ns + kk] * sl.p[0] - gN[sl.gj * Ns + kk] * vji;
        }
        havePfRatio = false;
        if (!lastLa && !dyn.prdOnly)
        {
            const int nContN = hNext.nMixed + hNext.nPure;
            if (t < nContN * Ns)
            {
                const int c = t / Ns;
                pfRatio = gRatio[(size_t)slotsNext[hNext.nLine + c].row * Ns + (t - c * Ns)];
            }
            havePfRatio = true;
        }
        wg_barrier<BIG>();
        STAMP(1);
        if (hasCont)
        {
            // row program: totals, per-atom emissivity and per-level sums over the continua, one
            // (row, depth) task per thread, continua added in reference order
            for (int task = t; task < progRows * Ns && (!(LWHIP_ABLATE & 8) || li == la0); task += T)
            {
                const int ri = divNs(task);
                const int kk = task - ri * Ns;
                const DevProgRow pr = sProgRows[ri];
                double v = 0.0;
                if (pr.row == 0)
                    v = sBgChi[kk];
                else if (pr.row == 1)
                    v = sBgEta[kk];
                for (int e = 0; e < pr.nEnt; ++e)
                {
                    const DevProgEnt en = sProgEnts[pr.entOff + e];
                    const double x = sChiCo[en.c * Ns + kk];
                    const double uji = sCp2[en.c] * sVji[en.c * Ns + kk];
                    if (en.kind == 0)
                        v += x;
                    else if (en.kind == 1)
                        v -= x;
                    else if (en.kind == 2)
                        v += uji;
                    else
                        v += gN[sCgj[en.c] * Ns + kk] * uji;
                }
                sRows[pr.row * Ns + kk] = v;
            }
            wg_barrier<BIG>();
        }
        STAMP(2);

        // ---- passes over groups of G rays -----------------------------------------------------------
        double mJ = 0.0, mM1 = 0.0, mE0 = 0.0, mE1 = 0.0; // this thread's share of J, M1, Meta0, Meta1
        for (int g0 = 0; g0 < dyn.nRaysActive; g0 += G)
        {
            const int nr = min(G, dyn.nRaysActive - g0);
            const bool active = isPoint && rl < nr;
            int mu = 0, toObs = 0;
            if (active)
            {
                const int r = sRayId[g0 + rl];
                mu = r >> 1;
                toObs = r & 1;
            }
            const double muz = active ? sRayMuz[g0 + rl] : 1.0;
            const double halfwmu = active ? sRayW[g0 + rl] : 0.0;
            double chi = 1.0, S = 0.0, eta = 0.0, rchi = 1.0;
            double Vij[MAXL], Vji[MAXL], xl[MAXL], el[MAXL], wph[MAXL];
#pragma unroll
            for (int l = 0; l < MAXL; ++l)
            {
                Vij[l] = 0.0;
                Vji[l] = 0.0;
                xl[l] = 0.0;
                el[l] = 0.0;
                wph[l] = 0.0;
            }
            // gather: Transition::uv + chi_eta_aux_accum for the (ray-dependent) lines
            if (active)
            {
                chi = hasCont ? sChiC[k] : sBgChi[k];
                eta = hasCont ? sEtaC[k] : sBgEta[k];
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                {
                    if (l < nLine && !(LWHIP_ABLATE & 64))
                    {
                        const CONST_AS DevSlot& sl = slots[l];
                        const double ph = havePfPhi ? pfPhi[l]
                                                    : gPhi[sl.phiOff + (size_t)(mu * 2 + toObs) * Ns + k];
                        wph[l] = gWphi[(size_t)sl.row * Ns + k]; // issued here, first used in the rates phase
                        const double vij = sl.p[0] * ph;
                        double g = sl.p[2];
                        if (sl.flags & 2)
                            g *= gRho[sl.rhoOff + k];
                        const double vji = g * vij;
                        const double uji = sl.p[3] * vji;
                        const double x = gN[sl.gi * Ns + k] * vij - gN[sl.gj * Ns + k] * vji;
                        const double em = gN[sl.gj * Ns + k] * uji;
                        Vij[l] = vij;
                        Vji[l] = vji;
                        xl[l] = x;
                        el[l] = em;
                        chi += x;
                        eta += em;
                    }
                }
                // compute_source_fn :169-179
                rchi = d_rcp(chi);
                S = (eta + sSca[k] * sJdag[k]) * rchi;
                sChi[t] = chi;
                sS[t] = S;
                if (RATES && a.storeDepth && !dyn.prdOnly)
                {
                    const size_t o = ((size_t)(la * Nrays + mu) * 2 + toObs) * Ns + k;
                    gDepthChi[o] = chi;
                    gDepthEta[o] = eta;
                }
            }
            // software prefetch of phi for the next pass / next wavelength
            {
                const bool nextSameLa = (g0 + G < dyn.nRaysActive);
                havePfPhi = false;
                if ((nextSameLa || (!lastLa && !dyn.prdOnly)) && !(LWHIP_ABLATE & 512))
                {
                    const int g0n = nextSameLa ? g0 + G : 0;
                    const CONST_AS DevSlot* sn = nextSameLa ? slots : slotsNext;
                    const int nLineN = nextSameLa ? nLine : hNext.nLine;
                    const int nrn = min(G, dyn.nRaysActive - g0n);
                    if (isPoint && rl < nrn)
                    {
                        const int rn = sRayId[g0n + rl];
#pragma unroll
                        for (int l = 0; l < MAXL; ++l)
                            if (l < nLineN)
                                pfPhi[l] = gPhi[sn[l].phiOff + (size_t)rn * Ns + k];
                    }
                    havePfPhi = true;
                }
            }
            wg_barrier<BIG>();
            STAMP(3);

            // ---- formal solver: per-point coefficients e, b, Psi of I_k = e I_upwind + b ------------
            const int dk = toObs ? -1 : 1;
            const int kS = toObs ? Ns - 1 : 0;
            const int kE = toObs ? 0 : Ns - 1;
            const int cb = rl * Ns; // column base
            double e = 0.0, b = 0.0, Psi = 0.0;

            // boundary intensity (wrappers FormalScalar.cpp:471-666)
            double Istart = 0.0;
            if (active && k == kS)
            {
                const double zb = (SOLVER == LWHIP_FS_LINEAR_1D ? 0.5 : 1.0) / muz;
                double dtau_b;
                if (SOLVER == LWHIP_FS_LINEAR_1D)
                    dtau_b = zb * (sChi[cb + kS] + sChi[cb + kS + dk]) * sDh[min(kS, kS + dk)];
                else
                    dtau_b = 0.5 * zb * (sChi[cb + kS] + sChi[cb + kS + dk]) * sDh[min(kS, kS + dk)];
                if (toObs)
                {
                    if (a.lowerBc == LWHIP_BC_THERMALISED)
                    {
                        const double B0 = sPlanck[0];
                        const double B1 = sPlanck[1];
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
                        const double B0 = sPlanck[2];
                        const double B1 = sPlanck[3];
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
                // piecewise_bezier3_1d_impl FormalScalar.cpp:209-325, restated per point: every
                // carried quantity except I_upw is a local stencil.  Each thread evaluates the
                // chi-derivative at its own point and both neighbours (division-free), which
                // gives it the optical depth of both adjacent intervals without an exchange.
                const double zmu = 1.0 / muz;
                const double* col = sChi + cb;
                auto ds_of = [&](int ka, int kb) { return sDh[min(ka, kb)] * zmu; };   // |dh| * zmu
                auto rds_of = [&](int ka, int kb) { return sRdh[min(ka, kb)] * muz; }; // 1 / ds
                auto dchi_at = [&](int p) {
                    if (p == kS)
                        return (col[p + dk] - col[p]) * rds_of(p, p + dk); // dx_uw :239
                    if (p == kE)
                        return (col[p] - col[p - dk]) * rds_of(p - dk, p); // dx_dw :288
                    return d_cent_deriv_r(ds_of(p - dk, p), ds_of(p, p + dk), rds_of(p - dk, p),
                                          rds_of(p, p + dk), sRsum[p] * muz, col[p - dk], col[p],
                                          col[p + dk]); // :240,251
                };
                // optical depth of the interval (pa -> pb), pb = pa + dk  :242-246, 261-263, 289-292
                auto dtau_of = [&](int pa, int pb, double dchi_a, double dchi_b) {
                    const double ds = ds_of(pa, pb);
                    const double ds3 = ds * (1.0 / 3.0);
                    const double cuw = col[pa] + ds3 * dchi_a;
                    const double c0 = col[pb] - ds3 * dchi_b;
                    return ds * (col[pa] + col[pb] + cuw + c0) * 0.25;
                };
                double dtau = 0.0, dtauLin = 0.0, dS = 0.0;
                if (active && (LWHIP_ABLATE & 2))
                {
                    dtau = 0.3;
                    dtauLin = 0.3;
                    sDS[t] = 0.0;
                }
                if (active && !(LWHIP_ABLATE & 2))
                {
                    const double d0 = dchi_at(k);
                    double dtauDw = 0.0;
                    if (k != kS)
                        dtau = dtau_of(k - dk, k, dchi_at(k - dk), d0);
                    if (k != kE)
                        dtauDw = dtau_of(k, k + dk, d0, dchi_at(k + dk));
                    // derivative of S with respect to tau  :247, 254-255, 268
                    if (k == kS)
                        dS = (sS[cb + k + dk] - S) * d_rcp(dtauDw);
                    else if (k != kE)
                        dS = d_cent_deriv_r(dtau, dtauDw, d_rcp(dtau), d_rcp(dtauDw), d_rcp(dtau + dtauDw),
                                            sS[cb + k - dk], S, sS[cb + k + dk]);
                    else // piecewise linear on end :309
                        dtauLin = 0.5 * zmu * (chi + col[k - dk]) * sDh[min(k, k - dk)];
                    sDS[t] = dS;
                }
                wg_barrier<BIG>();
                STAMP(4);
                // coefficients  :265-275, 294-304, 314-321.  One exp call site serves the interior
                // points (Bezier3_coeffs) and the end point (w2), so a wavefront never runs two.
                if (active && (LWHIP_ABLATE & 128))
                {
                    e = 0.5;
                    b = S;
                    Psi = 0.1;
                }
                if (active && !(LWHIP_ABLATE & 128))
                {
                    const bool isEnd = (k == kE);
                    const double x = isEnd ? dtauLin : dtau;
                    const bool needExp = (k != kS) && (isEnd ? (x >= 5.0E-4 && x <= 50.0) : (x >= 5e-2 && x <= 30.0));
                    const double ex = needExp ? ((LWHIP_ABLATE & 32) ? 0.5 : d_exp_neg(-x)) : 0.0;
                    if (k == kS)
                    {
                        e = 0.0;
                        b = Istart;
                        Psi = 0.0;
                    }
                    else if (isEnd)
                    {
                        const double rl_ = d_rcp(x);
                        const double dSuw = (S - sS[cb + k - dk]) * rl_;
                        double w0, w1;
                        if (x < 5.0E-4) // w2, LwInternal.hpp:90-110
                        {
                            w0 = x * (1.0 - 0.5 * x);
                            w1 = (x * x) * (0.5 - x * (1.0 / 3.0));
                        }
                        else if (x > 50.0)
                        {
                            w0 = 1.0;
                            w1 = 1.0;
                        }
                        else
                        {
                            w0 = 1.0 - ex;
                            w1 = w0 - x * ex;
                        }
                        e = 1.0 - w0;
                        b = w0 * S - w1 * dSuw;
                        Psi = w0 - w1 * rl_;
                    }
                    else
                    {
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
                            edt = ex; // 0 beyond dt = 30: the asymptotic branch is the exact one with edt = 0
                            const double r3 = d_rcp(dt3c);
                            al = (6.0 - edt * (6.0 + 6.0 * x + 3 * dt2 + dt3c)) * r3;
                            be = (6.0 * edt - 6.0 + 6.0 * x - 3.0 * dt2 + dt3c) * r3;
                            ga = 3.0 * (2.0 * x - 6.0 + edt * (6.0 + 4.0 * x + dt2)) * r3;
                            de = 3.0 * (6.0 - 4.0 * x + dt2 - 2.0 * edt * (3.0 + x)) * r3;
                        }
                        const double Suw = sS[cb + k - dk];
                        const double dt3 = x * (1.0 / 3.0);
                        const double cuw = Suw + dt3 * sDS[cb + k - dk];
                        const double c0 = S - dt3 * dS;
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
                        const double zmu = 0.5 / muz;
                        const double dtau = zmu * (sChi[cb + k - dk] + chi) * sDh[min(k, k - dk)];
                        const double rcp = d_rcp(dtau);
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
                    const double zmu = 1.0 / muz;
                    if (k == kS)
                    {
                        e = 0.0;
                        b = Istart;
                    }
                    else if (k == kE)
                    {
                        const double dtau = 0.5 * zmu * (chi + sChi[cb + k - dk]) * sDh[min(k, k - dk)];
                        const double rcp = d_rcp(dtau);
                        const double dSuw = (S - sS[cb + k - dk]) * rcp;
                        double w0, w1;
                        d_w2(dtau, w0, w1);
                        e = 1.0 - w0;
                        b = w0 * S - w1 * dSuw;
                        Psi = w0 - w1 * rcp;
                    }
                    else
                    {
                        const double ds_uw = sDh[min(k, k - dk)] * zmu;
                        const double ds_dw = sDh[min(k, k + dk)] * zmu;
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
            if (active)
            {
                sE[t] = e;
                sB[t] = b;
            }
            wg_barrier<BIG>();
            STAMP(5);

            // ---- the recurrence I_k = e_k I_{k-1} + b_k: one wavefront per ray, each lane composes
            // its contiguous run of points, a shuffle scan composes the lanes  :273,302,317 ----------
            {
                const int Lc = (Ns + 63) >> 6;
                for (int rr = wave; rr < nr && !(LWHIP_ABLATE & 4); rr += nWaves)
                {
                    const int up = sRayId[g0 + rr] & 1;
                    const int q0 = lane * Lc;
                    const int q1 = min(q0 + Lc, Ns);
                    double E = 1.0, B = 0.0;
                    for (int q = q0; q < q1; ++q)
                    {
                        const int kk = up ? Ns - 1 - q : q;
                        const double ee = sE[rr * Ns + kk];
                        B = ee * B + sB[rr * Ns + kk];
                        E = ee * E;
                    }
                    affine_scan_wave(E, B);
                    double Iin = __shfl_up(B, 1);
                    if (lane == 0)
                        Iin = 0.0;
                    for (int q = q0; q < q1; ++q)
                    {
                        const int kk = up ? Ns - 1 - q : q;
                        Iin = sE[rr * Ns + kk] * Iin + sB[rr * Ns + kk];
                        sB[rr * Ns + kk] = Iin;
                    }
                }
            }
            wg_barrier<BIG>();
            STAMP(6);

            double I = 0.0;
            if (active)
            {
                I = sB[t];
                Psi = Psi * rchi; // PsiStar :322-323
                if (toObs && k == 0)
                    gI[(size_t)la * Nrays + mu] = I; // spect.I(la, mu, 0) :349
                if (RATES && a.storeDepth && !dyn.prdOnly)
                    gDepthI[((size_t)(la * Nrays + mu) * 2 + toObs) * Ns + k] = I;
            }

            // ---- per-ray rates: lines and mixed continua  :192-234; angle moments for the rest -------
            if (RATES && active && !(LWHIP_ABLATE & 1))
            {
                const double PsiS = dyn.lambdaIterate ? 0.0 : Psi;
                const double wPsi = halfwmu * PsiS;
                mJ += halfwmu * I; // accumulate_J :181-190 and the Psi* moments, in registers over the passes
                mM1 += wPsi;
                if (nLine == 1 && nPerRayC == 0)
                {
                    // the common case, one active line and no mixed continuum: its level sums are
                    // the continuum rows plus its own chi / Uji, no searching
                    const CONST_AS DevSlot& sl = slots[0];
                    const double uji = sl.p[3] * Vji[0];
                    const double wlamu = ((sl.p[1] * wph[0]) * PI4_HC) * halfwmu;
                    double* acc = sAcc + (size_t)sl.cs * 4 * Ns + k;
                    if (sl.flags & 1)
                    {
                        if (sl.ms == 0)
                            mE0 += wPsi * el[0];
                        else if (sl.ms == 1)
                            mE1 += wPsi * el[0];
                        double etaA = 0.0, lci = 0.0, lcj = 0.0, luj = 0.0, lui = 0.0;
                        if (hasCont)
                        {
                            etaA = sEtaA[sl.atom * Ns + k];
                            lci = sLevChi[sl.gi * Ns + k];
                            lcj = sLevChi[sl.gj * Ns + k];
                            luj = sLevU[sl.gj * Ns + k];
                            lui = sLevU[sl.gi * Ns + k];
                        }
                        const double Ieff = I - PsiS * (etaA + el[0]);
                        const double chi_i = lci + xl[0];
                        const double chi_j = lcj - xl[0];
                        const double U_j = luj + uji;
                        const double U_i = lui;
                        lds_add(acc, ((uji + Vji[0] * Ieff) - (PsiS * chi_i * U_j)) * wlamu);
                        lds_add(acc + Ns, ((Vij[0] * Ieff) - (PsiS * chi_j * U_i)) * wlamu);
                    }
                    lds_add(acc + 2 * Ns, I * Vij[0] * wlamu);
                    lds_add(acc + 3 * Ns, (uji + I * Vji[0]) * wlamu);
                }
                else if (nLine + nPerRayC > 0)
                {
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
                            const CONST_AS DevSlot& sl = slots[l];
                            lgi[l] = sl.gi;
                            lgj[l] = sl.gj;
                            lat[l] = sl.atom;
                            lop[l] = sl.flags & 1;
                            luji[l] = sl.p[3] * Vji[l];
                            if (lop[l])
                            {
                                if (sl.ms == 0)
                                    mE0 += wPsi * el[l];
                                else if (sl.ms == 1)
                                    mE1 += wPsi * el[l];
                            }
                        }
                    }
                    auto levChi = [&](int g) {
                        double v = 0.0;
                        if (hasCont)
                            v = sLevChi[g * Ns + k];
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
                    auto levU = [&](int g) {
                        double v = 0.0;
                        if (hasCont)
                            v = sLevU[g * Ns + k];
#pragma unroll
                        for (int l = 0; l < MAXL; ++l)
                        {
                            if (lop[l] && lgj[l] == g)
                                v += luji[l];
                        }
                        return v;
                    };
                    auto atomEta = [&](int at) {
                        double v = 0.0;
                        if (hasCont)
                            v = sEtaA[at * Ns + k];
#pragma unroll
                        for (int l = 0; l < MAXL; ++l)
                        {
                            if (lop[l] && lat[l] == at)
                                v += el[l];
                        }
                        return v;
                    };
#pragma unroll
                    for (int l = 0; l < MAXL; ++l)
                    {
                        if (l < nLine)
                        {
                            const CONST_AS DevSlot& sl = slots[l];
                            const double wla = (sl.p[1] * wph[l]) * PI4_HC;
                            const double wlamu = wla * halfwmu;
                            double* acc = sAcc + (size_t)sl.cs * 4 * Ns + k;
                            if (sl.flags & 1)
                            {
                                const double Ieff = I - PsiS * atomEta(sl.atom);
                                const double g0v = ((luji[l] + Vji[l] * Ieff) - (PsiS * levChi(sl.gi) * levU(sl.gj))) * wlamu;
                                const double g1v = ((Vij[l] * Ieff) - (PsiS * levChi(sl.gj) * levU(sl.gi))) * wlamu;
                                lds_add(acc, g0v);
                                lds_add(acc + Ns, g1v);
                            }
                            lds_add(acc + 2 * Ns, I * Vij[l] * wlamu);
                            lds_add(acc + 3 * Ns, (luji[l] + I * Vji[l]) * wlamu);
                        }
                    }
                    for (int c = 0; c < nPerRayC; ++c) // mixed continua (always of active atoms)
                    {
                        const CONST_AS DevSlot& sl = slots[nLine + c];
                        const double vij = sl.p[0];
                        const double vji = sVji[c * Ns + k];
                        const double uji = sl.p[2] * vji;
                        const double wlamu = sl.p[1] * halfwmu;
                        double* acc = sAcc + (size_t)sl.cs * 4 * Ns + k;
                        const double Ieff = I - PsiS * atomEta(sl.atom);
                        const double g0v = ((uji + vji * Ieff) - (PsiS * levChi(sl.gi) * levU(sl.gj))) * wlamu;
                        const double g1v = ((vij * Ieff) - (PsiS * levChi(sl.gj) * levU(sl.gi))) * wlamu;
                        lds_add(acc, g0v);
                        lds_add(acc + Ns, g1v);
                        lds_add(acc + 2 * Ns, I * vij * wlamu);
                        lds_add(acc + 3 * Ns, (uji + I * vji) * wlamu);
                    }
                }
            }
            STAMP(7);
        } // ray groups

        // ---- J and the Psi* moments of the wavelength: each thread holds the partial sums of its own
        // rays over the passes; stage them and add the G partials in a fixed order -----------------------
        if (RATES)
        {
            if (isPoint)
            {
                sChi[t] = mJ;
                sS[t] = mM1;
                sDS[t] = mE0;
                sB[t] = mE1;
            }
            wg_barrier<BIG>();
            for (int task = t; task < 4 * Ns && !(LWHIP_ABLATE & 256); task += T)
            {
                const int m = divNs(task);
                const int kk = task - m * Ns;
                const double* src = (m == 0) ? sChi : (m == 1) ? sS : (m == 2) ? sDS : sB;
                double acc = 0.0;
                for (int r0 = 0; r0 < G; r0 += 8) // loads issued together, adds in a fixed order
                {
                    double v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        v[q] = (r0 + q < G) ? src[(r0 + q) * Ns + kk] : 0.0;
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (r0 + q < G)
                            acc += v[q];
                }
                sJ[m * Ns + kk] = acc;
            }
            wg_barrier<BIG>();
            STAMP(8);
        }

        if (RATES)
        {
            // ---- pure continua from the angle moments -----------------------------------------------------
            // sum_r w_r [Uji + Vji (I_r - Psi_r eta_r) - Psi_r chi_i U_j] with chi, U, Uji, Vji ray-
            // independent and eta_r = etaC + etaLine_r  ->  Uji W + Vji (J - M1 etaC - Meta) - M1 chi_i U_j
            for (int idx = t; idx < h.nPure * Ns && !(LWHIP_ABLATE & 16); idx += T)
            {
                const int cq = divNs(idx);
                const int c = h.nMixed + cq;
                const int kk = idx - cq * Ns;
                const CONST_AS DevSlot& sl = slots[nLine + c];
                const double vij = sl.p[0];
                const double vji = sVji[c * Ns + kk];
                const double uji = sl.p[2] * vji;
                const double wla = sl.p[1];
                const double Jk = sJ[kk];
                double* acc = sAcc + (size_t)sl.cs * 4 * Ns + kk;
                if (sl.flags & 1)
                {
                    const double M1 = sJ[Ns + kk];
                    const double Meta = (sl.ms >= 0) ? sJ[(2 + sl.ms) * Ns + kk] : 0.0;
                    const double Jeff = Jk - M1 * sEtaA[sl.atom * Ns + kk] - Meta;
                    const double g0v = ((uji * Wsum + vji * Jeff) - (M1 * sLevChi[sl.gi * Ns + kk] * sLevU[sl.gj * Ns + kk])) * wla;
                    const double g1v = ((vij * Jeff) - (M1 * sLevChi[sl.gj * Ns + kk] * sLevU[sl.gi * Ns + kk])) * wla;
                    if (BIG) // accumulators in global memory are only ever touched atomically (L2)
                    {
                        lds_add(acc, g0v);
                        lds_add(acc + Ns, g1v);
                    }
                    else
                    {
                        acc[0] += g0v;
                        acc[Ns] += g1v;
                    }
                }
                if (BIG)
                {
                    lds_add(acc + 2 * Ns, Jk * vij * wla);
                    lds_add(acc + 3 * Ns, (uji * Wsum + Jk * vji) * wla);
                }
                else
                {
                    acc[2 * Ns] += Jk * vij * wla;
                    acc[3 * Ns] += (uji * Wsum + Jk * vji) * wla;
                }
            }
            // ---- J out, dJ = max_k |1 - Jdag/J|  :477-486 ---------------------------------------------------
            if ((t & ~63) < Ns) // whole wavefronts, so the shuffles see every lane
            {
                double dj = 0.0;
                if (t < Ns)
                {
                    const double Jn = sJ[t];
                    gJ[(size_t)la * Ns + t] = Jn;
                    dj = fabs(1.0 - sJdag[t] / Jn);
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1)
                    dj = fmax(dj, __shfl_xor(dj, off));
                if ((t & 63) == 0)
                    sDjWave[t >> 6] = dj;
            }
        }
        wg_barrier<BIG>();
        // hand the prefetched inputs of the next wavelength over to LDS (made visible by the next
        // wavelength's first barrier; nothing reads these rows before it)
        if (pfInputs && inOwner && !lastLa)
            sIn[t] = pfK;
        if (RATES && t == 0) // this wavelength belongs to this workgroup alone: plain store, no atomics
        {
            double dj = 0.0;
            for (int w = 0; w < (Ns + 63) / 64; ++w)
                dj = fmax(dj, sDjWave[w]);
            GD(ARGS.dJ)[la] = dj;
        }
        STAMP(9);
    } // wavelengths

    if (RATES && !BIG)
    {
        if (ATOMIC)
        {
            // straight into the stage-1 buffer: part = chunk % SPLIT keeps 1/SPLIT of the chunks on one address
            const CONST_AS int32_t* slotTr = CTAB(int32_t, a.slotTr) + (size_t)chunk * maxCT;
            GLOBAL_AS double* r8 = GD(a.red8);
            const int n4 = 4 * Ns;
            const int part = chunk % LWHIP_REDUCE_SPLIT;
            for (int idx = t; idx < maxCT * n4; idx += T)
            {
                const int cs = divNs(idx) >> 2;           // idx / (4 Ns)
                const int tr = slotTr[cs];
                if (tr >= 0)
                    unsafeAtomicAdd((double*)(r8 + ((size_t)tr * LWHIP_REDUCE_SPLIT + part) * n4 + (idx - cs * n4)), sAcc[idx]);
            }
        }
        else
        {
            GLOBAL_AS double* out = gPartial + (size_t)chunk * maxCT * 4 * Ns;
            for (int idx = t; idx < maxCT * 4 * Ns; idx += T)
                out[idx] = sAcc[idx];
        }
    }
#ifdef LWHIP_STAMPS
    STAMP(10);
    if (t == 0 && a.stamps)
    {
        for (int i = 0; i < 10; ++i)
            a.stamps[(size_t)chunk * 12 + i] = stampAcc[i];
        a.stamps[(size_t)chunk * 12 + 10] = wallStart;     // absolute start / end (wall_clock64 ticks)
        a.stamps[(size_t)chunk * 12 + 11] = wall_clock64();
    }
#endif
}

#undef ARGS

template <int SOLVER, int MAXL, bool RATES, bool BIG>
__global__ void __launch_bounds__(1024) sweep_kernel(const SweepArgs* __restrict__ ap, const SweepDyn dyn)
{
    sweep_body<SOLVER, MAXL, RATES, BIG>(ap, dyn);
}

// the one-call iteration's instantiation: accumulators leave by atomic adds into the stage-1 buffer (a compile-time
// choice: as a run-time branch it cost every instantiation 3.5 us of register pressure)
template <int SOLVER, int MAXL>
__global__ void __launch_bounds__(1024) sweep_atomic_kernel(const SweepArgs* __restrict__ ap, const SweepDyn dyn)
{
    sweep_body<SOLVER, MAXL, true, false, true>(ap, dyn);
}

// column batches (1.5D): blockIdx.y picks the column's argument block; every column has the same structure
template <int SOLVER, int MAXL, bool RATES, bool BIG>
__global__ void __launch_bounds__(1024) sweep_batch_kernel(const SweepArgs* const* __restrict__ apList, const SweepDyn dyn)
{
    sweep_body<SOLVER, MAXL, RATES, BIG>(apList[blockIdx.y], dyn);
}

template <int SOLVER, int MAXL, bool RATES, bool BIG>
static hipError_t launch_sweep_t(const SweepArgs* devArgs, const SweepArgs& a, const SweepDyn& dyn,
                                 int nChunks, hipStream_t stream, const SweepArgs* const* apList, int nBatch)
{
    const size_t lds = sweep_lds_bytes(a, dyn.G);
    static size_t configured = 0, configuredBatch = 0, configuredAtomic = 0;
    const void* fn = apList ? (const void*)sweep_batch_kernel<SOLVER, MAXL, RATES, BIG>
                            : (const void*)sweep_kernel<SOLVER, MAXL, RATES, BIG>;
    bool atomic = false;
    if constexpr (RATES && !BIG)
        if (dyn.atomicOut && !apList)
        {
            atomic = true;
            fn = (const void*)sweep_atomic_kernel<SOLVER, MAXL>;
        }
    if (dyn.atomicOut && !atomic)
        return hipErrorInvalidValue; // the caller asked for an instantiation that does not exist
    size_t& conf = atomic ? configuredAtomic : (apList ? configuredBatch : configured);
    if (lds > conf)
    {
        hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess)
            return err;
        conf = lds;
    }
    // at least one wavefront per ray of a pass (the scan) and one thread per point
    int T = ((std::max(dyn.G * a.Ns, 64) + 63) / 64) * 64;
    if (T > 1024)
        return hipErrorInvalidConfiguration;
    if constexpr (RATES && !BIG)
        if (atomic)
        {
            hipLaunchKernelGGL((sweep_atomic_kernel<SOLVER, MAXL>), dim3(nChunks), dim3(T), lds, stream, devArgs, dyn);
            return hipGetLastError();
        }
    if (apList)
        hipLaunchKernelGGL((sweep_batch_kernel<SOLVER, MAXL, RATES, BIG>), dim3(nChunks, std::max(nBatch, 1)), dim3(T), lds,
                           stream, apList, dyn);
    else
        hipLaunchKernelGGL((sweep_kernel<SOLVER, MAXL, RATES, BIG>), dim3(nChunks), dim3(T), lds, stream, devArgs, dyn);
    return hipGetLastError();
}

template <int SOLVER, bool RATES>
static hipError_t launch_sweep_l(const SweepArgs* devArgs, const SweepArgs& a, const SweepDyn& dyn,
                                 int maxL, int nChunks, hipStream_t stream, const SweepArgs* const* apList, int nBatch)
{
    if (a.big)
    {
        if (maxL <= 2)
            return launch_sweep_t<SOLVER, 2, RATES, true>(devArgs, a, dyn, nChunks, stream, apList, nBatch);
        if (maxL <= 4)
            return launch_sweep_t<SOLVER, 4, RATES, true>(devArgs, a, dyn, nChunks, stream, apList, nBatch);
        return launch_sweep_t<SOLVER, 8, RATES, true>(devArgs, a, dyn, nChunks, stream, apList, nBatch);
    }
    if (maxL <= 2)
        return launch_sweep_t<SOLVER, 2, RATES, false>(devArgs, a, dyn, nChunks, stream, apList, nBatch);
    if (maxL <= 4)
        return launch_sweep_t<SOLVER, 4, RATES, false>(devArgs, a, dyn, nChunks, stream, apList, nBatch);
    return launch_sweep_t<SOLVER, 8, RATES, false>(devArgs, a, dyn, nChunks, stream, apList, nBatch);
}

hipError_t launch_sweep(const SweepArgs* devArgs, const SweepArgs& a, const SweepDyn& dyn, int solver,
                        int maxL, bool rates, int nChunks, hipStream_t stream, const SweepArgs* const* apList, int nBatch)
{
    switch (solver)
    {
        case LWHIP_FS_LINEAR_1D:
            return rates ? launch_sweep_l<LWHIP_FS_LINEAR_1D, true>(devArgs, a, dyn, maxL, nChunks, stream, apList, nBatch)
                         : launch_sweep_l<LWHIP_FS_LINEAR_1D, false>(devArgs, a, dyn, maxL, nChunks, stream, apList, nBatch);
        case LWHIP_FS_BESSER_1D:
            return rates ? launch_sweep_l<LWHIP_FS_BESSER_1D, true>(devArgs, a, dyn, maxL, nChunks, stream, apList, nBatch)
                         : launch_sweep_l<LWHIP_FS_BESSER_1D, false>(devArgs, a, dyn, maxL, nChunks, stream, apList, nBatch);
        default:
            return rates ? launch_sweep_l<LWHIP_FS_BEZIER3_1D, true>(devArgs, a, dyn, maxL, nChunks, stream, apList, nBatch)
                         : launch_sweep_l<LWHIP_FS_BEZIER3_1D, false>(devArgs, a, dyn, maxL, nChunks, stream, apList, nBatch);
    }
}

// ---- slab reduction: fixed-order sum of the per-chunk partials of each transition ----------------------
// replaces AtomStorageFactory::accumulate_Gamma / TransitionStorageFactory::accumulate_rates
// (ThreadStorage.cpp:73-88,150-166)
// (BATCH: column batches -- blockIdx.z picks the column's argument block from a table read through the constant
// address space; the single-problem instantiation keeps its kernel-argument block untouched)
template <class A> DEVINL void reduce_stage1_kernel_body(const A& a)
{
    const int tr = blockIdx.x;
    const int part = blockIdx.y;
    const int n4 = 4 * a.Ns;
    const int e0 = a.trOff[tr], e1 = a.trOff[tr + 1];
    const int per = (e1 - e0 + LWHIP_REDUCE_SPLIT - 1) / LWHIP_REDUCE_SPLIT;
    const int b0 = e0 + part * per;
    const int b1 = min(b0 + per, e1);
    for (int idx = threadIdx.x; idx < n4; idx += blockDim.x)
    {
        double s = 0.0;
        for (int e = b0; e < b1; ++e)
            s += a.partial[(size_t)a.trRow[e] * n4 + idx];
        a.red8[((size_t)tr * LWHIP_REDUCE_SPLIT + part) * n4 + idx] = s;
    }
}

template <bool BATCH>
__global__ void reduce_stage1_kernel(const ReduceArgs a0, const ReduceArgs* __restrict__ list)
{
    if constexpr (BATCH)
        reduce_stage1_kernel_body(CTAB(ReduceArgs, list)[blockIdx.z]);
    else
        reduce_stage1_kernel_body(a0);
}

__device__ void djmax_block(const double* dJ, int laEnd, int idxOffset, double* tail, int world, int rank,
                            double* mirror = nullptr /* host-mapped copy of the tail: no read-back copy needed */)
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
    for (int q = threadIdx.x; q < 2 * world; q += blockDim.x)
    {
        double v = 0.0;
        if (q == 2 * rank)
            v = sV[0] < 0.0 ? 0.0 : sV[0];
        else if (q == 2 * rank + 1)
            v = (double)(sI[0] + idxOffset);
        tail[q] = v;
        if (mirror)
            mirror[q] = v;
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
#pragma unroll
        for (int part = 0; part < LWHIP_REDUCE_SPLIT; ++part)
        {
            const size_t o = ((size_t)tr * LWHIP_REDUCE_SPLIT + part) * n4 + idx;
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
    if constexpr (BATCH)
        reduce_stage2_kernel_body(CTAB(ReduceArgs, list)[blockIdx.z]);
    else
        reduce_stage2_kernel_body(a0);
}

hipError_t launch_reduce_parts(const ReduceArgs& a, hipStream_t stream)
{
    if (a.Ntrans > 0)
        hipLaunchKernelGGL(reduce_stage1_kernel<false>, dim3(a.Ntrans, LWHIP_REDUCE_SPLIT), dim3(384), 0, stream, a, nullptr);
    return hipGetLastError();
}

hipError_t launch_reduce_sum(const ReduceArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(reduce_stage2_kernel<false>, dim3(a.Ntrans + 1), dim3(512), 0, stream, a, nullptr);
    return hipGetLastError();
}

hipError_t launch_reduce(const ReduceArgs& a, hipStream_t stream, const ReduceArgs* list, int nBatch)
{
    const int nz = std::max(nBatch, 1);
    if (list)
    {
        if (a.Ntrans > 0)
            hipLaunchKernelGGL(reduce_stage1_kernel<true>, dim3(a.Ntrans, LWHIP_REDUCE_SPLIT, nz), dim3(384), 0, stream, a, list);
        hipLaunchKernelGGL(reduce_stage2_kernel<true>, dim3(a.Ntrans + 1, 1, nz), dim3(512), 0, stream, a, list);
        return hipGetLastError();
    }
    if (a.Ntrans > 0)
        hipLaunchKernelGGL(reduce_stage1_kernel<false>, dim3(a.Ntrans, LWHIP_REDUCE_SPLIT), dim3(384), 0, stream, a, list);
    hipLaunchKernelGGL(reduce_stage2_kernel<false>, dim3(a.Ntrans + 1), dim3(512), 0, stream, a, list);
    return hipGetLastError();
}

// ---- apply: Gamma += radiative terms (Gamma keeps its collisional pre-fill), finalise_Gamma
// (SimdFullIterationTemplates.hpp:491-508), Rij/Rji out ----------------------------------------------------
// One block per (atom, slab of APPLY_K depth points): thread (e, kk) owns Gamma entry e = to*Nl+from
// at depth k; the extra block column x == Natom copies Rij/Rji.
enum { APPLY_K = 16 };
template <class A> DEVINL void apply_kernel_body(const A& a)
{
    const int Ns = a.Ns;
    const int k0 = blockIdx.y * APPLY_K;
    if ((int)blockIdx.x == a.Natom + 1) // dJMax (only launched when a.dJ is set)
    {
        if (blockIdx.y == 0)
            djmax_block(a.dJ, a.NlaDj, a.idxOffset, a.tail, 1, 0, a.tailHost);
        return;
    }
    // the reduced sum of transition tr, component q at depth k: from the reduce buffer, or from the stage-1 parts
    auto redAt = [&](int tr, int q, int k) -> double {
        if (!a.fromParts)
            return a.red[((size_t)tr * 4 + q) * Ns + k];
        double s = 0.0;
        double* r8 = const_cast<double*>(a.red8);
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
        const int Ntr = a.atomTrOff[a.Natom];
        for (int idx = threadIdx.x; idx < Ntr * APPLY_K; idx += blockDim.x)
        {
            const int tr = idx / APPLY_K;
            const int k = k0 + (idx - tr * APPLY_K);
            if (k < Ns)
            {
                if (a.prdOnly && !(a.trans[tr].prd && a.trans[tr].op && a.trans[tr].rhoOff >= 0))
                    continue;
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
    const int Nl = a.atomNlevel[at];
    double* G = a.Gamma + a.atomGammaOff[at];
    const double* Cm = a.Cmat + a.atomGammaOff[at];
    const int tr0 = a.atomTrOff[at], tr1 = a.atomTrOff[at + 1];
    // off-diagonal entries: pre-fill (optional) + radiative terms of the transitions that feed them
    for (int idx = threadIdx.x; idx < Nl * Nl * APPLY_K; idx += blockDim.x)
    {
        const int e = idx / APPLY_K;
        const int k = k0 + (idx - e * APPLY_K);
        if (k >= Ns)
            continue;
        const int to = e / Nl, from = e - to * Nl;
        double v = a.prefill ? a.crsw * Cm[(size_t)e * Ns + k] : G[(size_t)e * Ns + k];
        for (int tr = tr0; tr < tr1; ++tr) // transitions in reference order
        {
            const int i = a.transLi[tr], j = a.transLj[tr];
            if (to == i && from == j)
                v += redAt(tr, 0, k);
            if (to == j && from == i)
                v += redAt(tr, 1, k);
        }
        G[(size_t)e * Ns + k] = v;
    }
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
                gd += G[((size_t)j * Nl + i) * Ns + k];
        G[((size_t)i * Nl + i) * Ns + k] = -gd;
    }
}

template <bool BATCH>
__global__ void __launch_bounds__(1024) apply_kernel(const ApplyArgs a0, const ApplyArgs* __restrict__ list)
{
    if constexpr (BATCH)
        apply_kernel_body(CTAB(ApplyArgs, list)[blockIdx.z]);
    else
        apply_kernel_body(a0);
}

hipError_t launch_apply(const ApplyArgs& a, hipStream_t stream, const ApplyArgs* list, int nBatch)
{
    if (list)
        hipLaunchKernelGGL(apply_kernel<true>, dim3(a.Natom + 1, (a.Ns + APPLY_K - 1) / APPLY_K, std::max(nBatch, 1)), dim3(1024), 0,
                           stream, a, list);
    else
        hipLaunchKernelGGL(apply_kernel<false>, dim3(a.Natom + 1 + (a.dJ ? 1 : 0), (a.Ns + APPLY_K - 1) / APPLY_K), dim3(1024), 0,
                           stream, a, list);
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

}

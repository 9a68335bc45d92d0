// lwhip_raymarch.hip -- the ray-column sweep of the formal_sol_gamma_matrices iteration on gfx950 (MI355X).
//
// Mapping: the wavelength axis goes across the lanes, the depth axis is marched.
// A wavefront owns a *tile* of L = min(16, 64 / Nmu) structurally identical consecutive wavelengths in one
// direction; lane mu * L + j is the ray (wavelength la0 + j, angle mu).  It marches that ray through the
// atmosphere exactly as the reference's 1D solvers do (rolling three-point stencils in registers, the
// recurrence I_k = e_k I_k-1 + b_k carried in one register).  Consequences:
//   * everything that depends on depth only -- level populations, |dh|, w_phi, temperature -- is uniform
//     across the wavefront and comes in through *scalar* loads (one base address per depth: the depth table);
//   * everything that depends on wavelength only (B_ij h nu / 4 pi, quadrature weights, ...) is a lane
//     constant for the whole march;
//   * each stencil is evaluated once, there is no scan, no LDS column, no workgroup barrier in the march;
//   * the profile phi is stored wavelength-fastest per (direction, depth): one 512-byte row per wavefront
//     load; the ray-independent continuum terms come from a pre-pass, once per (wavelength, depth), in a
//     depth-major block so that one scalar base serves all rows of a depth;
//   * J and the Psi* moments are summed over the mu lanes of a wavelength through a wavefront-private LDS
//     row and stored per direction (the post-pass adds down + up in a fixed order: J is bit-reproducible);
//     the preconditioned rate integrals of the lines are summed over the rays of the step by a DPP
//     transpose-reduce and added to the workgroup's LDS accumulators; "pure" continua are integrated from the
//     angle moments by the post-pass;
//   * the loads of iteration t + 1 are issued at the top of iteration t; the steady state of the march is
//     one straight-line block (compile-time slot counts, branch-free Bezier coefficients).
// fp64 throughout, no MFMA: a streaming recurrence over fp64 columns, not a contraction.
//
// Reference restated (file:line under Lightweaver's Source/): intensity_core_opt
// SimdFullIterationTemplates.hpp:238-487, chi_eta_aux_accum :59-109, compute_full_operator_rates :192-234,
// piecewise_{linear,besser,bezier3}_1d FormalScalar.cpp:136-666, Bezier.hpp:58-127.
#include "lwhip_device.h"
#include "../../include/lwhip.h"

#include <algorithm>
#include <type_traits>

namespace lwhip
{
template <typename T> DEVINL const CONST_AS T* launder_c(const T* p)
{
    const CONST_AS T* q = (const CONST_AS T*)p;
    asm volatile("" : "+s"(q));
    return q;
}
#define TA (*launder_c(ap))

// ---- continuum terms of one (wavelength, depth) point ----------------------------------------------------------
// Atom::setup_wavelength (LwAtom.hpp:107-115) + Transition::uv (LwTransition.hpp:132-142) + chi_eta_aux_accum
// (SimdFullIterationTemplates.hpp:59-109) for the continua of one wavelength, visited in the reference's transition
// order: the totals chi_C, eta_C and, for the active atoms, the per-atom emissivity and the per-level sums
//   eta_A[atom] += n_j U_ji    chi[i] += chi_c    chi[j] -= chi_c    U[j] += U_ji
// accumulated in the thread's LDS column lev[slot * T] (slots: [eta_A x Natom][chi x NlevTot][U x NlevTot]).
// `us`: the continuum slot records of the tile's first wavelength (uniform structure: level rows, pool rows, flags,
// and in _pad the continuum's ordinal for position q of the transition order); `ls`: this thread's wavelength (p[]).
struct LevCols
{
    double* lev;
    int T, Natom, NlevTot;
    const int* remap; // pre-pass: dense slot -> column of this tile's compact set (unused slots share one trash column)
    DEVINL double* col(int slot) const { return lev + (size_t)remap[slot] * T; }
    // row id as the row programs number them: 0 chi_C, 1 eta_C, 2 + atom, 2 + Natom + level, 2 + Natom + NlevTot + level
    DEVINL double row(int id) const { return *col(id - 2); }
};
enum { RM_CB = 8 }; // continua of a wavelength the post-pass integrates from registers (more: a slower general walk)
// chi_C += sum chi_c, eta_C += sum eta_c and the level sums of the active atoms, continua in transition order.
// `cr`: the wavelength's continuum records (uniform); `lsB`: this thread's wavelength, byte address of its first slot record
// (the per-wavelength parameters p[]); kB = 8 k.  The next continuum's record and loads are requested before the
// current one is used.
DEVINL void cont_levels(const TileArgs* __restrict__ ap, const LevCols& lc, int nCols, const CONST_AS DevContRec* cr, const GLOBAL_AS char* lsB,
                        int nCont, unsigned kB, double boltz, double& chiC, double& etaC)
{
    for (int q = 0; q <= nCols; ++q) // the tile's compact columns and the trash column
        lc.lev[(size_t)q * lc.T] = 0.0;
    if (nCont <= 0)
        return;
    const GLOBAL_AS char* nB = (const GLOBAL_AS char*)TA.n;
    const GLOBAL_AS char* rB = (const GLOBAL_AS char*)TA.ratio;
    auto ldd = [](const GLOBAL_AS char* b, unsigned o) { return *(const GLOBAL_AS double*)(b + o); };
    DevContRec r = ld_c(cr);
    double ratio = ldd(rB + r.rat, kB), ni = ldd(nB + r.nI, kB), nj = ldd(nB + r.nJ, kB), p0 = ldd(lsB, r.pOff), p2 = ldd(lsB, r.pOff + 16);
    for (int q = 0; q < nCont; ++q)
    {
        const DevContRec rc = r;
        const double ratioC = ratio, niC = ni, njC = nj, p0C = p0, p2C = p2;
        if (q + 1 < nCont)
        {
            r = ld_c(cr + q + 1);
            ratio = ldd(rB + r.rat, kB);
            ni = ldd(nB + r.nI, kB);
            nj = ldd(nB + r.nJ, kB);
            p0 = ldd(lsB, r.pOff);
            p2 = ldd(lsB, r.pOff + 16);
        }
        const double vji = (ratioC * boltz) * p0C;
        const double chi_c = niC * p0C - njC * vji;
        const double uji = p2C * vji;
        const double eta_c = njC * uji;
        chiC += chi_c;
        etaC += eta_c;
        if (rc.flags & 1)
        {
            // LDS adds without a return value: no read-modify-write round trip, executed in program order
            lds_add(lc.col(rc.sEta), eta_c);
            lds_add(lc.col(rc.sChiI), chi_c);
            lds_add(lc.col(rc.sChiJ), -chi_c);
            lds_add(lc.col(rc.sUj), uji);
        }
    }
}

// ---- pre-pass: the tiles' row blocks and the depth table ----------------------------------------------------------
enum { RM_PRE_T = 256 };
enum { RM_ROW = 16 }; // doubles per row of a tile's row block (L <= 16 used): row offsets are instruction immediates
// the depth table: everything the march reads per depth only, contiguous per depth (scalar loads off one base)
DEVINL void depth_table(const TileArgs* __restrict__ ap, int t, int T)
{
    const int Ns = TA.Ns, st = TA.ktStride, NlevTot = TA.NlevTot, Nline = TA.Nline;
    for (int idx = t; idx < (Ns + 2) * st; idx += T)
    {
        const int k = idx / st - 1, q = idx - (k + 1) * st; // entries -1 and Ns are padding (read, never used)
        double v = 0.0;
        if (k < 0 || k >= Ns)
            v = 0.0;
        else if (q < 4)
            v = GCD(TA.geo)[4 * k + q];
        else if (q < 4 + NlevTot)
            v = GCD(TA.n)[(size_t)(q - 4) * Ns + k];
        else if (q < 4 + NlevTot + Nline)
            v = GCD(TA.wphi)[(size_t)(q - 4 - NlevTot) * Ns + k];
        GD(TA.kt)[idx] = v;
    }
}
// one point p = k * L + j of a tile (wavelength fastest: coalesced row writes): its rows of the tile's row block
DEVINL void pre_point(const TileArgs* __restrict__ ap, const DevTile& td, const DevLaHeader& h, const LevCols& lc, int nCols, int p)
{
    const int Ns = TA.Ns, L = TA.L;
    // ray-column march: wavelength fastest (rows of RM_ROW doubles per depth); lane sweep: depth fastest
    // ([row][wavelength][k] blocks): coalesced row writes either way
    const bool lanes = TA.laneD > 0;
    const int k = lanes ? p % Ns : p / L;
    const int j = lanes ? p / Ns : p - k * L;
    if (j >= td.nl)
        return;
    const int la = td.la0 + j;
    // structure from the tile's first wavelength (uniform), parameters from this thread's own
    const int off = ((const GLOBAL_AS DevLaHeader*)TA.laHdr)[la].off;
    const int nCont = h.nMixed + h.nPure;
    const CONST_AS DevSlot* us = CTAB(DevSlot, TA.slots) + h.off + h.nLine;
    const CONST_AS DevContRec* cr = CTAB(DevContRec, TA.contRec) + h.off + h.nLine;
    const GLOBAL_AS DevSlot* lsAll = (const GLOBAL_AS DevSlot*)TA.slots + off;
    const GLOBAL_AS DevSlot* ls = lsAll + h.nLine;
    const double rT = GCD(TA.geo)[4 * k + 3];
    const double hc_kl = HC_K / GCD(TA.wavelength)[la];
    const size_t o = (size_t)la * Ns + k;
    double chiC = GCD(TA.bgChi)[o], etaC = GCD(TA.bgEta)[o];
    const double scaJ = GCD(TA.bgSca)[o] * GCD(TA.J)[o]; // sca * Jdag of compute_source_fn :169-179
    const double boltz = d_exp_neg(-hc_kl * rT);
    cont_levels(ap, lc, nCols, cr, (const GLOBAL_AS char*)lsAll, nCont, (unsigned)k * 8u, boltz, chiC, etaC);
    const size_t rs = lanes ? (size_t)L * TA.laneLRD : (size_t)RM_ROW; // doubles between two rows of a point
    GLOBAL_AS double* out = GD(TA.rows) + td.rowBase + (lanes ? (size_t)j * TA.laneLRD + k : (size_t)k * td.nRowsT * RM_ROW + j);
    out[0] = chiC;
    out[rs] = etaC;
    out[2 * rs] = scaJ;
    const CONST_AS DevTileCopy* cp = CTAB(DevTileCopy, TA.copies) + td.copyOff;
    for (int q = 0; q < td.nCopy; ++q)
    {
        const DevTileCopy c = ld_c(cp + q);
        double v = 0.0;
        if (c.kind == 0)
            v = lc.row(c.src);
        else if (c.kind == 1) // V_ji of a mixed continuum
            v = (GCD(TA.ratio)[(size_t)us[c.src].row * Ns + k] * boltz) * ls[c.src].p[0];
        else if (c.kind == 2)
            v = GCD(TA.rho)[lsAll[c.src].rhoOff + k];
        out[(size_t)c.dst * rs] = v;
    }
}
// stand-alone pre-pass: grid (tile, part[, column]); block (0, 0) also refreshes the depth table
template <bool BATCH>
__global__ void __launch_bounds__(RM_PRE_T) tile_pre_kernel(const TileArgs* __restrict__ ap0, const TileArgs* const* __restrict__ apList,
                                                            const int32_t* __restrict__ tileList)
{
    dbg_poison_lds();
    extern __shared__ double lds[];
    const TileArgs* __restrict__ ap = BATCH ? apList[blockIdx.z] : ap0;
    if (blockIdx.x == 0 && blockIdx.y == 0)
        depth_table(ap, threadIdx.x, RM_PRE_T);
    const int ti = tileList ? CTAB(int32_t, tileList)[blockIdx.x] : (int)blockIdx.x;
    const DevTile td = ld_c(CTAB(DevTile, TA.tiles) + ti);
    // Only the level sums the tile's rows are copied from need a column: the dense slots [eta_A x Natom][chi x NlevTot]
    // [U x NlevTot] map to the tile's compact set (at most preCols columns, sized on the host), everything else to one
    // trash column.  LDS per thread drops from 26 to <= 11 doubles for H + Ca II: 3 -> 7 workgroups per CU.
    // (the map is a table of the host's, one row per tile)
    int* remap = (int*)(lds + (size_t)(TA.preCols + 1) * RM_PRE_T);
    const int nLev = TA.Natom + 2 * TA.NlevTot;
    const CONST_AS int32_t* rm = CTAB(int32_t, TA.tileRemap) + (size_t)ti * (nLev + 1);
    for (int q = threadIdx.x; q < nLev; q += RM_PRE_T)
        remap[q] = ((const GLOBAL_AS int32_t*)TA.tileRemap)[(size_t)ti * (nLev + 1) + q];
    const int nCols = rm[nLev];
    __syncthreads();
    const int p = blockIdx.y * RM_PRE_T + threadIdx.x;
    if (p >= TA.Ns * TA.L)
        return;
    const DevLaHeader h = ld_c(CTAB(DevLaHeader, TA.laHdr) + td.la0);
    const LevCols lc{ lds + threadIdx.x, RM_PRE_T, TA.Natom, TA.NlevTot, remap };
    pre_point(ap, td, h, lc, nCols, p);
}

// ---- cross-lane helpers -------------------------------------------------------------------------------------------
template <int CTRL> DEVINL double dpp_all64(double v) // every lane has a source (quad_perm, row_ror)
{
    const unsigned long long vv = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)vv, (int)(unsigned)vv, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(vv >> 32), (int)(unsigned)(vv >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// Sum four per-lane quantities over the 16 lanes of each DPP row; afterwards lane l holds the row total of
// quantity l & 3.  A transpose-reduce: every level halves the number of live quantities (27 VALU, 5 of them fp64).
DEVINL double row_reduce4(double v0, double v1, double v2, double v3, int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2;
    const double a = (b0 ? v1 : v0) + dpp_all64<0xB1>(b0 ? v0 : v1); // quad_perm [1,0,3,2]
    const double b = (b0 ? v3 : v2) + dpp_all64<0xB1>(b0 ? v2 : v3);
    double c = (b1 ? b : a) + dpp_all64<0x4E>(b1 ? a : b);           // quad_perm [2,3,0,1]
    c += dpp_all64<0x124>(c);                                         // row_ror:4
    c += dpp_all64<0x128>(c);                                         // row_ror:8
    return c;
}

// ---- the march ------------------------------------------------------------------------------------------------------
enum { RM_MAX_WAVES = 8 };
enum { PH_0 = 0, PH_1 = 1, PH_2 = 2, PH_MAIN = 3, PH_N = 4, PH_L = 5 };
enum { RM_WIN = 8, RM_WIN_PAD = 9, RM_WIN_Q = 16 * RM_WIN_PAD, RM_WIN_DOUBLES = 5 * RM_WIN_Q }; // moment window of a wave: [J, M1, Meta0, Meta1, unused][16][8 (+1 pad)]

typedef unsigned int rm_u2 __attribute__((ext_vector_type(2)));
// buffer access: resource in SGPRs + per-lane byte offset + scalar byte offset: no address arithmetic per load
DEVINL double buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
DEVINL void buf_st(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(rm_u2, v), r, (int)voff, (int)soff, 0);
}
// 1 / x from v_rcp_f64 and one Newton step (relative error ~1e-14 at worst: the seed is good to ~2^-26)
DEVINL double rm_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
// exp(x) for -700 < x <= 0 as d_exp_neg, the degree-13 polynomial split into its even and odd part (two independent
// Horner chains in r^2: half the dependent latency)
// (the constants are gathered in one struct; pinning them in VGPRs for the whole march was measured: 4 % fewer
// instructions, but 256 VGPRs and a spill)
struct ExpK
{
    double l2e, ln2h, ln2l, c12, c13, c10, c11, c8, c9, c6, c7, c4, c5, c3;
    DEVINL void init()
    {
        l2e = 1.4426950408889634074;
        ln2h = -6.93147180369123816490e-01;
        ln2l = -1.90821492927058770002e-10;
        c12 = 1.0 / 479001600.0;
        c13 = 1.0 / 6227020800.0;
        c10 = 1.0 / 3628800.0;
        c11 = 1.0 / 39916800.0;
        c8 = 1.0 / 40320.0;
        c9 = 1.0 / 362880.0;
        c6 = 1.0 / 720.0;
        c7 = 1.0 / 5040.0;
        c4 = 1.0 / 24.0;
        c5 = 1.0 / 120.0;
        c3 = 1.0 / 6.0;
    }
};
DEVINL double rm_exp_neg(double x, const ExpK& K)
{
    const double n = rint(x * K.l2e);
    double r = fma(n, K.ln2h, x);
    r = fma(n, K.ln2l, r);
    const double r2 = r * r;
    double pe = K.c12, po = K.c13;
    pe = fma(pe, r2, K.c10);
    po = fma(po, r2, K.c11);
    pe = fma(pe, r2, K.c8);
    po = fma(po, r2, K.c9);
    pe = fma(pe, r2, K.c6);
    po = fma(po, r2, K.c7);
    pe = fma(pe, r2, K.c4);
    po = fma(po, r2, K.c5);
    pe = fma(pe, r2, 0.5);
    po = fma(po, r2, K.c3);
    pe = fma(pe, r2, 1.0);
    po = fma(po, r2, 1.0);
    return ldexp(fma(po, r, pe), (int)n);
}
#define RM_KEEP(v) asm volatile("" : "+v"(v)) // the value is computed here, on every lane (keeps selects branch-free)

// What one iteration reads from memory; loaded one iteration ahead.
template <int NL> struct StepIn
{
    // gather of point t
    double chi, eta, sca;
    double phi[NL > 0 ? NL : 1], rho[NL > 0 ? NL : 1]; // per lane
    double ni[NL > 0 ? NL : 1], nj[NL > 0 ? NL : 1];   // uniform
    double dh, rdh, rsum;                              // uniform: interval (t-1 -> t); point t-1
};

// One task = one tile in one direction.  NL / NM: lines / mixed continua of the tile at compile time, or (GENERIC)
// their capacities with the counts taken from the tile.
// SEG (round 6, deep columns: TileArgs::depthSplit): the wavefront marches only the points m in [mA, mB) of the march order
// (m = 0: the boundary the ray starts from).  A segment that does not start at the boundary gathers from point mA - 3 on --
// five warm-up iterations after which every carried quantity (slopes, Steffen derivatives, optical depths) derives from real
// data exactly as in the unsplit march -- and takes the intensity behind point mA - 1 from the caller (Iin).  PASS1 (with
// RATES = false): nothing is stored; the segment's affine map I(mB - 1) = Eout * I(mA - 1) + Iout is formed instead, so that
// the caller can hand every later segment its starting intensity before any of them has run its real pass.
template <int SOLVER, int NL, int NM, bool RATES, bool GENERIC, bool SEG = false, bool PASS1 = false>
DEVINL void march_tile(const CONST_AS TileArgs& a, const TileDyn& dyn, const DevTile& td, const int toObs, double* sAcc,
                       double* win, const int lane, const int mA = 0, const int mB = 0, const double Iin = 0.0, double* Iout = nullptr,
                       double* Eout = nullptr)
{
    static_assert(!PASS1 || (SEG && !RATES), "the first pass of a segment stores nothing");
    constexpr int NLc = NL > 0 ? NL : 1, NMc = NM > 0 ? NM : 1;
    const int Ns = a.Ns, L = a.L, Nmu = a.Nmu;
    const int nLine = GENERIC ? (int)td.nLine : NL;
    const int nMixed = GENERIC ? (int)td.nMixed : NM;
#define RM_HASL(g, s) ((s) < NL && (!GENERIC || (s) < nLine))
#define RM_HASM(g, c) ((c) < NM && (!GENERIC || (c) < nMixed))
    const int muRaw = lane / L;
    const int jRaw = lane - muRaw * L;
    const bool active = (muRaw < Nmu) && (jRaw < td.nl);
    const int mu = min(muRaw, Nmu - 1);
    const int j = min(jRaw, td.nl - 1);
    const int la = td.la0 + j;
    const unsigned jb = (unsigned)j * 8u;                        // this lane's byte offset in a row of L doubles
    const unsigned srcB = (unsigned)(mu * L + j) * 8u;           // ... in a profile row of 64 doubles
    const unsigned L8 = (unsigned)L * 8u;
    const double muz = GCD(a.muz)[mu];
    const double zmu = 1.0 / muz;
    const double hw = active ? 0.5 * GCD(a.wmu)[mu] : 0.0;
    const double wav = GCD(a.wavelength)[la];
    const int dk = toObs ? -1 : 1;
    const int kS = toObs ? Ns - 1 : 0;

    // ---- slot structure (uniform: SGPRs) and per-lane parameters ------------------------------------------------
    const CONST_AS DevTileSlot* ts = CTAB(DevTileSlot, a.tslots) + td.slotOff;
    const int slotOffLa = ((const GLOBAL_AS DevLaHeader*)a.laHdr)[la].off;
    unsigned lnI[NLc], lnJ[NLc], lwO[NLc], lrow[NLc], lacc[NLc], laux[NLc], lphi[NLc];
    int lgi[NLc], lgj[NLc], lat[NLc], lfl[NLc];
    unsigned lmq[NLc];                                // window quantity of the line's Psi* eta moment (Meta0 / Meta1 / unused)
    double la0[NLc], la2[NLc], la3[NLc], lcw[NLc], lcwOp[NLc]; // per lane
    long long phi0 = 0;
#pragma unroll
    for (int s = 0; s < NLc; ++s)
    {
        lnI[s] = lnJ[s] = lwO[s] = lrow[s] = lacc[s] = laux[s] = lphi[s] = 0;
        lgi[s] = lgj[s] = lat[s] = -1 - s;
        lfl[s] = 0;
        lmq[s] = 4 * RM_WIN_Q;
        la0[s] = la2[s] = la3[s] = lcw[s] = lcwOp[s] = 0.0;
        if (RM_HASL(1, s))
        {
            const DevTileSlot u = ld_c(ts + s);
            if (s == 0)
                phi0 = u.phiOff;
            lnI[s] = (unsigned)(4 + u.gi) * 8u;
            lnJ[s] = (unsigned)(4 + u.gj) * 8u;
            lwO[s] = (unsigned)(4 + a.NlevTot + u.row) * 8u;
            lrow[s] = (unsigned)u.r0 * (RM_ROW * 8u);
            laux[s] = (unsigned)max((int)u.rAux, 0) * (RM_ROW * 8u);
            lacc[s] = (unsigned)(u.cs * 4 * Ns);
            lphi[s] = (unsigned)((u.phiOff - phi0) * 8) + (unsigned)(toObs * Ns) * 512u;
            lgi[s] = u.gi;
            lgj[s] = u.gj;
            lat[s] = u.atom;
            lfl[s] = u.flags;
            lmq[s] = (unsigned)(((u.flags & 1) && (u.ms == 0 || u.ms == 1)) ? 2 + u.ms : 4) * RM_WIN_Q;
            const GLOBAL_AS DevSlot* sl = (const GLOBAL_AS DevSlot*)a.slots + slotOffLa + u.e;
            la0[s] = sl->p[0];
            la2[s] = sl->p[2];
            la3[s] = sl->p[3];
            lcw[s] = (sl->p[1] * PI4_HC) * hw;
            lcwOp[s] = (u.flags & 1) ? lcw[s] : 0.0; // detailed atoms: rates only
        }
    }
    unsigned mrow[NMc], maux[NMc], macc[NMc];
    int mgi[NMc], mgj[NMc], mat[NMc];
    double mb0[NMc], mb2[NMc], mcw[NMc];
#pragma unroll
    for (int c = 0; c < NMc; ++c)
    {
        mrow[c] = maux[c] = macc[c] = 0;
        mgi[c] = mgj[c] = mat[c] = -100 - c;
        mb0[c] = mb2[c] = mcw[c] = 0.0;
        if (RM_HASM(1, c))
        {
            const DevTileSlot u = ld_c(ts + nLine + c);
            mrow[c] = (unsigned)u.r0 * (RM_ROW * 8u);
            maux[c] = (unsigned)max((int)u.rAux, 0) * (RM_ROW * 8u);
            macc[c] = (unsigned)(u.cs * 4 * Ns);
            mgi[c] = u.gi;
            mgj[c] = u.gj;
            mat[c] = u.atom;
            const GLOBAL_AS DevSlot* sl = (const GLOBAL_AS DevSlot*)a.slots + slotOffLa + u.e;
            mb0[c] = sl->p[0];
            mb2[c] = sl->p[2];
            mcw[c] = sl->p[1] * hw;
        }
    }
    // Which active lines feed which slot's level sums: one 5-bit code per (slot, line), evaluated once per tile.
    // bit 0: same atom (eta_A += eta_line)   bit 1: line.i == slot.i (chi_i += x)   bit 2: line.j == slot.i (chi_i -= x, U_i += u)
    // bit 3: line.i == slot.j (chi_j += x)   bit 4: line.j == slot.j (chi_j -= x, U_j += u)
    auto couple_code = [&](int l, int at, int gi, int gj) -> int {
        if (!(lfl[l] & 1))
            return 0;
        return (lat[l] == at ? 1 : 0) | (lgi[l] == gi ? 2 : 0) | (lgj[l] == gi ? 4 : 0) | (lgi[l] == gj ? 8 : 0)
               | (lgj[l] == gj ? 16 : 0);
    };
    int ccL[NLc][NLc], ccM[NMc][NLc];
#pragma unroll
    for (int s = 0; s < NLc; ++s)
#pragma unroll
        for (int l = 0; l < NLc; ++l)
        {
            ccL[s][l] = __builtin_amdgcn_readfirstlane((l != s && RM_HASL(1, s) && RM_HASL(1, l)) ? couple_code(l, lat[s], lgi[s], lgj[s]) : 0);
        }
#pragma unroll
    for (int c = 0; c < NMc; ++c)
#pragma unroll
        for (int l = 0; l < NLc; ++l)
        {
            ccM[c][l] = __builtin_amdgcn_readfirstlane((RM_HASM(1, c) && RM_HASL(1, l)) ? couple_code(l, mat[c], mgi[c], mgj[c]) : 0);
        }
    // boundary intensity of the ray (wrappers FormalScalar.cpp:471-666)
    double B0 = 0.0, B1 = 0.0;
    const int bcType = toObs ? a.lowerBc : a.upperBc;
    if (bcType == LWHIP_BC_THERMALISED)
    {
        const CONST_AS double* T = CTAB(double, a.temperature);
        B0 = d_planck(toObs ? T[Ns - 2] : T[0], wav);
        B1 = d_planck(toObs ? T[Ns - 1] : T[1], wav);
    }
    auto boundary_I = [&](double dtau_b) -> double {
        double Ib = 0.0;
        if (bcType == LWHIP_BC_THERMALISED)
            Ib = toObs ? B1 - (B0 - B1) / dtau_b : B0 - (B1 - B0) / dtau_b;
        else if (bcType == LWHIP_BC_CALLABLE)
        {
            const int mi = toObs ? CTAB(int32_t, a.lowerIdx)[mu * 2 + 1] : CTAB(int32_t, a.upperIdx)[mu * 2 + 0];
            if (mi >= 0)
                Ib = toObs ? GCD(a.lowerBcData)[(size_t)la * a.lowerNmu + mi] : GCD(a.upperBcData)[(size_t)la * a.upperNmu + mi];
        }
        return Ib;
    };

    // buffer resources: the tile's row block (depth-major), its profile blocks, the moment block of this direction
    const unsigned strideK = (unsigned)td.nRowsT * (RM_ROW * 8u);
    const __amdgpu_buffer_rsrc_t rRows
        = __builtin_amdgcn_make_buffer_rsrc((void*)(a.rows + td.rowBase), 0, (int)(strideK * (unsigned)Ns), 0x00020000);
    const __amdgpu_buffer_rsrc_t rPhi
        = __builtin_amdgcn_make_buffer_rsrc((void*)(a.phiT + phi0), 0, (int)((unsigned)max(nLine, 1) * 2u * (unsigned)Ns * 512u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rMom = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.mom + td.momBase + (size_t)toObs * 4 * L * Ns), 0, (int)(4u * L8 * (unsigned)Ns), 0x00020000);
    const CONST_AS char* kt0 = (const CONST_AS char*)a.kt; // entry -1 (padding) first: every offset below is unsigned
    const unsigned ktS = (unsigned)a.ktStride * 8u;
    const bool storeDepth = RATES && a.storeDepth && !dyn.prdOnly;
    ExpK expK;
    expK.init();
    const double lamf = dyn.lambdaIterate ? 0.0 : 1.0;
    auto kOf = [&](int t) { return kS + dk * t; };
    // one 64-bit base + one zero-extended 32-bit offset per scalar load (no 64-bit address arithmetic per depth)
#ifdef RM_DBG_HOTS
    auto ktAt = [&](int k, unsigned off) { return *(const CONST_AS double*)(kt0 + (size_t)((unsigned)((k & 1) + 1) * ktS + off)); }; // timing experiment
#else
    auto ktAt = [&](int k, unsigned off) { return *(const CONST_AS double*)(kt0 + (size_t)((unsigned)(k + 1) * ktS + off)); };
#endif
    auto dhAt = [&](int ka, int kb) { return ktAt(min(ka, kb), 0); };

    // ---- loads of iteration t: gather of point t, rate scalars of point t - 2 --------------------------------------------
    auto load_step = [&](int t, StepIn<NL>& g) {
#ifdef RM_DBG_HOTV
        const int k = 0; // timing experiment: every step reads depth 0 (cache-hot; wrong results)
#else
        const int k = kOf(t);
#endif
        const unsigned so = (unsigned)k * strideK;
        g.chi = buf_ld(rRows, jb, so);
        g.eta = buf_ld(rRows, jb + RM_ROW * 8u, so);
        g.sca = buf_ld(rRows, jb + 2 * RM_ROW * 8u, so);
#pragma unroll
        for (int s = 0; s < NLc; ++s)
        {
            g.phi[s] = 0.0;
            g.rho[s] = 1.0;
            g.ni[s] = g.nj[s] = 0.0;
            if (RM_HASL(2, s))
            {
                g.phi[s] = buf_ld(rPhi, srcB, lphi[s] + (unsigned)k * 512u);
                if (lfl[s] & 2)
                    g.rho[s] = buf_ld(rRows, jb, so + laux[s]);
                g.ni[s] = ktAt(k, lnI[s]);
                g.nj[s] = ktAt(k, lnJ[s]);
            }
        }
        const int kd = min(k, k - dk);
        g.dh = ktAt(kd, 0);
        g.rdh = ktAt(kd, 8);
        g.rsum = ktAt(k - dk, 16);
    };

    // ---- march state ----
    double c0 = 1.0, c1 = 1.0, c2 = 1.0, c3 = 1.0;  // chi(t), chi(t-1), chi(t-2), chi(t-3)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // S(t) .. S(t-3)
    double r0 = 1.0, r1 = 1.0, r2 = 1.0;            // 1 / chi
    double ds0 = 1.0, ds1 = 1.0;                    // path length of interval t, t-1
    double sl0 = 0.0, sl1 = 0.0;                    // chi slope of interval t, t-1
    double dx1 = 0.0, dx2 = 0.0;                    // d chi / d s at point t-1, t-2
    double dt1 = 1.0, dt2 = 1.0, rdt1 = 1.0, rdt2 = 1.0; // optical depth of interval t-1, t-2
    double sS1 = 0.0, sS2 = 0.0;                    // S slope of interval t-1, t-2 (per unit tau)
    double dS2 = 0.0, dS3 = 0.0;                    // dS/dtau at point t-2, t-3
    double dtL = 1.0, dSL = 0.0;                    // linear end point
    double I = 0.0;
    bool own = !SEG || mA == 0;                     // SEG: the point being finished belongs to this segment
    double Etot = 1.0;                              // PASS1: product of the steps' attenuations over the segment
    double vijQ[3][NLc], vjiQ[3][NLc];
#pragma unroll
    for (int s = 0; s < NLc; ++s)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            vijQ[q][s] = vjiQ[q][s] = 0.0;

    // ---- J and the Psi* moments (accumulate_J :181-190) -----------------------------------------------------------------
    // Every ray adds w I, w Psi*, w Psi* eta_line into the wave's LDS window [quantity][wavelength j][step mod 8] (the mu
    // lanes of a wavelength meet at one address); every 8 steps lane (q, j) moves its 8 sums to the moment block
    // [q][j][Ns] of this direction.  The post-pass adds the two directions in a fixed order.
    double* winLane = win + (jRaw * RM_WIN_PAD);
    const int qF = lane / L, jF = lane - qF * L;
    const bool flushLane = RATES && qF < 4 && jF < td.nl;
    double* winF = win + (qF & 3) * RM_WIN_Q + jF * RM_WIN_PAD;
    const unsigned momV = (unsigned)((qF & 3) * L + jF) * (unsigned)Ns * 8u;
    if (RATES)
    {
        for (int idx = lane; idx < RM_WIN_DOUBLES; idx += 64)
            win[idx] = 0.0;
    }
    auto flush_window = [&](int m) { // points m - (m & 7) .. m
        if (flushLane)
        {
            const int m0 = m & ~(RM_WIN - 1);
            double v[RM_WIN];
#pragma unroll
            for (int w = 0; w < RM_WIN; ++w)
            {
                v[w] = winF[w];
                winF[w] = 0.0;
            }
#pragma unroll
            for (int w = 0; w < RM_WIN; ++w)
                if (m0 + w <= m)
                    buf_st(rMom, momV, (unsigned)kOf(m0 + w) * 8u, v[w]);
        }
    };

    // ---- one iteration: gather point t (inputs `in`), finish point m = t - 2; the next iteration's loads go to `nx` ----
    auto step = [&](auto phTag, const int t, const StepIn<NL>& in, StepIn<NL>& nx) {
        constexpr int PH = decltype(phTag)::value;
        constexpr bool GATHER = PH <= PH_MAIN;
        constexpr bool DO_RATES = PH >= PH_2;
        const int km = kOf(t - 2);
        // loads of the rate integrals of point m: issued first (used at the end of this iteration) ...
        double lr[NLc][5], mr[NMc][5], mvji[NMc];
        double nim[NLc], njm[NLc], wphim[NLc];
        if (RATES && DO_RATES)
        {
#ifdef RM_DBG_HOTV
            const unsigned so = 0u;
#else
            const unsigned so = (unsigned)km * strideK;
#endif
#pragma unroll
            for (int s = 0; s < NLc; ++s)
            {
#pragma unroll
                for (int q = 0; q < 5; ++q)
                    lr[s][q] = 0.0;
                nim[s] = njm[s] = wphim[s] = 0.0;
                if (RM_HASL(3, s))
                {
#pragma unroll
                    for (int q = 0; q < 5; ++q)
                        lr[s][q] = buf_ld(rRows, jb + (unsigned)q * (RM_ROW * 8u), so + lrow[s]);
                    nim[s] = ktAt(km, lnI[s]);
                    njm[s] = ktAt(km, lnJ[s]);
                    wphim[s] = ktAt(km, lwO[s]);
                }
            }
#pragma unroll
            for (int c = 0; c < NMc; ++c)
            {
                mvji[c] = 0.0;
#pragma unroll
                for (int q = 0; q < 5; ++q)
                    mr[c][q] = 0.0;
                if (RM_HASM(3, c))
                {
#pragma unroll
                    for (int q = 0; q < 5; ++q)
                        mr[c][q] = buf_ld(rRows, jb + (unsigned)q * (RM_ROW * 8u), so + mrow[c]);
                    mvji[c] = buf_ld(rRows, jb, so + maux[c]);
                }
            }
        }
        // ... then everything the next iteration reads (out-of-range depths read zeros / the table's padding)
        if (PH < PH_L)
            load_step(t + 1, nx);

        // ---- A. gather point t: Transition::uv + chi_eta_aux_accum + compute_source_fn ----
        if (GATHER)
        {
            double chi = in.chi, eta = in.eta;
#pragma unroll
            for (int s = 0; s < NLc; ++s)
            {
                vijQ[0][s] = 0.0;
                vjiQ[0][s] = 0.0;
                if (RM_HASL(2, s))
                {
                    const double vij = la0[s] * in.phi[s];
                    const double vji = (la2[s] * in.rho[s]) * vij;
                    const double uji = la3[s] * vji;
                    chi += in.ni[s] * vij - in.nj[s] * vji;
                    eta += in.nj[s] * uji;
                    vijQ[0][s] = vij;
                    vjiQ[0][s] = vji;
                }
            }
            c0 = chi;
            r0 = rm_rcp(chi);
            s0 = (eta + in.sca) * r0;
            if (storeDepth && active)
            {
                const size_t o = ((size_t)(la * Nmu + mu) * 2 + toObs) * Ns + kOf(t);
                GD(a.depthChi)[o] = chi;
                GD(a.depthEta)[o] = eta;
            }
        }
        double Psi = 0.0;
        if (SOLVER == LWHIP_FS_BEZIER3_1D)
        {
            // piecewise_bezier3_1d_impl FormalScalar.cpp:209-325 with the loop-carried quantities in registers:
            // interval t is (point t-1 -> point t).  This iteration: slope(t), dx(t-1), dtau(t-1), dS(t-2), I(t-2).
            if (PH >= PH_1 && GATHER)
            {
                ds0 = in.dh * zmu;
                sl0 = (c0 - c1) * (in.rdh * muz);
            }
            if (PH == PH_1)
                dx1 = sl0; // dx_uw :239
            else if (PH == PH_2 || PH == PH_MAIN)
            {
                // cent_deriv (Steffen 1990, Bezier.hpp:58-65) at point t-1: Suw = slope(t-1), S0 = slope(t)
                const double P0 = fabs((sl1 * ds0 + sl0 * ds1) * (in.rsum * muz));
                dx1 = (copysign(1.0, sl0) + copysign(1.0, sl1)) * fmin(fabs(sl1), fmin(fabs(sl0), 0.5 * P0));
            }
            else if (PH == PH_N)
                dx1 = sl1; // dx_dw of the last interval :288
            if (PH >= PH_2 && PH <= PH_N)
            {
                // optical depth of interval t-1 (:242-246, 261-263, 289-292)
                const double ds3 = ds1 * (1.0 / 3.0);
                const double Cuw = c2 + ds3 * dx2;
                const double C0 = c1 - ds3 * dx1;
                dt1 = ds1 * (c2 + c1 + Cuw + C0) * 0.25;
                rdt1 = rm_rcp(dt1);
                sS1 = (s1 - s2) * rdt1;
                if (PH == PH_2)
                    dS2 = sS1; // dS_uw :247
                else
                {
                    // dS_central at point t-2 (:254-255): cent_deriv(dtau(t-2), dtau(t-1), S(t-3), S(t-2), S(t-1))
                    const double P0 = fabs((sS2 * dt1 + sS1 * dt2) * rm_rcp(dt1 + dt2));
                    dS2 = (copysign(1.0, sS1) + copysign(1.0, sS2)) * fmin(fabs(sS2), fmin(fabs(sS1), 0.5 * P0));
                }
            }
            if (PH == PH_2)
                I = boundary_I(0.5 * zmu * (c2 + c1) * dhAt(kOf(0), kOf(1)));
            else if (PH == PH_MAIN || PH == PH_N)
            {
                // Bezier3_coeffs of dtau(t-2) (Bezier.hpp:81-127) and the update :265-275.  Both branches are evaluated on
                // every lane and selected (no divergent control flow in the steady state); beyond dtau = 30 the
                // asymptotic branch is the exact one with edt = 0.
                const double x = dt2;
                const double x2 = x * x, x3 = x2 * x;
                const bool small = x < 5e-2;
                double edt = 0.0, al = 0.0, be = 0.0, ga = 0.0, de = 0.0;
                if (__builtin_amdgcn_ballot_w64(!small) != 0) // some ray of the wave is beyond the Taylor range
                {
                    const double ex = (x > 30.0) ? 0.0 : rm_exp_neg(-fmin(x, 64.0), expK);
                    const double r3 = (rdt2 * rdt2) * rdt2; // 1 / dtau^3 from the interval's 1 / dtau
                    edt = ex;
                    al = (6.0 - ex * (6.0 + 6.0 * x + 3.0 * x2 + x3)) * r3;
                    be = (6.0 * ex - 6.0 + 6.0 * x - 3.0 * x2 + x3) * r3;
                    ga = 3.0 * (2.0 * x - 6.0 + ex * (6.0 + 4.0 * x + x2)) * r3;
                    de = 3.0 * (6.0 - 4.0 * x + x2 - 2.0 * ex * (3.0 + x)) * r3;
                    RM_KEEP(edt);
                    RM_KEEP(al);
                    RM_KEEP(be);
                    RM_KEEP(ga);
                    RM_KEEP(de);
                }
                if (__builtin_amdgcn_ballot_w64(small) != 0) // some ray is within it
                {
                    double e1 = 1.0 - x + 0.5 * x2 - x3 * (1.0 / 6.0);
                    double a1 = 0.25 * x - 0.2 * x2 + x3 * (1.0 / 12.0);
                    double b1 = 0.25 * x - 0.05 * x2 + x3 * (1.0 / 120.0);
                    double g1 = 0.25 * x - 0.15 * x2 + 0.05 * x3;
                    double d1 = 0.25 * x - 0.1 * x2 + 0.025 * x3;
                    RM_KEEP(e1);
                    RM_KEEP(a1);
                    RM_KEEP(b1);
                    RM_KEEP(g1);
                    RM_KEEP(d1);
                    edt = small ? e1 : edt;
                    al = small ? a1 : al;
                    be = small ? b1 : be;
                    ga = small ? g1 : ga;
                    de = small ? d1 : de;
                }
                const double x_3 = x * (1.0 / 3.0);
                const double Cuw = s3 + x_3 * dS3;
                const double C0 = s2 - x_3 * dS2;
                I = I * edt + al * s3 + be * s2 + ga * Cuw + de * C0;
                if (PASS1)
                    Etot *= edt;
                Psi = be + de;
            }
            if (PH == PH_N)
            {
                // the last point is piecewise linear (:306-321); its inputs are at hand now
                dtL = 0.5 * zmu * (c1 + c2) * dhAt(kOf(Ns - 1), kOf(Ns - 2));
                dSL = (s1 - s2) * rm_rcp(dtL);
            }
            if (PH == PH_L)
            {
                double w0, w1;
                d_w2(dtL, w0, w1);
                I = (1.0 - w0) * I + w0 * s2 - w1 * dSL;
                Psi = w0 - w1 * d_rcp(dtL);
            }
        }
        else if (SOLVER == LWHIP_FS_LINEAR_1D)
        {
            // piecewise_linear_1d_impl FormalScalar.cpp:136-207.  Point m = t - 2 uses the interval (m-1 -> m), formed
            // one iteration earlier from c2 = chi(m-1), c1 = chi(m) and carried as dt2, rdt2, sS2
            if (PH >= PH_2 && PH <= PH_N)
            {
                dt1 = (0.5 * zmu) * (c2 + c1) * dhAt(kOf(t - 1), kOf(t - 2));
                rdt1 = d_rcp(dt1);
                sS1 = (s2 - s1) * rdt1; // dS_uw = (S(m-1) - S(m)) / dtau
            }
            if (PH == PH_2)
                I = boundary_I(dt1);
            else if (PH >= PH_MAIN)
            {
                double w0, w1;
                d_w2(dt2, w0, w1);
                I = (1.0 - w0) * I + w0 * s2 + w1 * sS2;
                if (PASS1)
                    Etot *= 1.0 - w0;
                Psi = w0 - w1 * rdt2;
            }
        }
        else
        {
            // piecewise_besser_1d_impl FormalScalar.cpp:395-467: point m = t - 2 needs chi, S of m-1, m, m+1
            // = (c3, c2, c1) and (s3, s2, s1)
            if (PH == PH_2)
                I = boundary_I(0.5 * zmu * (c2 + c1) * dhAt(kOf(0), kOf(1)));
            else if (PH == PH_MAIN || PH == PH_N)
            {
                const double ds_uw = dhAt(km, kOf(t - 3)) * zmu;
                const double ds_dw = dhAt(km, kOf(t - 1)) * zmu;
                const double chiC = d_besser_control_point(ds_uw, ds_dw, c3, c2, c1);
                const double dtauUw = (1.0 / 3.0) * (c3 + chiC + c2) * ds_uw;
                const double dtauDw = 0.5 * (c2 + c1) * ds_dw;
                const double SC = d_besser_control_point(dtauUw, dtauDw, s3, s2, s1);
                double cM, cO, cC, edt;
                d_besser_coeffs(dtauUw, cM, cO, cC, edt);
                I = I * edt + cM * s3 + cO * s2 + cC * SC;
                if (PASS1)
                    Etot *= edt;
                Psi = cO + cC;
            }
            else if (PH == PH_L)
            {
                // last point: linear (:449-466)
                const double dtau = 0.5 * zmu * (c2 + c3) * dhAt(kOf(Ns - 1), kOf(Ns - 2));
                const double rcp = d_rcp(dtau);
                const double dSuw = (s2 - s3) * rcp;
                double w0, w1;
                d_w2(dtau, w0, w1);
                I = (1.0 - w0) * I + w0 * s2 - w1 * dSuw;
                Psi = w0 - w1 * rcp;
            }
        }

        // ---- F. outputs and rate integrals of point m = t - 2 -----------------------------------------
        if (DO_RATES && !PASS1 && (!SEG || own))
        {
            const double PsiStar = Psi * r2; // PsiStar :322-323
            if (active)
            {
                if (toObs && PH == PH_L)
                    GD(a.I)[(size_t)la * Nmu + mu] = I; // spect.I(la, mu, 0) :349
                if (storeDepth)
                    GD(a.depthI)[((size_t)(la * Nmu + mu) * 2 + toObs) * Ns + km] = I;
                if (PH == PH_N)
                {
                    // ZPlaneDecomposition :351-361: the second-to-last point of the march is depth 1 of an up ray,
                    // depth Nz - 2 of a down ray
                    double* zp = toObs ? a.zUp : a.zDown;
                    if (zp)
                        GD(zp)[(size_t)la * Nmu + mu] = I;
                }
            }
            if (RATES)
            {
                const double PsiS = PsiStar * lamf;
                const double wPsi = hw * PsiS;
                // the lines' opacities at point m from the queued V_ij, V_ji and the populations of depth km
                double xl[NLc], el[NLc], uj[NLc];
#pragma unroll
                for (int s = 0; s < NLc; ++s)
                {
                    xl[s] = el[s] = uj[s] = 0.0;
                    if (RM_HASL(5, s))
                    {
                        uj[s] = la3[s] * vjiQ[2][s];
                        xl[s] = nim[s] * vijQ[2][s] - njm[s] * vjiQ[2][s];
                        el[s] = njm[s] * uj[s];
                    }
                }
                // level sums of a slot: continuum rows + the active lines of this wavelength, selected by the tile's
                // coupling codes (a line's coupling with itself is known at compile time)
                auto couple = [&](const int (&cc)[NLc], double& etaA, double& chi_i, double& chi_j, double& U_i, double& U_j) {
#pragma unroll
                    for (int l = 0; l < NLc; ++l)
                    {
                        if (l < NL)
                        {
                            const int c5 = cc[l];
                            etaA += (c5 & 1) ? el[l] : 0.0;
                            chi_i += (c5 & 2) ? xl[l] : 0.0;
                            chi_i -= (c5 & 4) ? xl[l] : 0.0;
                            chi_j += (c5 & 8) ? xl[l] : 0.0;
                            chi_j -= (c5 & 16) ? xl[l] : 0.0;
                            U_j += (c5 & 16) ? uj[l] : 0.0;
                            U_i += (c5 & 4) ? uj[l] : 0.0;
                        }
                    }
                };
#pragma unroll
                for (int s = 0; s < NLc; ++s)
                {
                    if (RM_HASL(7, s))
                    {
                        const double vij = vijQ[2][s], vji = vjiQ[2][s], uji = uj[s];
                        const double wlamu = lcw[s] * wphim[s];
                        double etaA = lr[s][0] + el[s], chi_i = lr[s][1] + xl[s], chi_j = lr[s][2] - xl[s];
                        double U_i = lr[s][3], U_j = lr[s][4] + uj[s];
                        if (NL > 1)
                            couple(ccL[s], etaA, chi_i, chi_j, U_i, U_j);
                        const double Ieff = I - PsiS * etaA; // compute_full_Ieff
                        const double wop = lcwOp[s] * wphim[s];
                        const double g0v = ((uji + vji * Ieff) - (PsiS * chi_i * U_j)) * wop;
                        const double g1v = ((vij * Ieff) - (PsiS * chi_j * U_i)) * wop;
                        const double r2v = I * vij * wlamu;
                        const double r3v = (uji + I * vji) * wlamu;
                        const double tot = row_reduce4(g0v, g1v, r2v, r3v, lane);
#ifdef RM_DBG_NOACC
                        asm volatile("" ::"v"(tot)); // timing experiment: no accumulator atomics (wrong results)
#else
                        if ((lane & 12) == 0)
                            lds_add(sAcc + lacc[s] + (unsigned)((lane & 3) * Ns + km), tot);
#endif
                    }
                }
#pragma unroll
                for (int c = 0; c < NMc; ++c)
                {
                    if (RM_HASM(8, c)) // mixed continua (always of active atoms)
                    {
                        const double vij = mb0[c], vji = mvji[c], uji = mb2[c] * vji;
                        const double wlamu = mcw[c];
                        double etaA = mr[c][0], chi_i = mr[c][1], chi_j = mr[c][2], U_i = mr[c][3], U_j = mr[c][4];
                        if (NL > 0)
                            couple(ccM[c], etaA, chi_i, chi_j, U_i, U_j);
                        const double Ieff = I - PsiS * etaA;
                        const double g0v = ((uji + vji * Ieff) - (PsiS * chi_i * U_j)) * wlamu;
                        const double g1v = ((vij * Ieff) - (PsiS * chi_j * U_i)) * wlamu;
                        const double r2v = I * vij * wlamu;
                        const double r3v = (uji + I * vji) * wlamu;
                        const double tot = row_reduce4(g0v, g1v, r2v, r3v, lane);
#ifdef RM_DBG_NOACC
                        asm volatile("" ::"v"(tot)); // timing experiment: no accumulator atomics (wrong results)
#else
                        if ((lane & 12) == 0)
                            lds_add(sAcc + macc[c] + (unsigned)((lane & 3) * Ns + km), tot);
#endif
                    }
                }
                // moments of the step into the window; a full window (or the last point) leaves for the moment block
                const int m = t - 2;
                if (active)
                {
                    double* wp = winLane + (m & (RM_WIN - 1));
#ifndef RM_DBG_NOWIN
                    lds_add(wp, hw * I);
                    lds_add(wp + RM_WIN_Q, wPsi);
#pragma unroll
                    for (int s = 0; s < NLc; ++s)
                        if (RM_HASL(7, s))
                            lds_add(wp + lmq[s], wPsi * el[s]); // Meta of the line's atom (or the unused row)
#else
                    double kp = hw * I + wPsi; // timing experiment: no window atomics (wrong results)
#pragma unroll
                    for (int s = 0; s < NLc; ++s)
                        kp += wPsi * el[s];
                    asm volatile("" ::"v"(kp));
#endif
                }
                if ((m & (RM_WIN - 1)) == RM_WIN - 1 || PH == PH_L)
                    flush_window(m);
            }
        }

        // ---- shift the windows ----
        c3 = c2;
        c2 = c1;
        c1 = c0;
        s3 = s2;
        s2 = s1;
        s1 = s0;
        r2 = r1;
        r1 = r0;
        if (SOLVER == LWHIP_FS_BEZIER3_1D)
        {
            ds1 = ds0;
            sl1 = sl0;
            dx2 = dx1;
            dt2 = dt1;
            rdt2 = rdt1;
            sS2 = sS1;
            dS3 = dS2;
        }
        else if (SOLVER == LWHIP_FS_LINEAR_1D)
        {
            dt2 = dt1;
            rdt2 = rdt1;
            sS2 = sS1;
        }
#pragma unroll
        for (int s = 0; s < NLc; ++s)
        {
            vijQ[2][s] = vijQ[1][s];
            vijQ[1][s] = vijQ[0][s];
            vjiQ[2][s] = vjiQ[1][s];
            vjiQ[1][s] = vjiQ[0][s];
        }
    };

    StepIn<NL> gA, gB;
    int t = 3;
    if (!SEG || mA == 0)
    {
        load_step(0, gA);
        gB = gA;
        step(std::integral_constant<int, PH_0>{}, 0, gA, gB);
        step(std::integral_constant<int, PH_1>{}, 1, gB, gA);
        step(std::integral_constant<int, PH_2>{}, 2, gA, gB);
    }
    else
    {
        t = mA - 3; // (>= 3: segments are at least 64 points long)
        load_step(t, gB);
    }
    const bool lastSeg = !SEG || mB >= Ns;
    const int tStop = lastSeg ? Ns : mB + 2; // main iterations t < tStop: the one at t finishes point t - 2
    // the iteration that finishes the segment's first point starts from the intensity behind it
    auto enter = [&](int tt) {
        if (SEG && mA > 0 && tt == mA + 2)
        {
            own = true;
            I = PASS1 ? 0.0 : Iin;
            Etot = 1.0;
        }
    };
    for (; t + 1 < tStop; t += 2) // ping-pong input buffers: no copies in the steady state
    {
        enter(t);
        step(std::integral_constant<int, PH_MAIN>{}, t, gB, gA);
        enter(t + 1);
        step(std::integral_constant<int, PH_MAIN>{}, t + 1, gA, gB);
    }
    if (t < tStop)
    {
        enter(t);
        step(std::integral_constant<int, PH_MAIN>{}, t, gB, gA);
        gB = gA;
    }
    if (lastSeg)
    {
        step(std::integral_constant<int, PH_N>{}, Ns, gB, gA);
        step(std::integral_constant<int, PH_L>{}, Ns + 1, gA, gB);
    }
    if (SEG)
    {
        if (Iout)
            *Iout = I;
        if (Eout)
            *Eout = Etot;
    }
}

// ---- post-pass: pure continua from the angle moments, J out, dJ ------------------------------------------------------
// sum_r w_r [Uji + Vji (I_r - Psi_r eta_r) - Psi_r chi_i U_j] with chi, U, Uji, Vji ray-independent and
// eta_r = etaC + etaLine_r  ->  Uji W + Vji (J - M1 etaC - Meta) - M1 chi_i U_j: an exact regrouping of
// compute_full_operator_rates :206-234 for continua whose level sums do not depend on the ray.
//
// post_tile: lane <-> depth (T threads cover depths k0 .. k0 + T - 1), the tile is walked wavelength by wavelength.
// What depends on depth only (nStar ratios, populations, 1 / T) is fetched once per tile and stays in registers for
// its wavelengths; what depends on the wavelength only is the same for every lane (broadcast loads, requested one
// wavelength ahead); the integrals of the tile's pure continua accumulate in registers (a lane owns its depth) and
// leave with one atomic add per (continuum, component, depth) into the stage-1 buffer.  Called by the sweep's
// workgroup right after its two wavefronts have marched the tile in both directions (the moments are still in L2),
// or by the stand-alone kernel below.
struct PostLa // this lane's moments of one wavelength (both directions), its J of the last iteration
{
    double Jd, Ju, M1d, M1u, E0d, E0u, E1d, E1u, Jdag;
};
struct PostP { double p0, p1, p2, p3; }; // per-(continuum, wavelength) parameters: one scalar load
template <int T>
DEVINL void post_tile(const TileArgs* __restrict__ ap, const DevTile& td, const bool doCont, const int part, double* levCols,
                      unsigned long long* sDj, const int t)
{
    const int Ns = TA.Ns, L = TA.L, Nmu = TA.Nmu;
    const LevCols lc{ levCols + t, T, TA.Natom, TA.NlevTot };
    const int nLev = lc.Natom + 2 * lc.NlevTot;
    double Wsum = 0.0; // total quadrature weight of the rays of the iteration, in ray order
    for (int m = 0; m < Nmu; ++m)
        for (int d = 0; d < 2; ++d)
            Wsum += 0.5 * CTAB(double, TA.wmu)[m];
    const DevLaHeader h = ld_c(CTAB(DevLaHeader, TA.laHdr) + td.la0); // the tile's (uniform) structure
    const int nCont = (doCont && h.nPure > 0) ? h.nMixed + h.nPure : 0;
    const CONST_AS DevContRec* cr = CTAB(DevContRec, TA.contRec) + h.off + h.nLine;
    const GLOBAL_AS double* momT = GCD(TA.mom) + td.momBase;
    const size_t ds = (size_t)Ns * 4 * L, qs = (size_t)L * Ns;
    GLOBAL_AS double* r8 = GD(TA.red8);
    const GLOBAL_AS char* nB = (const GLOBAL_AS char*)TA.n;
    const GLOBAL_AS char* rB = (const GLOBAL_AS char*)TA.ratio;
    const CONST_AS char* slB = (const CONST_AS char*)TA.slots;
    const int n4 = 4 * Ns;
    auto ldd = [](const GLOBAL_AS char* b, unsigned o) { return *(const GLOBAL_AS double*)(b + o); };
    if (t < 16)
        sDj[t] = 0ull;
    __syncthreads();
    for (int k0 = 0; k0 < Ns; k0 += T) // deep atmospheres: the depth axis in slices of the workgroup
    {
        const int k = k0 + t;
        const bool kOk = k < Ns;
        const int kc = min(k, Ns - 1);
        const unsigned kB = (unsigned)kc * 8u;
        const double rT = GCD(TA.geo)[4 * kc + 3];
        const bool single = nCont <= RM_CB;
        double acc[RM_CB][4]; // the tile's integrals of its pure continua at this depth
#pragma unroll
        for (int i = 0; i < RM_CB; ++i)
            acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.0;
        auto load_la = [&](int j, PostLa& w) {
            const GLOBAL_AS double* mp = momT + (size_t)j * Ns + kc;
            w.Jd = mp[0];
            w.Ju = mp[ds];
            w.M1d = mp[qs];
            w.M1u = mp[ds + qs];
            w.E0d = mp[2 * qs];
            w.E0u = mp[ds + 2 * qs];
            w.E1d = mp[3 * qs];
            w.E1u = mp[ds + 3 * qs];
            w.Jdag = GCD(TA.J)[(size_t)(td.la0 + j) * Ns + kc];
        };
        // populations and nStar ratios of the tile's continua at this depth: once per tile, not once per wavelength (the
        // J store of every wavelength keeps the compiler from holding on to anything it loaded through a global pointer)
        double cNi[RM_CB], cNj[RM_CB], cRat[RM_CB];
#pragma unroll
        for (int i = 0; i < RM_CB; ++i)
        {
            cNi[i] = cNj[i] = cRat[i] = 0.0;
            if (i < nCont)
            {
                const DevContRec rc = ld_c(cr + i);
                cNj[i] = ldd(nB + rc.nJ, kB);
                cNi[i] = ldd(nB + rc.nI, kB);
                cRat[i] = ldd(rB + rc.rat, kB);
            }
        }
        PostLa wA, wB;
        load_la(0, wA);
        wB = wA;
        auto one_la = [&](int j, const PostLa& w, PostLa& nx) {
            if (j + 1 < td.nl)
                load_la(j + 1, nx);
            const int la = td.la0 + j;
            // the two directions' sums (two wave tasks of the sweep), added in that order
            const double Jk = w.Jd + w.Ju;
            if (nCont > 0)
            {
                const double M1 = w.M1d + w.M1u, Me0 = w.E0d + w.E0u, Me1 = w.E1d + w.E1u;
                const double boltz = d_exp_neg(-(HC_K / CTAB(double, TA.wavelength)[la]) * rT);
                // this wavelength's slot records: the per-(continuum, wavelength) parameters are uniform -> scalar loads
                const CONST_AS char* lsB = slB + (size_t)CTAB(DevLaHeader, TA.laHdr)[la].off * sizeof(DevSlot);
                for (int q = 0; q < nLev; ++q)
                    lc.lev[(size_t)q * T] = 0.0;
                // level sums of the active atoms, continua in transition order (LDS adds without a return value: no
                // read-modify-write round trip, executed in program order)
                auto level_sums = [&](const DevContRec& rc, double njq, double niq, double rat) {
                    const PostP pp = ld_c((const CONST_AS PostP*)(lsB + rc.pOff));
                    const double vji = (rat * boltz) * pp.p0;
                    const double chi_c = niq * pp.p0 - njq * vji;
                    const double uji = pp.p2 * vji;
                    if (rc.flags & 1)
                    {
                        lds_add(lc.lev + (size_t)rc.sEta * T, njq * uji);
                        lds_add(lc.lev + (size_t)rc.sChiI * T, chi_c);
                        lds_add(lc.lev + (size_t)rc.sChiJ * T, -chi_c);
                        lds_add(lc.lev + (size_t)rc.sUj * T, uji);
                    }
                };
                if (single)
                {
#pragma unroll
                    for (int i = 0; i < RM_CB; ++i)
                        if (i < nCont)
                            level_sums(ld_c(cr + i), cNj[i], cNi[i], cRat[i]);
                }
                else
                    for (int q = 0; q < nCont; ++q)
                    {
                        const DevContRec rc = ld_c(cr + q);
                        level_sums(rc, ldd(nB + rc.nJ, kB), ldd(nB + rc.nI, kB), ldd(rB + rc.rat, kB));
                    }
                // the pure continua of the wavelength (every level sum is complete)
                if (single)
                {
#pragma unroll
                    for (int i = 0; i < RM_CB; ++i)
                    {
                        if (i < nCont)
                        {
                            const DevContRec rc = ld_c(cr + i);
                            if (rc.flags & 2)
                            {
                                const PostP pp = ld_c((const CONST_AS PostP*)(lsB + rc.pOff));
                                const double vij = pp.p0, wla = pp.p1;
                                const double vji = (cRat[i] * boltz) * vij;
                                const double uji = pp.p2 * vji;
                                if (rc.flags & 1)
                                {
                                    const double Meta = (rc.ms == 0) ? Me0 : (rc.ms == 1) ? Me1 : 0.0;
                                    const double Jeff = Jk - M1 * lc.lev[(size_t)rc.sEta * T] - Meta;
                                    acc[i][0] += ((uji * Wsum + vji * Jeff) - (M1 * lc.lev[(size_t)rc.sChiI * T] * lc.lev[(size_t)rc.sUj * T])) * wla;
                                    acc[i][1] += ((vij * Jeff) - (M1 * lc.lev[(size_t)rc.sChiJ * T] * lc.lev[(size_t)rc.sUi * T])) * wla;
                                }
                                acc[i][2] += Jk * vij * wla;
                                acc[i][3] += (uji * Wsum + Jk * vji) * wla;
                            }
                        }
                    }
                }
                else if (kOk)
                {
                    // more than 8 continua at one wavelength (rare): straight into the stage-1 buffer
                    for (int q = 0; q < nCont; ++q)
                    {
                        const DevContRec rc = ld_c(cr + q);
                        if (!(rc.flags & 2))
                            continue;
                        const PostP pp = ld_c((const CONST_AS PostP*)(lsB + rc.pOff));
                        const double vij = pp.p0, wla = pp.p1;
                        const double vji = (ldd(rB + rc.rat, kB) * boltz) * vij;
                        const double uji = pp.p2 * vji;
                        GLOBAL_AS double* dst = r8 + ((size_t)rc.tr * LWHIP_REDUCE_SPLIT + part) * n4 + k;
                        if (rc.flags & 1)
                        {
                            const double Meta = (rc.ms == 0) ? Me0 : (rc.ms == 1) ? Me1 : 0.0;
                            const double Jeff = Jk - M1 * lc.lev[(size_t)rc.sEta * T] - Meta;
                            unsafeAtomicAdd((double*)dst, ((uji * Wsum + vji * Jeff) - (M1 * lc.lev[(size_t)rc.sChiI * T] * lc.lev[(size_t)rc.sUj * T])) * wla);
                            unsafeAtomicAdd((double*)(dst + Ns), ((vij * Jeff) - (M1 * lc.lev[(size_t)rc.sChiJ * T] * lc.lev[(size_t)rc.sUi * T])) * wla);
                        }
                        unsafeAtomicAdd((double*)(dst + 2 * Ns), Jk * vij * wla);
                        unsafeAtomicAdd((double*)(dst + 3 * Ns), (uji * Wsum + Jk * vji) * wla);
                    }
                }
            }
            // J out, dJ = max_k |1 - Jdag / J|  :477-486
            double dj = 0.0;
            if (kOk)
            {
                GD(TA.J)[(size_t)la * Ns + k] = Jk;
                dj = fabs(1.0 - w.Jdag / Jk);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                dj = fmax(dj, __shfl_xor(dj, o));
            if ((t & 63) == 0) // non-negative doubles order as integers; no barrier here: it would drain the prefetch
                atomicMax(sDj + j, (unsigned long long)__double_as_longlong(dj));
        };
        for (int j = 0; j < td.nl; j += 2)
        {
            one_la(j, wA, wB);
            if (j + 1 < td.nl)
                one_la(j + 1, wB, wA);
        }
        if (single && kOk)
        {
            // the tile's integrals of this depth leave for the stage-1 buffer (part = tile % SPLIT spreads the adders)
#pragma unroll
            for (int i = 0; i < RM_CB; ++i)
                if (i < nCont)
                {
                    const DevContRec rc = ld_c(cr + i);
                    if (rc.flags & 2)
                    {
                        GLOBAL_AS double* dst = r8 + ((size_t)rc.tr * LWHIP_REDUCE_SPLIT + part) * n4 + k;
                        if (rc.flags & 1)
                        {
                            unsafeAtomicAdd((double*)dst, acc[i][0]);
                            unsafeAtomicAdd((double*)(dst + Ns), acc[i][1]);
                        }
                        unsafeAtomicAdd((double*)(dst + 2 * Ns), acc[i][2]);
                        unsafeAtomicAdd((double*)(dst + 3 * Ns), acc[i][3]);
                    }
                }
        }
        // the wavelengths' dJ: both waves' maxima are in
        __syncthreads();
        if (t < td.nl)
        {
            const double all = __longlong_as_double((long long)sDj[t]);
            GLOBAL_AS double* dJp = GD(TA.dJ) + td.la0 + t;
            *dJp = (k0 == 0) ? all : fmax(*dJp, all);
            sDj[t] = 0ull;
        }
        __syncthreads();
    }
}

// stand-alone post-pass (column batches, PRD rates pass, workgroups of the sweep that hold more than one tile)
enum { RM_POST_T = 128 };
template <bool BATCH>
__global__ void __launch_bounds__(RM_POST_T) tile_post_kernel(const TileArgs* __restrict__ ap0, const TileArgs* const* __restrict__ apList,
                                                              const TileDyn dyn, const int32_t* __restrict__ chunkTile,
                                                              const int32_t* __restrict__ tileList)
{
    dbg_poison_lds();
    extern __shared__ double lds[];
    const TileArgs* __restrict__ ap = BATCH ? apList[blockIdx.y] : ap0;
    unsigned long long* sDj = (unsigned long long*)lds; // [16]
    double* cols = lds + 16;                            // [Natom + 2 NlevTot][T]
    const int chunk = blockIdx.x;
    const int t0 = CTAB(int32_t, chunkTile)[chunk], t1 = CTAB(int32_t, chunkTile)[chunk + 1];
    const bool doCont = !dyn.prdOnly; // the PRD rates pass needs J and dJ only
    for (int li = t0; li < t1; ++li)
    {
        const int ti = tileList ? CTAB(int32_t, tileList)[li] : li;
        const DevTile td = ld_c(CTAB(DevTile, TA.tiles) + ti);
        post_tile<RM_POST_T>(ap, td, doCont, ti % LWHIP_REDUCE_SPLIT, cols, sDj, (int)threadIdx.x);
    }
}

// CAP == 0: the tiles of the launch are all of the kinds (0,0), (1,0), (1,1), (2,0) lines + mixed continua, each with
// its own straight-line march; otherwise every tile goes through the guarded generic march of capacity CAP lines + CAP
// mixed continua.
// FUSE: the workgroup is the two wavefronts (down, up) of ONE tile and finishes it with the post-pass (post_tile) while
// the tile's moments are still in L2: no separate post launch, its accumulators in the march's free registers.
// (-DRM_WAVES_PER_EU=n: the occupancy experiment of profiles/r03_occupancy.md -- the register allocator is held to
// 512 / n VGPRs; the shipped build leaves it at the 2 wavefronts per SIMD its 222 VGPRs give)
#ifdef RM_WAVES_PER_EU
#define RM_OCC __attribute__((amdgpu_waves_per_eu(RM_WAVES_PER_EU)))
#else
#define RM_OCC
#endif
template <int SOLVER, int CAP, bool RATES, bool BATCH, bool FUSE>
__global__ void __launch_bounds__(RM_MAX_WAVES * 64) RM_OCC raymarch_kernel(const TileArgs* __restrict__ ap0,
                                                                      const TileArgs* const* __restrict__ apList, const TileDyn dyn)
{
    dbg_poison_lds();
    extern __shared__ double lds[];
    const TileArgs* __restrict__ ap = BATCH ? apList[blockIdx.y] : ap0;
    const CONST_AS TileArgs& a = *(const CONST_AS TileArgs*)ap;
    const int Ns = a.Ns;
    const int maxCT = a.maxCT;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6); // uniform: everything derived from the tile stays scalar
    const int nWaves = blockDim.x >> 6;
    double* sAcc = lds;
    double* win = lds + (size_t)maxCT * 4 * Ns + (size_t)wave * RM_WIN_DOUBLES; // the wave's moment window
    const int chunk = blockIdx.x;
    const int t0 = CTAB(int32_t, a.chunkTile)[chunk], t1 = CTAB(int32_t, a.chunkTile)[chunk + 1];
    if (RATES)
    {
        for (int idx = t; idx < maxCT * 4 * Ns; idx += blockDim.x)
            sAcc[idx] = 0.0;
        __syncthreads();
    }
    const CONST_AS int32_t* tileList = CTAB(int32_t, a.tileList);
    // tasks of the chunk: (tile, direction); both directions unless the launch is up-only
    const int nDir = dyn.nPass;
    for (int task = t0 * nDir + wave; task < t1 * nDir; task += nWaves)
    {
        const int li = (nDir == 2) ? (task >> 1) : task;
        const int toObs = (nDir == 2) ? (task & 1) : 1;
        const int ti = tileList ? tileList[li] : li;
        const DevTile td = ld_c(CTAB(DevTile, a.tiles) + ti);
#ifdef RM_TIMING
        const unsigned long long tm0 = __builtin_amdgcn_s_memtime();
#endif
        if (CAP == 0)
        {
            if (td.nLine == 0)
                march_tile<SOLVER, 0, 0, RATES, false>(a, dyn, td, toObs, sAcc, win, lane);
            else if (td.nLine == 1 && td.nMixed == 0)
                march_tile<SOLVER, 1, 0, RATES, false>(a, dyn, td, toObs, sAcc, win, lane);
            else if (td.nLine == 1)
                march_tile<SOLVER, 1, 1, RATES, false>(a, dyn, td, toObs, sAcc, win, lane);
            else
                march_tile<SOLVER, 2, 0, RATES, false>(a, dyn, td, toObs, sAcc, win, lane);
        }
        else
            march_tile<SOLVER, (CAP > 0 ? CAP : 1), (CAP > 0 ? CAP : 1), RATES, true>(a, dyn, td, toObs, sAcc, win, lane);
#ifdef RM_TIMING
        // diagnostic build: shader clocks of this wavefront's march -> lsDbg[tile][kind, down, up, post]
        if (lane == 0 && a.lsDbg)
        {
            double* o = a.lsDbg + (size_t)ti * 8;
            o[0] = (double)(td.nLine * 10 + td.nMixed);
            o[1 + toObs] = (double)(__builtin_amdgcn_s_memtime() - tm0);
        }
#endif
    }
    if (RATES && FUSE)
    {
        __syncthreads(); // both directions of the tile are marched, their moment stores have left the wavefronts
        const int li = t0;
        const int ti = tileList ? tileList[li] : li;
        const DevTile td = ld_c(CTAB(DevTile, a.tiles) + ti);
        unsigned long long* sDj = (unsigned long long*)(lds + (size_t)maxCT * 4 * Ns);
#ifdef RM_TIMING
        const unsigned long long tp0 = __builtin_amdgcn_s_memtime();
#endif
        post_tile<128>(ap, td, !dyn.prdOnly, ti % LWHIP_REDUCE_SPLIT, (double*)(sDj + 16), sDj, t);
#ifdef RM_TIMING
        if (t == 0 && a.lsDbg)
            a.lsDbg[(size_t)ti * 8 + 3] = (double)(__builtin_amdgcn_s_memtime() - tp0);
#endif
    }
    if (RATES)
    {
        __syncthreads();
        // the chunk's accumulators go straight into the stage-1 buffer: part = chunk % SPLIT keeps 1/SPLIT of the
        // chunks on one address
        const CONST_AS int32_t* slotTr = CTAB(int32_t, a.slotTr) + (size_t)chunk * maxCT;
        GLOBAL_AS double* r8 = GD(a.red8);
        const int n4 = 4 * Ns;
        const int part = chunk % LWHIP_REDUCE_SPLIT;
        for (int idx = t; idx < maxCT * n4; idx += blockDim.x)
        {
            const int cs = idx / n4;
            const int tr = slotTr[cs];
            if (tr >= 0)
                unsafeAtomicAdd((double*)(r8 + ((size_t)tr * LWHIP_REDUCE_SPLIT + part) * n4 + (idx - cs * n4)), sAcc[idx]);
        }
    }
}

// The fused launch of a deep column with the depth points of a direction split over S wavefronts (see the comment inside): a
// kernel of its own, so that the eight further inlined marches do not weigh on the registers of raymarch_kernel.
template <int SOLVER>
__global__ void __launch_bounds__(RM_MAX_WAVES * 64) raymarch_split_kernel(const TileArgs* __restrict__ ap, const TileDyn dyn)
{
    dbg_poison_lds();
    extern __shared__ double lds[];
    const CONST_AS TileArgs& a = *(const CONST_AS TileArgs*)ap;
    const int Ns = a.Ns;
    const int maxCT = a.maxCT;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nWaves = blockDim.x >> 6;
    double* sAcc = lds;
    double* win = lds + (size_t)maxCT * 4 * Ns + (size_t)wave * RM_WIN_DOUBLES;
    const int chunk = blockIdx.x;
    const int t0 = CTAB(int32_t, a.chunkTile)[chunk];
    for (int idx = t; idx < maxCT * 4 * Ns; idx += blockDim.x)
        sAcc[idx] = 0.0;
    __syncthreads();
    const CONST_AS int32_t* tileList = CTAB(int32_t, a.tileList);
    // ---- deep columns (round 6, TileArgs::depthSplit = S > 1; fused launches of the compiled kinds): the workgroup is the
    // 2 S wavefronts of ONE tile, wavefront (segment, direction) marches a quarter (S = 4) of the depth points.  The march is a
    // chain of ~440 dependent-ish instructions per depth point: a 500-point column is 0.8 ms for ONE wavefront however few
    // wavelengths there are (the reference benchmark's grid: 484 wavefronts on 1 024 SIMDs).  The recurrence is affine in the
    // intensity, so the segments can run side by side: segment 0 of a direction does its real pass at once; the others first
    // walk their points without rates (65 % of a step) to get the segment's map I_out = E I_in + B, everybody meets at ONE
    // barrier, every later segment composes its starting intensity from segment 0's result and the maps in front of it and
    // does its real pass.  Critical path 2 T / S instead of T -- as long as the chip has idle SIMDs for the doubled work (measured:
    // 0.80 -> 0.65 ms at 500 points x 2 908 wavelengths with S = 4; lwhip_tables.hip).  Same arithmetic per point as the unsplit march (the warm-up
    // reproduces the carried state exactly); the intensity entering a segment is formed through the composed map, i.e. in
    // another order of roundings (1e-16).
    {
        {
            const int S = a.depthSplit;
            const int toObs = wave & 1, seg = wave >> 1;
            const int ti = tileList ? tileList[t0] : t0;
            const DevTile td = ld_c(CTAB(DevTile, a.tiles) + ti);
            const int mLen = (((Ns + S - 1) / S) + 7) & ~7;
            const int mA = seg * mLen, mB = min(Ns, mA + mLen);
            double* xch = lds + (size_t)maxCT * 4 * Ns + (size_t)nWaves * RM_WIN_DOUBLES; // [direction][segment][E | B][64]
            double* mine = xch + (size_t)((toObs * S + seg) * 2) * 64;
            auto run = [&](auto pass1Tag, const double Iin, double* Io, double* Eo) {
                constexpr bool P1 = decltype(pass1Tag)::value;
                if (td.nLine == 0)
                    march_tile<SOLVER, 0, 0, !P1, false, true, P1>(a, dyn, td, toObs, sAcc, win, lane, mA, mB, Iin, Io, Eo);
                else if (td.nLine == 1 && td.nMixed == 0)
                    march_tile<SOLVER, 1, 0, !P1, false, true, P1>(a, dyn, td, toObs, sAcc, win, lane, mA, mB, Iin, Io, Eo);
                else if (td.nLine == 1)
                    march_tile<SOLVER, 1, 1, !P1, false, true, P1>(a, dyn, td, toObs, sAcc, win, lane, mA, mB, Iin, Io, Eo);
                else
                    march_tile<SOLVER, 2, 0, !P1, false, true, P1>(a, dyn, td, toObs, sAcc, win, lane, mA, mB, Iin, Io, Eo);
            };
            // phase 0: segment 0 runs its real pass, the others their first pass; phase 1: the others' real pass (ONE call site of
            // the real march: it is inlined once per tile kind)
#pragma unroll 1
            for (int phase = 0; phase < 2; ++phase)
            {
                if (phase == 0 && seg > 0 && mA < mB && mB < Ns)
                {
                    double Io = 0.0, Eo = 1.0;
                    run(std::true_type{}, 0.0, &Io, &Eo);
                    mine[lane] = Eo;
                    mine[64 + lane] = Io;
                }
                if ((phase == 0) == (seg == 0) && mA < mB)
                {
                    double Iin = 0.0;
                    if (seg > 0)
                    {
                        Iin = xch[(size_t)((toObs * S) * 2 + 1) * 64 + lane]; // what segment 0 ends with
                        for (int q = 1; q < seg; ++q)
                            Iin = fma(xch[(size_t)((toObs * S + q) * 2) * 64 + lane], Iin, xch[(size_t)((toObs * S + q) * 2 + 1) * 64 + lane]);
                    }
                    double Io = 0.0, Eo = 1.0;
                    run(std::false_type{}, Iin, &Io, &Eo);
                    if (seg == 0)
                        mine[64 + lane] = Io;
                }
                if (phase == 0)
                    __syncthreads();
            }
            __syncthreads(); // every segment's moment stores have left the wavefronts
            unsigned long long* sDj = (unsigned long long*)(lds + (size_t)maxCT * 4 * Ns);
            if (S == 2)
                post_tile<256>(ap, td, !dyn.prdOnly, ti % LWHIP_REDUCE_SPLIT, (double*)(sDj + 16), sDj, t);
            else
                post_tile<512>(ap, td, !dyn.prdOnly, ti % LWHIP_REDUCE_SPLIT, (double*)(sDj + 16), sDj, t);
            __syncthreads();
            const CONST_AS int32_t* slotTr = CTAB(int32_t, a.slotTr) + (size_t)chunk * maxCT;
            GLOBAL_AS double* r8 = GD(a.red8);
            const int n4 = 4 * Ns;
            const int part = chunk % LWHIP_REDUCE_SPLIT;
            for (int idx = t; idx < maxCT * n4; idx += blockDim.x)
            {
                const int cs = idx / n4;
                const int tr = slotTr[cs];
                if (tr >= 0)
                    unsafeAtomicAdd((double*)(r8 + ((size_t)tr * LWHIP_REDUCE_SPLIT + part) * n4 + (idx - cs * n4)), sAcc[idx]);
            }
        }
    }
}

// ---- profiles into the tile layout ------------------------------------------------------------------------------------
// grid (tile, line slot); phiT[phiOff + (dir * Ns + k) * 64 + mu * L + j] = phi[lt0 + j][mu][dir][k]
// (blockIdx.z: the column of a batch, its arguments read from the list)
__global__ void __launch_bounds__(256) retile_kernel(const RetileArgs a0, const RetileArgs* __restrict__ list)
{
    dbg_poison_lds();
    const RetileArgs a = list ? ld_c(CTAB(RetileArgs, list) + blockIdx.z) : a0;
    if ((int)blockIdx.x >= a.nTiles)
        return;
    const DevTile td = a.tiles[blockIdx.x];
    if ((int)blockIdx.y >= td.nLine)
        return;
    const DevTileSlot sl = a.tslots[td.slotOff + blockIdx.y];
    const DevTrans tr = a.trans[sl.tr];
    const int lt0 = td.la0 - tr.Nblue;
    const int Ns = a.Ns, Nmu = a.Nmu, L = a.L;
    const double* src = a.phi + tr.phiOff;
    double* dst = a.phiT + sl.phiOff;
    for (int idx = threadIdx.x; idx < 2 * Ns * 64; idx += blockDim.x)
    {
        const int lane = idx & 63;
        const int dk = idx >> 6;
        const int dir = dk / Ns, k = dk - dir * Ns;
        const int mu = lane / L, j = lane - mu * L;
        double v = 0.0;
        if (mu < Nmu && j < td.nl)
            v = src[(((size_t)(lt0 + j) * Nmu + mu) * 2 + dir) * Ns + k];
        dst[idx] = v;
    }
}

hipError_t launch_retile(const RetileArgs& a, hipStream_t stream)
{
    if (a.nTiles <= 0)
        return hipSuccess;
    LWHIP_LAUNCH(retile_kernel, dim3(a.nTiles, 8), dim3(256), 0, stream, a, (const RetileArgs*)nullptr);
    return hipGetLastError();
}

hipError_t launch_retile_list(const RetileArgs* devList, const RetileArgs* hostList, int n, hipStream_t stream)
{
    int maxTiles = 0;
    for (int i = 0; i < n; ++i)
        maxTiles = std::max(maxTiles, hostList[i].nTiles);
    if (n <= 0 || maxTiles <= 0)
        return hipSuccess;
    LWHIP_LAUNCH(retile_kernel, dim3(maxTiles, 8, n), dim3(256), 0, stream, hostList[0], devList);
    return hipGetLastError();
}

size_t raymarch_lds_bytes(const TileArgs& a, int waves, bool fuse)
{
    // accumulators + the waves' moment windows; the fused post-pass reuses the windows' space for the dJ row and the
    // threads' level columns
    // (depth-split launches: 2 S wavefronts, the segments' exchange block behind the windows, the post-pass on all 128 S threads)
    const int S = (fuse && a.depthSplit > 1) ? a.depthSplit : 1;
    const size_t win = (size_t)waves * RM_WIN_DOUBLES + (S > 1 ? (size_t)2 * S * 2 * 64 : 0);
    const size_t post = fuse ? 16 + (size_t)(a.Natom + 2 * a.NlevTot) * 128 * S : 0;
    return sizeof(double) * ((size_t)a.maxCT * 4 * a.Ns + std::max(win, post));
}

hipError_t launch_tile_pre(const TileArgs* devArgs, const TileArgs& a, int nTilesLaunch, const TileArgs* const* apList, int nBatch,
                           hipStream_t stream)
{
    if (nTilesLaunch <= 0)
        return hipSuccess;
    const size_t lds = (size_t)(a.preCols + 1) * RM_PRE_T * sizeof(double) + (size_t)(a.Natom + 2 * a.NlevTot + 2) * sizeof(int);
    const int parts = (a.Ns * a.L + RM_PRE_T - 1) / RM_PRE_T;
    if (lds > 48 * 1024)
    {
        hipError_t e = hipFuncSetAttribute(apList ? (const void*)tile_pre_kernel<true> : (const void*)tile_pre_kernel<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
    }
    if (apList)
        LWHIP_LAUNCH(tile_pre_kernel<true>, dim3(nTilesLaunch, parts, std::max(nBatch, 1)), dim3(RM_PRE_T), lds, stream, devArgs,
                           apList, a.tileList);
    else
        LWHIP_LAUNCH(tile_pre_kernel<false>, dim3(nTilesLaunch, parts), dim3(RM_PRE_T), lds, stream, devArgs, apList, a.tileList);
    return hipGetLastError();
}

template <int SOLVER, int CAP, bool RATES, bool FUSE>
static hipError_t launch_rm_t(const TileArgs* devArgs, const TileArgs& a, const TileDyn& dyn, int nChunks, int waves,
                              const TileArgs* const* apList, int nBatch, hipStream_t stream)
{
    const size_t lds = raymarch_lds_bytes(a, waves, FUSE);
    if (lds > 48 * 1024)
    {
        hipError_t e = hipFuncSetAttribute(apList ? (const void*)raymarch_kernel<SOLVER, CAP, RATES, true, FUSE>
                                                  : (const void*)raymarch_kernel<SOLVER, CAP, RATES, false, FUSE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
    }
    if (apList)
        LWHIP_LAUNCH((raymarch_kernel<SOLVER, CAP, RATES, true, FUSE>), dim3(nChunks, std::max(nBatch, 1)), dim3(waves * 64), lds,
                           stream, devArgs, apList, dyn);
    else
        LWHIP_LAUNCH((raymarch_kernel<SOLVER, CAP, RATES, false, FUSE>), dim3(nChunks), dim3(waves * 64), lds, stream, devArgs, apList,
                           dyn);
    return hipGetLastError();
}

// cap: 0 = every tile is one of the fast kinds; otherwise the generic march's capacity (lines, and mixed continua)
template <int SOLVER, bool RATES, bool FUSE>
static hipError_t launch_rm_s(const TileArgs* devArgs, const TileArgs& a, const TileDyn& dyn, int cap, int nChunks, int waves,
                              const TileArgs* const* apList, int nBatch, hipStream_t stream)
{
    if (cap == 0)
        return launch_rm_t<SOLVER, 0, RATES, FUSE>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream);
    if (cap <= 2)
        return launch_rm_t<SOLVER, 2, RATES, FUSE>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream);
    if (cap <= 4)
        return launch_rm_t<SOLVER, 4, RATES, FUSE>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream);
    return launch_rm_t<SOLVER, 8, RATES, FUSE>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream);
}

template <int SOLVER>
static hipError_t launch_rm_r(const TileArgs* devArgs, const TileArgs& a, const TileDyn& dyn, int cap, bool rates, bool fuse,
                              int nChunks, int waves, const TileArgs* const* apList, int nBatch, hipStream_t stream)
{
    if (!rates)
        return launch_rm_s<SOLVER, false, false>(devArgs, a, dyn, cap, nChunks, waves, apList, nBatch, stream);
    if (fuse)
        return launch_rm_s<SOLVER, true, true>(devArgs, a, dyn, cap, nChunks, waves, apList, nBatch, stream);
    return launch_rm_s<SOLVER, true, false>(devArgs, a, dyn, cap, nChunks, waves, apList, nBatch, stream);
}

// fuse: every workgroup is the two wavefronts of one tile (waves == 2, one tile per chunk) and runs the post-pass itself
hipError_t launch_tile_sweep(const TileArgs* devArgs, const TileArgs& a, const TileDyn& dyn, int solver, int cap, bool rates,
                             bool fuse, int nChunks, int waves, const TileArgs* const* apList, int nBatch, hipStream_t stream)
{
    if (nChunks <= 0)
        return hipSuccess;
    const int S = (fuse && a.depthSplit > 1) ? a.depthSplit : 1;
    if (cap > 8 || waves < 1 || waves > RM_MAX_WAVES || (fuse && (waves != 2 * S || !rates || dyn.nPass != 2)) || (S > 1 && (cap != 0 || apList)))
        return hipErrorInvalidValue;
    if (S > 1)
    {
        const size_t lds = raymarch_lds_bytes(a, waves, true);
        auto go = [&](auto kernel) -> hipError_t {
            if (lds > 48 * 1024)
            {
                hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess)
                    return e;
            }
            LWHIP_LAUNCH(kernel, dim3(nChunks), dim3(waves * 64), lds, stream, devArgs, dyn);
            return hipGetLastError();
        };
        if (solver == LWHIP_FS_LINEAR_1D)
            return go(raymarch_split_kernel<LWHIP_FS_LINEAR_1D>);
        if (solver == LWHIP_FS_BESSER_1D)
            return go(raymarch_split_kernel<LWHIP_FS_BESSER_1D>);
        return go(raymarch_split_kernel<LWHIP_FS_BEZIER3_1D>);
    }
    switch (solver)
    {
        case LWHIP_FS_LINEAR_1D:
            return launch_rm_r<LWHIP_FS_LINEAR_1D>(devArgs, a, dyn, cap, rates, fuse, nChunks, waves, apList, nBatch, stream);
        case LWHIP_FS_BESSER_1D:
            return launch_rm_r<LWHIP_FS_BESSER_1D>(devArgs, a, dyn, cap, rates, fuse, nChunks, waves, apList, nBatch, stream);
        default:
            return launch_rm_r<LWHIP_FS_BEZIER3_1D>(devArgs, a, dyn, cap, rates, fuse, nChunks, waves, apList, nBatch, stream);
    }
}

hipError_t launch_tile_post(const TileArgs* devArgs, const TileArgs& a, const TileDyn& dyn, int nPostChunks,
                            const TileArgs* const* apList, int nBatch, hipStream_t stream)
{
    if (nPostChunks <= 0)
        return hipSuccess;
    const size_t lds = sizeof(double) * (16 + (size_t)(a.Natom + 2 * a.NlevTot) * RM_POST_T);
    if (lds > 48 * 1024)
    {
        hipError_t e = hipFuncSetAttribute(apList ? (const void*)tile_post_kernel<true> : (const void*)tile_post_kernel<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
    }
    if (apList)
        LWHIP_LAUNCH(tile_post_kernel<true>, dim3(nPostChunks, std::max(nBatch, 1)), dim3(RM_POST_T), lds, stream, devArgs, apList,
                           dyn, a.postChunkTile, a.tileList);
    else
        LWHIP_LAUNCH(tile_post_kernel<false>, dim3(nPostChunks), dim3(RM_POST_T), lds, stream, devArgs, apList, dyn,
                           a.postChunkTile, a.tileList);
    return hipGetLastError();
}
}

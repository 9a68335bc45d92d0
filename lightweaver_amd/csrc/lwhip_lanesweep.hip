// lwhip_lanesweep.hip -- the depth-across-lanes sweep of the formal_sol_gamma_matrices iteration on gfx950 (MI355X).
//
// Mapping: the DEPTH axis goes across the lanes.  A lane owns D consecutive depth points of one
// wavelength; LR = ceil(Ns / D) lanes make a ray; R = 64 / LR consecutive, structurally identical wavelengths (a *tile*)
// share a wavefront.  The wavefront walks the rays of its wavelengths one after the other -- mu outer, down then up,
// the reference's order (SimdFullIterationTemplates.hpp:300-304) -- and per ray
//   * gathers chi, eta, S at its own points (Transition::uv + chi_eta_aux_accum + compute_source_fn),
//   * forms every stencil quantity of the 1D solver for them (the halo points come from the neighbouring lanes by
//     DPP wave shifts): path lengths, chi slopes, Steffen derivatives, optical depths, dS/dtau, the Bezier
//     coefficients -- everything but the intensity itself, D independent instruction chains per lane,
//   * writes the solver's recurrence I_k = e_k I_k-1 + b_k as affine maps, composes its own D maps and scans the
//     composites over the lanes of the ray (the first point of a ray is a constant map, so one wave-wide scan serves all
//     rays of the wavefront); there is NO sequential march over depth any more,
//   * expands I at its points and integrates J, the Psi* moments and the preconditioned rate integrals of the lines
//     and mixed continua into REGISTERS (a lane keeps its (wavelength, depth) points for all rays, so the angle sums
//     need no cross-lane traffic and run in the reference's ray order).
// After the last ray the wavefront finishes its wavelengths itself: J out, dJ, and the pure continua from the angle
// moments (their level sums held in registers by a per-tile program of the host's), adds its integrals to the
// workgroup's LDS accumulators, and the workgroup adds those to the stage-1 reduce buffer.
// Memory: profiles are read from the caller's layout phi[lt][mu][dir][k] (32 contiguous bytes per lane, no tile-blocked
// copy), the ray-independent continuum rows from the pre-pass's [row][wavelength][k] blocks; no moment block.
// fp64 throughout, no MFMA: a recurrence over fp64 columns, not a contraction.
//
// Reference restated (file:line under Lightweaver's Source/): intensity_core_opt
// SimdFullIterationTemplates.hpp:238-487, chi_eta_aux_accum :59-109, compute_full_operator_rates :192-234,
// piecewise_bezier3_1d FormalScalar.cpp:209-325, Bezier.hpp:58-127, w2 LwInternal.hpp:90-110.
#include "lwhip_device.h"
#include "../../include/lwhip.h"

#include <algorithm>
#include <type_traits>

namespace lwhip
{
namespace
{
typedef unsigned int ls_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int ls_u4 __attribute__((ext_vector_type(4)));

DEVINL __amdgpu_buffer_rsrc_t ls_rsrc(const void* p, unsigned long long bytes)
{
    const unsigned n = bytes > 0xfffffff0ull ? 0xfffffff0u : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)n, 0x00020000);
}
// N consecutive doubles at byte offset voff (per lane) + soff (uniform) of a buffer: 16-byte loads (out-of-range reads
// return zero: the padding points of the last lane of a ray)
template <int N> DEVINL void ls_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double (&x)[N])
{
#pragma unroll
    for (int q = 0; q + 1 < N; q += 2)
    {
        const ls_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(voff + (unsigned)q * 8u), (int)soff, 0);
        x[q] = __builtin_bit_cast(double, ls_u2{ v.x, v.y });
        x[q + 1] = __builtin_bit_cast(double, ls_u2{ v.z, v.w });
    }
    if (N & 1)
        x[N - 1] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)(voff + (unsigned)(N - 1) * 8u), (int)soff, 0));
}
DEVINL double ls_ld1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
// the value of the neighbouring lane: FROM_LOWER: lane - 1 (wave_shr:1), else lane + 1 (wave_shl:1); lanes without a
// source get 0
template <bool FROM_LOWER> DEVINL double ls_shift1(double v)
{
    const unsigned long long vv = __double_as_longlong(v);
    constexpr int CTRL = FROM_LOWER ? 0x138 : 0x130;
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)vv, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(vv >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
DEVINL double ls_bperm(double v, int srcLane)
{
    const unsigned long long vv = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(srcLane << 2, (int)(unsigned)vv);
    const int hi = __builtin_amdgcn_ds_bpermute(srcLane << 2, (int)(unsigned)(vv >> 32));
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// Inclusive scan of the affine maps I -> E I + B along the march: lane order (down rays) or reverse lane order (up rays).
// Afterwards a lane holds the composition of every map from the start of the wave up to and including its own.
template <bool UP> DEVINL void ls_scan(double& E, double& B, const int lane)
{
    if (!UP)
    {
        affine_scan_wave(E, B); // DPP row shifts + row broadcasts (lwhip_device.h)
        return;
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1)
    {
        const int src = lane + s;
        const bool ok = src < 64;
        const double Ep = ls_bperm(E, src), Bp = ls_bperm(B, src);
        const double Bn = fma(E, Bp, B), En = E * Ep; // (Ep, Bp) is applied first
        B = ok ? Bn : B;
        E = ok ? En : E;
    }
}
// 1 / x from v_rcp_f64 and one Newton step
DEVINL double ls_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
// exp(x) for -700 < x <= 0: range reduction + the degree-13 polynomial split into its even and odd part
DEVINL double ls_exp_neg(double x)
{
    const double n = rint(x * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    const double r2 = r * r;
    double pe = 1.0 / 479001600.0, po = 1.0 / 6227020800.0;
    pe = fma(pe, r2, 1.0 / 3628800.0);
    po = fma(po, r2, 1.0 / 39916800.0);
    pe = fma(pe, r2, 1.0 / 40320.0);
    po = fma(po, r2, 1.0 / 362880.0);
    pe = fma(pe, r2, 1.0 / 720.0);
    po = fma(po, r2, 1.0 / 5040.0);
    pe = fma(pe, r2, 1.0 / 24.0);
    po = fma(po, r2, 1.0 / 120.0);
    pe = fma(pe, r2, 0.5);
    po = fma(po, r2, 1.0 / 6.0);
    pe = fma(pe, r2, 1.0);
    po = fma(po, r2, 1.0);
    return ldexp(fma(po, r, pe), (int)n);
}

enum { LS_MAX_WAVES = 8, LS_MAX_R = 16 };
#define LS_KEEP(v) asm volatile("" : "+v"(v)) /* the value is computed here, on every lane (keeps selects branch-free) */
#define LS_PHASE() __builtin_amdgcn_sched_barrier(0) /* the scheduler does not move code across phase boundaries */

// What a lane is in its tile, and the buffers every phase reads.
struct LaneCtx
{
    int Ns, Nmu, LRD, L;
    int lane, rRaw, blk, r, k0, la, nV, pBot; // nV: valid points of the block; pBot: index of depth Ns - 1 in the block (if in range)
    bool act;
    double wav;
    unsigned vK, vRow, rowStride, Ns8, gStride;
    __amdgpu_buffer_rsrc_t rRows, rN, rW, rG;
};

// The rays of one tile: NL / NM line / mixed-continuum slots.  Leaves J and the Psi* moments of the lane's points in
// Jk / M1 / Me and adds the slots' integrals to the workgroup's accumulators.
template <int SOLVER, int D, int NL, int NM, bool RATES>
DEVINL void lane_rays(const CONST_AS TileArgs& a, const TileDyn& dyn, const DevTile& td, const LaneCtx& c, double* sAcc,
                      double (&Jk)[D], double (&M1)[D], double (&Me)[2][D])
{
    constexpr int NLc = NL > 0 ? NL : 1, NMc = NM > 0 ? NM : 1;
    const int Ns = c.Ns, Nmu = c.Nmu, LRD = c.LRD;
    const int la = c.la, k0 = c.k0;
    const bool act = c.act;
    const unsigned vK = c.vK, vRow = c.vRow, rowStride = c.rowStride, Ns8 = c.Ns8;
    const __amdgpu_buffer_rsrc_t rRows = c.rRows, rN = c.rN, rW = c.rW, rG = c.rG;

    // ---- slot structure (uniform) and per-lane parameters -------------------------------------------------------------
    const CONST_AS DevTileSlot* ts = CTAB(DevTileSlot, a.tslots) + td.slotOff;
    const int slotOffLa = ((const GLOBAL_AS DevLaHeader*)a.laHdr)[la].off;
    unsigned lnI[NLc], lnJ[NLc], lwO[NLc], lrow[NLc], lacc[NLc];
    int lgi[NLc], lgj[NLc], lat[NLc], lfl[NLc], lms[NLc], lzm[NLc];
    double la0[NLc], la2[NLc], la3[NLc], lcw1[NLc]; // per lane
    __amdgpu_buffer_rsrc_t rPhi[NLc], rRho[NLc];
    const unsigned vPhi = (unsigned)c.r * (unsigned)(Nmu * 2) * Ns8 + vK;
    const unsigned vRho = (unsigned)c.r * Ns8 + vK;
#pragma unroll
    for (int s = 0; s < NLc; ++s)
    {
        lnI[s] = lnJ[s] = lwO[s] = lrow[s] = lacc[s] = 0;
        lgi[s] = lgj[s] = lat[s] = -1 - s;
        lfl[s] = 0;
        lms[s] = -1;
        lzm[s] = 31;
        la0[s] = la2[s] = la3[s] = lcw1[s] = 0.0;
        rPhi[s] = ls_rsrc(a.phi, 0);
        rRho[s] = ls_rsrc(a.rho, 0);
        if (s < NL)
        {
            const DevTileSlot u = ld_c(ts + s);
            lnI[s] = (unsigned)u.gi * Ns8;
            lnJ[s] = (unsigned)u.gj * Ns8;
            lwO[s] = (unsigned)u.row * Ns8;
            lrow[s] = (unsigned)u.r0 * rowStride;
            lacc[s] = (unsigned)(u.cs * 4 * LRD);
            lgi[s] = u.gi;
            lgj[s] = u.gj;
            lat[s] = u.atom;
            lfl[s] = u.flags;
            lms[s] = (u.flags & 1) ? u.ms : -1;
            lzm[s] = (u.flags & 1) ? u.zmask : 31;
            rPhi[s] = ls_rsrc(a.phi + u.phiOff, (a.phiDoubles - (unsigned long long)u.phiOff) * 8ull);
            if (u.flags & 2)
                rRho[s] = ls_rsrc(a.rho + u.rhoOff, (a.rhoDoubles - (unsigned long long)u.rhoOff) * 8ull);
            const GLOBAL_AS DevSlot* sl = (const GLOBAL_AS DevSlot*)a.slots + slotOffLa + u.e;
            la0[s] = sl->p[0];
            la2[s] = sl->p[2];
            la3[s] = sl->p[3];
            lcw1[s] = sl->p[1] * PI4_HC; // times 0.5 wmu per ray
        }
    }
    unsigned mrow[NMc], maux[NMc], macc[NMc];
    int mgi[NMc], mgj[NMc], mat[NMc], mzm[NMc];
    double mb0[NMc], mb2[NMc], mcw1[NMc];
#pragma unroll
    for (int m = 0; m < NMc; ++m)
    {
        mrow[m] = maux[m] = macc[m] = 0;
        mgi[m] = mgj[m] = mat[m] = -100 - m;
        mzm[m] = 31;
        mb0[m] = mb2[m] = mcw1[m] = 0.0;
        if (m < NM)
        {
            const DevTileSlot u = ld_c(ts + NL + m);
            mrow[m] = (unsigned)u.r0 * rowStride;
            maux[m] = (unsigned)max((int)u.rAux, 0) * rowStride;
            macc[m] = (unsigned)(u.cs * 4 * LRD);
            mgi[m] = u.gi;
            mgj[m] = u.gj;
            mat[m] = u.atom;
            mzm[m] = u.zmask;
            const GLOBAL_AS DevSlot* sl = (const GLOBAL_AS DevSlot*)a.slots + slotOffLa + u.e;
            mb0[m] = sl->p[0];
            mb2[m] = sl->p[2];
            mcw1[m] = sl->p[1];
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
            ccL[s][l] = __builtin_amdgcn_readfirstlane((l != s && s < NL && l < NL) ? couple_code(l, lat[s], lgi[s], lgj[s]) : 0);
#pragma unroll
    for (int m = 0; m < NMc; ++m)
#pragma unroll
        for (int l = 0; l < NLc; ++l)
            ccM[m][l] = __builtin_amdgcn_readfirstlane((m < NM && l < NL) ? couple_code(l, mat[m], mgi[m], mgj[m]) : 0);

    const bool storeDepth = RATES && a.storeDepth && !dyn.prdOnly;
    const double lamf = dyn.lambdaIterate ? 0.0 : 1.0;

    // boundary intensities of the rays (wrappers FormalScalar.cpp:471-666): Planck functions of the two end points
    double Bd0 = 0.0, Bd1 = 0.0, Bu0 = 0.0, Bu1 = 0.0;
    {
        const CONST_AS double* T = CTAB(double, a.temperature);
        if (a.upperBc == LWHIP_BC_THERMALISED && dyn.nPass == 2)
        {
            Bd0 = d_planck(T[0], c.wav);
            Bd1 = d_planck(T[1], c.wav);
        }
        if (a.lowerBc == LWHIP_BC_THERMALISED)
        {
            Bu0 = d_planck(T[Ns - 2], c.wav);
            Bu1 = d_planck(T[Ns - 1], c.wav);
        }
    }

    // the integrals of the slots at this lane's points, summed over the rays
    double accL[NLc][4][D], accM[NMc][4][D];
#pragma unroll
    for (int p = 0; p < D; ++p)
    {
#pragma unroll
        for (int s = 0; s < NLc; ++s)
            accL[s][0][p] = accL[s][1][p] = accL[s][2][p] = accL[s][3][p] = 0.0;
#pragma unroll
        for (int m = 0; m < NMc; ++m)
            accM[m][0][p] = accM[m][1][p] = accM[m][2][p] = accM[m][3][p] = 0.0;
    }

    // ---- one ray of every wavelength of the tile ------------------------------------------------------------------------
    auto pass = [&](auto upTag, const int mu) {
        constexpr bool UP = decltype(upTag)::value;
        constexpr int dirI = UP ? 1 : 0;
#define MI(i) (UP ? D - 1 - (i) : (i)) /* march index <-> index in the lane's k-ordered block (an involution) */
#define IV(i) (UP ? D - (i) : (i))     /* interval i (march point i-1 -> i) -> index in dhK / rdhK */
        const double muz = CTAB(double, a.muz)[mu];
        const double zmu = 1.0 / muz;
        const double hw = act ? 0.5 * CTAB(double, a.wmu)[mu] : 0.0;
        const unsigned sRay = (unsigned)(mu * 2 + dirI) * Ns8;
        const int bcType = UP ? a.lowerBc : a.upperBc;
        // the ray's first point is depth 0 (down: march index 0 of block 0) or depth Ns - 1 (up), its last the other one
        auto is_first = [&](int i) { return UP ? MI(i) == c.pBot : (i == 0 && c.blk == 0); };
        auto is_last = [&](int i) { return UP ? (i == D - 1 && c.blk == 0) : MI(i) == c.pBot; };
        auto is_ok = [&](int i) { return MI(i) < c.nV; };

        // ---- A. gather: Transition::uv + chi_eta_aux_accum + compute_source_fn at the lane's points (march order) ----
        double chi[D], S[D], rch[D];
        double vij[NLc][D], vji[NLc][D];
        {
            double cC[D], eC[D], sc[D];
            ls_ld<D>(rRows, vRow, 0u, cC);
            ls_ld<D>(rRows, vRow, rowStride, eC);
            ls_ld<D>(rRows, vRow, 2u * rowStride, sc);
#pragma unroll
            for (int s = 0; s < NLc; ++s)
            {
                if (s < NL)
                {
                    double phi[D], ni[D], nj[D], rho[D];
                    ls_ld<D>(rPhi[s], vPhi, sRay, phi);
                    ls_ld<D>(rN, vK, lnI[s], ni);
                    ls_ld<D>(rN, vK, lnJ[s], nj);
                    if (lfl[s] & 2)
                        ls_ld<D>(rRho[s], vRho, 0u, rho);
                    else
                    {
#pragma unroll
                        for (int p = 0; p < D; ++p)
                            rho[p] = 1.0;
                    }
#pragma unroll
                    for (int p = 0; p < D; ++p)
                    {
                        const double v1 = la0[s] * phi[p];
                        const double v2 = (la2[s] * rho[p]) * v1;
                        const double uji = la3[s] * v2;
                        cC[p] += ni[p] * v1 - nj[p] * v2;
                        eC[p] += nj[p] * uji;
                        vij[s][MI(p)] = v1;
                        vji[s][MI(p)] = v2;
                    }
                }
                else
                {
#pragma unroll
                    for (int p = 0; p < D; ++p)
                        vij[s][p] = vji[s][p] = 0.0;
                }
            }
#pragma unroll
            for (int p = 0; p < D; ++p)
            {
                const double rc = ls_rcp(cC[p]);
                chi[MI(p)] = cC[p];
                rch[MI(p)] = rc;
                S[MI(p)] = (eC[p] + sc[p]) * rc;
            }
            if (storeDepth && act)
            {
                const size_t o = ((size_t)(la * Nmu + mu) * 2 + dirI) * Ns + k0;
#pragma unroll
                for (int p = 0; p < D; ++p)
                    if (p < c.nV)
                    {
                        GD(a.depthChi)[o + p] = cC[p];
                        GD(a.depthEta)[o + p] = eC[p];
                    }
            }
        }
        LS_PHASE();
        // halo: the last point of the upwind lane (P), the first point of the downwind lane (N)
        const double chiP = ls_shift1<!UP>(chi[D - 1]), SP = ls_shift1<!UP>(S[D - 1]);
        const double chiN = ls_shift1<UP>(chi[0]), SN = ls_shift1<UP>(S[0]);
#define CHI(i) ((i) < 0 ? chiP : (i) >= D ? chiN : chi[(i) < 0 ? 0 : (i) >= D ? D - 1 : (i)])
#define SRC(i) ((i) < 0 ? SP : (i) >= D ? SN : S[(i) < 0 ? 0 : (i) >= D ? D - 1 : (i)])

        double ee[D], bb[D], PsiS[D]; // the points' maps I_i = ee I_i-1 + bb, and Psi* (:322-323)
        if (SOLVER == LWHIP_FS_BEZIER3_1D)
        {
            // piecewise_bezier3_1d_impl FormalScalar.cpp:209-325, every quantity of the march formed per point
            double x[D], dS[D], dSP; // optical depth of the intervals, dS / dtau at the points
            {
                double dhK[D + 1], rdhK[D + 1], rsumK[D];
                ls_ld<D + 1>(rG, vK, 0, dhK); // entry k + 1 holds |h_k - h_k+1|: dhK[q] = dh[k0 - 1 + q]
                ls_ld<D + 1>(rG, vK, c.gStride, rdhK);
                ls_ld<D>(rG, vK + 8u, 2u * c.gStride, rsumK); // rsumK[p] = rsum[k0 + p]
                double ds[D + 1], sl[D + 1];
#pragma unroll
                for (int i = 0; i <= D; ++i)
                {
                    ds[i] = dhK[IV(i)] * zmu;
                    sl[i] = (CHI(i) - CHI(i - 1)) * (rdhK[IV(i)] * muz);
                }
                // d chi / d s at the points: cent_deriv (Steffen 1990, Bezier.hpp:58-65); the end points take the slope
                // of their only interval (:239, :288)
                double dx[D];
#pragma unroll
                for (int i = 0; i < D; ++i)
                {
                    const double P0 = fabs((sl[i] * ds[i + 1] + sl[i + 1] * ds[i]) * (rsumK[MI(i)] * muz));
                    const double v = (copysign(1.0, sl[i + 1]) + copysign(1.0, sl[i])) * fmin(fabs(sl[i]), fmin(fabs(sl[i + 1]), 0.5 * P0));
                    dx[i] = is_first(i) ? sl[i + 1] : is_last(i) ? sl[i] : v;
                }
                const double dxP = ls_shift1<!UP>(dx[D - 1]);
                // optical depth of the intervals (:242-246, 261-263, 289-292) and the S slopes per unit tau
                double sS[D];
#pragma unroll
                for (int i = 0; i < D; ++i)
                {
                    const double cu = CHI(i - 1), cd = CHI(i);
                    const double dxu = i == 0 ? dxP : dx[i > 0 ? i - 1 : 0];
                    const double ds3 = ds[i] * (1.0 / 3.0);
                    const double Cuw = cu + ds3 * dxu;
                    const double C0 = cd - ds3 * dx[i];
                    double t = ds[i] * (cu + cd + Cuw + C0) * 0.25;
                    t = (is_first(i) || !is_ok(i)) ? 1.0 : t;
                    x[i] = t;
                    sS[i] = (SRC(i) - SRC(i - 1)) * ls_rcp(t);
                }
                const double dtN = ls_shift1<UP>(x[0]), sSN = ls_shift1<UP>(sS[0]);
                // dS / dtau at the points (:247, :254-255)
#pragma unroll
                for (int i = 0; i < D; ++i)
                {
                    const double dtd = i == D - 1 ? dtN : x[i < D - 1 ? i + 1 : 0];
                    const double sSd = i == D - 1 ? sSN : sS[i < D - 1 ? i + 1 : 0];
                    const double P0 = fabs((sS[i] * dtd + sSd * x[i]) * ls_rcp(x[i] + dtd));
                    const double v = (copysign(1.0, sSd) + copysign(1.0, sS[i])) * fmin(fabs(sS[i]), fmin(fabs(sSd), 0.5 * P0));
                    dS[i] = is_first(i) ? sSd : v;
                }
                dSP = ls_shift1<!UP>(dS[D - 1]);
                // the last point of the ray is piecewise linear (:306-321): its optical depth replaces the interval's
#pragma unroll
                for (int i = 0; i < D; ++i)
                {
                    if (__builtin_amdgcn_ballot_w64(is_last(i)) != 0)
                    {
                        const double dtL = ((0.5 * zmu) * (CHI(i) + CHI(i - 1))) * dhK[IV(i)];
                        x[i] = is_last(i) ? dtL : x[i];
                    }
                }
                // the first point is the boundary intensity: a constant map
#pragma unroll
                for (int i = 0; i < D; ++i)
                {
                    ee[i] = 0.0;
                    bb[i] = 0.0;
                    if (__builtin_amdgcn_ballot_w64(is_first(i)) != 0)
                    {
                        double Ib = 0.0;
                        if (bcType == LWHIP_BC_THERMALISED)
                        {
                            const double dtb = ((0.5 * zmu) * (CHI(i) + CHI(i + 1))) * dhK[IV(i + 1)];
                            Ib = UP ? Bu1 - (Bu0 - Bu1) / dtb : Bd0 - (Bd1 - Bd0) / dtb;
                        }
                        else if (bcType == LWHIP_BC_CALLABLE)
                        {
                            const int mi = UP ? CTAB(int32_t, a.lowerIdx)[mu * 2 + 1] : CTAB(int32_t, a.upperIdx)[mu * 2 + 0];
                            if (mi >= 0)
                                Ib = UP ? GCD(a.lowerBcData)[(size_t)la * a.lowerNmu + mi] : GCD(a.upperBcData)[(size_t)la * a.upperNmu + mi];
                        }
                        bb[i] = Ib;
                    }
                }
            }
            LS_PHASE();
            // Bezier3_coeffs (Bezier.hpp:81-127) and the update :265-275 as a map
#ifdef LS_DBG_SKIP_C
#pragma unroll
            for (int i = 0; i < D; ++i)
            {
                ee[i] = x[i] * SRC(i - 1);
                bb[i] = dS[i] + (i == 0 ? dSP : 0.0) + bb[i];
                PsiS[i] = rch[i];
            }
#else
#pragma unroll
            for (int i = 0; i < D; ++i)
            {
                const double Su = SRC(i - 1), S0 = S[i];
                const double xi = x[i];
                const double x2 = xi * xi, x3 = x2 * xi;
                const double exr = ls_exp_neg(-fmin(xi, 64.0));
                const double ex = (xi > 30.0) ? 0.0 : exr;
                const double r3 = ls_rcp(x3);
                const bool small = xi < 5e-2;
                // both branches are evaluated on every lane and selected: straight-line code
                double alG = (6.0 - ex * (6.0 + 6.0 * xi + 3.0 * x2 + x3)) * r3;
                double beG = (6.0 * ex - 6.0 + 6.0 * xi - 3.0 * x2 + x3) * r3;
                double gaG = 3.0 * (2.0 * xi - 6.0 + ex * (6.0 + 4.0 * xi + x2)) * r3;
                double deG = 3.0 * (6.0 - 4.0 * xi + x2 - 2.0 * ex * (3.0 + xi)) * r3;
                double alT = 0.25 * xi - 0.2 * x2 + x3 * (1.0 / 12.0);
                double beT = 0.25 * xi - 0.05 * x2 + x3 * (1.0 / 120.0);
                double gaT = 0.25 * xi - 0.15 * x2 + 0.05 * x3;
                double deT = 0.25 * xi - 0.1 * x2 + 0.025 * x3;
                double edT = 1.0 - xi + 0.5 * x2 - x3 * (1.0 / 6.0);
                LS_KEEP(alG);
                LS_KEEP(beG);
                LS_KEEP(gaG);
                LS_KEEP(deG);
                LS_KEEP(alT);
                LS_KEEP(beT);
                LS_KEEP(gaT);
                LS_KEEP(deT);
                LS_KEEP(edT);
                const double al = small ? alT : alG, be = small ? beT : beG, ga = small ? gaT : gaG, de = small ? deT : deG;
                const double edt = small ? edT : ex;
                const double x_3 = xi * (1.0 / 3.0);
                const double dSu = i == 0 ? dSP : dS[i > 0 ? i - 1 : 0];
                const double Cuw = Su + x_3 * dSu;
                const double C0 = S0 - x_3 * dS[i];
                double e1 = edt, b1 = al * Su + be * S0 + ga * Cuw + de * C0, ps = be + de;
                if (__builtin_amdgcn_ballot_w64(is_last(i)) != 0)
                {
                    // w2 (LwInternal.hpp:90-110) of the linear closing step, xi = its optical depth
                    double w0 = 1.0 - exr, w1 = (1.0 - exr) - xi * exr;
                    if (xi < 5.0E-4)
                    {
                        w0 = xi * (1.0 - 0.5 * xi);
                        w1 = (xi * xi) * (0.5 - xi * (1.0 / 3.0));
                    }
                    else if (xi > 50.0)
                        w0 = w1 = 1.0;
                    const double rL = d_rcp(xi);
                    const double dSL = (S0 - Su) * rL;
                    const bool lst = is_last(i);
                    e1 = lst ? 1.0 - w0 : e1;
                    b1 = lst ? w0 * S0 - w1 * dSL : b1;
                    ps = lst ? w0 - w1 * rL : ps;
                }
                if (__builtin_amdgcn_ballot_w64(is_first(i)) != 0)
                {
                    const bool fst = is_first(i);
                    e1 = fst ? 0.0 : e1;
                    b1 = fst ? bb[i] : b1;
                    ps = fst ? 0.0 : ps;
                }
                if (__builtin_amdgcn_ballot_w64(!is_ok(i)) != 0)
                {
                    const bool okk = is_ok(i);
                    e1 = okk ? e1 : 1.0;
                    b1 = okk ? b1 : 0.0;
                    ps = okk ? ps : 0.0;
                }
                ee[i] = e1;
                bb[i] = b1;
                PsiS[i] = (ps * rch[i]) * lamf;
#ifdef LS_SERIAL_C
                if ((i % LS_SERIAL_C) == LS_SERIAL_C - 1)
                    LS_PHASE();
#endif
            }
#endif
        }
        LS_PHASE();

        // ---- the recurrence: compose the lane's maps, scan the composites along the ray, expand ----
        double Iv[D];
        {
            double E = 1.0, B = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i)
            {
                B = fma(ee[i], B, bb[i]);
                E = ee[i] * E;
            }
            ls_scan<UP>(E, B, c.lane);
            double I = ls_shift1<!UP>(B); // the intensity entering the lane (a ray's first point multiplies it by 0)
#pragma unroll
            for (int i = 0; i < D; ++i)
            {
                I = fma(ee[i], I, bb[i]);
                Iv[i] = I;
            }
        }
        LS_PHASE();

        // ---- outputs and the sums over the rays ----
        if (UP && act && c.blk == 0)
            GD(a.I)[(size_t)la * Nmu + mu] = Iv[D - 1]; // spect.I(la, mu, 0) :349
        if ((UP ? a.zUp : a.zDown) != nullptr && act)
        {
            // ZPlaneDecomposition :351-361: depth 1 of an up ray, depth Nz - 2 of a down ray
#pragma unroll
            for (int i = 0; i < D; ++i)
                if (k0 + MI(i) == (UP ? 1 : Ns - 2))
                    GD(UP ? a.zUp : a.zDown)[(size_t)la * Nmu + mu] = Iv[i];
        }
        if (storeDepth && act)
        {
            const size_t o = ((size_t)(la * Nmu + mu) * 2 + dirI) * Ns + k0;
#pragma unroll
            for (int i = 0; i < D; ++i)
                if (is_ok(i))
                    GD(a.depthI)[o + MI(i)] = Iv[i];
        }
        if (RATES)
        {
            double wPsi[D];
#pragma unroll
            for (int i = 0; i < D; ++i)
            {
                wPsi[i] = hw * PsiS[i];
                Jk[MI(i)] += hw * Iv[i]; // accumulate_J :181-190
                M1[MI(i)] += wPsi[i];
            }
            // the lines' opacities at the points (populations of the depth, V_ij / V_ji of the gather)
            double xl[NLc][D], el[NLc][D], uj[NLc][D];
#pragma unroll
            for (int s = 0; s < NLc; ++s)
            {
                if (s < NL)
                {
                    double ni[D], nj[D];
                    ls_ld<D>(rN, vK, lnI[s], ni);
                    ls_ld<D>(rN, vK, lnJ[s], nj);
#pragma unroll
                    for (int i = 0; i < D; ++i)
                    {
                        uj[s][i] = la3[s] * vji[s][i];
                        xl[s][i] = ni[MI(i)] * vij[s][i] - nj[MI(i)] * vji[s][i];
                        el[s][i] = nj[MI(i)] * uj[s][i];
                    }
                }
                else
                {
#pragma unroll
                    for (int i = 0; i < D; ++i)
                        xl[s][i] = el[s][i] = uj[s][i] = 0.0;
                }
            }
            // level sums of a slot: continuum rows + the active lines of this wavelength, selected by the tile's
            // coupling codes (a line's coupling with itself is known at compile time)
            auto couple = [&](const int (&cc)[NLc], const int i, double& etaA, double& chi_i, double& chi_j, double& U_i, double& U_j) {
#pragma unroll
                for (int l = 0; l < NLc; ++l)
                {
                    if (l < NL)
                    {
                        const int c5 = cc[l];
                        etaA += (c5 & 1) ? el[l][i] : 0.0;
                        chi_i += (c5 & 2) ? xl[l][i] : 0.0;
                        chi_i -= (c5 & 4) ? xl[l][i] : 0.0;
                        chi_j += (c5 & 8) ? xl[l][i] : 0.0;
                        chi_j -= (c5 & 16) ? xl[l][i] : 0.0;
                        U_j += (c5 & 16) ? uj[l][i] : 0.0;
                        U_i += (c5 & 4) ? uj[l][i] : 0.0;
                    }
                }
            };
            // the five continuum rows of a slot (rows that no continuum feeds are not read), k order
            auto load_rows = [&](const unsigned rowOff, const int zm, double (&lr)[5][D]) {
#pragma unroll
                for (int q = 0; q < 5; ++q)
                {
                    if (!(zm & (1 << q)))
                        ls_ld<D>(rRows, vRow, rowOff + (unsigned)q * rowStride, lr[q]);
                    else
                    {
#pragma unroll
                        for (int p = 0; p < D; ++p)
                            lr[q][p] = 0.0;
                    }
                }
            };
#pragma unroll
            for (int s = 0; s < NLc; ++s)
            {
                if (s < NL)
                {
                    double lr[5][D], wphi[D];
                    load_rows(lrow[s], lzm[s], lr);
                    ls_ld<D>(rW, vK, lwO[s], wphi);
                    const double lcw = lcw1[s] * hw;
                    const double lcwOp = (lfl[s] & 1) ? lcw : 0.0; // detailed atoms: rates only
#pragma unroll
                    for (int i = 0; i < D; ++i)
                    {
                        const double v1 = vij[s][i], v2 = vji[s][i], uji = uj[s][i];
                        const double wlamu = lcw * wphi[MI(i)];
                        double etaA = lr[0][MI(i)] + el[s][i], chi_i = lr[1][MI(i)] + xl[s][i], chi_j = lr[2][MI(i)] - xl[s][i];
                        double U_i = lr[3][MI(i)], U_j = lr[4][MI(i)] + uj[s][i];
                        if (NL > 1)
                            couple(ccL[s], i, etaA, chi_i, chi_j, U_i, U_j);
                        const double Ieff = Iv[i] - PsiS[i] * etaA; // compute_full_Ieff
                        const double wop = lcwOp * wphi[MI(i)];
                        accL[s][0][MI(i)] += ((uji + v2 * Ieff) - (PsiS[i] * chi_i * U_j)) * wop;
                        accL[s][1][MI(i)] += ((v1 * Ieff) - (PsiS[i] * chi_j * U_i)) * wop;
                        accL[s][2][MI(i)] += Iv[i] * v1 * wlamu;
                        accL[s][3][MI(i)] += (uji + Iv[i] * v2) * wlamu;
                    }
                    if (lms[s] == 0)
                    {
#pragma unroll
                        for (int i = 0; i < D; ++i)
                            Me[0][MI(i)] += wPsi[i] * el[s][i]; // Meta of the line's atom
                    }
                    else if (lms[s] == 1)
                    {
#pragma unroll
                        for (int i = 0; i < D; ++i)
                            Me[1][MI(i)] += wPsi[i] * el[s][i];
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < NMc; ++m)
            {
                if (m < NM) // mixed continua (always of active atoms)
                {
                    double mr[5][D], mvji[D];
                    load_rows(mrow[m], mzm[m], mr);
                    ls_ld<D>(rRows, vRow, maux[m], mvji);
                    const double wlamu = mcw1[m] * hw;
#pragma unroll
                    for (int i = 0; i < D; ++i)
                    {
                        const double v1 = mb0[m], v2 = mvji[MI(i)], uji = mb2[m] * v2;
                        double etaA = mr[0][MI(i)], chi_i = mr[1][MI(i)], chi_j = mr[2][MI(i)], U_i = mr[3][MI(i)], U_j = mr[4][MI(i)];
                        if (NL > 0)
                            couple(ccM[m], i, etaA, chi_i, chi_j, U_i, U_j);
                        const double Ieff = Iv[i] - PsiS[i] * etaA;
                        accM[m][0][MI(i)] += ((uji + v2 * Ieff) - (PsiS[i] * chi_i * U_j)) * wlamu;
                        accM[m][1][MI(i)] += ((v1 * Ieff) - (PsiS[i] * chi_j * U_i)) * wlamu;
                        accM[m][2][MI(i)] += Iv[i] * v1 * wlamu;
                        accM[m][3][MI(i)] += (uji + Iv[i] * v2) * wlamu;
                    }
                }
            }
        }
        LS_PHASE();
#undef MI
#undef IV
#undef CHI
#undef SRC
    };

    for (int mu = 0; mu < Nmu; ++mu)
    {
        if (dyn.nPass == 2)
            pass(std::false_type{}, mu);
        pass(std::true_type{}, mu);
    }
    if (!RATES)
        return;
    // the slots' integrals of this tile to the workgroup's sums
    if (act)
    {
#pragma unroll
        for (int s = 0; s < NLc; ++s)
            if (s < NL)
            {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int p = 0; p < D; ++p)
                        if (p < c.nV)
                            lds_add(sAcc + lacc[s] + (unsigned)(q * LRD + k0 + p), accL[s][q][p]);
            }
#pragma unroll
        for (int m = 0; m < NMc; ++m)
            if (m < NM)
            {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int p = 0; p < D; ++p)
                        if (p < c.nV)
                            lds_add(sAcc + macc[m] + (unsigned)(q * LRD + k0 + p), accM[m][q][p]);
            }
    }
}

// The tile's wavelengths are complete: J out, dJ, and the pure continua from the angle moments.
// sum_r w_r [Uji + Vji (I_r - Psi_r eta_r) - Psi_r chi_i U_j] with chi, U, Uji, Vji ray-independent and
// eta_r = etaC + etaLine_r  ->  Uji W + Vji (J - M1 etaC - Meta) - M1 chi_i U_j: an exact regrouping of
// compute_full_operator_rates :206-234 for continua whose level sums do not depend on the ray.  The level sums
// (chi_eta_aux_accum :59-109, continua in transition order) live in registers: per tile the host lists which sums have
// several contributors (DevPostProg).
template <int D>
DEVINL void lane_finish(const CONST_AS TileArgs& a, const TileDyn& dyn, const DevTile& td, const int ti, const LaneCtx& c,
                        double* sAcc, unsigned long long* sDj, const double (&Jk)[D], const double (&M1)[D], const double (&Me)[2][D])
{
    const int Ns = c.Ns, Nmu = c.Nmu, LRD = c.LRD, k0 = c.k0, la = c.la;
    const unsigned vK = c.vK, Ns8 = c.Ns8;
    {
        double dj = 0.0;
        if (c.act)
        {
            const GLOBAL_AS double* Jrow = GCD(a.J) + (size_t)la * Ns + k0;
#pragma unroll
            for (int p = 0; p < D; ++p)
                if (p < c.nV)
                {
                    const double Jdag = Jrow[p];
                    GD(a.J)[(size_t)la * Ns + k0 + p] = Jk[p];
                    dj = fmax(dj, fabs(1.0 - Jdag / Jk[p])); // :477-486
                }
            // non-negative doubles order as integers
            atomicMax(sDj + c.rRaw, (unsigned long long)__double_as_longlong(dj));
        }
    }
    const DevLaHeader h = ld_c(CTAB(DevLaHeader, a.laHdr) + td.la0); // the tile's (uniform) structure
    const int nCont = (!dyn.prdOnly && h.nPure > 0) ? h.nMixed + h.nPure : 0;
    if (nCont <= 0)
        return;
    double Wsum = 0.0; // total quadrature weight of the rays of the iteration, in ray order
    for (int m = 0; m < Nmu; ++m)
        for (int d = 0; d < 2; ++d)
            Wsum += 0.5 * CTAB(double, a.wmu)[m];
    const CONST_AS DevContRec* cr = CTAB(DevContRec, a.contRec) + h.off + h.nLine;
    const CONST_AS DevPostProg* pg = CTAB(DevPostProg, a.postProg) + ti;
    const __amdgpu_buffer_rsrc_t rRat = ls_rsrc(a.ratio, a.ratioDoubles * 8ull);
    const int slotOffLa = ((const GLOBAL_AS DevLaHeader*)a.laHdr)[la].off;
    const GLOBAL_AS char* lsB = (const GLOBAL_AS char*)a.slots + (size_t)slotOffLa * sizeof(DevSlot);
    double boltz[D];
    {
        double rT[D];
        ls_ld<D>(c.rG, vK + 8u, 3u * c.gStride, rT);
        const double hc_kl = HC_K / c.wav;
#pragma unroll
        for (int p = 0; p < D; ++p)
            boltz[p] = d_exp_neg(-hc_kl * rT[p]);
    }
    double S8[LWHIP_POST_SUMS][D];
#pragma unroll
    for (int m = 0; m < LWHIP_POST_SUMS; ++m)
#pragma unroll
        for (int p = 0; p < D; ++p)
            S8[m][p] = 0.0;
    // walk 1: the shared level sums, continua in transition order
    for (int q = 0; q < nCont; ++q)
    {
        const DevContRec rc = ld_c(cr + q);
        const unsigned code = (rc.flags & 1) ? pg->code[q] : 0u;
        if (code == 0u)
            continue;
        double ni[D], nj[D], rat[D];
        ls_ld<D>(c.rN, vK, rc.nI, ni);
        ls_ld<D>(c.rN, vK, rc.nJ, nj);
        ls_ld<D>(rRat, vK, rc.rat, rat);
        const double p0 = *(const GLOBAL_AS double*)(lsB + rc.pOff), p2 = *(const GLOBAL_AS double*)(lsB + rc.pOff + 16);
        double chi_c[D], uji[D], eta_c[D];
#pragma unroll
        for (int p = 0; p < D; ++p)
        {
            const double vji = (rat[p] * boltz[p]) * p0;
            chi_c[p] = ni[p] * p0 - nj[p] * vji;
            uji[p] = p2 * vji;
            eta_c[p] = nj[p] * uji[p];
        }
#pragma unroll
        for (int m = 0; m < LWHIP_POST_SUMS; ++m)
        {
            const unsigned kind = (code >> (3 * m)) & 7u;
            if (kind == 0u)
                continue;
#pragma unroll
            for (int p = 0; p < D; ++p)
                S8[m][p] += kind == 1u ? chi_c[p] : kind == 2u ? -chi_c[p] : kind == 3u ? uji[p] : eta_c[p];
        }
    }
    // walk 2: the pure continua
    for (int q = 0; q < nCont; ++q)
    {
        const DevContRec rc = ld_c(cr + q);
        if (!(rc.flags & 2))
            continue;
        double ni[D], nj[D], rat[D];
        ls_ld<D>(c.rN, vK, rc.nI, ni);
        ls_ld<D>(c.rN, vK, rc.nJ, nj);
        ls_ld<D>(rRat, vK, rc.rat, rat);
        const double vij = *(const GLOBAL_AS double*)(lsB + rc.pOff), wla = *(const GLOBAL_AS double*)(lsB + rc.pOff + 8),
                     p2 = *(const GLOBAL_AS double*)(lsB + rc.pOff + 16);
        double vji[D], chi_c[D], uji[D], eta_c[D];
#pragma unroll
        for (int p = 0; p < D; ++p)
        {
            vji[p] = (rat[p] * boltz[p]) * vij;
            chi_c[p] = ni[p] * vij - nj[p] * vji[p];
            uji[p] = p2 * vji[p];
            eta_c[p] = nj[p] * uji[p];
        }
        const unsigned ops = pg->ops[q];
        double o5[5][D]; // eta_A, chi_i, U_j, chi_j, U_i
#pragma unroll
        for (int w = 0; w < 5; ++w)
        {
            const unsigned sel = (ops >> (4 * w)) & 15u;
#pragma unroll
            for (int p = 0; p < D; ++p)
                o5[w][p] = sel == 9u ? chi_c[p] : sel == 10u ? -chi_c[p] : sel == 11u ? uji[p] : sel == 12u ? eta_c[p] : 0.0;
#pragma unroll
            for (int m = 0; m < LWHIP_POST_SUMS; ++m)
                if (sel == (unsigned)(m + 1))
                {
#pragma unroll
                    for (int p = 0; p < D; ++p)
                        o5[w][p] = S8[m][p];
                }
        }
        double* dst = sAcc + (unsigned)(rc.cs * 4 * LRD) + k0;
#pragma unroll
        for (int p = 0; p < D; ++p)
        {
            if (!(c.act && p < c.nV))
                continue;
            if (rc.flags & 1)
            {
                const double Meta = (rc.ms == 0) ? Me[0][p] : (rc.ms == 1) ? Me[1][p] : 0.0;
                const double Jeff = Jk[p] - M1[p] * o5[0][p] - Meta;
                lds_add(dst + p, ((uji[p] * Wsum + vji[p] * Jeff) - (M1[p] * o5[1][p] * o5[2][p])) * wla);
                lds_add(dst + LRD + p, ((vij * Jeff) - (M1[p] * o5[3][p] * o5[4][p])) * wla);
            }
            lds_add(dst + 2 * LRD + p, Jk[p] * vij * wla);
            lds_add(dst + 3 * LRD + p, (uji[p] * Wsum + Jk[p] * vji[p]) * wla);
        }
    }
}

// One task = one tile (up to R wavelengths), every ray of it.
template <int SOLVER, int D, bool RATES>
DEVINL void lane_task(const CONST_AS TileArgs& a, const TileDyn& dyn, const DevTile& td, const int ti, double* sAcc,
                      unsigned long long* sDj, const int lane)
{
    LaneCtx c;
    c.Ns = a.Ns;
    c.Nmu = a.Nmu;
    c.LRD = a.laneLRD;
    c.L = a.L;
    c.lane = lane;
    c.rRaw = lane / a.laneLR;
    c.blk = lane - c.rRaw * a.laneLR;
    c.act = c.rRaw < td.nl;
    c.r = min(c.rRaw, td.nl - 1);
    c.k0 = c.blk * D;
    c.la = td.la0 + c.r;
    c.nV = min(max(c.Ns - c.k0, 0), D);
    c.pBot = c.Ns - 1 - c.k0;
    c.wav = GCD(a.wavelength)[c.la];
    c.vK = (unsigned)c.k0 * 8u;
    c.vRow = (unsigned)(c.r * c.LRD + c.k0) * 8u;
    c.rowStride = (unsigned)(c.L * c.LRD) * 8u;
    c.Ns8 = (unsigned)c.Ns * 8u;
    c.gStride = (unsigned)(c.LRD + 2) * 8u;
    c.rRows = ls_rsrc(a.rows + td.rowBase, (unsigned long long)td.nRowsT * c.rowStride);
    c.rN = ls_rsrc(a.n, (unsigned long long)a.NlevTot * c.Ns8);
    c.rW = ls_rsrc(a.wphi, (unsigned long long)max(a.Nline, 1) * c.Ns8);
    c.rG = ls_rsrc(a.geoT, 4ull * c.gStride);
    double Jk[D], M1[D], Me[2][D];
#pragma unroll
    for (int p = 0; p < D; ++p)
        Jk[p] = M1[p] = Me[0][p] = Me[1][p] = 0.0;
#ifdef LS_ONLY_NL
    lane_rays<SOLVER, D, LS_ONLY_NL, LS_ONLY_NM, RATES>(a, dyn, td, c, sAcc, Jk, M1, Me);
#else
    if (td.nLine == 0)
        lane_rays<SOLVER, D, 0, 0, RATES>(a, dyn, td, c, sAcc, Jk, M1, Me);
    else if (td.nLine == 1 && td.nMixed == 0)
        lane_rays<SOLVER, D, 1, 0, RATES>(a, dyn, td, c, sAcc, Jk, M1, Me);
    else if (td.nLine == 1)
        lane_rays<SOLVER, D, 1, 1, RATES>(a, dyn, td, c, sAcc, Jk, M1, Me);
    else
        lane_rays<SOLVER, D, 2, 0, RATES>(a, dyn, td, c, sAcc, Jk, M1, Me);
#endif
    if (RATES)
        lane_finish<D>(a, dyn, td, ti, c, sAcc, sDj, Jk, M1, Me);
}

// Workgroup = W wavefronts, each takes tiles of the workgroup's chunk in turn.
template <int SOLVER, int D, bool RATES, bool BATCH>
__global__ void __launch_bounds__(LS_MAX_WAVES * 64) lanesweep_kernel(const TileArgs* __restrict__ ap0,
                                                                       const TileArgs* const* __restrict__ apList, const TileDyn dyn)
{
    extern __shared__ double lds[];
    const TileArgs* __restrict__ ap = BATCH ? apList[blockIdx.y] : ap0;
    const CONST_AS TileArgs& a = *(const CONST_AS TileArgs*)ap;
    const int Ns = a.Ns, LRD = a.laneLRD;
    const int maxCT = a.maxCT;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nWaves = blockDim.x >> 6;
    double* sAcc = lds;                                                             // [maxCT][4][LRD]
    unsigned long long* sDj = (unsigned long long*)(lds + (size_t)maxCT * 4 * LRD) + wave * LS_MAX_R; // per wave [R]
    const int chunk = blockIdx.x;
    const int t0 = CTAB(int32_t, a.chunkTile)[chunk], t1 = CTAB(int32_t, a.chunkTile)[chunk + 1];
    if (RATES)
    {
        for (int idx = t; idx < maxCT * 4 * LRD; idx += blockDim.x)
            sAcc[idx] = 0.0;
        __syncthreads();
    }
    const CONST_AS int32_t* tileList = CTAB(int32_t, a.tileList);
    for (int li = t0 + wave; li < t1; li += nWaves)
    {
        const int ti = tileList ? tileList[li] : li;
        const DevTile td = ld_c(CTAB(DevTile, a.tiles) + ti);
        if (RATES && lane < LS_MAX_R)
            sDj[lane] = 0ull;
        lane_task<SOLVER, D, RATES>(a, dyn, td, ti, sAcc, sDj, lane);
        if (RATES && lane < td.nl)
            GD(a.dJ)[td.la0 + lane] = __longlong_as_double((long long)sDj[lane]);
    }
    if (RATES)
    {
        __syncthreads();
        // the chunk's sums go straight into the stage-1 buffer: part = chunk % SPLIT keeps 1/SPLIT of the chunks on one address
        const CONST_AS int32_t* slotTr = CTAB(int32_t, a.slotTr) + (size_t)chunk * maxCT;
        GLOBAL_AS double* r8 = GD(a.red8);
        const int n4 = 4 * Ns;
        const int part = chunk % LWHIP_REDUCE_SPLIT;
        for (int idx = t; idx < maxCT * 4 * LRD; idx += blockDim.x)
        {
            const int cs = idx / (4 * LRD);
            const int rem = idx - cs * 4 * LRD;
            const int q = rem / LRD, k = rem - q * LRD;
            const int tr = slotTr[cs];
            if (tr >= 0 && k < Ns)
                unsafeAtomicAdd((double*)(r8 + ((size_t)tr * LWHIP_REDUCE_SPLIT + part) * n4 + q * Ns + k), sAcc[idx]);
        }
    }
}

template <int SOLVER, int D, bool RATES>
hipError_t launch_ls_t(const TileArgs* devArgs, const TileArgs& a, const TileDyn& dyn, int nChunks, int waves,
                       const TileArgs* const* apList, int nBatch, hipStream_t stream)
{
    const size_t lds = lane_sweep_lds_bytes(a, waves);
    if (lds > 48 * 1024)
    {
        hipError_t e = hipFuncSetAttribute(apList ? (const void*)lanesweep_kernel<SOLVER, D, RATES, true>
                                                  : (const void*)lanesweep_kernel<SOLVER, D, RATES, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
    }
    if (apList)
        hipLaunchKernelGGL((lanesweep_kernel<SOLVER, D, RATES, true>), dim3(nChunks, std::max(nBatch, 1)), dim3(waves * 64), lds, stream,
                           devArgs, apList, dyn);
    else
        hipLaunchKernelGGL((lanesweep_kernel<SOLVER, D, RATES, false>), dim3(nChunks), dim3(waves * 64), lds, stream, devArgs, apList, dyn);
    return hipGetLastError();
}
}

// D: depth points per lane; LR: lanes per ray; R: rays (wavelengths) per wavefront
bool lane_sweep_supported(int Ns, int Nrays, int solver, int* D, int* LR, int* R)
{
    (void)Nrays;
    if (solver != LWHIP_FS_BEZIER3_1D)
        return false;
    const int d = 4;
    const int lr = (Ns + d - 1) / d;
    if (lr < 4 || lr > 64)
        return false;
    *D = d;
    *LR = lr;
    *R = std::min(64 / lr, (int)LS_MAX_R);
    return true;
}

size_t lane_sweep_lds_bytes(const TileArgs& a, int waves)
{
    return sizeof(double) * ((size_t)a.maxCT * 4 * a.laneLRD + (size_t)waves * LS_MAX_R);
}

hipError_t launch_lane_sweep(const TileArgs* devArgs, const TileArgs& a, const TileDyn& dyn, int solver, bool rates, int nChunks,
                             int waves, const TileArgs* const* apList, int nBatch, hipStream_t stream)
{
    if (nChunks <= 0)
        return hipSuccess;
    if (waves < 1 || waves > LS_MAX_WAVES || solver != LWHIP_FS_BEZIER3_1D || a.laneD != 4)
        return hipErrorInvalidValue;
    if (rates)
        return launch_ls_t<LWHIP_FS_BEZIER3_1D, 4, true>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream);
    return launch_ls_t<LWHIP_FS_BEZIER3_1D, 4, false>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream);
}
}

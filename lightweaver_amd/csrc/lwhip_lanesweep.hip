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
#include <atomic>
#include <cstdlib>
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
// N consecutive doubles of the workgroup's LDS copy of the depth arena: byte offset voff (per lane) + soff (uniform)
template <int N> DEVINL void ls_lds(const char* base, unsigned voff, unsigned soff, double (&x)[N])
{
    const double* p = (const double*)(base + voff + soff);
#pragma unroll
    for (int q = 0; q < N; ++q)
        x[q] = p[q];
}
// N consecutive doubles at base (uniform) + soff (uniform) + voff (per lane) bytes of a global array: a 64-bit scalar base and
// a 32-bit lane offset per load, no buffer descriptor (four SGPRs each, and the kernel is short of them).  The pools these
// loads touch are padded: a ray's last block reads up to 24 bytes past its row.
template <int N> DEVINL void ls_gld(const double* base, unsigned voff, unsigned soff, double (&x)[N])
{
    const GLOBAL_AS char* b = (const GLOBAL_AS char*)base + soff;
    asm volatile("" : "+s"(b)); // (the uniform part of the address stays a scalar add: folded into the lane offset it costs a VGPR per load)
    const GLOBAL_AS double* p = (const GLOBAL_AS double*)(b + voff);
#pragma unroll
    for (int q = 0; q < N; ++q)
        x[q] = p[q];
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
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)vv, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(vv >> 32), CTRL, 0xf, 0xf, true);
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
// Afterwards a lane holds the composition of every map from the start of the wave up to and including its own.  Each
// step fetches the partial composite of the lane 1, 2, 4 ... positions upwind (ds_bpermute): the combination tree of a
// lane depends only on its distance from its ray's first point -- everything upwind of that point is multiplied by an
// exact 0 --, so the intensities do not depend on which ray slot of the wavefront a wavelength sits in (row-relative DPP
// shifts would make them: J must not change in the last bit when the wavelength grid is sharded differently).
// (A ray of LR lanes is complete after ceil(log2 LR) steps: a step that reaches beyond the ray's first point multiplies by
// that point's exact 0 and changes nothing, so the last step is skipped -- uniformly -- where no ray is longer than 32 lanes.)
template <bool UP> DEVINL void ls_scan(double& E, double& B, const int lane, const int LR)
{
#pragma unroll
    for (int s = 1; s < 64; s <<= 1)
    {
        if (s == 32 && LR <= 32)
            break;
        // (a lane closer than s to its ray's first point already holds E = 0 exactly -- its composite contains that point's
        // constant map --, so whatever the fetch brings from beyond the ray, or wrapped around the wavefront, changes
        // nothing: no validity select)
        const int src = UP ? lane + s : lane - s;
        const double Ep = ls_bperm(E, src), Bp = ls_bperm(B, src);
        B = fma(E, Bp, B); // (Ep, Bp) is applied first
        E = E * Ep;
    }
}
// 1 / x from v_rcp_f64 and one Newton step
DEVINL double ls_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
// exp(x) for -700 < x <= 0: x = (32 m + j) ln2 / 32 + r, |r| <= ln2 / 64; 2^(j/32) from a 32-entry LDS table, exp(r) by a
// degree-6 polynomial (truncation 3e-18), 2^m by ldexp: seven constants and six dependent FMAs instead of fifteen and
// thirteen
enum { LS_EXP_TAB = 32 };
DEVINL void ls_exp_table_init(double* tab, int t)
{
    if (t < LS_EXP_TAB)
        tab[t] = exp2((double)t * (1.0 / LS_EXP_TAB));
}
DEVINL double ls_exp_neg(double x, const double* tab)
{
    // (the constants are pinned to SGPR pairs where they are used: left alone the compiler parks them in VGPRs for the
    // whole kernel)
    double k0 = 46.166241308446828384, k1 = -2.16608493865351192653e-02, k2 = -5.96317165397058656257e-12; // 32 / ln2; ln2 / 32 head, tail
    double c6 = 1.0 / 720.0, c5 = 1.0 / 120.0, c4 = 1.0 / 24.0, c3 = 1.0 / 6.0;
    asm volatile("" : "+s"(k0), "+s"(k1), "+s"(k2), "+s"(c6), "+s"(c5), "+s"(c4), "+s"(c3));
    const double n = rint(x * k0);
    double r = fma(n, k1, x);
    r = fma(n, k2, r);
    const int ni = (int)n;
    const double T = tab[ni & (LS_EXP_TAB - 1)];
    double p = fma(r, c6, c5);
    p = fma(p, r, c4);
    p = fma(p, r, c3);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(T * p, ni >> 5);
}
enum { LS_MAX_WAVES = 8, LS_MAX_R = 16 };
// The workgroup's LDS accumulators through an LDS-address-space pointer: an address that has been through an `asm` (to keep its
// arithmetic where it is used) is a GENERIC pointer to the compiler otherwise, and the adds become flat_atomic_add_f64 -- correct,
// but ~750 clocks each once somebody waits for them (the fixed-order mode hands a slot on only when its adds are complete:
// measured 12 k clocks per hand-over against ~1 k with ds_add_f64).
#ifdef LS_FLAT_ACC /* diagnostic build: rounds 3 and 4's generic pointer -- flat_atomic_add_f64 -- for the slots' adds */
typedef double ls_lds_double;
DEVINL void ls_lds_add(ls_lds_double* addr, double v) { unsafeAtomicAdd(addr, v); }
#else
typedef __attribute__((address_space(3))) double ls_lds_double;
DEVINL void ls_lds_add(ls_lds_double* addr, double v) { (void)__hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif
#define LS_KEEP(v) asm volatile("" : "+v"(v)) /* the value is computed here, on every lane (keeps selects branch-free) */
#define LS_KEEPI(v) asm volatile("" : "+v"(v))
#define LS_KEEPS(v)                                                                                                                        \
    do                                                                                                                                     \
    {                                                                                                                                      \
        v = __builtin_amdgcn_readfirstlane(v);                                                                                             \
        asm volatile("" : "+s"(v));                                                                                                        \
    } while (0) /* a uniform int re-issued here, in an SGPR: what is derived from it (flag tests) is formed at its use -- hoisted in    \
                   front of the ray loop the tests become 64-bit masks, two dozen SGPRs that are spilled to VGPR lanes and read back   \
                   (v_readlane) in every phase */
#define LS_PHASE() __builtin_amdgcn_sched_barrier(0) /* the scheduler does not move code across phase boundaries */

// Ablation switches of diagnostic builds (tools/build_variant.py; results are WRONG on purpose, used with the instruction
// counters): -DLS_ABL_NO_RAYS skips the ray loop, -DLS_ABL_NO_SLOTFIN the slots' rates from the moments, -DLS_ABL_NO_FINISH the
// pure continua.  The shipped build defines none of them.
#if defined(LS_ABL_NO_RAYS) || defined(LS_ABL_NO_SLOTFIN) || defined(LS_ABL_NO_FINISH)
#define LS_ABLATION 1
#endif
// Phase timers of a diagnostic build (-DLS_TIMING): shader clocks per phase, summed over the rays of a tile, written to
// TileArgs::lsDbg[tile][8] = setup, gather, stencils, coefficients, scan, rates, finish, total.
#ifdef LS_TIMING
#define LS_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define LS_TACC(acc, a0, a1) acc += (a1) - (a0)
#else
#define LS_T(var)
#define LS_TACC(acc, a0, a1)
#endif
// What a lane is in its tile, and the buffers every phase reads.
struct LaneCtx
{
    int Ns, Nmu, LRD, LR, L;
    int lane, rRaw, blk, r, k0, la, nV, pBot, ti; // nV: valid points of the block; pBot: index of depth Ns - 1 in the block (if in range)
    bool act;
    double wav;
    unsigned vK, vRow, rowStride, Ns8, gStride;
    unsigned oN, oW, oRat, oG;                 // byte offsets of the population, w_phi, nStar-ratio pools and the geometry in the depth arena
    const double* rRows;                       // the tile's row block
    const char* sDep;                          // the workgroup's LDS copy of the depth arena
    double* sRow3;                             // the wavefront's LDS copy of its tile's rows chi_C, eta_C, sigma J: [3][64][D]
    double* sB;                                // the wavefront's LDS row [R][4]: Planck functions of the rays' end points
    const double* sExp;                        // the workgroup's 2^(j/32) table
    double* sPP;                               // the wavefront's LDS block [continuum][ray][3]: per-(continuum, wavelength) parameters of the finish
    int* sTurn;                                // fixed-order mode: the workgroup's turn counters [slot][D], else null
    double* dbgWait;                           // -DLS_TIMING: clocks the workgroup's wavefronts waited for their turns at line / mixed slots
    int sub;                                   // which wavefront of the tile's group this is (0 unless the rays are split)
};

// ---- fixed-order accumulation with several wavefronts per workgroup (LWHIP_OPT_DETERMINISTIC, round 5) --------------------
// The wavefronts of a workgroup add their tiles' integrals to the SAME LDS accumulators (neighbouring tiles share their
// continua, a line spans many tiles); left alone, the order of those atomic adds follows the wavefronts' timing and the
// last bits of Gamma / Rij / Rji change from run to run.  In the fixed-order mode every (accumulator slot, point of the
// block) has a turn counter in LDS, and the host gives every task, per slot, the value the counter holds when it is this
// task's turn (lwhip_tables.hip: tasks of a round in order of expected cost, line / mixed-continuum adds of a round before
// its pure-continuum adds): a wavefront waits for that value before its first add to the slot and counts the slot on after
// its last one.  Nobody waits for a task that is behind it in the order, so the waits cannot deadlock (also across the
// barriers of a split tile: the order never puts an add that follows a barrier in front of one that precedes it); the
// wavefronts walk the shared continua in the same (transition) order, so they follow each other through the slots like a
// pipeline.  The sums are then formed in ONE order, whatever the timing: bit-reproducible, with four wavefronts per
// workgroup (round 3's fixed-order mode ran one, at a quarter of the occupancy: 2.7 x the default's time).
// turn waits of this workgroup that gave up (see ls_turn_wait): the workgroup adds the count to TileArgs::turnLate when it ends,
// and the host turns a non-zero total into an error instead of silently unordered sums
DEVINL int* ls_late_counter()
{
    __shared__ int late;
    return &late;
}
template <int D> DEVINL void ls_turn_wait(int* sTurn, const int cs, const int p0, const int np, const int turn, double* dbgClocks = nullptr)
{
    if (!sTurn)
        return;
#ifdef LS_TIMING
    const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
    // (LDS only: the wavefront's LDS operations reach the LDS in program order, so the adds that follow are behind the read
    // that saw the turn -- no fence, which would also wait for the task's global loads and stores)
    // (bounded: a wait that never ends would take the device down with it; ~2^20 polls are tens of milliseconds, a thousand
    // times a task -- past that the wavefront goes on, and the sums are merely in timing order)
    // (an LDS pointer, not a generic one: volatile loads through a generic pointer are flat_load ... sc0 sc1 with a wait for
    // ALL of the wavefront's global memory traffic behind each -- measured: 21 k clocks per hand-over instead of ~1 k)
    volatile __attribute__((address_space(3))) int* t = (volatile __attribute__((address_space(3))) int*)(sTurn + cs * D + p0);
    bool seen = false;
    for (int spin = 0; spin < (1 << 20); ++spin)
    {
        int lo = 0x7fffffff;
        for (int p = 0; p < np; ++p)
            lo = min(lo, t[p]);
        if (__builtin_amdgcn_readfirstlane(lo) >= turn)
        {
            seen = true;
            break;
        }
#ifndef LS_TURN_SLEEP
#define LS_TURN_SLEEP 1
#endif
        __builtin_amdgcn_s_sleep(LS_TURN_SLEEP);
    }
    if (!seen && (threadIdx.x & 63) == 0) // (the wait timed out: the add goes ahead out of turn -- counted, reported by the host)
        __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)ls_late_counter(), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
#ifdef LS_TIMING
    if (dbgClocks && (threadIdx.x & 63) == 0)
        unsafeAtomicAdd(dbgClocks, (double)(__builtin_amdgcn_s_memtime() - tw0));
#endif
}
template <int D> DEVINL void ls_turn_done(int* sTurn, const int cs, const int p0, const int np, const int lane)
{
    if (!sTurn)
        return;
    // (the adds are LDS atomics without return: counted by lgkmcnt, complete when it reaches zero; no fence -- it would wait
    // for the task's outstanding global loads and stores as well, a memory round trip per slot)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < np)
        __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)(sTurn + cs * D + p0 + lane), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The rays of one tile: NL / NM line / mixed-continuum slots.  Leaves J and the Psi* moments of the lane's points in
// Jk / M1 / Me (Me[0] alone if the tile has fewer than two lines: its moment slot is returned) and adds the slots'
// integrals to the workgroup's accumulators.
//
// Rates of the slots (compute_full_operator_rates, SimdFullIterationTemplates.hpp:192-234).  Per ray r every integrand of a
// line or mixed-continuum slot is a polynomial in the rays' profiles phi_l(r) with ray-independent coefficients:
//   V_ij = c1 phi, V_ji = c2 phi, U_ji = cu phi, chi_line = cx phi, eta_line = ce phi   (c* from n_i, n_j, rho and the line's constants),
//   eta_A, chi_i, chi_j, U_i, U_j = continuum rows + multiples of the coupled lines' phi_l,
// so the sums over the rays need only the angle moments
//   P0_l = sum w phi_l,  P1_l = sum w phi_l I,  Q_l = sum w Psi* phi_l,  Q_ll' = sum w Psi* phi_l phi_l'   (+ J, M1 = sum w Psi*, W = sum w)
// -- an exact regrouping of the reference's ray sum, as the pure continua's (lane_finish).  The ray loop accumulates the
// moments in registers (6 flops per point, ray and line; no continuum rows, no populations, no LDS atomics); the slots'
// integrals are formed ONCE per tile after the last ray (slot_finish_half) and are linear in the moments, so the
// wavefronts of a split tile each add theirs.  Hybrid PRD (rho follows the ray) keeps the ray-by-ray form.
template <int SOLVER, int D, int NL, int NM, int MODE>
DEVINL int lane_rays(const CONST_AS TileArgs& a, const TileDyn& dyn, const DevLaneTile& tr, const LaneCtx& c, double* sAcc, const int q0,
                     const int q1, double (&Jk)[D], double (&M1)[D], double (&Me)[2][D])
{
    constexpr bool RATES = (MODE & 1) != 0, HPRD = (MODE & 2) != 0; // MODE: bit 0 rates, bit 1 hybrid PRD (compiled in only where used)
    constexpr bool MOM = RATES && !HPRD;                            // the slots' rates from angle moments
    // Two lines (a blend): the moments of both would be 72 registers, and this kernel has none to spare and must not spill
    // (what a value spilled inside divergent control flow did).  Every integral is LINEAR in the moments,
    // so such a tile walks its rays TWICE -- run 0 accumulates J, M1, W and the moments of line 0 (and Q_01), run 1 those of
    // line 1 alone -- and the slots' integrals are formed after each run from what that run accumulated; the gather takes the
    // populations from LDS ray by ray and the moments re-read the profiles (an L1 hit) instead of holding cx / ce / phi.
    // Blends are rare (28 of 3 429 tiles on the bench grid); they pay twice the ray work, the other kinds nothing.
    // NL < 0: the GENERIC kind (round 5) -- any number of lines (<= 8) and mixed continua (<= 8), for the tiles no compiled kind
    // covers.  The gather walks all lines of the tile per ray (a uniform loop over the tile's slot records); the rays are
    // walked ONCE, whatever the number of lines: the moments that do not fit the registers live in a scratch block in global
    // memory (below); after the last ray every slot's integrals are formed from all of them (gen_slot_finish).  Not for
    // hybrid PRD (rates ray by ray).
    constexpr bool GEN = NL < 0;
    static_assert(!(GEN && HPRD), "the generic kind forms its rates from the angle moments");
    constexpr bool HOLD = !HPRD && !GEN && NL < 2;
    constexpr int NRUN = (MOM && NL > 1) ? 2 : 1;
    int run = 0;
    constexpr int NLc = NL > 0 ? NL : 1, NMc = NM > 0 ? NM : 1;
    // (generic kind: the tile's slot records and, per lane, its wavelength's slot parameters)
    const int gnL = GEN ? (int)tr.td.nLine : 0, gnM = GEN ? (int)tr.td.nMixed : 0;
    const CONST_AS DevTileSlot* gSlots = CTAB(DevTileSlot, a.tslots) + tr.td.slotOff;
    const GLOBAL_AS DevSlot* gPar = nullptr; // the lane's wavelength: slot record e of the tile is gPar[gSlots[e].e]
    if (GEN)
        gPar = (const GLOBAL_AS DevSlot*)a.slots + ((const GLOBAL_AS DevLaHeader*)a.laHdr)[c.la].off;
#ifdef LS_TIMING
    unsigned long long lsTurnExit = 0; // (when the wavefront left its first turn wait: the critical section's length is reported)
#endif
    // The moments of a generic tile: line 0's in the registers every kind uses (mP0, mP1, mQ1, mQ2), those of the further lines
    // and every cross moment in the wavefront's block of the moment scratch in global memory (TileArgs::momS), updated ray by
    // ray by the lane that owns them (an L1 / L2 resident read-modify-write: 20 doubles per lane and ray for two lines --
    // against walking the rays once more per further line, which is what round 5's first version did).  Arrays, each [D][64]:
    //   line l >= 1:  4 (l - 1) + { P0, P1, Q, Q_ll }        pair m < l:  4 (nL - 1) + l (l - 1) / 2 + m  (Q_ml)
    //   then, filled by the slots' integrals: per line l { c2, cu, cx, ce } of the point being finished
    GLOBAL_AS double* gMom = nullptr;
    bool gFirstRay = true; // (the first ray of the tile stores, the others add: the scratch needs no clearing)
    if (GEN && a.momS)
        gMom = (GLOBAL_AS double*)a.momS + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * ((size_t)a.momA * D * 64) + c.lane;
    const int gPairBase = 4 * (gnL - 1), gCoefBase = gPairBase + gnL * (gnL - 1) / 2;
    auto gm_add = [&](const int arr, const int p, const double v) {
        GLOBAL_AS double* q = gMom + (size_t)(arr * D + p) * 64;
        *q = gFirstRay ? v : *q + v;
    };
    const int Ns = c.Ns, Nmu = c.Nmu, LR = c.LR;
    const int la = c.la, k0 = c.k0;
    const bool act = c.act;
    const unsigned rowStride = c.rowStride, Ns8 = c.Ns8;
    const double* rRows = c.rRows;
    const char* sDep = c.sDep;

    // ---- slot structure (uniform: the tile record) and per-lane parameters (the lane's ray record) ------------------------
    const DevTile& td = tr.td;
    const GLOBAL_AS DevLaneRay* lray = (const GLOBAL_AS DevLaneRay*)a.laneRays + (size_t)c.ti * c.L + min(c.rRaw, c.L - 1);
    unsigned lnI[NLc], lnJ[NLc], lwO[NLc], lrow[NLc], lacc[NLc];
    int lgi[NLc], lgj[NLc], lat[NLc], lfl[NLc], lms[NLc], lzm[NLc];
    double la0[NLc], la2[NLc], la3[NLc], lcw1[NLc]; // per lane
    const double* rPhi[NLc];
    const double* rRho[NLc];
    const lwhip_rho_coeff* rHc[NLc]; // hybrid PRD: the line's rho interpolation coefficients at the tile's first wavelength
    const unsigned vPhi0 = (unsigned)c.r * (unsigned)(Nmu * 2) * Ns8 + c.vK;
    const unsigned vRho0 = (unsigned)c.r * Ns8 + c.vK;
#pragma unroll
    for (int s = 0; s < NLc; ++s)
    {
        lnI[s] = lnJ[s] = lwO[s] = lrow[s] = lacc[s] = 0;
        lgi[s] = lgj[s] = lat[s] = -1 - s;
        lfl[s] = 0;
        lms[s] = -1;
        lzm[s] = 31;
        la0[s] = la2[s] = la3[s] = lcw1[s] = 0.0;
        rPhi[s] = a.phi;
        rRho[s] = a.rho;
        rHc[s] = nullptr;
        if (s < NL)
        {
            const DevTileSlot& u = tr.slot[s];
            lnI[s] = c.oN + (unsigned)u.gi * Ns8;
            lnJ[s] = c.oN + (unsigned)u.gj * Ns8;
            lwO[s] = c.oW + (unsigned)u.row * Ns8;
            lrow[s] = (unsigned)u.r0 * rowStride;
            lacc[s] = (unsigned)(u.cs * 4 * c.LRD);
            lgi[s] = u.gi;
            lgj[s] = u.gj;
            lat[s] = u.atom;
            lfl[s] = u.flags;
            lms[s] = (u.flags & 1) ? u.ms : -1;
            lzm[s] = (u.flags & 1) ? u.zmask : 31;
            rPhi[s] = a.phi + u.phiOff;
            if (u.flags & 2)
                rRho[s] = a.rho + u.rhoOff;
            if (HPRD && (u.flags & 2) && u.hCoefOff >= 0 && a.hRho)
                rHc[s] = a.hRho + u.hCoefOff;
            if (HPRD)
            {
                la0[s] = lray->p[s][0];
                la2[s] = lray->p[s][2];
                la3[s] = lray->p[s][3];
                lcw1[s] = lray->p[s][1] * PI4_HC; // times 0.5 wmu per ray
            }
        }
    }
    // (not hybrid PRD) a line's per-(transition, wavelength) parameters hnu/4pi Bij, Bji/Bij, Aji/Bji, w_lambda 4pi/hc are
    // re-read from the lane's ray record where they are used (setup, a blend's gather, the slots' integrals): an L1 / L2 hit
    // instead of 8 registers per line held over the rays
    struct LinePar
    {
        double a0, a2, a3, cw1;
    };
    auto line_par = [&](const int s) -> LinePar {
        const GLOBAL_AS DevLaneRay* q = lray;
        asm volatile("" : "+v"(q)); // (re-issued here: not kept from an earlier use)
        return LinePar{ q->p[s][0], q->p[s][2], q->p[s][3], q->p[s][1] * PI4_HC };
    };
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
            const DevTileSlot& u = tr.slot[NL + m < 2 ? NL + m : 1];
            mrow[m] = (unsigned)u.r0 * rowStride;
            maux[m] = (unsigned)max((int)u.rAux, 0) * rowStride;
            macc[m] = (unsigned)(u.cs * 4 * c.LRD);
            mgi[m] = u.gi;
            mgj[m] = u.gj;
            mat[m] = u.atom;
            mzm[m] = u.zmask;
            mb0[m] = lray->p[NL + m < 2 ? NL + m : 1][0];
            mb2[m] = lray->p[NL + m < 2 ? NL + m : 1][2];
            mcw1[m] = lray->p[NL + m < 2 ? NL + m : 1][1];
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

    // rho of a PRD line at N points of the block starting at point H, ray independent: its row of this wavelength
    auto load_rho0 = [&](const int s, auto nTag, const unsigned H, double* rho) {
        constexpr int N = decltype(nTag)::value;
        if (!(lfl[s] & 2))
        {
#pragma unroll
            for (int p = 0; p < N; ++p)
                rho[p] = 1.0;
        }
        else
        {
            double tmp[N];
            ls_gld<N>(rRho[s], (unsigned)c.r * Ns8 + c.vK + H * 8u, 0u, tmp);
#pragma unroll
            for (int p = 0; p < N; ++p)
                rho[p] = tmp[p];
        }
    };
    // (not hybrid PRD) the lines' opacity and emissivity per unit profile at the lane's points, for the gather of every ray:
    // chi_line = n_i V_ij - n_j V_ji = cx phi, eta_line = n_j U_ji = ce phi with V_ij = la0 phi, V_ji = la2 rho V_ij,
    // U_ji = la3 V_ji (Transition::uv LwTransition.hpp:93-144, chi_eta_aux_accum :59-109)
    double cxr[NLc][D], cer[NLc][D];
#pragma unroll
    for (int s = 0; s < NLc; ++s)
    {
#pragma unroll
        for (int p = 0; p < D; ++p)
            cxr[s][p] = cer[s][p] = 0.0;
        if (HOLD && s < NL)
        {
            double ni[D], nj[D], rho[D];
            ls_lds<D>(sDep, c.vK, lnI[s], ni);
            ls_lds<D>(sDep, c.vK, lnJ[s], nj);
            load_rho0(s, std::integral_constant<int, D>{}, 0u, rho);
            const LinePar lp = line_par(s);
#pragma unroll
            for (int p = 0; p < D; ++p)
            {
                const double c2 = (lp.a2 * rho[p]) * lp.a0;
                cxr[s][p] = ni[p] * lp.a0 - nj[p] * c2;
                cer[s][p] = nj[p] * (lp.a3 * c2);
            }
        }
    }
    // the angle moments of the slots' rates (see above) of the line the current run accumulates: P0, P1, Q, Q_ll; Q_01 of a
    // blend (run 0); W
    double mP0[D], mP1[D], mQ1[D], mQ2[D], mQ12[D], mW = 0.0;
#pragma unroll
    for (int p = 0; p < D; ++p)
        mP0[p] = mP1[p] = mQ1[p] = mQ2[p] = mQ12[p] = 0.0;

    const bool storeDepth = RATES && a.storeDepth && !dyn.prdOnly;
    const double lamf = dyn.lambdaIterate ? 0.0 : 1.0;
    // which moment row a line's Psi* eta goes to: with fewer than two lines row 0 stands for the line's moment slot
    const int meSlot0 = NL == 1 ? lms[0] : 0;

    // (the ray-independent rows chi_C, eta_C, sigma J of the lane's points are in the wavefront's LDS block: lane_pre)
    // boundary intensities of the rays (wrappers FormalScalar.cpp:471-666): Planck functions of the two end points of
    // this lane's wavelength, parked in the wavefront's LDS row (read by the lane that holds a ray's first point)
    // (the host's table: once per atmosphere and wavelength, not per task)
    // (EVERY lane of a ray writes the ray's four values -- the same ones, to the same address: no lane-divergent `if` here.
    // The register allocator of ROCm 7.2 put spill stores at the top of the block that joined this `if`, in front of the
    // instruction that switches the other lanes back on: stored for three lanes, reloaded for 64 -- tools/spill_lint.py.)
    // (c.rRaw < LS_MAX_R for every lane: LR >= 4 lanes per ray, lane_sweep_supported)
    {
        const GLOBAL_AS double* bp = GCD(a.bcPlanck) + (size_t)la * 4;
        double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
        if (a.upperBc == LWHIP_BC_THERMALISED && dyn.nPass == 2)
        {
            b0 = bp[0];
            b1 = bp[1];
        }
        if (a.lowerBc == LWHIP_BC_THERMALISED)
        {
            b2 = bp[2];
            b3 = bp[3];
        }
        c.sB[c.rRaw * 4 + 0] = b0;
        c.sB[c.rRaw * 4 + 1] = b1;
        c.sB[c.rRaw * 4 + 2] = b2;
        c.sB[c.rRaw * 4 + 3] = b3;
    }

    // The integrals of the slots at this lane's points go to the workgroup's LDS accumulators ray by ray:
    // [slot][quantity][point of the block][block], so that the lanes of a ray hit consecutive addresses (the R rays of
    // the wavefront meet at one address: an LDS atomic without return serialises them).
    double* accBase = sAcc + c.blk;

    // the profiles are the one stream that comes from HBM: a ray's block is requested one ray ahead
    double phiN[NLc][D];
    auto prefetch_first = [&]() {
        const unsigned sRay0 = (unsigned)(dyn.nPass == 2 ? q0 : 2 * q0 + 1) * Ns8; // the wavefront's first ray
#pragma unroll
        for (int s = 0; s < NLc; ++s)
        {
            if (s < NL && (HOLD || HPRD))
                ls_gld<D>(rPhi[s], vPhi0, sRay0, phiN[s]);
            else
            {
#pragma unroll
                for (int p = 0; p < D; ++p)
                    phiN[s][p] = 0.0;
            }
        }
    };

    // ---- continuum-only tiles (no line, no mixed continuum; round 5) ---------------------------------------------------------
    // chi and S of such a wavelength are the same for every ray, the slopes per unit path scale with mu and the path with 1 / mu:
    // the control points of chi (ds/3 d chi/ds) and of S (dtau/3 dS/dtau) do not depend on the ray at all, the intervals'
    // optical depths are T / mu, dS/dtau is mu G.  And the up direction walks the same intervals: T is symmetric, the
    // derivatives change sign (also at the end points: the one-sided rules of the two directions are each other's mirror).
    // So gather + stencils -- 31 k of such a task's 77 k clocks -- are formed ONCE per tile at mu = 1, in depth order
    // (cont_unit), and a ray scales and, going up, mirrors them.
#ifdef LS_NO_CONT_HOIST
    constexpr bool CONT = false;
#else
    constexpr bool CONT = !GEN && !HPRD && NL == 0 && NM == 0 && SOLVER == LWHIP_FS_BEZIER3_1D;
#endif
    // ---- one-line tiles whose profiles do not depend on the direction (a static atmosphere: phi(mu, up) = phi(mu, down)) ----
    // The two rays of an angle then see the same chi and S, so the same holds for the pair: gather and stencils once per
    // ANGLE (cont_unit with the line's opacity of that angle), the down and the up ray scale / mirror them.  The host sets
    // TileDyn::phiSym when every line's profile is direction-symmetric (checked when the profiles are uploaded or generated);
    // wavefronts that hold only one direction of an angle (split rays) take the general path.
#ifdef LS_NO_PAIR_HOIST
    constexpr bool PAIRABLE = false;
#else
    constexpr bool PAIRABLE = !GEN && HOLD && NL == 1 && SOLVER == LWHIP_FS_BEZIER3_1D;
#endif
    double hChi[D], hS[D], hRch[D], hDS[D]; // the lane's points, depth order
    double hPhi[D];                                  // (pairs: the line's profile of the angle, for the moments)
    double hT[D + 1], hRT[D + 1];                    // hT[q]: optical depth at mu = 1 of the interval (k0 + q - 1, k0 + q), and 1 / hT
    double hChiLo = 0.0, hChiHi = 0.0, hSLo = 0.0, hSHi = 0.0, hDSLo = 0.0, hDSHi = 0.0; // the neighbouring lanes' edge points
    auto cont_unit = [&](const int muLine) {
        const double* r3 = c.sRow3 + c.lane * D;
        double dh[D + 1], rdh[D + 1], rsum[D];
        ls_lds<D + 1>(sDep, c.vK, c.oG, dh); // entry k + 1 holds |h_k - h_k+1|: dh[q] = the interval (k0 + q - 1, k0 + q)
        ls_lds<D + 1>(sDep, c.vK, c.oG + c.gStride, rdh);
        ls_lds<D>(sDep, c.vK + 8u, c.oG + 2u * c.gStride, rsum);
        double lphi[D];
#pragma unroll
        for (int p = 0; p < D; ++p)
            lphi[p] = 0.0;
        if constexpr (PAIRABLE)
        {
            // the line's profile of this angle (its down ray's block; the up ray's is the same) -- requested here, not one ray
            // ahead (a request one ANGLE ahead was measured: 14 more spilled registers, no gain)
            ls_gld<D>(rPhi[0], vPhi0, (unsigned)(muLine * 2) * Ns8, lphi);
#pragma unroll
            for (int p = 0; p < D; ++p)
                hPhi[p] = lphi[p];
        }
#pragma unroll
        for (int p = 0; p < D; ++p)
        {
            double cC = r3[p], eC = r3[64 * D + p];
            const double sc = r3[2 * 64 * D + p];
            if constexpr (PAIRABLE)
            {
                cC = fma(cxr[0][p], lphi[p], cC);
                eC = fma(cer[0][p], lphi[p], eC);
            }
            const double rc = ls_rcp(cC);
            hChi[p] = cC;
            hRch[p] = rc;
            hS[p] = (eC + sc) * rc;
        }
        hChiLo = ls_shift1<true>(hChi[D - 1]);
        hSLo = ls_shift1<true>(hS[D - 1]);
        hChiHi = ls_shift1<false>(hChi[0]);
        hSHi = ls_shift1<false>(hS[0]);
        const int pBot = c.pBot, blk = c.blk, nV = c.nV;
        auto dFirst = [&](int i) { return i == 0 && blk == 0; };
        auto dLast = [&](int i) { return i == pBot; };
        auto dOk = [&](int i) { return i < nV; };
        auto chiK = [&](int i) { return i < 0 ? hChiLo : i >= D ? hChiHi : hChi[i < 0 ? 0 : i >= D ? D - 1 : i]; };
        auto srcK = [&](int i) { return i < 0 ? hSLo : i >= D ? hSHi : hS[i < 0 ? 0 : i >= D ? D - 1 : i]; };
        // (the stencils of the ray loop below for the down direction with zmu = muz = 1: FormalScalar.cpp:209-325)
        double sl[D + 1];
#pragma unroll
        for (int i = 0; i <= D; ++i)
            sl[i] = (chiK(i) - chiK(i - 1)) * rdh[i];
        double dx[D];
#pragma unroll
        for (int i = 0; i < D; ++i)
        {
            const double P0 = fabs((sl[i] * dh[i + 1] + sl[i + 1] * dh[i]) * rsum[i]);
            const double v = (copysign(1.0, sl[i + 1]) + copysign(1.0, sl[i])) * fmin(fabs(sl[i]), fmin(fabs(sl[i + 1]), 0.5 * P0));
            dx[i] = dFirst(i) ? sl[i + 1] : dLast(i) ? sl[i] : v;
        }
        const double dxP = ls_shift1<true>(dx[D - 1]);
        double sS[D];
#pragma unroll
        for (int i = 0; i < D; ++i)
        {
            const double cu = chiK(i - 1), cd = chiK(i);
            const double dxu = i == 0 ? dxP : dx[i > 0 ? i - 1 : 0];
            const double ds3 = dh[i] * (1.0 / 3.0);
            const double Cuw = cu + ds3 * dxu;
            const double C0 = cd - ds3 * dx[i];
            double t = dh[i] * (cu + cd + Cuw + C0) * 0.25;
            t = (dFirst(i) || !dOk(i)) ? 1.0 : t;
            hT[i] = t;
            hRT[i] = ls_rcp(t);
            sS[i] = (srcK(i) - srcK(i - 1)) * hRT[i];
        }
        const double dtN = ls_shift1<false>(hT[0]), rtN = ls_shift1<false>(hRT[0]), sSN = ls_shift1<false>(sS[0]);
        // (the interval to the next lane's first point; past the column's end it is as inert as a padding point's)
        const bool okN = c.k0 + D < Ns;
        hT[D] = okN ? dtN : 1.0;
        hRT[D] = okN ? rtN : 1.0;
#pragma unroll
        for (int i = 0; i < D; ++i)
        {
            const double dtd = i == D - 1 ? dtN : hT[i < D - 1 ? i + 1 : 0];
            const double sSd = i == D - 1 ? sSN : sS[i < D - 1 ? i + 1 : 0];
            const double P0 = fabs((sS[i] * dtd + sSd * hT[i]) * ls_rcp(hT[i] + dtd));
            const double v = (copysign(1.0, sSd) + copysign(1.0, sS[i])) * fmin(fabs(sS[i]), fmin(fabs(sSd), 0.5 * P0));
            // the column's first point takes the slope of the interval below it, its last point that of the interval above it
            // (what the up direction, whose first point it is, takes with the opposite sign: the down direction closes linearly)
            hDS[i] = dFirst(i) ? sSd : dLast(i) ? sS[i] : v;
        }
        hDSLo = ls_shift1<true>(hDS[D - 1]);
        hDSHi = ls_shift1<false>(hDS[0]);
    };

#ifdef LS_TIMING
    unsigned long long tmA = 0, tmB = 0, tmC = 0, tmS = 0, tmD = 0, tmBprev = 0;
#endif
    // ---- one ray of every wavelength of the tile ------------------------------------------------------------------------
    auto pass = [&](auto upTag, auto reuseTag, const int mu, const unsigned sRayNext) {
        constexpr bool UP = decltype(upTag)::value;
        constexpr bool REUSE = CONT || decltype(reuseTag)::value; // chi, S and the stencils come from cont_unit
        constexpr int dirI = UP ? 1 : 0;
        LS_T(tp0);
#define MI(i) (UP ? D - 1 - (i) : (i)) /* march index <-> index in the lane's k-ordered block (an involution) */
#define IV(i) (UP ? D - (i) : (i))     /* interval i (march point i-1 -> i) -> index in dhK / rdhK */
        const double muz = CTAB(double, a.muz)[mu];
        const double zmu = CTAB(double, a.muz)[Nmu + mu]; // 1 / muz (the host's division)
        const double hw = act ? 0.5 * CTAB(double, a.wmu)[mu] : 0.0;
        const unsigned sRay = (unsigned)(mu * 2 + dirI) * Ns8;
        const int bcType = UP ? a.lowerBc : a.upperBc;
        // the ray's first point is depth 0 (down: march index 0 of block 0) or depth Ns - 1 (up), its last the other one
        // (re-derived per ray from the lane's integers: hoisted out of the ray loop they would sit in two dozen SGPRs)
        int pBot = c.pBot, blk = c.blk, nV = c.nV;
        LS_KEEPI(pBot);
        LS_KEEPI(blk);
        LS_KEEPI(nV);
        auto is_first = [&](int i) { return UP ? MI(i) == pBot : (i == 0 && blk == 0); };
        auto is_last = [&](int i) { return UP ? (i == D - 1 && blk == 0) : MI(i) == pBot; };
        auto is_ok = [&](int i) { return MI(i) < nV; };

        // (the address registers are re-issued per phase: otherwise the compiler keeps what one phase loaded alive for the
        // next phase, the other direction and the next ray instead of re-reading 32 bytes from the L1)
        // rho of a PRD line at N points of the block starting at point H: its row of this wavelength, or -- hybrid PRD --
        // interpolated to the ray's rest wavelength between two rows of the line (Transition::uv, LwTransition.hpp:116-127)
        auto load_rho = [&](const int s, auto nTag, const unsigned H, const unsigned vRho, double* rho) {
            constexpr int N = decltype(nTag)::value;
            int fl = lfl[s];
            LS_KEEPS(fl);
            if (!(fl & 2))
            {
#pragma unroll
                for (int p = 0; p < N; ++p)
                    rho[p] = 1.0;
            }
            else if (HPRD && rHc[s])
            {
                const GLOBAL_AS lwhip_rho_coeff* cc = (const GLOBAL_AS lwhip_rho_coeff*)rHc[s]
                                                      + ((size_t)c.r * (Nmu * 2) + (size_t)(mu * 2 + dirI)) * Ns + (k0 + H);
                const GLOBAL_AS double* rl = (const GLOBAL_AS double*)rRho[s] + (k0 + H);
#pragma unroll
                for (int p = 0; p < N; ++p)
                {
                    const int i0 = cc[p].i0, i1 = cc[p].i1;
                    const double fr = cc[p].frac;
                    rho[p] = (1.0 - fr) * rl[(size_t)i0 * Ns + p] + fr * rl[(size_t)i1 * Ns + p];
                }
            }
            else
            {
                double tmp[N];
                ls_gld<N>(rRho[s], vRho, 0u, tmp);
#pragma unroll
                for (int p = 0; p < N; ++p)
                    rho[p] = tmp[p];
            }
        };
        // ---- A. gather: Transition::uv + chi_eta_aux_accum + compute_source_fn at the lane's points (march order) ----
        double chi[D], S[D], rch[D];
        double phiK[NLc][D]; // the ray's profiles at the lane's points (k order): kept for the moments
        double dhK[D + 1], rdhK[D + 1], rsumK[D]; // geometry of the block (used by the stencils, requested with the gather's loads)
        if constexpr (REUSE)
        {
            // (continuum-only tile, or the pair of a direction-symmetric angle: chi, S, 1 / chi from cont_unit, in march order;
            // dh for the boundary steps)
#pragma unroll
            for (int p = 0; p < D; ++p)
            {
                chi[MI(p)] = hChi[p];
                rch[MI(p)] = hRch[p];
                S[MI(p)] = hS[p];
                if constexpr (NL > 0)
                    phiK[0][p] = hPhi[p];
            }
            ls_lds<D + 1>(sDep, c.vK, c.oG, dhK);
            if (storeDepth && act)
            {
                const size_t o = ((size_t)(la * Nmu + mu) * 2 + dirI) * Ns + k0;
#pragma unroll
                for (int p = 0; p < D; ++p)
                    if (p < c.nV)
                    {
                        // (eta formed again, with cont_unit's operations, instead of four registers held for a rare output)
                        double eC = (c.sRow3 + c.lane * D)[64 * D + p];
                        if constexpr (PAIRABLE)
                            eC = fma(cer[0][p], hPhi[p], eC);
                        GD(a.depthChi)[o + p] = hChi[p];
                        GD(a.depthEta)[o + p] = eC;
                    }
            }
        }
        else
        {
            unsigned vK = c.vK, vRow = c.vRow, vPhi = vPhi0, vRho = vRho0;
            LS_KEEPI(vK);
            LS_KEEPI(vRow);
            LS_KEEPI(vPhi);
            LS_KEEPI(vRho);
            double cC[D], eC[D];
            const double* r3 = c.sRow3 + c.lane * D;
#pragma unroll
            for (int p = 0; p < D; ++p)
            {
                cC[p] = r3[p];
                eC[p] = r3[64 * D + p];
            }
            if constexpr (GEN)
            {
                // every line of the tile: Transition::uv + chi_eta_aux_accum (LwTransition.hpp:93-144, :59-109)
                auto gather_line = [&](const int l) {
                    const DevTileSlot u = ld_c(gSlots + l);
                    double phi[D], ni[D], nj[D], rho[D];
                    ls_lds<D>(sDep, vK, c.oN + (unsigned)u.gi * Ns8, ni);
                    ls_lds<D>(sDep, vK, c.oN + (unsigned)u.gj * Ns8, nj);
                    if (u.flags & 2)
                        ls_gld<D>(a.rho + u.rhoOff, vRho, 0u, rho);
                    else
                    {
#pragma unroll
                        for (int p = 0; p < D; ++p)
                            rho[p] = 1.0;
                    }
                    ls_gld<D>(a.phi + u.phiOff, vPhi, sRay, phi);
                    const GLOBAL_AS DevSlot* q = gPar + u.e;
                    const double a0 = q->p[0], a2 = q->p[2], a3 = q->p[3];
#pragma unroll
                    for (int p = 0; p < D; ++p)
                    {
                        const double v1 = a0 * phi[p];
                        const double v2 = (a2 * rho[p]) * v1;
                        const double uji = a3 * v2;
                        cC[p] += ni[p] * v1 - nj[p] * v2;
                        eC[p] += nj[p] * uji;
                    }
                };
                // (the first two lines as straight-line code -- their loads are in flight together --, further ones in a loop)
                if (gnL > 0)
                    gather_line(0);
                if (gnL > 1)
                    gather_line(1);
                for (int l = 2; l < gnL; ++l)
                    gather_line(l);
            }
#pragma unroll
            for (int s = 0; s < NLc; ++s)
            {
                if (s < NL && HOLD)
                {
#pragma unroll
                    for (int p = 0; p < D; ++p)
                        phiK[s][p] = phiN[s][p];
                    ls_gld<D>(rPhi[s], vPhi, sRayNext, phiN[s]); // the next ray's profile
#pragma unroll
                    for (int p = 0; p < D; ++p)
                    {
                        cC[p] = fma(cxr[s][p], phiK[s][p], cC[p]);
                        eC[p] = fma(cer[s][p], phiK[s][p], eC[p]);
                    }
                }
                else if (s < NL)
                {
                    double phi[D], ni[D], nj[D], rho[D];
                    ls_lds<D>(sDep, vK, lnI[s], ni);
                    ls_lds<D>(sDep, vK, lnJ[s], nj);
                    load_rho(s, std::integral_constant<int, D>{}, 0u, vRho, rho);
                    if (HPRD)
                    {
#pragma unroll
                        for (int p = 0; p < D; ++p)
                            phi[p] = phiN[s][p];
                        ls_gld<D>(rPhi[s], vPhi, sRayNext, phiN[s]); // the next ray's profile
                    }
                    else
                        ls_gld<D>(rPhi[s], vPhi, sRay, phi); // (a blend: no register for a profile held one ray ahead)
                    LinePar lp{ la0[s], la2[s], la3[s], lcw1[s] };
                    if (!HPRD)
                        lp = line_par(s);
#pragma unroll
                    for (int p = 0; p < D; ++p)
                    {
                        const double v1 = lp.a0 * phi[p];
                        const double v2 = (lp.a2 * rho[p]) * v1;
                        const double uji = lp.a3 * v2;
                        cC[p] += ni[p] * v1 - nj[p] * v2;
                        eC[p] += nj[p] * uji;
                    }
                }
                else
                {
#pragma unroll
                    for (int p = 0; p < D; ++p)
                        phiK[s][p] = 0.0;
                }
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
            double sc[D];
#pragma unroll
            for (int p = 0; p < D; ++p)
                sc[p] = r3[2 * 64 * D + p];
            ls_lds<D + 1>(sDep, vK, c.oG, dhK); // entry k + 1 holds |h_k - h_k+1|: dhK[q] = dh[k0 - 1 + q]
            ls_lds<D + 1>(sDep, vK, c.oG + c.gStride, rdhK);
            ls_lds<D>(sDep, vK + 8u, c.oG + 2u * c.gStride, rsumK); // rsumK[p] = rsum[k0 + p]
#pragma unroll
            for (int p = 0; p < D; ++p)
            {
                const double rc = ls_rcp(cC[p]);
                chi[MI(p)] = cC[p];
                rch[MI(p)] = rc;
                S[MI(p)] = (eC[p] + sc[p]) * rc;
            }
        }
        LS_PHASE();
        LS_T(tp1);
        // halo: the last point of the upwind lane (P), the first point of the downwind lane (N)
        const double chiP = REUSE ? (UP ? hChiHi : hChiLo) : ls_shift1<!UP>(chi[D - 1]), SP = REUSE ? (UP ? hSHi : hSLo) : ls_shift1<!UP>(S[D - 1]);
        const double chiN = REUSE ? (UP ? hChiLo : hChiHi) : ls_shift1<UP>(chi[0]), SN = REUSE ? (UP ? hSLo : hSHi) : ls_shift1<UP>(S[0]);
#define CHI(i) ((i) < 0 ? chiP : (i) >= D ? chiN : chi[(i) < 0 ? 0 : (i) >= D ? D - 1 : (i)])
#define SRC(i) ((i) < 0 ? SP : (i) >= D ? SN : S[(i) < 0 ? 0 : (i) >= D ? D - 1 : (i)])

        double ee[D], bb[D], PsiS[D]; // the points' maps I_i = ee I_i-1 + bb, and Psi* (:322-323)
        if (SOLVER == LWHIP_FS_BEZIER3_1D)
        {
            // piecewise_bezier3_1d_impl FormalScalar.cpp:209-325, every quantity of the march formed per point
            double x[D], dS[D], dSP; // optical depth of the intervals, dS / dtau at the points
            double rx[D];            // 1 / x: serves the S slopes here and, cubed, the coefficients' 1 / x^3
            if constexpr (REUSE)
            {
                // (the intervals of this direction, scaled; going up the derivatives change sign)
#pragma unroll
                for (int i = 0; i < D; ++i)
                {
                    x[i] = hT[UP ? D - i : i] * zmu;
                    rx[i] = hRT[UP ? D - i : i] * muz;
                    dS[i] = UP ? -(hDS[D - 1 - i] * muz) : hDS[i] * muz;
                }
                dSP = UP ? -(hDSHi * muz) : hDSLo * muz;
            }
            else
            {
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
                    rx[i] = ls_rcp(t);
                    sS[i] = (SRC(i) - SRC(i - 1)) * rx[i];
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
            }
            {
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
                            const double rdtb = ls_rcp(dtb);
                            const double* sb = c.sB + min(c.rRaw, LS_MAX_R - 1) * 4 + (UP ? 2 : 0);
                            const double B0 = sb[0], B1 = sb[1];
                            Ib = UP ? B1 - (B0 - B1) * rdtb : B0 - (B1 - B0) * rdtb;
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
            LS_T(tp2);
            LS_TACC(tmB, tp1, tp2);
            // Bezier3_coeffs (Bezier.hpp:81-127) and the update :265-275 as a map
#pragma unroll
            for (int i = 0; i < D; ++i)
            {
                const double Su = SRC(i - 1), S0 = S[i];
                const double xi = x[i];
                const double x2 = xi * xi, x3 = x2 * xi;
                const double exr = ls_exp_neg(-fmin(xi, 64.0), c.sExp);
                const double ex = (xi > 30.0) ? 0.0 : exr;
                // (1 / x^3 from the stencils' 1 / x: two multiplications instead of a quarter-rate reciprocal and its Newton step;
                // at the ray's last point x was replaced by the closing step's optical depth, whose coefficients do not use it)
                const double r3 = (rx[i] * rx[i]) * rx[i];
                const bool small = xi < 5e-2;
                // both branches are evaluated on every lane and selected at once: straight-line code, two values in flight
                auto pick = [&](double g, double t) {
                    LS_KEEP(g);
                    LS_KEEP(t);
                    double r = small ? t : g;
                    LS_KEEP(r);
                    return r;
                };
                // (the Taylor branch with two constants: q = x^2 / 20, t = x^3 / 120)
                const double xq = 0.25 * xi, q = 0.05 * x2, t3 = x3 * (1.0 / 120.0);
                const double al = pick((6.0 - ex * (6.0 + 6.0 * xi + 3.0 * x2 + x3)) * r3, xq - 4.0 * q + 10.0 * t3);
                const double be = pick((6.0 * ex - 6.0 + 6.0 * xi - 3.0 * x2 + x3) * r3, xq - q + t3);
                const double ga = pick(3.0 * (2.0 * xi - 6.0 + ex * (6.0 + 4.0 * xi + x2)) * r3, xq - 3.0 * q + 6.0 * t3);
                const double de = pick(3.0 * (6.0 - 4.0 * xi + x2 - 2.0 * ex * (3.0 + xi)) * r3, xq - 2.0 * q + 3.0 * t3);
                const double edt = pick(ex, 1.0 - xi + 0.5 * x2 - 20.0 * t3);
                const double x_3 = xi * (1.0 / 3.0);
                const double dSu = i == 0 ? dSP : dS[i > 0 ? i - 1 : 0];
                const double Cuw = Su + x_3 * dSu;
                const double C0 = S0 - x_3 * dS[i];
                double e1 = edt, b1 = al * Su + be * S0 + ga * Cuw + de * C0, ps = be + de;
                LS_KEEP(b1); // (the point's coefficients are consumed here, before the next point's are formed)
                LS_KEEP(ps);
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
                    const double rL = ls_rcp(xi);
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
                LS_PHASE();
            }
        }
        else
        {
            // The other two solvers (not the benchmark's: written for coverage, not tuned).  The boundary intensity of the
            // ray's first point, as above:
            auto boundary_I = [&](const int i) -> double {
                double Ib = 0.0;
                if (bcType == LWHIP_BC_THERMALISED)
                {
                    const double dtb = ((0.5 * zmu) * (CHI(i) + CHI(i + 1))) * dhK[IV(i + 1)];
                    const double rdtb = ls_rcp(dtb);
                    const double* sb = c.sB + min(c.rRaw, LS_MAX_R - 1) * 4 + (UP ? 2 : 0);
                    const double B0 = sb[0], B1 = sb[1];
                    Ib = UP ? B1 - (B0 - B1) * rdtb : B0 - (B1 - B0) * rdtb;
                }
                else if (bcType == LWHIP_BC_CALLABLE)
                {
                    const int mi = UP ? CTAB(int32_t, a.lowerIdx)[mu * 2 + 1] : CTAB(int32_t, a.upperIdx)[mu * 2 + 0];
                    if (mi >= 0)
                        Ib = UP ? GCD(a.lowerBcData)[(size_t)la * a.lowerNmu + mi] : GCD(a.upperBcData)[(size_t)la * a.upperNmu + mi];
                }
                return Ib;
            };
#pragma unroll
            for (int i = 0; i < D; ++i)
            {
                const double cu = CHI(i - 1), c0 = CHI(i), cd = CHI(i + 1);
                const double Su = SRC(i - 1), S0 = S[i], Sd = SRC(i + 1);
                double e1, b1, ps;
                // the linear step over the upwind interval: piecewise_linear_1d_impl FormalScalar.cpp:136-207, and the last
                // point of piecewise_besser_1d_impl :449-466
                auto linear_step = [&]() {
                    double dtau = ((0.5 * zmu) * (cu + c0)) * dhK[IV(i)];
                    dtau = (is_first(i) || !is_ok(i)) ? 1.0 : dtau;
                    const double rdt = d_rcp(dtau);
                    const double dSuw = (Su - S0) * rdt;
                    double w0, w1;
                    d_w2(dtau, w0, w1);
                    e1 = 1.0 - w0;
                    b1 = w0 * S0 + w1 * dSuw;
                    ps = w0 - w1 * rdt;
                };
                if (SOLVER == LWHIP_FS_LINEAR_1D)
                    linear_step();
                else
                {
                    // piecewise_besser_1d_impl FormalScalar.cpp:395-467: chi, S of the upwind, local and downwind point
                    const double ds_uw = dhK[IV(i)] * zmu, ds_dw = dhK[IV(i + 1)] * zmu;
                    const bool interior = !is_first(i) && !is_last(i) && is_ok(i);
                    // (points without both neighbours get harmless arguments; their maps are replaced below)
                    const double hU = interior ? ds_uw : 1.0, hD = interior ? ds_dw : 1.0;
                    const double chiC = d_besser_control_point(hU, hD, cu, c0, cd);
                    const double dtauUw = interior ? (1.0 / 3.0) * (cu + chiC + c0) * ds_uw : 1.0;
                    const double dtauDw = interior ? 0.5 * (c0 + cd) * ds_dw : 1.0;
                    const double SC = d_besser_control_point(dtauUw, dtauDw, Su, S0, Sd);
                    double cM, cO, cC, edt;
                    d_besser_coeffs(dtauUw, cM, cO, cC, edt);
                    e1 = edt;
                    b1 = cM * Su + cO * S0 + cC * SC;
                    ps = cO + cC;
                    if (__builtin_amdgcn_ballot_w64(is_last(i)) != 0)
                    {
                        // last point: linear, w1 with the sign of the reference's dS_uw = (S_0 - S_uw) / dtau (:449-466)
                        const double dtau = is_last(i) ? ((0.5 * zmu) * (c0 + cu)) * dhK[IV(i)] : 1.0;
                        const double rcp = d_rcp(dtau);
                        const double dSuw = (S0 - Su) * rcp;
                        double w0, w1;
                        d_w2(dtau, w0, w1);
                        const bool lst = is_last(i);
                        e1 = lst ? 1.0 - w0 : e1;
                        b1 = lst ? w0 * S0 - w1 * dSuw : b1;
                        ps = lst ? w0 - w1 * rcp : ps;
                    }
                }
                if (__builtin_amdgcn_ballot_w64(is_first(i)) != 0)
                {
                    const bool fst = is_first(i);
                    const double Ib = boundary_I(i);
                    e1 = fst ? 0.0 : e1;
                    b1 = fst ? Ib : b1;
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
            }
            LS_T(tp2);
            LS_TACC(tmB, tp1, tp2);
        }
        LS_PHASE();
        LS_T(tp3);

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
            ls_scan<UP>(E, B, c.lane, LR);
            double I = ls_shift1<!UP>(B); // the intensity entering the lane (a ray's first point multiplies it by 0)
#pragma unroll
            for (int i = 0; i < D; ++i)
            {
                I = fma(ee[i], I, bb[i]);
                Iv[i] = I;
            }
        }
        LS_PHASE();
        LS_T(tp4);

        // ---- outputs and the sums over the rays ----
        int la = c.la;
        LS_KEEPI(la); // (output addresses are formed here, not held from the start of the tile)
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
        if (HPRD && RATES && a.hJOff && td.hasPrd && act)
        {
            // hybrid PRD: the ray's intensity into the rest-frame mean intensity, JRest(idx, k) += 0.5 wmu frac I(k)
            // (SimdFullIterationTemplates.hpp:397-408); fp64 atomics: other wavelengths and rays add to the same rows
            const int hq = ((const GLOBAL_AS int32_t*)a.hLa2h)[la];
            if (hq >= 0)
            {
                const GLOBAL_AS int64_t* off = (const GLOBAL_AS int64_t*)a.hJOff + ((size_t)(hq * Nmu + mu) * 2 + dirI) * Ns + k0;
#pragma unroll
                for (int p = 0; p < D; ++p)
                {
                    if (p >= c.nV)
                        continue;
                    for (int64_t e = off[p]; e < off[p + 1]; ++e)
                    {
                        const GLOBAL_AS lwhip_j_coeff* jc = (const GLOBAL_AS lwhip_j_coeff*)a.hJCoef + e;
                        const double fr = jc->frac;
                        const int row = jc->idx;
                        unsafeAtomicAdd(a.JRest + (size_t)row * Ns + (k0 + p), (hw * fr) * Iv[MI(p)]);
                    }
                }
            }
        }
        if (RATES)
        {
            double wPsi[D];
#pragma unroll
            for (int i = 0; i < D; ++i)
            {
                wPsi[i] = hw * PsiS[i];
                if ((!GEN && NRUN == 1) || run == 0)
                {
                    Jk[MI(i)] += hw * Iv[i]; // accumulate_J :181-190
                    M1[MI(i)] += wPsi[i];
                }
            }
            if (MOM)
            {
                // the angle moments of the slots' rates: of line 0, or (a blend) of the line this run accumulates
                if ((!GEN && NRUN == 1) || run == 0)
                    mW += hw;
                if constexpr (GEN)
                {
                    if (gnL > 0)
                    {
                        // (the ray's profiles again: an L1 hit)
                        double ph0[D];
                        ls_gld<D>(a.phi + ld_c(gSlots + 0).phiOff, vPhi0, sRay, ph0);
#pragma unroll
                        for (int p = 0; p < D; ++p)
                        {
                            const int i = MI(p);
                            const double wa = hw * ph0[p], wb = wPsi[i] * ph0[p];
                            mP0[p] += wa;
                            mP1[p] = fma(wa, Iv[i], mP1[p]);
                            mQ1[p] += wb;
                            mQ2[p] = fma(wb, ph0[p], mQ2[p]);
                        }
                        if (gnL > 1)
                        {
                            // line 1 and its pair with line 0: the five arrays' old values requested together, then stored
                            double ph1[D], old[5][D];
                            ls_gld<D>(a.phi + ld_c(gSlots + 1).phiOff, vPhi0, sRay, ph1);
#pragma unroll
                            for (int k = 0; k < 5; ++k)
#pragma unroll
                                for (int p = 0; p < D; ++p)
                                    old[k][p] = gFirstRay ? 0.0 : gMom[(size_t)(((k < 4 ? k : gPairBase)) * D + p) * 64];
#pragma unroll
                            for (int p = 0; p < D; ++p)
                            {
                                const int i = MI(p);
                                const double wa = hw * ph1[p], wb = wPsi[i] * ph1[p];
                                gMom[(size_t)(0 * D + p) * 64] = old[0][p] + wa;
                                gMom[(size_t)(1 * D + p) * 64] = fma(wa, Iv[i], old[1][p]);
                                gMom[(size_t)(2 * D + p) * 64] = old[2][p] + wb;
                                gMom[(size_t)(3 * D + p) * 64] = fma(wb, ph1[p], old[3][p]);
                                gMom[(size_t)(gPairBase * D + p) * 64] = fma(wb, ph0[p], old[4][p]);
                            }
                        }
                        for (int l = 2; l < gnL; ++l)
                        {
                            double phl[D];
                            ls_gld<D>(a.phi + ld_c(gSlots + l).phiOff, vPhi0, sRay, phl);
#pragma unroll
                            for (int p = 0; p < D; ++p)
                            {
                                const int i = MI(p);
                                const double wa = hw * phl[p], wb = wPsi[i] * phl[p];
                                gm_add(4 * (l - 1) + 0, p, wa);
                                gm_add(4 * (l - 1) + 1, p, wa * Iv[i]);
                                gm_add(4 * (l - 1) + 2, p, wb);
                                gm_add(4 * (l - 1) + 3, p, wb * phl[p]);
                            }
                            for (int m = 0; m < l; ++m)
                            {
                                double phm[D];
                                ls_gld<D>(a.phi + ld_c(gSlots + m).phiOff, vPhi0, sRay, phm);
#pragma unroll
                                for (int p = 0; p < D; ++p)
                                    gm_add(gPairBase + l * (l - 1) / 2 + m, p, (wPsi[MI(p)] * phl[p]) * phm[p]);
                            }
                        }
                    }
                    gFirstRay = false;
                }
                else if (NL > 0)
                {
                    double ph[D], phB[D];
                    if (HOLD)
                    {
#pragma unroll
                        for (int p = 0; p < D; ++p)
                            ph[p] = phB[p] = phiK[0][p];
                    }
                    else
                    {
                        // (the ray's profiles again: an L1 hit)
                        const double* rp = (NL > 1 && run != 0) ? rPhi[NLc - 1] : rPhi[0];
                        ls_gld<D>(rp, vPhi0, sRay, ph);
                        if (NL > 1 && run == 0)
                            ls_gld<D>(rPhi[NLc - 1], vPhi0, sRay, phB);
                        else
                        {
#pragma unroll
                            for (int p = 0; p < D; ++p)
                                phB[p] = 0.0;
                        }
                    }
#pragma unroll
                    for (int p = 0; p < D; ++p)
                    {
                        const int i = MI(p);
                        const double wa = hw * ph[p], wb = wPsi[i] * ph[p];
                        mP0[p] += wa;
                        mP1[p] = fma(wa, Iv[i], mP1[p]);
                        mQ1[p] += wb;
                        mQ2[p] = fma(wb, ph[p], mQ2[p]);
                        if (NL > 1)
                            mQ12[p] = fma(wb, phB[p], mQ12[p]);
                    }
                }
            }
            // Hybrid PRD: the rate integrals ray by ray, half a block at a time (bounds what is in flight): V_ij / V_ji again
            // from the profile (an L1 / L2 hit), the populations, the slots' continuum rows.
            auto rates_half = [&](auto hTag) {
                constexpr int H = decltype(hTag)::value; // points p = H .. H + HP - 1 of the block (k order)
                constexpr int HP = D / 2;
                unsigned vK = c.vK + (unsigned)H * 8u, vRow = c.vRow + (unsigned)H * 8u, vPhi = vPhi0 + (unsigned)H * 8u,
                         vRho = vRho0 + (unsigned)H * 8u;
                LS_KEEPI(vK);
                LS_KEEPI(vRow);
                LS_KEEPI(vPhi);
                LS_KEEPI(vRho);
                int zmL[NLc], flL[NLc], msL[NLc], zmM[NMc];
#pragma unroll
                for (int s = 0; s < NLc; ++s)
                {
                    zmL[s] = lzm[s];
                    flL[s] = lfl[s];
                    msL[s] = lms[s];
                    LS_KEEPS(zmL[s]);
                    LS_KEEPS(flL[s]);
                    LS_KEEPS(msL[s]);
                }
#pragma unroll
                for (int m = 0; m < NMc; ++m)
                {
                    zmM[m] = mzm[m];
                    LS_KEEPS(zmM[m]);
                }
                int ccLL[NLc][NLc], ccML[NMc][NLc]; // the coupling codes, likewise
#pragma unroll
                for (int l = 0; l < NLc; ++l)
                {
#pragma unroll
                    for (int s = 0; s < NLc; ++s)
                    {
                        ccLL[s][l] = ccL[s][l];
                        if (NL > 1)
                            LS_KEEPS(ccLL[s][l]);
                    }
#pragma unroll
                    for (int m = 0; m < NMc; ++m)
                    {
                        ccML[m][l] = ccM[m][l];
                        if (NM > 0 && NL > 0)
                            LS_KEEPS(ccML[m][l]);
                    }
                }
                double v1[NLc][HP], v2[NLc][HP], xl[NLc][HP], el[NLc][HP], uj[NLc][HP];
#pragma unroll
                for (int s = 0; s < NLc; ++s)
                {
                    if (s < NL)
                    {
                        double phi[HP], ni[HP], nj[HP], rho[HP];
                        ls_gld<HP>(rPhi[s], vPhi, sRay, phi);
                        ls_lds<HP>(sDep, vK, lnI[s], ni);
                        ls_lds<HP>(sDep, vK, lnJ[s], nj);
                        load_rho(s, std::integral_constant<int, HP>{}, (unsigned)H, vRho, rho);
#pragma unroll
                        for (int p = 0; p < HP; ++p)
                        {
                            v1[s][p] = la0[s] * phi[p];
                            v2[s][p] = (la2[s] * rho[p]) * v1[s][p];
                            uj[s][p] = la3[s] * v2[s][p];
                            xl[s][p] = ni[p] * v1[s][p] - nj[p] * v2[s][p];
                            el[s][p] = nj[p] * uj[s][p];
                        }
                    }
                    else
                    {
#pragma unroll
                        for (int p = 0; p < HP; ++p)
                            v1[s][p] = v2[s][p] = xl[s][p] = el[s][p] = uj[s][p] = 0.0;
                    }
                }
                // level sums of a slot: continuum rows + the active lines of this wavelength, selected by the tile's
                // coupling codes (a line's coupling with itself is known at compile time)
                auto couple = [&](const int (&cc)[NLc], const int p, double& etaA, double& chi_i, double& chi_j, double& U_i, double& U_j) {
#pragma unroll
                    for (int l = 0; l < NLc; ++l)
                    {
                        if (l < NL)
                        {
                            const int c5 = cc[l];
                            etaA += (c5 & 1) ? el[l][p] : 0.0;
                            chi_i += (c5 & 2) ? xl[l][p] : 0.0;
                            chi_i -= (c5 & 4) ? xl[l][p] : 0.0;
                            chi_j += (c5 & 8) ? xl[l][p] : 0.0;
                            chi_j -= (c5 & 16) ? xl[l][p] : 0.0;
                            U_j += (c5 & 16) ? uj[l][p] : 0.0;
                            U_i += (c5 & 4) ? uj[l][p] : 0.0;
                        }
                    }
                };
                // the five continuum rows of a slot (rows that no continuum feeds are not read)
                auto load_rows = [&](const unsigned rowOff, const int zm, double (&lr)[5][HP]) {
#pragma unroll
                    for (int q = 0; q < 5; ++q)
                    {
                        if (!(zm & (1 << q)))
                            ls_gld<HP>(rRows, vRow, rowOff + (unsigned)q * rowStride, lr[q]);
                        else
                        {
#pragma unroll
                            for (int p = 0; p < HP; ++p)
                                lr[q][p] = 0.0;
                        }
                    }
                };
#pragma unroll
                for (int s = 0; s < NLc; ++s)
                {
                    if (s < NL)
                    {
                        double lr[5][HP], wphi[HP];
                        load_rows(lrow[s], zmL[s], lr);
                        ls_lds<HP>(sDep, vK, lwO[s], wphi);
                        const double lcw = lcw1[s] * hw;
                        const double lcwOp = (flL[s] & 1) ? lcw : 0.0; // detailed atoms: rates only
#pragma unroll
                        for (int p = 0; p < HP; ++p)
                        {
                            const int i = MI(H + p); // march index of the point
                            const double uji = uj[s][p];
                            const double wlamu = lcw * wphi[p];
                            double etaA = lr[0][p] + el[s][p], chi_i = lr[1][p] + xl[s][p], chi_j = lr[2][p] - xl[s][p];
                            double U_i = lr[3][p], U_j = lr[4][p] + uj[s][p];
                            if (NL > 1)
                                couple(ccLL[s], p, etaA, chi_i, chi_j, U_i, U_j);
                            const double Ieff = Iv[i] - PsiS[i] * etaA; // compute_full_Ieff
                            const double wop = lcwOp * wphi[p];
                            const double g0 = ((uji + v2[s][p] * Ieff) - (PsiS[i] * chi_i * U_j)) * wop;
                            const double g1 = ((v1[s][p] * Ieff) - (PsiS[i] * chi_j * U_i)) * wop;
                            const double g2 = Iv[i] * v1[s][p] * wlamu;
                            const double g3 = (uji + Iv[i] * v2[s][p]) * wlamu;
                            if (act)
                            {
                                double* dst = accBase + lacc[s] + (unsigned)((H + p) * LR);
                                lds_add(dst, g0);
                                lds_add(dst + D * LR, g1);
                                lds_add(dst + 2 * D * LR, g2);
                                lds_add(dst + 3 * D * LR, g3);
                            }
                            // Meta of the line's atom
                            if (msL[s] >= 0)
                            {
                                if (NL < 2 || msL[s] == 0)
                                    Me[0][H + p] += wPsi[i] * el[s][p];
                                else
                                    Me[1][H + p] += wPsi[i] * el[s][p];
                            }
                        }
                    }
                }
#pragma unroll
                for (int m = 0; m < NMc; ++m)
                {
                    if (m < NM) // mixed continua (always of active atoms)
                    {
                        double mr[5][HP], mvji[HP];
                        load_rows(mrow[m], zmM[m], mr);
                        ls_gld<HP>(rRows, vRow, maux[m], mvji);
                        const double wlamu = mcw1[m] * hw;
#pragma unroll
                        for (int p = 0; p < HP; ++p)
                        {
                            const int i = MI(H + p);
                            const double w1 = mb0[m], w2 = mvji[p], uji = mb2[m] * w2;
                            double etaA = mr[0][p], chi_i = mr[1][p], chi_j = mr[2][p], U_i = mr[3][p], U_j = mr[4][p];
                            if (NL > 0)
                                couple(ccML[m], p, etaA, chi_i, chi_j, U_i, U_j);
                            const double Ieff = Iv[i] - PsiS[i] * etaA;
                            if (act)
                            {
                                double* dst = accBase + macc[m] + (unsigned)((H + p) * LR);
                                lds_add(dst, ((uji + w2 * Ieff) - (PsiS[i] * chi_i * U_j)) * wlamu);
                                lds_add(dst + D * LR, ((w1 * Ieff) - (PsiS[i] * chi_j * U_i)) * wlamu);
                                lds_add(dst + 2 * D * LR, Iv[i] * w1 * wlamu);
                                lds_add(dst + 3 * D * LR, (uji + Iv[i] * w2) * wlamu);
                            }
                        }
                    }
                }
            };
            static_assert(D % 2 == 0, "the block's rate integrals are formed in halves");
            if (!MOM)
            {
                rates_half(std::integral_constant<int, 0>{});
                LS_PHASE();
                rates_half(std::integral_constant<int, D / 2>{});
            }
        }
        LS_PHASE();
        LS_T(tp5);
        LS_TACC(tmA, tp0, tp1);
        LS_TACC(tmS, tp3, tp4);
        LS_TACC(tmD, tp4, tp5);
#ifdef LS_TIMING
        tmC += tp3 - tp1 - (tmB - tmBprev);
        tmBprev = tmB;
#endif
#undef MI
#undef IV
#undef CHI
#undef SRC
    };

    // rays q0 .. q1 - 1 of the iteration's list (mu outer, down then up: SimdFullIterationTemplates.hpp:300-304; up only
    // for formal_sol's upOnly) -- all of them, or this wavefront's share when the tile's rays are split over S wavefronts
    // (written as the loop over mu with its two directions guarded, not as a loop over q with a direction branch: the
    // latter makes the compiler hoist three dozen registers of addresses and flags in front of the loop, and spill)
    const int nRun = NRUN;
    for (run = 0; run < nRun; ++run)
    {
    // (pairs: only a wavefront that holds both rays of every angle -- no split -- and only if the host found the profiles symmetric)
    const bool pairMode = PAIRABLE && dyn.phiSym != 0 && dyn.nPass == 2 && q0 == 0 && q1 == 2 * Nmu;
    if (!pairMode)
        prefetch_first();
    if constexpr (CONT)
        cont_unit(0);
#ifdef LS_ABL_NO_RAYS
    if (dyn.nPass == 77)
#else
    if (dyn.nPass == 2)
#endif
    {
        if (pairMode)
        {
            if constexpr (PAIRABLE)
            {
                for (int mu = 0; mu < Nmu; ++mu)
                {
                    cont_unit(mu);
                    pass(std::false_type{}, std::true_type{}, mu, 0u);
                    pass(std::true_type{}, std::true_type{}, mu, 0u);
                }
            }
        }
        else
        {
            for (int mu = q0 >> 1; 2 * mu < q1; ++mu)
            {
                if (2 * mu >= q0)
                    pass(std::false_type{}, std::false_type{}, mu, (unsigned)(mu * 2 + 1) * Ns8);
                if (2 * mu + 1 < q1)
                    pass(std::true_type{}, std::false_type{}, mu, (unsigned)(mu * 2 + 2) * Ns8);
            }
        }
    }
    else
#ifdef LS_ABL_NO_RAYS
        if (dyn.nPass == 78)
#endif
        for (int mu = q0; mu < q1; ++mu)
            pass(std::true_type{}, std::false_type{}, mu, (unsigned)(mu * 2 + 3) * Ns8);
#ifdef LS_ABL_NO_SLOTFIN
    if (MOM && dyn.nPass == 77)
#else
    if constexpr (MOM)
#endif
    {
        // ---- the slots' rate integrals from the moments, half a block at a time -------------------------------------------
        // A level sum of a slot is affine in the profiles: X = X0 + sum_l X_l phi_l (X0: the slot's continuum row; X_l: the
        // coupled line's cx / ce / cu, selected by the tile's coupling codes).  With
        //   L(X)    = sum w Psi* X          = X0 M1 + sum_l X_l Q_l
        //   L_s(X)  = sum w Psi* phi_s X    = X0 Q_s + sum_l X_l Q_sl
        //   B(X, Y) = sum w Psi* X Y        = X0 Y0 M1 + sum_l (X0 Y_l + X_l Y0) Q_l + sum_ll' X_l Y_l' Q_ll'
        // the sums of compute_full_operator_rates :206-234 over the rays are
        //   line s:   Gij += K_op [cu_s P0_s + c2_s P1_s - c2_s L_s(eta_A) - B(chi_i, U_j)]      Rij += K c1_s P1_s
        //             Gji += K_op [c1_s P1_s - c1_s L_s(eta_A) - B(chi_j, U_i)]                  Rji += K (cu_s P0_s + c2_s P1_s)
        //   mixed c:  Gij += K_c [U_c W + V_c (J - L(eta_A)) - B(chi_i, U_j)]                    Rij += K_c alpha J
        //             Gji += K_c [alpha (J - L(eta_A)) - B(chi_j, U_i)]                          Rji += K_c (U_c W + V_c J)
        // (K = w_lambda 4 pi / hc w_phi; K_op = K for active atoms, 0 for detailed ones; the quadrature weights are in the moments)
        auto slot_finish_half = [&](auto hTag, auto npTag) {
            constexpr int H = decltype(hTag)::value, HP = decltype(npTag)::value; // points H .. H + HP - 1 of the block
            const unsigned vK = c.vK + (unsigned)H * 8u, vRow = c.vRow + (unsigned)H * 8u;
            // (the accumulator addresses are formed here: hoisted in front of a blend's run loop they are two dozen registers
            // held -- spilled -- over the rays)
            ls_lds_double* accB = (ls_lds_double*)accBase;
            asm volatile("" : "+v"(accB));
            // the lines' coefficients per unit profile at these points
            double c2[NLc][HP], cu[NLc][HP], cx[NLc][HP], ce[NLc][HP];
#pragma unroll
            for (int s = 0; s < NLc; ++s)
            {
                if (s < NL)
                {
                    double ni[HP], nj[HP], rho[HP];
                    ls_lds<HP>(sDep, vK, lnI[s], ni);
                    ls_lds<HP>(sDep, vK, lnJ[s], nj);
                    load_rho0(s, std::integral_constant<int, HP>{}, (unsigned)H, rho);
                    const LinePar lp = line_par(s);
#pragma unroll
                    for (int p = 0; p < HP; ++p)
                    {
                        c2[s][p] = (lp.a2 * rho[p]) * lp.a0;
                        cu[s][p] = lp.a3 * c2[s][p];
                        cx[s][p] = ni[p] * lp.a0 - nj[p] * c2[s][p];
                        ce[s][p] = nj[p] * cu[s][p];
                    }
                }
                else
                {
#pragma unroll
                    for (int p = 0; p < HP; ++p)
                        c2[s][p] = cu[s][p] = cx[s][p] = ce[s][p] = 0.0;
                }
            }
            // a slot's level sums: X0 from the continuum rows, X_l from the coupling code of line l with the slot
            // (own: the slot is line l itself); bits as couple_code
            struct Aff
            {
                double e[NLc], xi[NLc], xj[NLc], ui[NLc], uj[NLc];
            };
            auto coupled = [&](const int code, const bool own, const int l, const int p, Aff& f) {
                const double sE = (own || (code & 1)) ? 1.0 : 0.0;
                const double sXi = own ? 1.0 : ((code & 2) ? 1.0 : 0.0) - ((code & 4) ? 1.0 : 0.0);
                const double sXj = own ? -1.0 : ((code & 8) ? 1.0 : 0.0) - ((code & 16) ? 1.0 : 0.0);
                const double sUi = own ? 0.0 : (code & 4) ? 1.0 : 0.0;
                const double sUj = (own || (code & 16)) ? 1.0 : 0.0;
                f.e[l] = sE * ce[l][p];
                f.xi[l] = sXi * cx[l][p];
                f.xj[l] = sXj * cx[l][p];
                f.ui[l] = sUi * cu[l][p];
                f.uj[l] = sUj * cu[l][p];
            };
            // what this run accumulated (a blend's run 0: J, M1, W, line 0 and Q_01; its run 1: line 1 alone)
            auto cur = [&](const int l) { return NRUN == 1 ? l == 0 : l == run; };
            auto Q1of = [&](const int l, const int P) { return cur(l) ? mQ1[P] : 0.0; };
            auto Q2of = [&](const int l, const int m, const int P) { return l == m ? (cur(l) ? mQ2[P] : 0.0) : (run == 0 ? mQ12[P] : 0.0); };
            auto P0of = [&](const int l, const int P) { return cur(l) ? mP0[P] : 0.0; };
            auto P1of = [&](const int l, const int P) { return cur(l) ? mP1[P] : 0.0; };
            auto M1of = [&](const int P) { return (NRUN == 1 || run == 0) ? M1[P] : 0.0; };
            auto Jof = [&](const int P) { return (NRUN == 1 || run == 0) ? Jk[P] : 0.0; };
            const double Wof = (NRUN == 1 || run == 0) ? mW : 0.0;
            auto Lin = [&](const double X0, const double (&X)[NLc], const int P) {
                double r = X0 * M1of(P);
#pragma unroll
                for (int l = 0; l < NLc; ++l)
                    if (l < NL)
                        r = fma(X[l], Q1of(l, P), r);
                return r;
            };
            auto LinS = [&](const int s, const double X0, const double (&X)[NLc], const int P) {
                double r = X0 * Q1of(s, P);
#pragma unroll
                for (int l = 0; l < NLc; ++l)
                    if (l < NL)
                        r = fma(X[l], Q2of(s, l, P), r);
                return r;
            };
            auto Bil = [&](const double X0, const double (&X)[NLc], const double Y0, const double (&Y)[NLc], const int P) {
                double r = (X0 * Y0) * M1of(P);
#pragma unroll
                for (int l = 0; l < NLc; ++l)
                {
                    if (l < NL)
                    {
                        r = fma(fma(X0, Y[l], X[l] * Y0), Q1of(l, P), r);
#pragma unroll
                        for (int m = 0; m < NLc; ++m)
                            if (m < NL)
                                r = fma(X[l] * Y[m], Q2of(l, m, P), r);
                    }
                }
                return r;
            };
            // the five continuum rows of a slot (rows that no continuum feeds are not read)
            auto load_rows = [&](const unsigned rowOff, const int zm, double (&lr)[5][HP]) {
#pragma unroll
                for (int q = 0; q < 5; ++q)
                {
                    if (!(zm & (1 << q)))
                        ls_gld<HP>(rRows, vRow, rowOff + (unsigned)q * rowStride, lr[q]);
                    else
                    {
#pragma unroll
                        for (int p = 0; p < HP; ++p)
                            lr[q][p] = 0.0;
                    }
                }
            };
#pragma unroll
            for (int s = 0; s < NLc; ++s)
            {
                if (s < NL)
                {
                    double lr[5][HP], wphi[HP], gq[4][HP];
                    load_rows(lrow[s], lzm[s], lr);
                    ls_lds<HP>(sDep, vK, lwO[s], wphi);
                    const LinePar lp = line_par(s);
#pragma unroll
                    for (int p = 0; p < HP; ++p)
                    {
                        const int P = H + p;
                        Aff f;
#pragma unroll
                        for (int l = 0; l < NLc; ++l)
                        {
                            f.e[l] = f.xi[l] = f.xj[l] = f.ui[l] = f.uj[l] = 0.0;
                            if (l < NL)
                                coupled(ccL[s][l], l == s, l, p, f);
                        }
                        const double K = lp.cw1 * wphi[p];
                        const double Kop = (lfl[s] & 1) ? K : 0.0; // detailed atoms: rates only
                        const double up = fma(cu[s][p], P0of(s, P), c2[s][p] * P1of(s, P)); // sum w (U_ji + V_ji I)
                        const double LsE = LinS(s, lr[0][p], f.e, P);
                        const double g0 = ((up - c2[s][p] * LsE) - Bil(lr[1][p], f.xi, lr[4][p], f.uj, P)) * Kop;
                        const double g1 = ((lp.a0 * (P1of(s, P) - LsE)) - Bil(lr[2][p], f.xj, lr[3][p], f.ui, P)) * Kop;
                        const double g2 = (lp.a0 * P1of(s, P)) * K;
                        const double g3 = up * K;
                        gq[0][p] = g0;
                        gq[1][p] = g1;
                        gq[2][p] = g2;
                        gq[3][p] = g3;
                        // Meta of the line's atom for the pure continua (lane_finish): sum w Psi* eta_line
                        if (lms[s] >= 0)
                        {
                            if (NL < 2 || lms[s] == 0)
                                Me[0][P] = fma(ce[s][p], Q1of(s, P), Me[0][P]);
                            else
                                Me[1][P] = fma(ce[s][p], Q1of(s, P), Me[1][P]);
                        }
                    }
                    // (fixed-order mode: the task's turn at the slot's accumulator -- waited for HERE, with the integrals formed:
                    // only the adds themselves are serialised between the wavefronts of a workgroup)
                    if (c.sTurn && H == 0)
                    {
                        ls_turn_wait<D>(c.sTurn, tr.slot[s].cs, 0, D, (int)(((unsigned)tr.slot[s].flags >> (run == 0 ? 8 : 20)) & 0xfffu) + c.sub, nullptr);
#ifdef LS_TIMING
                        if (s == 0)
                            lsTurnExit = __builtin_amdgcn_s_memtime();
#endif
                    }
                    if (act)
                    {
#pragma unroll
                        for (int p = 0; p < HP; ++p)
                        {
                            ls_lds_double* dst = accB + lacc[s] + (unsigned)((H + p) * LR);
                            ls_lds_add(dst, gq[0][p]);
                            ls_lds_add(dst + D * LR, gq[1][p]);
                            ls_lds_add(dst + 2 * D * LR, gq[2][p]);
                            ls_lds_add(dst + 3 * D * LR, gq[3][p]);
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < NMc; ++m)
            {
                if (m < NM) // mixed continua (always of active atoms)
                {
                    double mr[5][HP], mvji[HP], gq[4][HP];
                    load_rows(mrow[m], mzm[m], mr);
                    ls_gld<HP>(rRows, vRow, maux[m], mvji);
#pragma unroll
                    for (int p = 0; p < HP; ++p)
                    {
                        const int P = H + p;
                        Aff f;
#pragma unroll
                        for (int l = 0; l < NLc; ++l)
                        {
                            f.e[l] = f.xi[l] = f.xj[l] = f.ui[l] = f.uj[l] = 0.0;
                            if (l < NL)
                                coupled(ccM[m][l], false, l, p, f);
                        }
                        const double w1 = mb0[m], w2 = mvji[p], uji = mb2[m] * w2;
                        const double Jeff = Jof(P) - Lin(mr[0][p], f.e, P);
                        const double uw = fma(uji, Wof, w2 * Jof(P)); // sum w (U_ji + V_ji I)
                        gq[0][p] = ((uji * Wof + w2 * Jeff) - Bil(mr[1][p], f.xi, mr[4][p], f.uj, P)) * mcw1[m];
                        gq[1][p] = ((w1 * Jeff) - Bil(mr[2][p], f.xj, mr[3][p], f.ui, P)) * mcw1[m];
                        gq[2][p] = (w1 * Jof(P)) * mcw1[m];
                        gq[3][p] = uw * mcw1[m];
                    }
                    if (c.sTurn && H == 0)
                        ls_turn_wait<D>(c.sTurn, tr.slot[NL + m < 2 ? NL + m : 1].cs, 0, D,
                                        (int)(((unsigned)tr.slot[NL + m < 2 ? NL + m : 1].flags >> (run == 0 ? 8 : 20)) & 0xfffu) + c.sub, c.dbgWait);
                    if (act)
                    {
#pragma unroll
                        for (int p = 0; p < HP; ++p)
                        {
                            ls_lds_double* dst = accB + macc[m] + (unsigned)((H + p) * LR);
                            ls_lds_add(dst, gq[0][p]);
                            ls_lds_add(dst + D * LR, gq[1][p]);
                            ls_lds_add(dst + 2 * D * LR, gq[2][p]);
                            ls_lds_add(dst + 3 * D * LR, gq[3][p]);
                        }
                    }
                }
            }
        };
        // ---- the generic kind: every slot's integrals from the moments of all lines, a point at a time --------------------------
        // A slot's level sums are X = X0 + sum_l X_l phi_l over the tile's lines (X0: its continuum row; X_l: line l's cx / ce / cu
        // through the coupling code of (slot, l)), so with the moments Q_l, Q_lm (Q_ll on the diagonal), P0_l, P1_l, M1, J, W
        //   B(X, Y)  = X0 Y0 M1 + sum_l (X0 Y_l + X_l Y0) Q_l + sum_lm X_l Y_m Q_lm
        //   L(X)     = X0 M1 + sum_l X_l Q_l                  L_s(X) = X0 Q_s + sum_l X_l Q_sl
        // and the sums of compute_full_operator_rates :206-234 are those of slot_finish_half above.  Per point the lines'
        // coefficients go to the scratch block once; the slots' loops read them back (an L1 hit) instead of holding 4 nL values.
        auto gen_slot_finish = [&]() {
            ls_lds_double* accB = (ls_lds_double*)accBase;
            asm volatile("" : "+v"(accB));
            const GLOBAL_AS char* rowsB = (const GLOBAL_AS char*)rRows + c.vRow;
            const GLOBAL_AS double* rhoB = (const GLOBAL_AS double*)a.rho + (vRho0 >> 3);
            const int nS = gnL + gnM;
            auto momAt = [&](const int arr, const int P) -> double { return gMom[(size_t)(arr * D + P) * 64]; };
#pragma unroll
            for (int P = 0; P < D; ++P)
            {
                // the lines' coefficients per unit profile at this point (Transition::uv, chi_eta_aux_accum) -> scratch
                for (int l = 0; l < gnL; ++l)
                {
                    const DevTileSlot u = ld_c(gSlots + l);
                    const GLOBAL_AS DevSlot* q = gPar + u.e;
                    const double a0 = q->p[0], a2 = q->p[2], a3 = q->p[3];
                    const double ni = ((const double*)(sDep + c.vK + c.oN + (unsigned)u.gi * Ns8))[P];
                    const double nj = ((const double*)(sDep + c.vK + c.oN + (unsigned)u.gj * Ns8))[P];
                    const double rho = (u.flags & 2) ? rhoB[u.rhoOff + P] : 1.0;
                    const double c2 = (a2 * rho) * a0, cu = a3 * c2;
                    GLOBAL_AS double* o = gMom + (size_t)((gCoefBase + 4 * l) * D + P) * 64;
                    o[0] = c2;
                    o[(size_t)D * 64] = cu;
                    o[(size_t)2 * D * 64] = ni * a0 - nj * c2;
                    o[(size_t)3 * D * 64] = nj * cu;
                }
                auto coefAt = [&](const int l, const int k) -> double { return gMom[(size_t)((gCoefBase + 4 * l + k) * D + P) * 64]; };
                // the moments of line l and of the pair (l, m)
                auto Q1at = [&](const int l) -> double { return l == 0 ? mQ1[P] : momAt(4 * (l - 1) + 2, P); };
                auto Q2at = [&](const int l, const int m) -> double {
                    if (l == m)
                        return l == 0 ? mQ2[P] : momAt(4 * (l - 1) + 3, P);
                    const int lo = l < m ? l : m, hi = l < m ? m : l;
                    return momAt(gPairBase + hi * (hi - 1) / 2 + lo, P);
                };
                const double M1P = M1[P], JP = Jk[P], WP = mW;
                for (int sI = 0; sI < nS; ++sI)
                {
                    const DevTileSlot us = ld_c(gSlots + sI);
                    const bool isLine = sI < gnL;
                    const bool op = (us.flags & 1) != 0;
                    // how line l enters this slot's level sums (couple_code's bits; the slot's own line as in `coupled`)
                    auto cpl = [&](const int l, double& e, double& xi, double& xj, double& ui, double& uj) {
                        const DevTileSlot ul = ld_c(gSlots + l);
                        const bool own = isLine && l == sI;
                        const int code = (ul.flags & 1) ? ((ul.atom == us.atom ? 1 : 0) | (ul.gi == us.gi ? 2 : 0) | (ul.gj == us.gi ? 4 : 0)
                                                           | (ul.gi == us.gj ? 8 : 0) | (ul.gj == us.gj ? 16 : 0))
                                                        : 0;
                        const double cx = coefAt(l, 2), ce = coefAt(l, 3), cu = coefAt(l, 1);
                        e = ((own || (code & 1)) ? 1.0 : 0.0) * ce;
                        xi = (own ? 1.0 : ((code & 2) ? 1.0 : 0.0) - ((code & 4) ? 1.0 : 0.0)) * cx;
                        xj = (own ? -1.0 : ((code & 8) ? 1.0 : 0.0) - ((code & 16) ? 1.0 : 0.0)) * cx;
                        ui = (own ? 0.0 : (code & 4) ? 1.0 : 0.0) * cu;
                        uj = ((own || (code & 16)) ? 1.0 : 0.0) * cu;
                    };
                    double lr[5];
                    const int zm = (isLine && !op) ? 31 : (int)us.zmask;
#pragma unroll
                    for (int q = 0; q < 5; ++q)
                        lr[q] = (zm & (1 << q)) ? 0.0 : ((const GLOBAL_AS double*)(rowsB + (size_t)(us.r0 + q) * rowStride))[P];
                    // L(eta_A), L_s(eta_A), B(chi_i, U_j), B(chi_j, U_i)
                    double LinE = lr[0] * M1P, LsE = isLine ? lr[0] * Q1at(sI) : 0.0;
                    double BilA = (lr[1] * lr[4]) * M1P, BilB = (lr[2] * lr[3]) * M1P;
                    for (int l = 0; l < gnL; ++l)
                    {
                        double el, xil, xjl, uil, ujl;
                        cpl(l, el, xil, xjl, uil, ujl);
                        const double Ql = Q1at(l);
                        LinE = fma(el, Ql, LinE);
                        if (isLine)
                            LsE = fma(el, Q2at(sI, l), LsE);
                        BilA = fma(fma(lr[1], ujl, xil * lr[4]), Ql, BilA);
                        BilB = fma(fma(lr[2], uil, xjl * lr[3]), Ql, BilB);
                        for (int m = 0; m < gnL; ++m)
                        {
                            double em, xim, xjm, uim, ujm;
                            cpl(m, em, xim, xjm, uim, ujm);
                            const double Qlm = Q2at(l, m);
                            BilA = fma(xil * ujm, Qlm, BilA);
                            BilB = fma(xjl * uim, Qlm, BilB);
                        }
                    }
                    const GLOBAL_AS DevSlot* qs = gPar + us.e;
                    double g0, g1, g2, g3;
                    if (isLine)
                    {
                        const double wphi = ((const double*)(sDep + c.vK + c.oW + (unsigned)us.row * Ns8))[P];
                        const double K = (qs->p[1] * PI4_HC) * wphi;
                        const double Kop = op ? K : 0.0; // detailed atoms: rates only
                        const double c2s = coefAt(sI, 0), cus = coefAt(sI, 1), a0s = qs->p[0];
                        const double P0s = sI == 0 ? mP0[P] : momAt(4 * (sI - 1) + 0, P), P1s = sI == 0 ? mP1[P] : momAt(4 * (sI - 1) + 1, P);
                        const double up = fma(cus, P0s, c2s * P1s);
                        g0 = ((up - c2s * LsE) - BilA) * Kop;
                        g1 = ((a0s * (P1s - LsE)) - BilB) * Kop;
                        g2 = (a0s * P1s) * K;
                        g3 = up * K;
                        if (us.ms >= 0) // Meta of the line's atom for the pure continua (lane_finish): sum w Psi* eta_line
                        {
                            if (us.ms == 0)
                                Me[0][P] = fma(coefAt(sI, 3), Q1at(sI), Me[0][P]);
                            else
                                Me[1][P] = fma(coefAt(sI, 3), Q1at(sI), Me[1][P]);
                        }
                    }
                    else
                    {
                        const double w1 = qs->p[0], mcw = qs->p[1];
                        const double w2 = ((const GLOBAL_AS double*)(rowsB + (size_t)max((int)us.rAux, 0) * rowStride))[P];
                        const double uji = qs->p[2] * w2;
                        const double Jeff = JP - LinE;
                        g0 = ((uji * WP + w2 * Jeff) - BilA) * mcw;
                        g1 = ((w1 * Jeff) - BilB) * mcw;
                        g2 = (w1 * JP) * mcw;
                        g3 = fma(uji, WP, w2 * JP) * mcw;
                    }
                    if (c.sTurn && P == 0)
                        ls_turn_wait<D>(c.sTurn, us.cs, 0, D, (int)(((unsigned)us.flags >> 8) & 0xfffu) + c.sub);
                    if (act)
                    {
                        ls_lds_double* dst = accB + (unsigned)(us.cs * 4 * c.LRD) + (unsigned)(P * LR);
                        ls_lds_add(dst, g0);
                        ls_lds_add(dst + D * LR, g1);
                        ls_lds_add(dst + 2 * D * LR, g2);
                        ls_lds_add(dst + 3 * D * LR, g3);
                    }
                }
            }
        };
        // The same for tiles with at most TWO lines (any number of mixed continua: overlapping lines with the continua coupled to
        // them, the realistic case): the lines' coefficients and every moment of the point in registers -- the general form above
        // reads them back from the scratch inside its loops, a chain of memory round trips per slot.
        auto gen_slot_finish2 = [&]() {
            ls_lds_double* accB = (ls_lds_double*)accBase;
            asm volatile("" : "+v"(accB));
            const GLOBAL_AS char* rowsB = (const GLOBAL_AS char*)rRows + c.vRow;
            const GLOBAL_AS double* rhoB = (const GLOBAL_AS double*)a.rho + (vRho0 >> 3);
            const int nS = gnL + gnM;
            const bool hasB = gnL > 1;
            DevTileSlot ul[2] = { ld_c(gSlots + 0), ld_c(gSlots + (hasB ? 1 : 0)) };
            double pl[2][4];
#pragma unroll
            for (int l = 0; l < 2; ++l)
            {
                const GLOBAL_AS DevSlot* q = gPar + ul[l].e;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    pl[l][k] = q->p[k];
            }
#pragma unroll
            for (int P = 0; P < D; ++P)
            {
                double c2[2], cu[2], cx[2], ce[2];
#pragma unroll
                for (int l = 0; l < 2; ++l)
                {
                    const double ni = ((const double*)(sDep + c.vK + c.oN + (unsigned)ul[l].gi * Ns8))[P];
                    const double nj = ((const double*)(sDep + c.vK + c.oN + (unsigned)ul[l].gj * Ns8))[P];
                    const double rho = (ul[l].flags & 2) ? rhoB[ul[l].rhoOff + P] : 1.0;
                    const double on = (l == 0 || hasB) ? 1.0 : 0.0; // (a tile with one line: line "1" is absent)
                    c2[l] = on * ((pl[l][2] * rho) * pl[l][0]);
                    cu[l] = pl[l][3] * c2[l];
                    cx[l] = on * (ni * pl[l][0]) - nj * c2[l];
                    ce[l] = nj * cu[l];
                }
                // moments: line 0 in registers, line 1 and the pair from the scratch (five loads in flight)
                const double M1P = M1[P], JP = Jk[P], WP = mW;
                double Ql[2] = { mQ1[P], 0.0 }, Qll[2] = { mQ2[P], 0.0 }, P0l[2] = { mP0[P], 0.0 }, P1l[2] = { mP1[P], 0.0 }, Q01 = 0.0;
                if (hasB)
                {
                    P0l[1] = gMom[(size_t)(0 * D + P) * 64];
                    P1l[1] = gMom[(size_t)(1 * D + P) * 64];
                    Ql[1] = gMom[(size_t)(2 * D + P) * 64];
                    Qll[1] = gMom[(size_t)(3 * D + P) * 64];
                    Q01 = gMom[(size_t)(gPairBase * D + P) * 64];
                }
                for (int sI = 0; sI < nS; ++sI)
                {
                    const DevTileSlot us = ld_c(gSlots + sI);
                    const bool isLine = sI < gnL;
                    const bool op = (us.flags & 1) != 0;
                    double e[2], xi[2], xj[2], ui[2], uj[2];
#pragma unroll
                    for (int l = 0; l < 2; ++l)
                    {
                        const bool own = isLine && l == sI;
                        const int code = (ul[l].flags & 1) ? ((ul[l].atom == us.atom ? 1 : 0) | (ul[l].gi == us.gi ? 2 : 0) | (ul[l].gj == us.gi ? 4 : 0)
                                                              | (ul[l].gi == us.gj ? 8 : 0) | (ul[l].gj == us.gj ? 16 : 0))
                                                           : 0;
                        e[l] = ((own || (code & 1)) ? 1.0 : 0.0) * ce[l];
                        xi[l] = (own ? 1.0 : ((code & 2) ? 1.0 : 0.0) - ((code & 4) ? 1.0 : 0.0)) * cx[l];
                        xj[l] = (own ? -1.0 : ((code & 8) ? 1.0 : 0.0) - ((code & 16) ? 1.0 : 0.0)) * cx[l];
                        ui[l] = (own ? 0.0 : (code & 4) ? 1.0 : 0.0) * cu[l];
                        uj[l] = ((own || (code & 16)) ? 1.0 : 0.0) * cu[l];
                    }
                    double lr[5];
                    const int zm = (isLine && !op) ? 31 : (int)us.zmask;
#pragma unroll
                    for (int q = 0; q < 5; ++q)
                        lr[q] = (zm & (1 << q)) ? 0.0 : ((const GLOBAL_AS double*)(rowsB + (size_t)(us.r0 + q) * rowStride))[P];
                    auto Bil = [&](const double X0, const double (&X)[2], const double Y0, const double (&Y)[2]) {
                        double r = (X0 * Y0) * M1P;
                        r = fma(fma(X0, Y[0], X[0] * Y0), Ql[0], r);
                        r = fma(fma(X0, Y[1], X[1] * Y0), Ql[1], r);
                        r = fma(X[0] * Y[0], Qll[0], r);
                        r = fma(X[1] * Y[1], Qll[1], r);
                        r = fma(fma(X[0], Y[1], X[1] * Y[0]), Q01, r);
                        return r;
                    };
                    const double LinE = fma(e[1], Ql[1], fma(e[0], Ql[0], lr[0] * M1P));
                    const GLOBAL_AS DevSlot* qs = gPar + us.e;
                    double g0, g1, g2, g3;
                    if (isLine)
                    {
                        const int sl = sI; // 0 or 1
                        const double wphi = ((const double*)(sDep + c.vK + c.oW + (unsigned)us.row * Ns8))[P];
                        const double K = (qs->p[1] * PI4_HC) * wphi;
                        const double Kop = op ? K : 0.0; // detailed atoms: rates only
                        const double c2s = sl == 0 ? c2[0] : c2[1], cus = sl == 0 ? cu[0] : cu[1], a0s = sl == 0 ? pl[0][0] : pl[1][0];
                        const double P0s = sl == 0 ? P0l[0] : P0l[1], P1s = sl == 0 ? P1l[0] : P1l[1];
                        const double up = fma(cus, P0s, c2s * P1s);
                        const double LsE = sl == 0 ? fma(e[1], Q01, fma(e[0], Qll[0], lr[0] * Ql[0])) : fma(e[0], Q01, fma(e[1], Qll[1], lr[0] * Ql[1]));
                        g0 = ((up - c2s * LsE) - Bil(lr[1], xi, lr[4], uj)) * Kop;
                        g1 = ((a0s * (P1s - LsE)) - Bil(lr[2], xj, lr[3], ui)) * Kop;
                        g2 = (a0s * P1s) * K;
                        g3 = up * K;
                        if (us.ms >= 0) // Meta of the line's atom for the pure continua (lane_finish): sum w Psi* eta_line
                        {
                            const double me = (sl == 0 ? ce[0] : ce[1]) * (sl == 0 ? Ql[0] : Ql[1]);
                            if (us.ms == 0)
                                Me[0][P] += me;
                            else
                                Me[1][P] += me;
                        }
                    }
                    else
                    {
                        const double w1 = qs->p[0], mcw = qs->p[1];
                        const double w2 = ((const GLOBAL_AS double*)(rowsB + (size_t)max((int)us.rAux, 0) * rowStride))[P];
                        const double uji = qs->p[2] * w2;
                        const double Jeff = JP - LinE;
                        g0 = ((uji * WP + w2 * Jeff) - Bil(lr[1], xi, lr[4], uj)) * mcw;
                        g1 = ((w1 * Jeff) - Bil(lr[2], xj, lr[3], ui)) * mcw;
                        g2 = (w1 * JP) * mcw;
                        g3 = fma(uji, WP, w2 * JP) * mcw;
                    }
                    if (c.sTurn && P == 0)
                        ls_turn_wait<D>(c.sTurn, us.cs, 0, D, (int)(((unsigned)us.flags >> 8) & 0xfffu) + c.sub);
                    if (act)
                    {
                        ls_lds_double* dst = accB + (unsigned)(us.cs * 4 * c.LRD) + (unsigned)(P * LR);
                        ls_lds_add(dst, g0);
                        ls_lds_add(dst + D * LR, g1);
                        ls_lds_add(dst + 2 * D * LR, g2);
                        ls_lds_add(dst + 3 * D * LR, g3);
                    }
                }
            }
        };
        LS_PHASE();
        // (fixed-order mode: a task waits for its turn at a slot right before its first add to it, inside the functions below)
        if constexpr (GEN)
        {
            if (gnL <= 2)
                gen_slot_finish2();
            else
                gen_slot_finish();
        }
        else if constexpr (NL > 1)
        {
            // (a blend: a point at a time -- two lines' coefficients and level sums are twice the registers)
            slot_finish_half(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            LS_PHASE();
            slot_finish_half(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            LS_PHASE();
            slot_finish_half(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
            LS_PHASE();
            slot_finish_half(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
        }
        else
        {
            // (the whole block at once: every load of the slots' rows, populations and parameters is in flight together --
            // the registers of the ray loop are free here; measured 0.1342 -> 0.1321 ms against two halves)
            slot_finish_half(std::integral_constant<int, 0>{}, std::integral_constant<int, D>{});
        }
        LS_PHASE();
        if (c.sTurn && (run == 0 || run == nRun - 1)) // (the first run's turn is handed on at once, the later runs' after the last)
        {
            if constexpr (GEN)
            {
                for (int e = 0; e < gnL + gnM; ++e)
                    ls_turn_done<D>(c.sTurn, ld_c(gSlots + e).cs, 0, D, c.lane);
            }
            else
            {
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    if (e < NL + NM)
                        ls_turn_done<D>(c.sTurn, tr.slot[e].cs, 0, D, c.lane);
#ifdef LS_TIMING
                if (NL > 0 && c.dbgWait && c.lane == 0 && lsTurnExit)
                    unsafeAtomicAdd(c.dbgWait, (double)(__builtin_amdgcn_s_memtime() - lsTurnExit));
#endif
            }
        }
        if (NRUN > 1)
        {
#pragma unroll
            for (int p = 0; p < D; ++p)
                mP0[p] = mP1[p] = mQ1[p] = mQ2[p] = mQ12[p] = 0.0;
        }
    }
    } // run
#ifdef LS_TIMING
    if (c.lane == 0 && a.lsDbg)
    {
        double* o = a.lsDbg + (size_t)c.ti * 8;
        o[1] = (double)tmA;
        o[2] = (double)tmB;
        o[3] = (double)tmC;
        o[4] = (double)tmS;
        o[5] = (double)tmD;
    }
#endif
    return meSlot0;
}

// The tile's wavelengths are complete: J out, dJ, and the pure continua from the angle moments.
// sum_r w_r [Uji + Vji (I_r - Psi_r eta_r) - Psi_r chi_i U_j] with chi, U, Uji, Vji ray-independent and
// eta_r = etaC + etaLine_r  ->  Uji W + Vji (J - M1 etaC - Meta) - M1 chi_i U_j: an exact regrouping of
// compute_full_operator_rates :206-234 for continua whose level sums do not depend on the ray.  The level sums
// (chi_eta_aux_accum :59-109, continua in transition order) live in registers: per tile the host lists which sums have
// several contributors (DevPostProg).  P0, NP: the points of the lane's block this call handles (the block is
// finished in pieces to bound the registers).
// The tile's continuum records for the finish: lane q holds record q (and the tile program's words for it), fetched with
// ONE vector round trip; the walks pick a continuum's words with v_readlane (q is uniform).  As scalar loads inside the
// walks they were ~40 dependent memory waits per task (every walk of every half re-read them, and a tile's records
// rarely sit in the scalar cache).
struct LaneFinRecs
{
    unsigned nI, nJ, rat, w6, w7, code, ops; // DevContRec dwords 0, 1, 2, 6 (sUj | flags << 16 | ms << 24), 7 (tr | cs << 16)
    unsigned feed;                           // DevLaneFin::feed
};
DEVINL LaneFinRecs lane_fin_recs(const CONST_AS TileArgs& a, const int ti, const int nCont, const int lane)
{
    LaneFinRecs r{};
    if (lane < nCont)
    {
        // (the tile's flat record of continuum `lane`: one 32-byte load, its address known from the tile index alone)
        const GLOBAL_AS unsigned* w = (const GLOBAL_AS unsigned*)(const void*)(a.laneFin + (size_t)ti * LWHIP_POST_MAXC + lane);
        r.nI = w[0];
        r.nJ = w[1];
        r.rat = w[2];
        r.w6 = w[3];
        r.w7 = w[4];
        r.code = w[5];
        r.ops = w[6];
        r.feed = w[7];
    }
    return r;
}
// the per-(continuum, wavelength) parameters alpha, w_lambda / lambda 4 pi / h, 2hc / lambda^3 of the tile's wavelengths into
// the wavefront's LDS block: lane (ray r, block q) fetches continuum q of its wavelength -- one memory round trip for all of
// them, from the tile's flat table
DEVINL void lane_fin_params(const CONST_AS TileArgs& a, const LaneCtx& c, const int ti, const int nCont)
{
    for (int q0 = 0; q0 < nCont; q0 += c.LR)
    {
        const int q = q0 + c.blk;
        if (q < nCont && c.act)
        {
            const GLOBAL_AS double* src = GCD(a.laneFinPar) + (((size_t)ti * LWHIP_POST_MAXC + q) * c.L + c.r) * 3;
            double* dst = c.sPP + (q * c.L + c.r) * 3;
            dst[0] = src[0];
            dst[1] = src[1];
            dst[2] = src[2];
        }
    }
}
struct LaneFinRec1 // one continuum's words, uniform
{
    unsigned nI, nJ, rat, flags, code, ops;
    int ms, cs;
};
DEVINL LaneFinRec1 lane_fin_rec(const LaneFinRecs& v, const int q)
{
    LaneFinRec1 r;
    r.nI = (unsigned)__builtin_amdgcn_readlane((int)v.nI, q);
    r.nJ = (unsigned)__builtin_amdgcn_readlane((int)v.nJ, q);
    r.rat = (unsigned)__builtin_amdgcn_readlane((int)v.rat, q);
    const unsigned w6 = (unsigned)__builtin_amdgcn_readlane((int)v.w6, q), w7 = (unsigned)__builtin_amdgcn_readlane((int)v.w7, q);
    r.flags = (w6 >> 16) & 0xffu;
    r.ms = (int)(int8_t)(w6 >> 24);
    r.cs = (int)(int16_t)(w7 >> 16);
    r.code = (unsigned)__builtin_amdgcn_readlane((int)v.code, q);
    r.ops = (unsigned)__builtin_amdgcn_readlane((int)v.ops, q);
    return r;
}

// The task's own pre-pass (round 4; until then a launch of its own, tile_pre_kernel): the ray-independent continuum terms
// of the lane's points -- Atom::setup_wavelength (LwAtom.hpp:107-115) + Transition::uv (LwTransition.hpp:132-142) +
// chi_eta_aux_accum (SimdFullIterationTemplates.hpp:59-109) for the wavelength's continua in the reference's transition order:
//   chi_C = bgChi + sum chi_c,  eta_C = bgEta + sum eta_c,  sca J  -> the wavefront's LDS block (rows 0 .. 2),
//   per line / mixed slot the level sums eta_A, chi(i), chi(j), U(i), U(j) of the active continua (DevLaneFin::feed says
//   which) and V_ji of a mixed continuum -> the tile's row block in global memory (read back ONCE, by the slots' integrals
//   after the rays: the same lane, tens of microseconds later).
// One launch, 60 MB of row writes and 60 MB of reads per 10 240 wavelengths less; the walk is ~600 vector instructions of a
// task's ~11 000.
template <int D, bool GEN>
DEVINL void lane_pre(const CONST_AS TileArgs& a, const DevLaneTile& tr, const LaneCtx& c, const int ti, const bool slotRows)
{
    const int Ns = c.Ns;
    const int nC = tr.nContAll;
    // (generic kind: any number of slots -- their records from the launch's slot table, the continua's feed bytes from the
    // host's table for generic tiles; the walk below serves slots 0 and 1, gen_more_rows the rest, two at a time)
    const CONST_AS DevTileSlot* gSlots = CTAB(DevTileSlot, a.tslots) + tr.td.slotOff;
    const CONST_AS uint8_t* gFeed = CTAB(uint8_t, a.laneFeedG) + (size_t)(GEN ? tr.gen : 0) * (LWHIP_POST_MAXC * LWHIP_GEN_MAXS);
    auto slot_rec = [&](const int e) -> DevTileSlot { return GEN ? ld_c(gSlots + e) : tr.slot[e < 2 ? e : 1]; };
    // background, J and the finish's records / parameters are requested first
    double chiC[D], etaC[D], scaJ[D];
    {
        const size_t o = (size_t)c.la * Ns + c.k0;
        double bs[D], bj[D];
        const GLOBAL_AS double* pc = GCD(a.bgChi) + o;
        const GLOBAL_AS double* pe = GCD(a.bgEta) + o;
        const GLOBAL_AS double* ps = GCD(a.bgSca) + o;
        const GLOBAL_AS double* pj = GCD(a.J) + o;
#pragma unroll
        for (int p = 0; p < D; ++p) // (the pools are padded: the last lane of a ray reads past its row)
        {
            chiC[p] = pc[p];
            etaC[p] = pe[p];
            bs[p] = ps[p];
            bj[p] = pj[p];
        }
#pragma unroll
        for (int p = 0; p < D; ++p)
            scaJ[p] = bs[p] * bj[p]; // sca * Jdag of compute_source_fn :169-179
    }
    const LaneFinRecs recs = lane_fin_recs(a, ti, nC, c.lane);
    lane_fin_params(a, c, ti, nC); // (stays in the wavefront's LDS block for the finish)
    double boltz[D];
    {
        double rT[D];
        ls_lds<D>(c.sDep, c.vK + 8u, c.oG + 3u * c.gStride, rT);
        const double hc_kl = HC_K / c.wav;
#pragma unroll
        for (int p = 0; p < D; ++p)
            boltz[p] = ls_exp_neg(-hc_kl * rT[p], c.sExp);
    }
    // the slots' level sums: [slot][eta_A, chi_i, chi_j, U_i, U_j] and the mixed continuum's V_ji
    double ls[2][5][D], vj[2][D];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int p = 0; p < D; ++p)
        {
            vj[e][p] = 0.0;
#pragma unroll
            for (int q = 0; q < 5; ++q)
                ls[e][q][p] = 0.0;
        }
    for (int q = 0; q < nC; ++q)
    {
        const LaneFinRec1 rc = lane_fin_rec(recs, q);
        unsigned feed = (unsigned)__builtin_amdgcn_readlane((int)recs.feed, q) & 0xffffu;
        if (GEN)
            feed = (unsigned)gFeed[q * LWHIP_GEN_MAXS + 0] | ((unsigned)gFeed[q * LWHIP_GEN_MAXS + 1] << 8);
        double ni[D], nj[D], rat[D];
        ls_lds<D>(c.sDep, c.vK, c.oN + rc.nI, ni);
        ls_lds<D>(c.sDep, c.vK, c.oN + rc.nJ, nj);
        ls_lds<D>(c.sDep, c.vK, c.oRat + rc.rat, rat);
        const double* ppq = c.sPP + (q * c.L + c.r) * 3;
        const double p0 = ppq[0], p2 = ppq[2];
        double vji[D], chi_c[D], uji[D], eta_c[D];
#pragma unroll
        for (int p = 0; p < D; ++p)
        {
            vji[p] = (rat[p] * boltz[p]) * p0;
            chi_c[p] = ni[p] * p0 - nj[p] * vji[p];
            uji[p] = p2 * vji[p];
            eta_c[p] = nj[p] * uji[p];
            chiC[p] += chi_c[p];
            etaC[p] += eta_c[p];
        }
        if (slotRows && feed != 0u)
        {
#pragma unroll
            for (int e = 0; e < 2; ++e)
            {
                const unsigned f = (feed >> (8 * e)) & 0xffu;
                if (f == 0u)
                    continue;
#pragma unroll
                for (int p = 0; p < D; ++p)
                {
                    if (f & 1u)
                        ls[e][0][p] += eta_c[p];
                    if (f & 2u)
                        ls[e][1][p] += chi_c[p];
                    if (f & 4u)
                        ls[e][1][p] -= chi_c[p];
                    if (f & 8u)
                        ls[e][2][p] += chi_c[p];
                    if (f & 16u)
                        ls[e][2][p] -= chi_c[p];
                    if (f & 32u)
                        ls[e][3][p] += uji[p];
                    if (f & 64u)
                        ls[e][4][p] += uji[p];
                    if (f & 128u)
                        vj[e][p] = vji[p];
                }
            }
        }
    }
    {
        double* r3 = c.sRow3 + c.lane * D;
#pragma unroll
        for (int p = 0; p < D; ++p)
        {
            r3[p] = chiC[p];
            r3[64 * D + p] = etaC[p];
            r3[2 * 64 * D + p] = scaJ[p];
        }
    }
    // the slots' rows into the tile's row block (rows no continuum feeds are neither written nor read: zmask)
    const int nSl = tr.td.nLine + tr.td.nMixed;
    auto write_rows = [&](const int e0) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
        {
            if (e0 + e >= nSl)
                continue;
            const DevTileSlot u = slot_rec(e0 + e);
            if (!(u.flags & 1))
                continue;
            GLOBAL_AS double* dst = (GLOBAL_AS double*)c.rRows + (size_t)(c.vRow >> 3);
#pragma unroll
            for (int q = 0; q < 5; ++q)
            {
                if (u.zmask & (1 << q))
                    continue;
                GLOBAL_AS double* d = dst + (size_t)(u.r0 + q) * (c.rowStride >> 3);
#pragma unroll
                for (int p = 0; p < D; ++p)
                    d[p] = ls[e][q][p];
            }
            if (e0 + e >= tr.td.nLine && u.rAux >= 0)
            {
                GLOBAL_AS double* d = dst + (size_t)u.rAux * (c.rowStride >> 3);
#pragma unroll
                for (int p = 0; p < D; ++p)
                    d[p] = vj[e][p];
            }
        }
    };
    if (slotRows)
        write_rows(0);
    if constexpr (GEN)
    {
        // the rows of slots 2, 3, ...: the continua are walked again for every further pair of slots (a rare kind of tile)
        for (int e0 = 2; slotRows && e0 < nSl; e0 += 2)
        {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int p = 0; p < D; ++p)
                {
                    vj[e][p] = 0.0;
#pragma unroll
                    for (int q = 0; q < 5; ++q)
                        ls[e][q][p] = 0.0;
                }
            for (int q = 0; q < nC; ++q)
            {
                const unsigned feed = (unsigned)gFeed[q * LWHIP_GEN_MAXS + e0] | ((unsigned)gFeed[q * LWHIP_GEN_MAXS + e0 + 1] << 8);
                if (feed == 0u)
                    continue;
                const LaneFinRec1 rc = lane_fin_rec(recs, q);
                double ni[D], nj[D], rat[D];
                ls_lds<D>(c.sDep, c.vK, c.oN + rc.nI, ni);
                ls_lds<D>(c.sDep, c.vK, c.oN + rc.nJ, nj);
                ls_lds<D>(c.sDep, c.vK, c.oRat + rc.rat, rat);
                const double* ppq = c.sPP + (q * c.L + c.r) * 3;
                const double p0 = ppq[0], p2 = ppq[2];
#pragma unroll
                for (int e = 0; e < 2; ++e)
                {
                    const unsigned f = (feed >> (8 * e)) & 0xffu;
                    if (f == 0u)
                        continue;
#pragma unroll
                    for (int p = 0; p < D; ++p)
                    {
                        const double vji = (rat[p] * boltz[p]) * p0;
                        const double chi_c = ni[p] * p0 - nj[p] * vji;
                        const double uji = p2 * vji;
                        const double eta_c = nj[p] * uji;
                        if (f & 1u)
                            ls[e][0][p] += eta_c;
                        if (f & 2u)
                            ls[e][1][p] += chi_c;
                        if (f & 4u)
                            ls[e][1][p] -= chi_c;
                        if (f & 8u)
                            ls[e][2][p] += chi_c;
                        if (f & 16u)
                            ls[e][2][p] -= chi_c;
                        if (f & 32u)
                            ls[e][3][p] += uji;
                        if (f & 64u)
                            ls[e][4][p] += uji;
                        if (f & 128u)
                            vj[e][p] = vji;
                    }
                }
            }
            write_rows(e0);
        }
    }
}

template <int D, int P0, int NP, bool WIDE = false>
DEVINL void lane_finish_part(const CONST_AS TileArgs& a, const LaneCtx& c, const int nCont, const LaneFinRecs& recs, double* sAcc,
                             const int ms0, const int ms1, const double (&Jk)[D], const double (&M1)[D], const double (&Me)[2][D])
{
    // WIDE: the tile's program words are in the wide encoding (more than 8 shared sums -- continua of three and more active
    // atoms at one wavelength): 8 bits per target (0xFF none), 6 bits per operand (60 .. 63 the continuum's own terms)
    constexpr int NSUMS = WIDE ? (int)LWHIP_POST_SUMS_WIDE : (int)LWHIP_POST_SUMS;
    constexpr unsigned CODE_NONE = WIDE ? 0xFFFFFFFFu : 0x9999u;
    const int Nmu = c.Nmu, LR = c.LR;
    const unsigned vK = c.vK + (unsigned)P0 * 8u;
    double Wsum = 0.0; // total quadrature weight of the rays of the iteration, in ray order
    for (int m = 0; m < Nmu; ++m)
        for (int d = 0; d < 2; ++d)
            Wsum += 0.5 * CTAB(double, a.wmu)[m];
    double boltz[NP];
    {
        double rT[NP];
        ls_lds<NP>(c.sDep, vK + 8u, c.oG + 3u * c.gStride, rT);
        const double hc_kl = HC_K / c.wav;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            boltz[p] = ls_exp_neg(-hc_kl * rT[p], c.sExp);
    }
    // entry 0: zero (never written); 1 .. 8: the shared sums; 9: the spare entry that takes the terms no shared sum wants.
    // One register vector per point, so that an entry picked by a (uniform) index is an indexed register access
    // (v_movrel), not a trip through scratch memory as for an array.
    typedef double ls_sums_t __attribute__((ext_vector_type(NSUMS + 2)));
    ls_sums_t S8[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p)
        S8[p] = (ls_sums_t)(0.0);
    // walk 1: the shared level sums, continua in transition order
    for (int q = 0; q < nCont; ++q)
    {
        const LaneFinRec1 rc = lane_fin_rec(recs, q);
        const unsigned code = (rc.flags & 1) ? rc.code : CODE_NONE;
        if (code == CODE_NONE)
            continue;
        double ni[NP], nj[NP], rat[NP];
        ls_lds<NP>(c.sDep, vK, c.oN + rc.nI, ni);
        ls_lds<NP>(c.sDep, vK, c.oN + rc.nJ, nj);
        ls_lds<NP>(c.sDep, vK, c.oRat + rc.rat, rat);
        const double* ppq = c.sPP + (q * c.L + c.r) * 3;
        const double p0 = ppq[0], p2 = ppq[2];
        double chi_c[NP], uji[NP], eta_c[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
        {
            const double vji = (rat[p] * boltz[p]) * p0;
            chi_c[p] = ni[p] * p0 - nj[p] * vji;
            uji[p] = p2 * vji;
            eta_c[p] = nj[p] * uji[p];
        }
        // the program word names, for each of the continuum's four terms, the entry it goes to (1 .. 8 a shared sum, 9: none
        // -- the spare entry): four indexed register updates instead of a test per (sum, term)
        // (the two words almost every continuum of a one- or two-atom problem has -- its atom's eta, U(j), chi(j) are the
        // shared sums 1, 2, 3 or 4, 5, 6, its chi(i) is its own -- are recognised and served with fixed registers; an
        // indexed register access costs a mode switch each)
        if (!WIDE && code == 0x1239u)
        {
#pragma unroll
            for (int p = 0; p < NP; ++p)
            {
                S8[p][3] -= chi_c[p];
                S8[p][2] += uji[p];
                S8[p][1] += eta_c[p];
            }
        }
        else if (!WIDE && code == 0x4569u)
        {
#pragma unroll
            for (int p = 0; p < NP; ++p)
            {
                S8[p][6] -= chi_c[p];
                S8[p][5] += uji[p];
                S8[p][4] += eta_c[p];
            }
        }
        else
        {
            // (wide: a target of 0xFF -- none -- goes to the spare entry NSUMS + 1)
            auto tgt = [&](const int f) -> unsigned {
                if (!WIDE)
                    return (code >> (4 * f)) & 15u;
                const unsigned v = (code >> (8 * f)) & 0xffu;
                return v == 0xffu ? (unsigned)(NSUMS + 1) : v;
            };
            const unsigned t0 = tgt(0), t1 = tgt(1), t2 = tgt(2), t3 = tgt(3);
#pragma unroll
            for (int p = 0; p < NP; ++p)
            {
                S8[p][t0] += chi_c[p];
                S8[p][t1] -= chi_c[p];
                S8[p][t2] += uji[p];
                S8[p][t3] += eta_c[p];
            }
        }
    }
    // walk 2: the pure continua
    for (int q = 0; q < nCont; ++q)
    {
        const LaneFinRec1 rc = lane_fin_rec(recs, q);
        if (!(rc.flags & 2))
            continue;
        double ni[NP], nj[NP], rat[NP];
        ls_lds<NP>(c.sDep, vK, c.oN + rc.nI, ni);
        ls_lds<NP>(c.sDep, vK, c.oN + rc.nJ, nj);
        ls_lds<NP>(c.sDep, vK, c.oRat + rc.rat, rat);
        const double* ppq = c.sPP + (q * c.L + c.r) * 3;
        const double vij = ppq[0], wla = ppq[1], p2 = ppq[2];
        double vji[NP], chi_c[NP], uji[NP], eta_c[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
        {
            vji[p] = (rat[p] * boltz[p]) * vij;
            chi_c[p] = ni[p] * vij - nj[p] * vji[p];
            uji[p] = p2 * vji[p];
            eta_c[p] = nj[p] * uji[p];
        }
        const unsigned ops = rc.ops;
        double o5[5][NP]; // eta_A, chi_i, U_j, chi_j, U_i
        // operand w of the continuum: sel_w = 0 nothing, 1 .. 8 a shared sum (one indexed register read: the index is
        // uniform), 9 .. 12 the continuum's own term -- instead of a chain of tests per operand
        if (!WIDE && (ops == 0x03291u || ops == 0x06594u || ops == 0x0AB9Cu))
        {
            // the common words: the atom's shared sums 1, 2, 3 / 4, 5, 6, or a continuum alone in its atom at this wavelength
            const bool a0 = ops == 0x03291u, alone = ops == 0x0AB9Cu;
#pragma unroll
            for (int p = 0; p < NP; ++p)
            {
                o5[0][p] = alone ? eta_c[p] : a0 ? S8[p][1] : S8[p][4];
                o5[1][p] = chi_c[p];
                o5[2][p] = alone ? uji[p] : a0 ? S8[p][2] : S8[p][5];
                o5[3][p] = alone ? -chi_c[p] : a0 ? S8[p][3] : S8[p][6];
                o5[4][p] = 0.0;
            }
        }
        else
        {
#pragma unroll
            for (int w = 0; w < 5; ++w)
            {
                const unsigned sel = WIDE ? (ops >> (6 * w)) & 63u : (ops >> (4 * w)) & 15u;
                constexpr unsigned OWN0 = WIDE ? 60u : 9u; // the continuum's own +chi, -chi, U, eta: OWN0 .. OWN0 + 3
                const unsigned si = sel >= OWN0 ? 0u : sel;
#pragma unroll
                for (int p = 0; p < NP; ++p)
                {
                    const double own = sel == OWN0 ? chi_c[p] : sel == OWN0 + 1u ? -chi_c[p] : sel == OWN0 + 2u ? uji[p] : eta_c[p];
                    const double shared = S8[p][si];
                    o5[w][p] = sel >= OWN0 ? own : shared;
                }
            }
        }
        double* dst = sAcc + (unsigned)(rc.cs * 4 * c.LRD) + c.blk;
        // (the integrals first, then -- fixed-order mode -- the task's turn at the continuum's accumulator, then the adds: only
        // these are serialised between the wavefronts of a workgroup)
        double gq[4][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
        {
            const double Jp = Jk[P0 + p], M1p = M1[P0 + p];
            const double Meta = (rc.ms < 0) ? 0.0 : (rc.ms == ms0) ? Me[0][P0 + p] : (rc.ms == ms1) ? Me[1][P0 + p] : 0.0;
            const double Jeff = Jp - M1p * o5[0][p] - Meta;
            gq[0][p] = ((uji[p] * Wsum + vji[p] * Jeff) - (M1p * o5[1][p] * o5[2][p])) * wla;
            gq[1][p] = ((vij * Jeff) - (M1p * o5[3][p] * o5[4][p])) * wla;
            gq[2][p] = Jp * vij * wla;
            gq[3][p] = (uji[p] * Wsum + Jp * vji[p]) * wla;
        }
        if (c.sTurn)
            ls_turn_wait<D>(c.sTurn, rc.cs, P0, NP, (int)(__builtin_amdgcn_readlane((int)recs.feed, q) >> 16) & 0xffff);
#pragma unroll
        for (int p = 0; p < NP; ++p)
        {
            if (!(c.act && P0 + p < c.nV))
                continue;
            if (rc.flags & 1)
            {
                lds_add(dst + (P0 + p) * LR, gq[0][p]);
                lds_add(dst + (D + P0 + p) * LR, gq[1][p]);
            }
            lds_add(dst + (2 * D + P0 + p) * LR, gq[2][p]);
            lds_add(dst + (3 * D + P0 + p) * LR, gq[3][p]);
        }
        ls_turn_done<D>(c.sTurn, rc.cs, P0, NP, c.lane);
    }
}

// lane_finish_part for the tiles whose continua all carry the common program words (the host's DevLaneTile::finFast: every
// contributing continuum feeds its atom's sums 1, 2, 3 or 4, 5, 6 -- eta_A, U(j), chi(j) --, every pure one reads them, or stands
// alone in its atom; 97 % of the continua on the bench grid): six NAMED sums per point instead of a ten-entry register vector,
// so the whole block (NP = D) fits the registers and is finished in ONE pair of walks -- half the record reads, scalar branches
// and LDS round trips of two half-block passes, and four points of independent arithmetic per instruction stream.
template <int D, int P0, int NP>
DEVINL void lane_finish_fast(const CONST_AS TileArgs& a, const LaneCtx& c, const int nCont, const LaneFinRecs& recs, double* sAcc,
                             const int ms0, const int ms1, const double (&Jk)[D], const double (&M1)[D], const double (&Me)[2][D])
{
    const int Nmu = c.Nmu, LR = c.LR;
    const unsigned vK = c.vK + (unsigned)P0 * 8u;
    double Wsum = 0.0; // total quadrature weight of the rays of the iteration, in ray order
    for (int m = 0; m < Nmu; ++m)
        for (int d = 0; d < 2; ++d)
            Wsum += 0.5 * CTAB(double, a.wmu)[m];
    double boltz[NP];
    {
        double rT[NP];
        ls_lds<NP>(c.sDep, vK + 8u, c.oG + 3u * c.gStride, rT);
        const double hc_kl = HC_K / c.wav;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            boltz[p] = ls_exp_neg(-hc_kl * rT[p], c.sExp);
    }
    double e1[NP], u1[NP], x1[NP], e2[NP], u2[NP], x2[NP]; // eta_A, U(j), chi(j) of the two atom slots
#pragma unroll
    for (int p = 0; p < NP; ++p)
        e1[p] = u1[p] = x1[p] = e2[p] = u2[p] = x2[p] = 0.0;
    // walk 1: the shared level sums, continua in transition order (chi_eta_aux_accum :59-109)
    for (int q = 0; q < nCont; ++q)
    {
        const LaneFinRec1 rc = lane_fin_rec(recs, q);
        const unsigned code = (rc.flags & 1) ? rc.code : 0x9999u;
        if (code == 0x9999u)
            continue;
        double ni[NP], nj[NP], rat[NP];
        ls_lds<NP>(c.sDep, vK, c.oN + rc.nI, ni);
        ls_lds<NP>(c.sDep, vK, c.oN + rc.nJ, nj);
        ls_lds<NP>(c.sDep, vK, c.oRat + rc.rat, rat);
        const double* ppq = c.sPP + (q * c.L + c.r) * 3;
        const double p0 = ppq[0], p2 = ppq[2];
        const bool first = code == 0x1239u;
#pragma unroll
        for (int p = 0; p < NP; ++p)
        {
            const double vji = (rat[p] * boltz[p]) * p0;
            const double chi_c = ni[p] * p0 - nj[p] * vji;
            const double uji = p2 * vji;
            const double eta_c = nj[p] * uji;
            if (first)
            {
                x1[p] -= chi_c;
                u1[p] += uji;
                e1[p] += eta_c;
            }
            else
            {
                x2[p] -= chi_c;
                u2[p] += uji;
                e2[p] += eta_c;
            }
        }
    }
    // walk 2: the pure continua (compute_full_operator_rates :206-234 from the angle moments)
    for (int q = 0; q < nCont; ++q)
    {
        const LaneFinRec1 rc = lane_fin_rec(recs, q);
        if (!(rc.flags & 2))
            continue;
        double ni[NP], nj[NP], rat[NP];
        ls_lds<NP>(c.sDep, vK, c.oN + rc.nI, ni);
        ls_lds<NP>(c.sDep, vK, c.oN + rc.nJ, nj);
        ls_lds<NP>(c.sDep, vK, c.oRat + rc.rat, rat);
        const double* ppq = c.sPP + (q * c.L + c.r) * 3;
        const double vij = ppq[0], wla = ppq[1], p2 = ppq[2];
        const unsigned ops = rc.ops;
        const bool a0 = ops == 0x03291u, alone = ops == 0x0AB9Cu;
        const bool active = (rc.flags & 1) != 0;
        const int msel = (rc.ms < 0) ? -1 : (rc.ms == ms0) ? 0 : (rc.ms == ms1) ? 1 : -1;
        double* dst = sAcc + (unsigned)(rc.cs * 4 * c.LRD) + c.blk;
        auto integrals = [&](const int p, double& g0, double& g1, double& g2, double& g3) {
            const double vji = (rat[p] * boltz[p]) * vij;
            const double chi_c = ni[p] * vij - nj[p] * vji;
            const double uji = p2 * vji;
            const double eta_c = nj[p] * uji;
            const double oEta = alone ? eta_c : a0 ? e1[p] : e2[p];
            const double oUj = alone ? uji : a0 ? u1[p] : u2[p];
            const double Jp = Jk[P0 + p], M1p = M1[P0 + p];
            const double Meta = msel < 0 ? 0.0 : msel == 0 ? Me[0][P0 + p] : Me[1][P0 + p];
            const double Jeff = Jp - M1p * oEta - Meta;
            g0 = ((uji * Wsum + vji * Jeff) - (M1p * chi_c * oUj)) * wla;
            g1 = (vij * Jeff) * wla; // (U_i of a pure continuum's lower level is zero here)
            g2 = Jp * vij * wla;
            g3 = (uji * Wsum + Jp * vji) * wla;
        };
        if (c.sTurn)
        {
            // fixed-order mode: the integrals first, then the task's turn at the continuum's accumulator, then the adds -- only
            // these are serialised between the wavefronts of a workgroup
            double gq[4][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p)
                integrals(p, gq[0][p], gq[1][p], gq[2][p], gq[3][p]);
            ls_turn_wait<D>(c.sTurn, rc.cs, P0, NP, (int)(__builtin_amdgcn_readlane((int)recs.feed, q) >> 16) & 0xffff);
#pragma unroll
            for (int p = 0; p < NP; ++p)
            {
                if (!(c.act && P0 + p < c.nV))
                    continue;
                if (active)
                {
                    lds_add(dst + (P0 + p) * LR, gq[0][p]);
                    lds_add(dst + (D + P0 + p) * LR, gq[1][p]);
                }
                lds_add(dst + (2 * D + P0 + p) * LR, gq[2][p]);
                lds_add(dst + (3 * D + P0 + p) * LR, gq[3][p]);
            }
            ls_turn_done<D>(c.sTurn, rc.cs, P0, NP, c.lane);
            continue;
        }
#pragma unroll
        for (int p = 0; p < NP; ++p)
        {
            if (!(c.act && P0 + p < c.nV))
                continue;
            double g0, g1, g2, g3;
            integrals(p, g0, g1, g2, g3);
            if (active)
            {
                lds_add(dst + (P0 + p) * LR, g0);
                lds_add(dst + (D + P0 + p) * LR, g1);
            }
            lds_add(dst + (2 * D + P0 + p) * LR, g2);
            lds_add(dst + (3 * D + P0 + p) * LR, g3);
        }
    }
}

template <int D>
DEVINL void lane_finish(const CONST_AS TileArgs& a, const TileDyn& dyn, const DevLaneTile& tr, const int ti, const LaneCtx& c,
                        double* sAcc, unsigned long long* sDj, const int ms0, const int ms1, const double (&Jk)[D],
                        const double (&M1)[D], const double (&Me)[2][D])
{
    const int Ns = c.Ns, k0 = c.k0;
    int la = c.la;
    LS_KEEPI(la); // (nothing of the finish is computed before the rays are done)
    // the finish's records and parameters are requested first: they arrive while J goes out
    const int nCont = dyn.prdOnly ? 0 : tr.nCont;
    const LaneFinRecs recs = lane_fin_recs(a, ti, nCont, c.lane);
    // (the continua's per-wavelength parameters are in the wavefront's LDS block since lane_pre)
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
                    if (a.Jhost) // (lwhip_map_host_J: the caller's array is an output of the sweep, no copy afterwards)
                        GD(a.Jhost)[(size_t)la * Ns + k0 + p] = Jk[p];
                    dj = fmax(dj, fabs(1.0 - Jdag * ls_rcp(Jk[p]))); // :477-486 (the quotient by a refined reciprocal: 1e-16 of a convergence measure)
                }
            // non-negative doubles order as integers
            atomicMax(sDj + c.rRaw, (unsigned long long)__double_as_longlong(dj));
        }
    }
    if (nCont <= 0)
        return;
    static_assert(D % 2 == 0, "the block is finished in halves");
    if (tr.finFast == 1)
        lane_finish_fast<D, 0, D>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
    else if (tr.finFast == 2)
    {
        // (the wide encoding: 18 sums per point -- a point at a time)
        lane_finish_part<D, 0, 1, true>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
        LS_PHASE();
        lane_finish_part<D, 1, 1, true>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
        LS_PHASE();
        lane_finish_part<D, 2, 1, true>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
        LS_PHASE();
        lane_finish_part<D, 3, 1, true>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
    }
    else
    {
        lane_finish_part<D, 0, D / 2>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
        LS_PHASE();
        lane_finish_part<D, D / 2, D / 2>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
    }
}

// ---- a tile's rays split over the S wavefronts of a group (small problems: more, shorter tasks) -----------------------------
// Each wavefront of the group has marched its share of the rays and holds partial sums J, M1, Meta0, Meta1 of its lanes'
// D points.  Wavefront d finishes the points d * D/S .. of every block: it needs the other wavefronts' partials at those
// points.  Exchange area X[d][source][quantity][point][lane] (over the group's row blocks, which nobody reads any more);
// the totals are formed in wavefront order 0 .. S - 1, so J does not depend on timing (it does depend on S: a context
// pins S, LWHIP_LANE_SPLIT overrides).
template <int D, int S>
DEVINL void lane_split_exchange_write(double* sX, const int sub, const int lane, const double (&Jk)[D], const double (&M1)[D],
                                      const double (&Me)[2][D])
{
    constexpr int NP = D / S;
#pragma unroll
    for (int d = 0; d < S; ++d)
    {
        if (d == sub)
            continue;
        double* x = sX + (size_t)((d * (S - 1) + (sub < d ? sub : sub - 1)) * 4 * NP) * 64 + lane;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
        {
            x[(0 * NP + pp) * 64] = Jk[d * NP + pp];
            x[(1 * NP + pp) * 64] = M1[d * NP + pp];
            x[(2 * NP + pp) * 64] = Me[0][d * NP + pp];
            x[(3 * NP + pp) * 64] = Me[1][d * NP + pp];
        }
    }
}
template <int D, int S>
DEVINL void lane_split_exchange_read(const double* sX, const int sub, const int lane, double (&Jk)[D], double (&M1)[D], double (&Me)[2][D])
{
    constexpr int NP = D / S;
    double tot[4][NP];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
            tot[q][pp] = 0.0;
#pragma unroll
    for (int src = 0; src < S; ++src)
    {
        if (src == sub)
        {
#pragma unroll
            for (int dd = 0; dd < S; ++dd)
                if (dd == sub)
                {
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp)
                    {
                        tot[0][pp] += Jk[dd * NP + pp];
                        tot[1][pp] += M1[dd * NP + pp];
                        tot[2][pp] += Me[0][dd * NP + pp];
                        tot[3][pp] += Me[1][dd * NP + pp];
                    }
                }
        }
        else
        {
            const double* x = sX + (size_t)((sub * (S - 1) + (src < sub ? src : src - 1)) * 4 * NP) * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int pp = 0; pp < NP; ++pp)
                    tot[q][pp] += x[(q * NP + pp) * 64];
        }
    }
#pragma unroll
    for (int dd = 0; dd < S; ++dd)
        if (dd == sub)
        {
#pragma unroll
            for (int pp = 0; pp < NP; ++pp)
            {
                Jk[dd * NP + pp] = tot[0][pp];
                M1[dd * NP + pp] = tot[1][pp];
                Me[0][dd * NP + pp] = tot[2][pp];
                Me[1][dd * NP + pp] = tot[3][pp];
            }
        }
}
// lane_finish for wavefront `sub` of S: its D / S points of every block
template <int D, int S>
DEVINL void lane_finish_split(const CONST_AS TileArgs& a, const TileDyn& dyn, const DevLaneTile& tr, const int ti, const LaneCtx& c,
                              const int sub, double* sAcc, unsigned long long* sDj, const int ms0, const int ms1, const double (&Jk)[D],
                              const double (&M1)[D], const double (&Me)[2][D])
{
    constexpr int NP = D / S;
    const int Ns = c.Ns, k0 = c.k0;
    int la = c.la;
    LS_KEEPI(la);
    const int nCont = dyn.prdOnly ? 0 : tr.nCont;
    const LaneFinRecs recs = lane_fin_recs(a, ti, nCont, c.lane);
    {
        double dj = 0.0;
        if (c.act)
        {
            const GLOBAL_AS double* Jrow = GCD(a.J) + (size_t)la * Ns + k0;
#pragma unroll
            for (int dd = 0; dd < S; ++dd)
                if (dd == sub)
                {
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp)
                    {
                        const int p = dd * NP + pp;
                        if (p < c.nV)
                        {
                            const double Jdag = Jrow[p];
                            GD(a.J)[(size_t)la * Ns + k0 + p] = Jk[p];
                            if (a.Jhost)
                                GD(a.Jhost)[(size_t)la * Ns + k0 + p] = Jk[p];
                            dj = fmax(dj, fabs(1.0 - Jdag * ls_rcp(Jk[p]))); // :477-486 (the quotient by a refined reciprocal: 1e-16 of a convergence measure)
                        }
                    }
                }
            atomicMax(sDj + c.rRaw, (unsigned long long)__double_as_longlong(dj)); // non-negative doubles order as integers
        }
    }
    if (nCont <= 0)
        return;
    static_assert(D == 4, "the wide form below finishes the wavefront's points one at a time");
    if (tr.finFast == 2)
    {
        // (the wide encoding: a point at a time; this wavefront's NP points)
        if (sub * NP == 0)
            lane_finish_part<D, 0, 1, true>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
        if (sub * NP <= 1 && 1 < (sub + 1) * NP)
            lane_finish_part<D, 1, 1, true>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
        if (sub * NP <= 2 && 2 < (sub + 1) * NP)
            lane_finish_part<D, 2, 1, true>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
        if (sub * NP <= 3 && 3 < (sub + 1) * NP)
            lane_finish_part<D, 3, 1, true>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
        return;
    }
    if (sub == 0)
        lane_finish_part<D, 0, NP>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
    else if (sub == 1)
        lane_finish_part<D, NP, NP>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
    else if (S > 2 && sub == 2)
        lane_finish_part<D, (S > 2 ? 2 * NP : 0), NP>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
    else if (S > 2)
        lane_finish_part<D, (S > 2 ? 3 * NP : 0), NP>(a, c, nCont, recs, sAcc, ms0, ms1, Jk, M1, Me);
}

// One task = one tile (up to R wavelengths): every ray of it, or (S > 1) the share of wavefront `sub` of the tile's group.
template <int SOLVER, int D, int MODE>
DEVINL void lane_task(const CONST_AS TileArgs& a, const TileDyn& dyn, const DevLaneTile& tr, const int ti, const bool have, const int S,
                      const int sub, double* sAcc, unsigned long long* sDj, double* sB, const double* sExp, const char* sDep,
                      double* sRow3, double* sX, double* sPP, int* sTurn, const int lane)
{
    constexpr bool RATES = (MODE & 1) != 0;
    LS_T(tt0);
    const DevTile& td = tr.td;
    // this wavefront's rays of the iteration's list
    const int nRays = a.Nmu * dyn.nPass;
    const int shS = S == 4 ? 2 : S == 2 ? 1 : 0;
    const int q0 = (sub * nRays) >> shS, q1 = ((sub + 1) * nRays) >> shS;
    // what a lane is in its tile: formed for the rays and formed AGAIN for the finish from the lane number (a dozen integer
    // operations) -- kept alive over the ray loop it would be a dozen registers that the kernel does not have
    auto lane_ctx = [&](int ln) {
        asm volatile("" : "+v"(ln)); // (recomputed from here on)
        LaneCtx c;
        c.sPP = sPP;
        c.sTurn = sTurn;
        c.sub = sub;
        c.dbgWait = a.lsDbg ? a.lsDbg + (size_t)a.nTiles * 8 + (size_t)blockIdx.x * 4 + 3 : nullptr;
        c.ti = ti;
        c.sExp = sExp;
        c.sDep = sDep;
        c.sRow3 = sRow3;
        c.Ns = a.Ns;
        c.Nmu = a.Nmu;
        c.LR = a.laneLR;
        c.LRD = a.laneLRD;
        c.L = a.L;
        c.lane = ln;
        c.rRaw = ln / a.laneLR;
        c.blk = ln - c.rRaw * a.laneLR;
        c.act = c.rRaw < td.nl;
        c.r = min(c.rRaw, td.nl - 1);
        c.k0 = c.blk * D;
        c.la = td.la0 + c.r;
        c.nV = min(max(c.Ns - c.k0, 0), D);
        c.pBot = c.Ns - 1 - c.k0;
        c.vK = (unsigned)c.k0 * 8u;
        c.vRow = (unsigned)(c.r * c.LRD + c.k0) * 8u;
        c.rowStride = (unsigned)(c.L * c.LRD) * 8u;
        c.Ns8 = (unsigned)c.Ns * 8u;
        c.gStride = (unsigned)(c.LRD + 2) * 8u;
        c.oN = a.depN;
        c.oW = a.depW;
        c.oRat = a.depRat;
        c.oG = a.depG;
        c.rRows = a.rows + td.rowBase;
        c.sB = sB;
        c.wav = 0.0;
        return c;
    };
    LaneCtx c = lane_ctx(lane);
    double Jk[D], M1[D], Me[2][D];
#pragma unroll
    for (int p = 0; p < D; ++p)
        Jk[p] = M1[p] = Me[0][p] = Me[1][p] = 0.0;
    int ms0 = 0, ms1 = 1;
    if (have)
    {
        c.wav = ((const GLOBAL_AS DevLaneRay*)a.laneRays)[(size_t)ti * a.L + min(c.rRaw, a.L - 1)].wav;
        if (tr.gen >= 0)
            lane_pre<D, true>(a, tr, c, ti, RATES);
        else
            lane_pre<D, false>(a, tr, c, ti, RATES);
        LS_PHASE();
#if defined(LS_ISA_ONLY) && defined(LS_ONLY_NL) /* one tile kind, for register / instruction counts of the listing */
        ms0 = lane_rays<SOLVER, D, LS_ONLY_NL, LS_ONLY_NM, MODE>(a, dyn, tr, c, sAcc, q0, q1, Jk, M1, Me);
        if (LS_ONLY_NL < 2)
            ms1 = -2;
#else
        if (tr.gen >= 0)
        {
            if constexpr ((MODE & 2) == 0)
                lane_rays<SOLVER, D, -1, -1, MODE>(a, dyn, tr, c, sAcc, q0, q1, Jk, M1, Me); // the generic kind (not hybrid PRD)
        }
        else if (td.nLine == 0)
        {
            lane_rays<SOLVER, D, 0, 0, MODE>(a, dyn, tr, c, sAcc, q0, q1, Jk, M1, Me);
            ms0 = ms1 = -2;
        }
        else if (td.nLine == 1 && td.nMixed == 0)
        {
            ms0 = lane_rays<SOLVER, D, 1, 0, MODE>(a, dyn, tr, c, sAcc, q0, q1, Jk, M1, Me);
            ms1 = -2;
        }
        else if (td.nLine == 1)
        {
            ms0 = lane_rays<SOLVER, D, 1, 1, MODE>(a, dyn, tr, c, sAcc, q0, q1, Jk, M1, Me);
            ms1 = -2;
        }
        else
            lane_rays<SOLVER, D, 2, 0, MODE>(a, dyn, tr, c, sAcc, q0, q1, Jk, M1, Me);
#endif
    }
    LS_T(tt1);
    if (RATES && S > 1)
    {
        // (barriers of the whole workgroup: every wavefront passes here once per round, with or without a tile)
        __syncthreads(); // the group's wavefronts are done with their row blocks: the exchange area lies over them
        if (have)
        {
            if (S == 2)
                lane_split_exchange_write<D, 2>(sX, sub, lane, Jk, M1, Me);
            else
                lane_split_exchange_write<D, 4>(sX, sub, lane, Jk, M1, Me);
        }
        __syncthreads();
        if (have)
        {
            if (S == 2)
                lane_split_exchange_read<D, 2>(sX, sub, lane, Jk, M1, Me);
            else
                lane_split_exchange_read<D, 4>(sX, sub, lane, Jk, M1, Me);
        }
    }
#ifdef LS_ABL_NO_FINISH
    if (RATES && have && dyn.nPass == 77)
#else
    if (RATES && have)
#endif
    {
        c = lane_ctx(lane);
        c.wav = ((const GLOBAL_AS DevLaneRay*)a.laneRays)[(size_t)ti * a.L + min(c.rRaw, a.L - 1)].wav;
        if (S == 1)
            lane_finish<D>(a, dyn, tr, ti, c, sAcc, sDj, ms0, ms1, Jk, M1, Me);
        else if (S == 2)
            lane_finish_split<D, 2>(a, dyn, tr, ti, c, sub, sAcc, sDj, ms0, ms1, Jk, M1, Me);
        else
            lane_finish_split<D, 4>(a, dyn, tr, ti, c, sub, sAcc, sDj, ms0, ms1, Jk, M1, Me);
    }
    if (RATES && S > 1)
        __syncthreads(); // the group's dJ maxima are complete; the row blocks may be overwritten by the next round
#ifdef LS_TIMING
    {
        LS_T(tt2);
        if (lane == 0 && a.lsDbg)
        {
            double* o = a.lsDbg + (size_t)ti * 8;
            o[6] = (double)(tt2 - tt1);
            o[7] = (double)(tt2 - tt0) + 0.01 * (double)(td.nLine * 10 + td.nMixed); // (the tile's kind rides in the fraction)
            o[0] = (double)(tt1 - tt0) - o[1] - o[2] - o[3] - o[4] - o[5];
        }
    }
#endif
}

// Workgroup = W wavefronts, each takes tiles of the workgroup's chunk in turn.
template <int SOLVER, int D, int MODE, bool BATCH>
#ifndef LS_WAVES_PER_EU
#define LS_WAVES_PER_EU 2
#endif
__global__ void __launch_bounds__(LS_MAX_WAVES * 64) __attribute__((amdgpu_waves_per_eu(LS_WAVES_PER_EU))) lanesweep_kernel(const TileArgs* __restrict__ ap0,
                                                                       const TileArgs* const* __restrict__ apList, const TileDyn dyn)
{
    dbg_poison_lds();
    if (dyn.stopCtl)
    {
        // (the PRD rates pass of a pipelined sub-iteration: an earlier sub-iteration met the tolerance -- nothing to do)
        const int stopAt = __builtin_amdgcn_readfirstlane(*(const volatile int32_t*)dyn.stopCtl);
        if (stopAt != 0 && stopAt < dyn.stopIter)
            return;
    }
    LS_T(tk0);
    constexpr bool RATES = (MODE & 1) != 0;
    extern __shared__ double lds[];
    const TileArgs* __restrict__ ap = BATCH ? apList[blockIdx.y] : ap0;
    const CONST_AS TileArgs& a = *(const CONST_AS TileArgs*)ap;
    const int Ns = a.Ns, LRD = a.laneLRD;
    const int maxCT = a.maxCT;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nWaves = blockDim.x >> 6;
    // the workgroup's record: its chunk, the chunk's tiles and split factor (S > 1: the rays of a tile are split over the S
    // consecutive wavefronts of a group, sub = 0 .. S - 1 -- small launches, and the short chunks of a long launch's tail)
    const DevLaneWg wg = ld_c(CTAB(DevLaneWg, a.laneWg) + blockIdx.x);
    const int chunk = wg.chunk;
    const int S = wg.split;
    // (S is 1, 2 or 4: shifts, and the results pinned in SGPRs -- an integer division is done in the vector unit and would
    // leave every pointer derived from it in VGPRs)
    const int sh = S == 4 ? 2 : S == 2 ? 1 : 0;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> sh), sub = __builtin_amdgcn_readfirstlane(wave & (S - 1)),
              nGroups = __builtin_amdgcn_readfirstlane(nWaves >> sh);
    double* sAcc = lds;                                                             // [maxCT][4][LRD]
    unsigned long long* sDj = (unsigned long long*)(lds + (size_t)maxCT * 4 * LRD) + grp * LS_MAX_R; // per group [R]
    double* sB = lds + (size_t)maxCT * 4 * LRD + (size_t)nWaves * LS_MAX_R + (size_t)wave * (LS_MAX_R * 4); // per wave [R][4]
    double* sExp = lds + (size_t)maxCT * 4 * LRD + (size_t)nWaves * LS_MAX_R * 5;                              // [LS_EXP_TAB]
    double* sRow3 = sExp + LS_EXP_TAB + (size_t)wave * (3 * 64 * D);                                            // per wave [3][64][D]
    double* sX = sExp + LS_EXP_TAB + (size_t)(grp * S) * (3 * 64 * D); // the group's exchange area: over its row blocks, after the rays
    // (the per-wave parameter blocks hold the most continua any wavelength of THIS context has, not the kernel's capacity)
    const int nContLds = min(a.maxC, (int)LWHIP_POST_MAXC);
    double* sPP = sExp + LS_EXP_TAB + (size_t)nWaves * (3 * 64 * D) + (size_t)wave * (nContLds * a.L * 3); // per wave [continuum][ray][3]
    double* sDep = sExp + LS_EXP_TAB + (size_t)nWaves * (3 * 64 * D) + (size_t)nWaves * (nContLds * a.L * 3); // [depBytes / 8]
    // fixed-order mode with several wavefronts per workgroup: the accumulators' turn counters [maxCT][D] behind the depth arena
    // (one wavefront alone adds in program order anyway)
    constexpr bool HPRD_K = (MODE & 2) != 0;
    int* sTurn = (RATES && !HPRD_K && a.detSlab && nWaves > 1) ? (int*)((char*)sDep + a.depBytes) : nullptr;
    if (sTurn)
    {
        for (int idx = t; idx < maxCT * D; idx += blockDim.x)
            sTurn[idx] = 0;
        if (t == 0)
            *ls_late_counter() = 0;
    }
#ifdef LS_TIMING
    if (t == 0 && a.lsDbg && !BATCH)
        a.lsDbg[(size_t)a.nTiles * 8 + (size_t)blockIdx.x * 4 + 3] = 0.0;
#endif
    ls_exp_table_init(sExp, t);
    {
        // the depth-only pools (populations, w_phi, nStar ratios, geometry) are the same for every tile: one LDS copy per
        // workgroup serves all rays of all its tiles
        // (eight loads in flight per thread: one after the other the copy is ~13 dependent memory round trips per workgroup)
        const GLOBAL_AS double* src = GCD(a.depBase);
        const int nDep = (int)(a.depBytes >> 3);
        const int T = blockDim.x;
        for (int base = t; base < nDep; base += 8 * T)
        {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                v[q] = base + q * T < nDep ? src[base + q * T] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (base + q * T < nDep)
                    sDep[base + q * T] = v[q];
        }
    }
    const int t0 = wg.t0, t1 = wg.t1;
    if (RATES)
    {
        for (int idx = t; idx < maxCT * 4 * LRD; idx += blockDim.x)
            sAcc[idx] = 0.0;
    }
    __syncthreads();
    const CONST_AS int32_t* tileList = CTAB(int32_t, a.tileList);
    LS_T(tk1);
    // rounds of the workgroup: each group takes a tile of the chunk (with S > 1 every wavefront passes the round's barriers,
    // also a group that has no tile left)
    const int nRounds = __builtin_amdgcn_readfirstlane((t1 - t0 + nGroups - 1) / nGroups);
    for (int rd = 0; rd < nRounds; ++rd)
    {
        const int li0 = t0 + rd * nGroups + grp;
        const bool have = li0 < t1;
        if (S == 1 && !have)
            break;
        const int li = have ? li0 : t0;
        const int ti = tileList ? tileList[li] : li;
        const DevLaneTile tr = ld_c(CTAB(DevLaneTile, a.laneTiles) + ti);
        if (RATES && lane < LS_MAX_R && (S == 1 || sub == 0))
            sDj[lane] = 0ull; // (S > 1: before the task's first barrier, the maxima are added after it)
        lane_task<SOLVER, D, MODE>(a, dyn, tr, ti, have, S, sub, sAcc, sDj, sB, sExp, (const char*)sDep, sRow3, sX, sPP, sTurn, lane);
        if (RATES && have && sub == 0 && lane < tr.td.nl)
            GD(a.dJ)[tr.td.la0 + lane] = __longlong_as_double((long long)sDj[lane]);
    }
    LS_T(tk2);
    if (RATES)
    {
        __syncthreads();
        // the chunk's sums go straight into the stage-1 buffer: part = chunk % SPLIT keeps 1/SPLIT of the chunks on one address.
        // [slot][quantity][point of the block][block] -> depth k = block * D + point: a wavefront takes a quantity, a lane a
        // depth -- no integer division per element (the flat loop's three per element were 8 k of a workgroup's 130 k clocks)
        const CONST_AS int32_t* slotTr = CTAB(int32_t, a.slotTr) + (size_t)chunk * maxCT;
        GLOBAL_AS double* r8 = GD(a.red8);
        const int n4 = 4 * Ns;
        const int part = chunk % LWHIP_REDUCE_SPLIT;
        for (int k = lane; k < Ns; k += 64) // (consecutive lanes, consecutive depths: a wavefront's atomics cover whole lines)
        {
            const int pb = (k % D) * a.laneLR + k / D;
            for (int q = wave; q < 4; q += nWaves)
            {
                for (int cs = 0; cs < maxCT; ++cs)
                {
                    const int tr = slotTr[cs];
                    const double v = sAcc[(cs * 4 + q) * LRD + pb];
                    if (a.detSlab)
                    {
                        // deterministic mode (LWHIP_OPT_DETERMINISTIC): the workgroup -- one wavefront, so its LDS sums were
                        // formed in program order -- stores its slab; det_reduce_kernel adds the slabs in workgroup order
                        GD(a.detSlab)[((size_t)chunk * maxCT + cs) * n4 + q * Ns + k] = v;
                    }
                    else if (tr >= 0)
                        unsafeAtomicAdd((double*)(r8 + ((size_t)tr * LWHIP_REDUCE_SPLIT + part) * n4 + q * Ns + k), v);
                }
            }
        }
    }
    if (sTurn && t == 0 && a.turnLate)
    {
        const int late = *ls_late_counter();
        if (late)
            atomicAdd_system(a.turnLate, late);
    }
#ifdef LS_TIMING
    if (t == 0 && a.lsDbg && !BATCH)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LS_T(tk3);
        double* o = a.lsDbg + (size_t)a.nTiles * 8 + (size_t)blockIdx.x * 4;
        o[0] = (double)(tk1 - tk0);
        o[1] = (double)(tk2 - tk1);
        o[2] = (double)(tk3 - tk2);
    }
#endif
}

// Deterministic mode: red8[tr][part 0][q][k] = sum of the workgroups' slabs that hold transition tr, in a FIXED order
// (the list of (workgroup, slot) pairs per transition is the host's): bit-reproducible Gamma and rates, as the reference's
// single-thread path and its fixed-order thread reduction are (ThreadStorage.cpp:343-396).  Round 4: a two-level sum of fixed
// shape instead of one thread walking a whole list -- a continuum's list has one entry per workgroup, 3 429 at the bench size, and
// the walk was 0.7 ms of latency.  Segment s of DET_SEG covers list positions [L s / DET_SEG, L (s + 1) / DET_SEG) -- a function
// of the list's length alone --, is added up in list order (eight loads in flight), and the DET_SEG partial sums are added in
// segment order.
enum { DET_SEG = LWHIP_DET_SEG };
__global__ void __launch_bounds__(256) det_reduce_kernel(const double* __restrict__ slab, const int32_t* __restrict__ off,
                                                         const int32_t* __restrict__ ent, double* __restrict__ part, int Ns)
{
    dbg_poison_lds();
    const int tr = blockIdx.y, seg = blockIdx.z;
    const int n4 = 4 * Ns;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4)
        return;
    const int e0 = off[tr];
    const long long L = off[tr + 1] - e0;
    int e = e0 + (int)((L * seg) / DET_SEG);
    const int e1 = e0 + (int)((L * (seg + 1)) / DET_SEG);
    double sum = 0.0;
    for (; e + 8 <= e1; e += 8)
    {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            v[q] = slab[(size_t)ent[e + q] * n4 + i]; // ent = workgroup * maxCT + slot
#pragma unroll
        for (int q = 0; q < 8; ++q)
            sum += v[q];
    }
    for (; e < e1; ++e)
        sum += slab[(size_t)ent[e] * n4 + i];
    part[((size_t)tr * DET_SEG + seg) * n4 + i] = sum;
}

#ifdef LS_ISA_ONLY /* one instance, for instruction counts of the listing (tools/isa_loops.py) */
#ifndef LS_ISA_SOLVER
#define LS_ISA_SOLVER LWHIP_FS_BEZIER3_1D
#define LS_ISA_MODE 1
#define LS_ISA_BATCH false
#endif
template __global__ void lanesweep_kernel<LS_ISA_SOLVER, 4, LS_ISA_MODE, LS_ISA_BATCH>(const TileArgs* __restrict__, const TileArgs* const* __restrict__,
                                                                                       const TileDyn);
#else
// Diagnosis (LWHIP_DEBUG=1 LWHIP_SCRATCH_POISON=1): every wavefront slot's private-segment ("scratch") memory is filled with a
// finite pattern in front of EVERY kernel launch of the library (LWHIP_LAUNCH).  A register the compiler spills while some lanes are switched off is
// stored for the active lanes only; reloaded with all lanes on, the others get what the scratch slot held before -- what an
// earlier kernel of this or another process left there.  Under this fill such a reload reaches the results on every run.
__global__ void __launch_bounds__(256) scratch_poison_kernel(double* sink, const double pat, const int n)
{
    volatile double buf[320]; // (2560 bytes per lane: more than any kernel's private segment in this library)
    for (int i = 0; i < 320; ++i)
        buf[i] = pat * (double)(1 + ((threadIdx.x + i) & 7));
    double sum = 0.0;
    for (int i = 0; i < n; ++i) // (n = 0: nothing is read back; the stores above stay because the array is volatile)
        sum += buf[i % 320];
    if (n > 0)
        sink[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

} // namespace

void dbg_scratch_poison(hipStream_t stream)
{
    static const bool poison = std::getenv("LWHIP_DEBUG") && std::getenv("LWHIP_SCRATCH_POISON");
    if (poison)
        hipLaunchKernelGGL(scratch_poison_kernel, dim3(8192), dim3(256), 0, stream, (double*)nullptr, 1.0e-3, 0);
}

namespace
{
template <int SOLVER, int D, int MODE>
hipError_t launch_ls_t(const TileArgs* devArgs, const TileArgs& a, const TileDyn& dyn, int nChunks, int waves,
                       const TileArgs* const* apList, int nBatch, hipStream_t stream)
{
    const size_t lds = lane_sweep_lds_bytes(a, waves);
    if (lds > 48 * 1024)
    {
        // (once per kernel instance, device and size: the call is a driver round trip of a microsecond or two in front of
        // EVERY launch otherwise, while the device idles between two iterations)
        static std::atomic<size_t> granted[2][16];
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<size_t>& g = granted[apList ? 1 : 0][dev & 15];
        if (g.load(std::memory_order_relaxed) < lds)
        {
            hipError_t e = hipFuncSetAttribute(apList ? (const void*)lanesweep_kernel<SOLVER, D, MODE, true>
                                                      : (const void*)lanesweep_kernel<SOLVER, D, MODE, false>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess)
                return e;
            g.store(lds, std::memory_order_relaxed);
        }
    }
    if (apList)
        LWHIP_LAUNCH((lanesweep_kernel<SOLVER, D, MODE, true>), dim3(nChunks, std::max(nBatch, 1)), dim3(waves * 64), lds, stream,
                           devArgs, apList, dyn);
    else
        LWHIP_LAUNCH((lanesweep_kernel<SOLVER, D, MODE, false>), dim3(nChunks), dim3(waves * 64), lds, stream, devArgs, apList, dyn);
    return hipGetLastError();
}
#endif
}

// D: depth points per lane; LR: lanes per ray; R: rays (wavelengths) per wavefront
bool lane_sweep_supported(int Ns, int Nrays, int solver, int* D, int* LR, int* R)
{
    (void)Nrays;
    if (solver != LWHIP_FS_BEZIER3_1D && solver != LWHIP_FS_LINEAR_1D && solver != LWHIP_FS_BESSER_1D)
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

size_t det_reduce_scratch_doubles(int Ns, int Ntrans) { return (size_t)std::max(Ntrans, 1) * DET_SEG * 4 * Ns; }

// (round 5: the DET_SEG segment sums are what the apply launch -- or stage 2 of a sharded iteration -- reads as the transition's
// "parts", in segment order: the second-level launch of round 4 is gone)
hipError_t launch_det_reduce(const double* slab, const int32_t* off, const int32_t* ent, double* part, int Ns, int Ntrans,
                             hipStream_t stream)
{
    if (Ntrans <= 0)
        return hipSuccess;
    LWHIP_LAUNCH(det_reduce_kernel, dim3((4 * Ns + 255) / 256, Ntrans, DET_SEG), dim3(256), 0, stream, slab, off, ent, part, Ns);
    return hipGetLastError();
}

size_t lane_sweep_lds_bytes(const TileArgs& a, int waves)
{
    return sizeof(double) * ((size_t)a.maxCT * 4 * a.laneLRD + (size_t)waves * LS_MAX_R * 5 + LS_EXP_TAB + (size_t)waves * 3 * 64 * a.laneD
                             + (size_t)waves * std::min(a.maxC, (int)LWHIP_POST_MAXC) * a.L * 3)
           + a.depBytes + (a.detSlab ? ((size_t)a.maxCT * a.laneD * sizeof(int) + 7) / 8 * 8 : 0); // (+ the fixed-order mode's turn counters)
}

#ifndef LS_ISA_ONLY
hipError_t launch_lane_sweep(const TileArgs* devArgs, const TileArgs& a, const TileDyn& dyn, int solver, bool rates, int nChunks,
                             int waves, const TileArgs* const* apList, int nBatch, hipStream_t stream)
{
    if (nChunks <= 0)
        return hipSuccess;
    if (waves < 1 || waves > LS_MAX_WAVES || a.laneD != 4)
        return hipErrorInvalidValue;
    // (the linear and BESSER solvers: one instance each per rates / no rates, with the hybrid-PRD paths compiled in)
    if (solver == LWHIP_FS_LINEAR_1D)
        return rates ? launch_ls_t<LWHIP_FS_LINEAR_1D, 4, 3>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream)
                     : launch_ls_t<LWHIP_FS_LINEAR_1D, 4, 2>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream);
    if (solver == LWHIP_FS_BESSER_1D)
        return rates ? launch_ls_t<LWHIP_FS_BESSER_1D, 4, 3>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream)
                     : launch_ls_t<LWHIP_FS_BESSER_1D, 4, 2>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream);
    if (solver != LWHIP_FS_BEZIER3_1D)
        return hipErrorInvalidValue;
    const bool hprd = a.hRho != nullptr || a.hJOff != nullptr;
    if (rates)
        return hprd ? launch_ls_t<LWHIP_FS_BEZIER3_1D, 4, 3>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream)
                    : launch_ls_t<LWHIP_FS_BEZIER3_1D, 4, 1>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream);
    return hprd ? launch_ls_t<LWHIP_FS_BEZIER3_1D, 4, 2>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream)
                : launch_ls_t<LWHIP_FS_BEZIER3_1D, 4, 0>(devArgs, a, dyn, nChunks, waves, apList, nBatch, stream);
}
#endif
}

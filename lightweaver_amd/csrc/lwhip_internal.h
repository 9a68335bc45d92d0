// lwhip_internal.h -- device-side tables shared by the host API (lwhip_host.h and the files it lists) and the kernels
// (lwhip_kernels.hip).  Not part of the public ABI (that is include/lwhip.h).
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lwhip.h"

namespace lwhip
{
// physical constants, bit-identical to the reference (Source/Constants.hpp:6-47)
constexpr double CLight = 2.99792458E+08;
constexpr double HPlanck = 6.6260755E-34;
constexpr double HC = HPlanck * CLight;
constexpr double KBoltzmann = 1.380658E-23;
constexpr double Pi = 3.14159265358979323846264338327950288;
constexpr double NM_TO_M = 1.0E-09;
constexpr double HC_K = HC / (KBoltzmann * NM_TO_M);           // hc_k, LwAtom.hpp:88
constexpr double HC_4PI = 0.25 * HC / Pi;                      // hc_4pi, LwAtom.hpp:86
constexpr double PI4_HC = 1.0 / HC_4PI;                        // pi4_hc, LwAtom.hpp:87
constexpr double PI4_H = 4.0 * Pi / HPlanck;                   // pi4_h,  LwAtom.hpp:85
constexpr double TWO_HC_NM3 = 2.0 * HC / (NM_TO_M * NM_TO_M * NM_TO_M); // twoHc, LwTransition.hpp:132

// One radiative transition, global index over all atoms in reference order
// (atoms in descriptor order, transitions in kr order).
struct DevTrans
{
    int32_t type;       // LWHIP_LINE / LWHIP_CONTINUUM
    int32_t atom;       // atom index
    int32_t gi, gj;     // global level rows (levelOff[atom] + i / j) into the n pool
    int32_t Nblue, Nred;
    int32_t prd;        // rhoPrd multiplies gij
    int32_t op;         // 1: atom is active (Gamma accumulated); 0: detailed (rates only)
    int32_t row;        // lines: row in the wphi pool; continua: row in the nStar-ratio pool
    int32_t _pad;
    int64_t parOff;     // offset (doubles) of this transition's per-wavelength params, 4 per lt
    int64_t phiOff;     // offset (doubles) into the phi pool (lines)
    int64_t rhoOff;     // offset (doubles) into the rhoPrd pool or -1
};
// per-(transition, lt) params, 4 doubles:
//   line:      { hnu_4pi*Bij, wlambda, Bji/Bij, Aji/Bji }      LwTransition.hpp:107-113,128-131; LwAtom.hpp:97-105
//   continuum: { alpha(lt),   wlambda/lambda*pi4_h, twoHc/lambda^3, 0 }   LwTransition.hpp:132-142; LwAtom.hpp:107-115

// Per-wavelength activity: header + slot records.  Slots of one wavelength are stored as
//   [lines (reference order)] [mixed continua] [pure continua]
// where a continuum of an active atom is "mixed" when an active line of the same atom shares one
// of its levels at this wavelength (its Gamma terms then depend on the ray through the line
// opacity and it is integrated ray by ray like a line); every other continuum is "pure": all of
// its ray dependence is in I and Psi*, so its Gamma / rate integrals follow from the angle moments
// J, M1 = sum w Psi*, Meta = sum w Psi* eta_line.
struct DevLaHeader
{
    int32_t off;        // first slot record
    int16_t nLine;
    int16_t nMixed;
    int16_t nPure;
    int16_t prog;       // row program of the continuum sums
    int32_t hasPrd;     // a PRD line (rhoPrd) is active at this wavelength
    int32_t rowBase;    // first row of this wavelength's block in the continuum-row buffer, -1: no continua
    int32_t nRows;      // rows of the program (the block holds nRows + nMixed + nPure rows)
};
struct DevSlot          // 96 bytes, read with scalar loads
{
    int32_t cs;         // chunk-local accumulator slot
    int32_t gi, gj;     // global level rows
    int32_t atom;
    int32_t flags;      // bit 0: atom is active (op); bit 1: rhoPrd multiplies gij
    int32_t row;        // lines: row in the wphi pool; continua: row in the nStar-ratio pool
    int32_t ms;         // moment slot (0/1) of the atom's active lines at this wavelength, or -1
    int32_t _pad;
    double p[4];        // per-(transition, lt) params (see above)
    int64_t phiOff;     // lines: offset of this wavelength's [Nrays,2,Ns] block in the phi pool
    int64_t rhoOff;     // PRD lines: offset of this wavelength's [Ns] row in the rho pool
    // rows of this wavelength's block in the continuum-row buffer (march kernel), -1: identically zero
    int16_t rEtaA, rChiI, rChiJ, rUI, rUJ;
    int16_t rVji;       // continua: row of Vji_c (= nRows + continuum ordinal)
    int16_t _pad2[2];
};
static_assert(sizeof(DevSlot) == 96, "DevSlot is read with fixed-size scalar loads");
// Row program: which continua feed which ray-independent per-depth row.  Rows are laid out
// contiguously in LDS as [chiC][etaC][etaA x Natom][levChi x NlevTot][levU x NlevTot].
struct DevProgRow
{
    int16_t row;        // row id in the layout above
    int16_t nEnt;
    int32_t entOff;     // first entry (relative to the program's entry base)
};
struct DevProgEnt
{
    int16_t c;          // continuum ordinal at this wavelength (0 .. nMixed+nPure-1)
    int16_t kind;       // 0: +chi_c  1: -chi_c  2: +Uji_c  3: +eta_c
};
struct DevProgram
{
    int32_t rowOff, nRows, entOff, nEnt;
};
enum { LWHIP_MAX_PROG_ROWS = 80, LWHIP_MAX_PROG_ENTS = 320 };

// Arguments of the continuum-row kernel of the 2D path (lwhip_2d.hip); lives in device memory.
struct ContArgs
{
    int32_t Ns, maxC;      // maxC: max continua active at one wavelength
    int32_t nContLa, _pad;
    const double* temperature;
    const double* wavelength;
    const double* bgChi;
    const double* bgEta;
    const double* n;               // [NlevTot, Ns] pool
    const double* ratio;           // [Ncont, Ns]  nStar_i / nStar_j
    const DevLaHeader* laHdr;      // [Nla]
    const DevSlot* slots;
    const DevProgram* progs;
    const DevProgRow* progRows;
    const DevProgEnt* progEnts;
    double* rows;                  // continuum rows written by cont_kernel: [rowBase + r][Ns] per wavelength
    const int32_t* contLa;         // [nContLa] wavelengths that have continua
};

// ---- the ray-column sweep (lwhip_raymarch.hip) -------------------------------------------------------------
// A *tile* is a run of up to L = min(16, 64 / Nmu) consecutive wavelengths with identical structure (same active
// lines, same "mixed" continua, same continuum row program).  One wavefront marches a tile through depth:
// lane = mu * L + j owns the ray (wavelength la0 + j, angle mu); every depth-dependent quantity is then uniform
// across the wavefront (scalar loads), every wavelength-dependent one is a lane constant.  Everything a tile
// reads per (wavelength, depth) sits in a tile-blocked, depth-major, wavelength-fastest row block written by the
// pre-pass:
//   rows[rowBase + (k * nRowsT + r) * L + j]    r = 0: chi_C  1: eta_C  2: sca * Jdag  3..: per-slot rows
// (one base address per depth, the rows of a depth contiguous), the line profiles in
//   phiT[phiOff + ((dir * Ns + k) * 64 + lane)]
// and the depth-only scalars (geometry, populations, w_phi) in the depth table kt[k][...].
struct DevTile
{
    int32_t la0;        // first wavelength (context-local)
    int16_t nl;         // wavelengths in the tile (<= L)
    int16_t nLine;      // line slots (first in the per-wavelength slot order)
    int16_t nMixed;     // mixed-continuum slots (follow the lines)
    int16_t hasPrd;
    int32_t slotOff;    // first DevTileSlot
    int32_t copyOff;    // pre-pass: first DevTileCopy
    int32_t nCopy;
    int32_t nRowsT;     // rows of the tile's block
    int32_t _pad;
    int64_t rowBase;    // offset (doubles) of the row block in the row pool
    int64_t momBase;    // offset (doubles) of the [down, up][J, M1, Meta0, Meta1][L][Ns] moment block
};
struct DevTileSlot
{
    int32_t tr;         // global transition
    int32_t cs;         // chunk-local accumulator slot
    int32_t gi, gj;     // global level rows
    int32_t atom;
    int32_t flags;      // bit 0: active atom; bit 1: rhoPrd multiplies gij
    int32_t row;        // lines: row in the wphi pool
    int32_t ms;         // moment slot of the atom's lines (0 / 1) or -1
    int16_t r0;         // first of the slot's five tile rows: eta_A, chi_i, chi_j, U_i, U_j (continuum parts; zero-filled
                        // where the wavelength has none)
    int16_t rAux;       // PRD lines: row of rho; mixed continua: row of Vji; -1: none
    int16_t e;          // ordinal among the wavelength's slot records (DevLaHeader::off + e)
    int16_t zmask;      // bit q: row r0 + q is identically zero at this tile's wavelengths (no continuum feeds it)
    int64_t phiOff;     // lines: offset (doubles) of the tile's [2][Ns][64] block in the phiT pool; lane sweep: of the
                        // first wavelength's [Nmu][2][Ns] block in the phi pool itself
    int64_t rhoOff;     // PRD lines: offset (doubles) of the first wavelength's [Ns] row in the rho pool, else -1
                        // (hybrid PRD: of the LINE's first row -- the interpolation indices count from there)
    int64_t hCoefOff;   // hybrid PRD lines: offset (entries) of the first wavelength's [Nmu][2][Ns] block of rho
                        // interpolation coefficients in TileArgs::hRho, else -1
};
// Pure continua of a tile for the lane sweep's in-wave post-pass: the level sums they need that have more than one
// contributor (or a foreign one) live in up to 8 registers per point ("shared sums"); everything else is the
// continuum's own term or zero.  Per continuum ordinal q of the wavelength's transition order:
//   code[q]: 4 x 4 bits, the entry (1 .. 8: a shared sum; 9: none) that continuum q's terms +chi_c, -chi_c, +U_ji, +eta_c are added to
//   ops[q]:  5 x 4 bits (eta_A, chi_i, U_j, chi_j, U_i of pure continuum q): 0 zero, 1..8 shared sum m = value - 1,
//            9 +chi_c, 10 -chi_c, 11 U_ji, 12 eta_c of the continuum itself
enum { LWHIP_POST_MAXC = 24, LWHIP_POST_SUMS = 8, LWHIP_POST_SUMS_WIDE = 14 }; // (wide encoding: 8 / 6 bits per field instead of 4, see lwhip_tables.hip; 14 sums + the zero and the
                                            // spare entry are a 16-entry register vector: other lengths are legalised through scratch memory)
struct DevPostProg
{
    uint32_t code[LWHIP_POST_MAXC];
    uint32_t ops[LWHIP_POST_MAXC];
};
// One continuum of a wavelength for the pre / post passes: everything uniform about it ready to use (byte offsets, level
// column slots), 32 bytes = one scalar load.  Stored per wavelength in the reference's transition order (the order in
// which chi_eta_aux_accum adds the continua), at DevLaHeader::off + nLine.
struct DevContRec
{
    uint32_t nI, nJ;            // byte offsets of n[gi][0], n[gj][0] in the population pool
    uint32_t rat;               // byte offset of ratio[row][0] in the nStar-ratio pool
    uint32_t pOff;              // byte offset of this continuum's p[0] relative to the wavelength's first slot record
    uint16_t sEta, sChiI, sChiJ, sUi, sUj; // level-column slots: eta_A(atom), chi(i), chi(j), U(i), U(j)
    uint8_t flags;              // bit 0: active atom   bit 1: "pure" (integrated by the post-pass from the angle moments)
    int8_t ms;                  // moment slot of the atom's lines (0 / 1) or -1
    int16_t tr;                 // global transition
    int16_t cs;                 // lane sweep: accumulator slot of a pure continuum inside its wavelength's workgroup chunk
};
static_assert(sizeof(DevContRec) == 32, "DevContRec is fetched with one scalar load");
struct DevTileCopy      // what the pre-pass writes into the tile's row block besides rows 0..2
{
    int16_t kind;       // 0: program row (dense index src)  1: Vji of continuum ordinal src  2: rho of line slot src  3: zero
    int16_t src;
    int16_t dst;        // tile row
    int16_t _pad;
};
// ---- lane sweep: what a task needs about its workgroup and tile behind ONE index each.  The chain kernel argument ->
// workgroup record -> tile record -> data is three dependent memory round trips; through the general tables
// (chunkOrder -> chunkSplit / chunkTile -> tiles -> tslots, laHdr -> slots, contRec -> slots) it was eight, ~30 k clocks of a
// 130 k-clock task (profiles/r04_lanesweep.md).  Built once per context (lwhip_create); atmosphere independent.
struct DevLaneWg            // per workgroup, in dispatch order
{
    int32_t chunk;          // the chunk it works on (slotTr row, reduce part)
    int32_t split;          // wavefronts a tile's rays are split over (1, 2, 4)
    int32_t t0, t1;         // the chunk's tiles [t0, t1) (indices into tileList where the launch has one)
};
struct DevLaneTile          // per tile: uniform, one burst of scalar loads
{
    DevTile td;
    int32_t nCont;          // continua the finish walks (0: none are pure): the first wavelength's nMixed + nPure
    int32_t finFast;        // 1: every continuum carries one of the common program words: the finish's named-register form;
                            // 2: the program words are in the WIDE encoding (more than 8 shared sums): the general form's wide instance
    int32_t nContAll;       // continua of the wavelength (the task's own pre-pass walks them all): nMixed + nPure
    int32_t gen;            // -1: one of the compiled kinds (slot[] holds its slots); >= 0: the generic kind -- any number of lines
                            // and mixed continua, slots read from TileArgs::tslots --, index into TileArgs::laneFeedG
    DevTileSlot slot[2];    // the tile's line / mixed-continuum slots (cs: of this launch's chunking; flags bits 8 ..: the task's
                            // turn at the slot's accumulator in the fixed-order mode)
};
enum { LWHIP_GEN_MAXS = 16 }; // line + mixed-continuum slots of a generic tile (8 + 8)
static_assert(sizeof(DevLaneTile) % 16 == 0, "DevLaneTile is read with wide scalar loads");
struct DevLaneRay           // per (tile, ray slot r < L): what depends on the wavelength (r >= nl repeats the last one)
{
    double wav;
    double p[2][4];         // the slots' per-(transition, wavelength) parameters (DevSlot::p)
    double _pad[3];
};
struct DevLaneFin           // per (tile, continuum q of the transition order): the finish's words, one per lane
{
    uint32_t nI, nJ, rat;   // DevContRec byte offsets
    uint32_t w6, w7;        // DevContRec dwords 6, 7: sUj | flags << 16 | ms << 24, tr | cs << 16
    uint32_t code, ops;     // DevPostProg words
    // which of the tile's slots' continuum rows this continuum feeds (the task's own pre-pass: Atom::setup_wavelength +
    // chi_eta_aux_accum for the continua, SimdFullIterationTemplates.hpp:59-109): byte e = slot e, bit 0 eta_A += eta_c,
    // bit 1 / 2 chi_i += / -= chi_c, bit 3 / 4 chi_j += / -= chi_c, bit 5 U_i += U_ji, bit 6 U_j += U_ji, bit 7: the
    // continuum IS mixed slot e (its V_ji row)
    uint32_t feed;
};
struct TileArgs
{
    int32_t Ns, Nmu, L, nTiles, Natom, NlevTot, Ntrans, maxCT, maxC;
    int32_t lowerBc, upperBc, lowerNmu, upperNmu, storeDepth;
    int32_t Nla, Nline;
    int32_t ktStride, _padK;       // doubles per depth of the depth table
    double* kt;                    // [Ns + 2][ktStride] (entry 0 and Ns + 1 are padding): |h_k - h_k+1|, its reciprocal, 1 / (|h_k-1 - h_k| + |h_k - h_k+1|), 1 / T_k,
                                   // n[0 .. NlevTot)[k], wphi[0 .. Nline)[k]  (refreshed by the pre-pass)
    const DevTile* tiles;
    const DevTileSlot* tslots;
    const DevTileCopy* copies;
    const int32_t* tileRemap; // [tile][Natom + 2 NlevTot + 1]: level-sum slot -> LDS column of the pre-pass (unused slots: the
                              // trash column), last entry the tile's column count
    const int32_t* chunkTile;      // [nChunks + 1] tile boundaries of the sweep's workgroups
    const int32_t* tileList;       // null, or the tiles this launch visits (PRD rates pass); chunkTile indexes it
    const int32_t* slotTr;         // [nChunks * maxCT] transition of every chunk-local accumulator slot, -1: unused
    const int32_t* postChunkTile;  // [nPostChunks + 1] tile boundaries of the post-pass workgroups
    const int32_t* postSlotTr;     // [nPostChunks * maxCTPost]
    const int32_t* postCs;         // per (wavelength slot record): accumulator slot inside its post chunk
    const DevContRec* contRec;     // per wavelength: at DevLaHeader::off + nLine, its continua in transition order
    int32_t maxCTPost;
    int32_t preCols;               // most level-sum columns a tile's rows are copied from (pre-pass LDS columns)
    const double* geo;             // [Ns][4]: |h_k - h_k+1|, its reciprocal, 1 / (|h_k-1 - h_k| + |h_k - h_k+1|), 1 / T_k
    const double* temperature;
    const double* muz;
    const double* wmu;
    const double* wavelength;
    const int32_t* lowerIdx;
    const int32_t* upperIdx;
    const double* lowerBcData;
    const double* upperBcData;
    const double* bgChi;
    const double* bgEta;
    const double* bgSca;
    double* J;                     // [Nla, Ns]
    double* Jhost;                 // the same rows of the caller's page-locked J array, or null (lwhip_map_host_J): written beside J
    int32_t depthSplit;            // the march, fused launches of deep columns: wavefronts a direction's depth points are split over (1, 2, 4)
    int32_t _padDS;
    double* I;                     // [Nla, Nmu]
    double* depthChi;
    double* depthEta;
    double* depthI;
    double* zDown;                 // ZPlaneDecomposition outputs [Nla, Nmu] (or null): I at depth Nz - 2 of the down rays,
    double* zUp;                   // at depth 1 of the up rays (SimdFullIterationTemplates.hpp:351-361)
    const double* n;               // [NlevTot, Ns]
    const double* ratio;           // [Ncont, Ns]
    const double* wphi;            // [Nline, Ns]
    const double* phiT;            // tile-blocked profiles
    const double* rho;             // PRD: [lt][Ns] rows per line (standard layout)
    double* rows;                  // row pool
    double* mom;                   // moment pool
    const DevLaHeader* laHdr;
    const DevSlot* slots;
    const DevProgram* progs;
    const DevProgRow* progRows;
    const DevProgEnt* progEnts;
    double* red8;                  // [Ntrans, SPLIT, 4, Ns]
    double* dJ;                    // [Nla]
    // ---- depth-across-lanes sweep (lwhip_lanesweep.hip): a lane owns laneD consecutive depth points of one wavelength,
    // laneLR = ceil(Ns / laneD) lanes make a ray, laneR = L rays (wavelengths) share a wavefront.  Row blocks are
    // rows[rowBase + (row * L + j) * laneLRD + k] (laneLRD = laneLR * laneD), profiles are read from the phi pool itself.
    int32_t laneD, laneLR, laneR, laneLRD;
    const int32_t* chunkOrder;     // lane sweep: workgroup -> chunk (dispatch order), or null
    const int32_t* chunkSplit;     // lane sweep: per chunk the split factor of its tiles' rays (overrides laneSplit), or null
    int32_t laneSplit, _padLS;      // S = 1, 2, 4: a tile's rays are split over S wavefronts of its workgroup (small problems)
    const double* phi;             // [line][lt][Nmu][2][Ns] pool (the caller's layout)
    const double* geoT;            // [4][laneLRD + 2], entry k + 1: |h_k - h_k+1|, its reciprocal, 1 / (|h_k-1 - h_k| + |h_k - h_k+1|), 1 / T_k
    const DevPostProg* postProg;   // [nTiles]
    uint64_t phiDoubles, rhoDoubles; // pool sizes (buffer ranges)
    // the depth arena: populations, w_phi, nStar ratios and the transposed geometry in ONE allocation (one buffer
    // resource, byte offsets)
    const double* depBase;
    uint32_t depBytes, depN, depW, depRat, depG, _padD;
    // hybrid PRD (lwhip_options.hprd), null without: rho interpolation coefficients of the PRD lines, global wavelength ->
    // ordinal among the wavelengths that scatter into the PRD region (or -1), their (row of JRest, weight) lists
    // [ordinal][mu][dir][k] in CSR form, the rest-frame mean intensity [NprdLambda][Ns]
    const lwhip_rho_coeff* hRho;
    const int32_t* hLa2h;
    const int64_t* hJOff;
    const lwhip_j_coeff* hJCoef;
    double* JRest;
    const DevLaneWg* laneWg;       // [workgroups of the launch]
    const DevLaneTile* laneTiles;  // [nTiles]
    const DevLaneRay* laneRays;    // [nTiles][L]
    const DevLaneFin* laneFin;     // [nTiles][LWHIP_POST_MAXC]
    const double* laneFinPar;      // [nTiles][LWHIP_POST_MAXC][L][3]: alpha, w_lambda / lambda 4 pi / h, 2hc / lambda^3 of (continuum, wavelength)
    const double* bcPlanck;        // [Nla][4]: Planck function of the wavelength at depths 0, 1, Ns - 2, Ns - 1 (thermalised boundaries)
    const uint8_t* laneFeedG;      // generic tiles: [gen][LWHIP_POST_MAXC][LWHIP_GEN_MAXS] which rows of slot e continuum q feeds (DevLaneFin::feed's byte)
    double* momS;                  // generic tiles: the moment scratch, [workgroup of the launch][wavefront][momA][laneD][64] (null: no generic tile)
    int32_t momA, _padM;           // arrays per wavefront: the most any generic tile needs (4 (nL - 1) + nL (nL - 1) / 2 moments + 4 nL coefficients)
    int32_t* turnLate;             // fixed-order mode: host-mapped count of turn waits that timed out (lwhip_lanesweep.hip: ls_turn_wait)
    double* detSlab;               // deterministic mode: [workgroup][maxCT][4][Ns] partial sums (no atomics), else null
    double* lsDbg;                 // diagnostic builds (-DLS_TIMING): [nTiles][8] phase clocks, host-mapped; else null
};
struct TileDyn
{
    int32_t nPass;          // 2: down then up; 1: up only
    int32_t lambdaIterate;
    int32_t prdOnly;
    // pipelined PRD sub-iterations (lwhip_redistribute_prd, round 5): this launch belongs to sub-iteration stopIter and does
    // nothing if an EARLIER sub-iteration met the tolerance (stopCtl[0] = that sub-iteration's number, 0: none yet)
    int32_t stopIter;
    const int32_t* stopCtl;
    // 1: every line's profile is the same for the two directions of an angle (a static atmosphere): the lane sweep forms
    // chi, S and the stencils once per angle for its one-line tiles (lwhip_lanesweep.hip, "pairs")
    int32_t phiSym, _padS;
};
// retile one line's profiles: standard [nlt][Nmu][2][Ns] -> the tiles' [2][Ns][64] blocks
struct RetileArgs
{
    int32_t Ns, Nmu, L, nTiles;
    const DevTile* tiles;
    const DevTileSlot* tslots;
    const DevTrans* trans;
    const double* phi;
    double* phiT;
};
size_t raymarch_lds_bytes(const TileArgs& a, int waves, bool fuse);
hipError_t launch_tile_pre(const TileArgs* devArgs, const TileArgs& hostArgs, int nTilesLaunch, const TileArgs* const* apList,
                           int nBatch, hipStream_t stream);
hipError_t launch_tile_sweep(const TileArgs* devArgs, const TileArgs& hostArgs, const TileDyn& dyn, int solver, int cap,
                             bool rates, bool fuse, int nChunks, int waves, const TileArgs* const* apList, int nBatch,
                             hipStream_t stream);
hipError_t launch_tile_post(const TileArgs* devArgs, const TileArgs& hostArgs, const TileDyn& dyn, int nPostChunks,
                            const TileArgs* const* apList, int nBatch, hipStream_t stream);
// the depth-across-lanes sweep (lwhip_lanesweep.hip); D = hostArgs.laneD
bool lane_sweep_supported(int Ns, int Nrays, int solver, int* D, int* LR, int* R);
size_t lane_sweep_lds_bytes(const TileArgs& a, int waves);
hipError_t launch_lane_sweep(const TileArgs* devArgs, const TileArgs& hostArgs, const TileDyn& dyn, int solver, bool rates,
                             int nChunks, int waves, const TileArgs* const* apList, int nBatch, hipStream_t stream);
// Diagnosis (LWHIP_DEBUG=1 LWHIP_SCRATCH_POISON=1): fills every wavefront slot's private-segment ("scratch") memory with a
// finite pattern in front of a kernel launch (lwhip_lanesweep.hip: scratch_poison_kernel).  Nothing otherwise.
void dbg_scratch_poison(hipStream_t stream);
hipError_t launch_det_reduce(const double* slab, const int32_t* off, const int32_t* ent, double* part, int Ns, int Ntrans,
                             hipStream_t stream); // -> part[Ntrans][LWHIP_DET_SEG][4][Ns]: read like red8 with nParts = LWHIP_DET_SEG
size_t det_reduce_scratch_doubles(int Ns, int Ntrans); // `part`: the partial sums of the fixed-shape reduction
hipError_t launch_retile(const RetileArgs& a, hipStream_t stream);
hipError_t launch_retile_list(const RetileArgs* devList, const RetileArgs* hostList, int n, hipStream_t stream);

// parts of a transition's sums in the stage-1 buffer (a workgroup adds to part chunk % SPLIT): contention of the sweep's atomics
// against the loads of whoever sums the parts.  Measured at the bench size (round 4): the sweep is the same with 2, 4, 8 and 16 parts
// (0.122 ms), apply_kernel takes 9.0 / 9.5 / 11.2 / 13.9 us
#ifndef LWHIP_REDUCE_SPLIT_N
#define LWHIP_REDUCE_SPLIT_N 4
#endif
enum { LWHIP_REDUCE_SPLIT = LWHIP_REDUCE_SPLIT_N };
enum { LWHIP_DET_SEG = 32 }; // fixed-order mode: segments of a transition's slab list that are summed in parallel (lwhip_lanesweep.hip)
struct ReduceArgs
{
    int32_t Ns, Ntrans;
    double* red8;                  // [Ntrans, SPLIT, 4, Ns] stage-1 sums (the sweep's workgroups add into them)
    double* red;                   // [Ntrans, 4, Ns] + gather tail [2 * world]
    const double* dJ;              // [Nla] per-wavelength dJ of this shard
    int32_t Nla, idxOffset, world, rank;
    double* batchTail;             // column batches: this column's (dJMax, idx) slot of the batch's read-back buffer
    int32_t zeroParts;             // stage 2 zeroes the parts it has summed (they were accumulated by atomics)
    int32_t nParts;                // parts per transition in red8: LWHIP_REDUCE_SPLIT, or -- fixed-order mode -- the LWHIP_DET_SEG
                                   // segment sums of the slab reduce (red8 then points at them)
};

struct ApplyArgs
{
    int32_t Ns, Natom, Nla, laStart, laEnd;
    const int32_t* atomTrOff;      // [Natom+1] global transition range of each atom
    const int32_t* atomNlevel;     // [Natom]
    const int32_t* atomDetailed;   // [Natom]
    const int64_t* atomGammaOff;   // [Natom] offset into Gamma pool
    const DevTrans* trans;
    const int32_t* transLi;        // [Ntrans] local i
    const int32_t* transLj;        // [Ntrans] local j
    const double* red;             // [Ntrans,4,Ns] (reduced)
    double* Gamma;                 // pool
    double* Rij;                   // [Ntrans, Ns]
    double* Rji;                   // [Ntrans, Ns]
    const double* Cmat;            // pool, same layout as Gamma
    double crsw;
    int32_t prefill;               // Gamma <- crsw * C before adding the radiative terms
    int32_t prdOnly;               // only Rij/Rji of PRD lines are written, Gamma untouched
    int32_t prdDetailed;           // ... including the PRD lines of detailed atoms (include_detailed_atoms); otherwise
                                   // those lines' sums are ADDED to their rates, as the reference does (PrdTemplates.hpp:30-52
                                   // zeroes only the included lines, SimdFullIterationTemplates.hpp:455-466 integrates all)
    // single-device iteration without the second reduce stage: the sums are taken from the stage-1 parts (added in
    // the same order) and one extra workgroup does the dJMax reduction
    const double* red8;            // [Ntrans, SPLIT, 4, Ns] or null
    const double* dJ;              // [NlaDj] per-wavelength dJ, or null (no dJMax workgroup)
    double* tail;                  // (dJMax, idx) slot
    double* tailHost;              // host-mapped mirror of that slot [dJMax, idx, ticket], or null
    double tailTicket;             // written last (system-scope release): the host polls for it instead of waiting for the stream
    int32_t fromParts, NlaDj, idxOffset;
    int32_t zeroParts;             // the parts are accumulated by atomics: whoever reads an element zeroes it
    int32_t scalarIdx;             // 1: dJMaxIdx as the reference's single-thread loop records it (lwhip_set_djmax_index_mode)
    int32_t gatherWorld;           // > 0 (sharded iteration, dJ == null): `tail` holds every shard's (dJMax, idx) after the
                                   // all-reduce; the extra workgroup takes their maximum and publishes it through tailHost
    // PRD sub-iteration on one device (round 5): the apply launch also reduces every PRD line's per-(depth, slice) maxima of
    // |d rho / rho| (what prd_change_kernel does as a launch of its own on shards): nPrdChange extra workgroups
    const struct PrdLineArgs* prdList;
    int32_t nPrdChange;
    int32_t nParts;                // parts per transition behind red8 (see ReduceArgs::nParts); 0: LWHIP_REDUCE_SPLIT
    // pipelined PRD sub-iterations: prdCtl = { number of the first sub-iteration whose max |d rho / rho| was below prdTol (0:
    // none yet), lines below the tolerance so far, lines reduced so far, sub-iterations run }.  The launch does nothing if
    // an earlier sub-iteration stopped the loop; its last PRD-line workgroup records whether THIS one does
    int32_t* prdCtl;
    double prdTol;
    int32_t prdIter;
    // sharded iteration with the peer exchange (lwhip_peer_attach, lwhip_kernels.hip: peer_publish_kernel): instead of `red`
    // after an all-reduce, the sums are taken from the `peerWorld` slots of THIS rank's window -- slot r holds what rank r's
    // publish launch stored there -- added in rank order (the same on every rank: sharded Gamma / rates are then bit-reproducible).
    // Every workgroup first waits until all `peerWorld` flags of the window have reached peerSeq.
    int32_t peerWorld;
    const double* peerSlots;                 // [peerWorld][peerStride] of the exchange's current buffer
    int64_t peerStride;
    const unsigned long long* peerFlags;     // [peerWorld] of the current buffer
    unsigned long long peerSeq;
};

enum { LWHIP_PEER_MAX = 16 }; // ranks of one peer exchange (one node: 8 GPUs)

// peer_publish_kernel: this rank's reduce buffer into its slot of every rank's window, then its flag there
struct PeerPublishArgs
{
    const double* red;
    int64_t n;                               // doubles to publish (sums + gather tail)
    int32_t world, rank;
    unsigned long long seq;
    int32_t* arrive;                         // workgroup arrival counter (zero between launches)
    double* slot[LWHIP_PEER_MAX];            // slot `rank` of rank r's window, current buffer
    unsigned long long* flag[LWHIP_PEER_MAX]; // flag `rank` of rank r's window, current buffer
};

// One line's profile evaluation (lwhip_voigt.hip)
struct VoigtLineArgs
{
    int32_t Ns, Nrays;
    int32_t nlt;        // rows of this context's shard
    int32_t ltStart;    // first own-grid index of the shard
    int32_t nltFull;    // full length of the line's grid
    int32_t _pad;
    double lambda0;
    const double* wave;     // [nltFull] the line's own wavelength grid
    const double* wlam;     // [nltFull] Transition::wlambda
    const double* vlosMu;   // [Nrays, Ns]
    const double* wmu;      // [Nrays]
    const double* vBroad;   // [Ns] of the line's atom
    const double* aDamp;    // [Ns]
    double* phi;            // [nlt, Nrays, 2, Ns]
    double* wphi;           // [Ns]
    double* wphiScratch;    // [wphiSlices, Ns] partial sums of the wavelength slices (null: one slice)
    int* wphiTicket;        // [point tiles] arrival counters, zero between launches
    int32_t wphiSlices, _pad2;
};
hipError_t launch_voigt_line(const VoigtLineArgs& a, hipStream_t stream);
// every line of the list in one launch pair (same Ns and slice count; hostList = what devList holds)
hipError_t launch_voigt_lines(const VoigtLineArgs* devList, const VoigtLineArgs* hostList, int nLines, hipStream_t stream);

// one piece of a gathered upload (H2DBatch, lwhip_host.h): `bytes` from offset `off` of the inbox to `dst`
struct H2DRec
{
    unsigned long long dst, off, bytes;
};
enum { H2D_CHUNK = 64 * 1024 }; // a record moves at most this much: one workgroup
// recs = inbox + recsOff, n of them (lwhip_kernels.hip)
hipError_t launch_h2d_scatter(const unsigned char* inbox, size_t recsOff, int n, hipStream_t stream);

enum { PRD_MAX_SLICES = 32 }; // workgroups per depth point in the PRD scattering integral

// One PRD line's scattering integral (lwhip_prd.hip)
struct PrdLineArgs
{
    int32_t Ns, Nl, Nblue, Nlevel;
    int32_t tr;             // global transition index of the line
    int32_t gi, gj;         // global level rows
    int32_t lj;             // local upper level
    int32_t trBegin, trEnd; // global transition range of the line's atom
    int32_t slices, _padS;  // workgroups per depth point (slices of the emission wavelengths)
    int32_t laLo, laHi;     // emission wavelengths handled here (the shard's part of the line), own-grid indices
    int32_t world, rank, q, Nprd; // where the line's (max change, index) goes in the reduce tail
    double* tail;           // [world][Nprd][2]
    double lambda0, Bij;
    const double* wave;     // [Nl] the line's own grid
    const double* J;        // [Nla, Ns]
    double* Jt;             // [Ns, Nl] scratch: J of the line, wavelength fastest
    const double* vBroad;   // [Ns]
    const double* aDamp;    // [Ns]
    const double* Qelast;   // [Ns]
    const double* Cmat;     // [Nlevel, Nlevel, Ns] of the atom
    const double* n;        // pool
    const double* Rij;      // [Ntrans, Ns]
    const double* Rji;
    const int32_t* transLi;
    const int32_t* transLj;
    double* rho;            // [laHi - laLo, Ns]: the shard's rows
    double* gII;            // [Ns, 88, Nl] cached gII weights of the line, or null (always recompute)
    int32_t* gWin;          // [Ns, PRD_MAX_SLICES, 2] first / last bracket a slice's fine grids touch (written with the cache), or null
    int32_t* gIT;           // [Ns, 88, Nl] cached bracketing indices, 16 bits each
    int32_t gIIFill, _padG; // 1: compute (and store) the weights; 0: read them from the cache
    double* changeVal;      // [Ns * slices] scratch
    int32_t* changeIdx;     // [Ns * slices] scratch
};
// every PRD line of the sub-iteration in one set of launches: devList = the device copy of hostList [nLines]
// reduceChange = false: the per-line maxima are reduced by the apply launch that follows (ApplyArgs::prdList), no prd_change launch
hipError_t launch_prd_scatter(const PrdLineArgs* devList, const PrdLineArgs* hostList, int nLines, hipStream_t stream, bool reduceChange = true,
                              const int32_t* stopCtl = nullptr, int stopIter = 0);

// nr_post_update (lwhip_pops.hip): one listed atom, and the launch arguments
struct NrAtom
{
    int32_t atom;       // index into the context's atoms
    int32_t Nlevel;
    int32_t levelOff;   // first row in the n pool
    int32_t eqOff;      // first equation (= offset into stages / nPrev rows)
    int32_t trBegin, trEnd;
    int64_t gammaOff;   // offset into the Gamma / C pools
    int64_t dcOff;      // offset (rows of Ns) into the dC scratch
};
struct NrArgs
{
    int32_t Ns, Natoms, Neqn, timeDep;
    const NrAtom* atoms;
    const double* Gamma;
    const double* Cmat;
    double* n;
    const double* nTotal;       // [Natom, Ns]
    const double* stages;       // [sum Nlevel]
    const double* nPrev;        // [sum Nlevel, Ns] (time dependent)
    const double* dC;           // per atom [Nlevel, Nlevel, Ns] back to back, or null
    const double* backgroundNe; // [Ns]
    double* ne;                 // [Ns]
    const int32_t* transType;   // [Ntrans]
    const int32_t* transLi;
    const int32_t* transLj;
    double dt, crsw;
    int32_t* status;
    int32_t k0, k1;             // depth range solved [k0, k1) (spaceStart / spaceEnd of the reference)
};
struct StatEqArgs
{
    int32_t Ns, Natoms;
    int32_t k0, k1;             // depth range solved [k0, k1)
    const NrAtom* atoms;        // the atoms to solve (atom, Nlevel, levelOff, gammaOff used)
    double* n;                  // pool
    const double* nTotal;       // [Natom, Ns]
    const double* Gamma;        // pool
    int32_t* status;            // host-mapped: LWHIP_ERR_SINGULAR is written straight into pinned memory
    double* change;             // host-mapped [Natoms][blocks][2]: max |dn/n| and its flattened index per block, or null
};
hipError_t launch_stat_eq(const StatEqArgs& a, int maxNlevel, hipStream_t stream, const StatEqArgs* list = nullptr,
                          int nBatch = 0);
int stat_eq_blocks(int Ns, int maxNlevel);
hipError_t launch_peer_publish(const PeerPublishArgs& a, hipStream_t stream);
// self-test of the exchange: fill `buf` with the rank's pattern / wait for the world's flags (bounded by spinLimit polls) and
// compare every slot with its rank's pattern; *result = 0 ok, 1 a flag never came, 2 a slot holds something else
hipError_t launch_peer_pattern(double* buf, int64_t n, int rank, hipStream_t stream);
hipError_t launch_peer_check(const double* slots, int64_t stride, const unsigned long long* flags, int world, unsigned long long seq, int64_t n,
                             long spinLimit, int32_t* result, hipStream_t stream);
hipError_t launch_fp_blocks(const double* p, size_t n, unsigned long long* sums /* [nBlocks][16] */, hipStream_t stream);
// 2D formal solver (lwhip_fs2d.hip): one workgroup per problem
struct Fs2dArgs
{
    int32_t Nx, Nz, Nrays, zLowerBc, zUpperBc;
    int32_t nSolve;        // number of solves of the launch, a multiple of nRayCycle
    int32_t nRayCycle;     // problem p solves ray rays[p % nRayCycle] at wavelength wavs[p / nRayCycle]
    float rNx;             // 1.0f / Nx (the point -> plane division of pass 1)
    double wav;            // used when wavs is null
    const double* wavs;
    const double* mux;
    const double* temperature;
    const lwhip_intersection* uw;
    const lwhip_intersection* dw;
    const int32_t* longCharIdx;
    const int32_t* substepOff;
    // the same records packed ([2][nRec] doubles: fractional part, distance; one word: flat index of the hit's lower
    // grid point | axis << 26 | (plane of the hit - plane of the point + 1) << 28) -- pass 1 is bound by the texture
    // addresser, and a 32-byte record per lane costs it three times the cache lines of these loads; null: uw / dw
    const double* uwS;
    const double* dwS;
    const int32_t* uwA;
    const int32_t* dwA;
    size_t nRec;
    // long characteristics walked by their own launch (null lcBuf: inline in pass 1): owner (ray, point) of every
    // characteristic, results [wavelength of the batch][NlongChar][Ecum, acc, chiC]
    const int32_t* lcOwner; // [NlongChar][2]
    double* lcBuf;
    int32_t NlongChar;
    int32_t lcUpOnly;       // the launch's ray list is the up-going rays (index ray >> 1) instead of every ray
    const lwhip_intersection* substeps;
    const int32_t* rays;
    const double2* cs;     // [nSolve, Nz, Nx] (chi, S) pairs: one 16-byte load per stencil point
    double* I;
    double* PsiStar;
    // the solve's two passes meet here: per point, in the ray's processing order, the affine map
    // I(k, p) = A I(k, p - 1) + c1 I(idx) + c2 I(idx + 1) + src of the point
    double* coef;          // [nSolve, 4 (A, c1, c2, src), Nz, Nx]
    int32_t* cidx;         // [nSolve, Nz, Nx]: x of the hit on the previous plane, or -(flat index) - 1
    // fixed (CALLABLE) x boundaries (FormalScalar2d.cpp:806-852): the column a ray enters through is prescribed
    int32_t periodic;      // 1: x-periodic grid (the fields below are unused)
    int32_t la0;           // wavelength (row of the boundary data) of solve 0; solve s: la0 + s / nRayCycle
    int32_t xNmuLow, xNmuUp;
    const double* xbcLow;  // [Nla, xNmuLow, Nz] (mux > 0: column 0)
    const double* xbcUp;   // [Nla, xNmuUp, Nz]  (mux < 0: column Nx - 1)
    const int32_t* xIdxLow; // [Nrays, 2] -> row of the boundary data
    const int32_t* xIdxUp;
    // CALLABLE z boundaries (:930-938): I(kStart, j) = bcData(la, idxs(mu, toObs), j), data [Nla, Nmu, Nx]
    int32_t zNmuLow, zNmuUp;
    const double* zbcLow;
    const double* zbcUp;
    const int32_t* zIdxLow;
    const int32_t* zIdxUp;
};
hipError_t launch_fs2d(const Fs2dArgs& a, int nSolve, hipStream_t stream);
// packed copy of the n = 2 Nrays Nx Nz intersection records (layout: Fs2dArgs::uwS / uwA); false if a record does not
// fit the packing (a hit further than one plane from its point), in which case pass 1 reads the records themselves
bool fs2d_records_packed(const lwhip_intersection* rec, size_t n, int Nx, int Nz, std::vector<double>& s, std::vector<int32_t>& w);
bool fs2d_long_chars_ok(const lwhip_grid2d* g);

// the 2D iteration's pointwise kernels (lwhip_2d.hip)
struct Batch2dArgs
{
    int32_t Ns, Nx, Nrays, nRaysActive, la0, nLa, lambdaIterate, storeDepth, NlevTot, Natom;
    const int32_t* rayList;        // [nRaysActive] -> 2 * mu + toObs
    const double* muz;
    const double* wmu;
    const DevLaHeader* laHdr;
    const DevSlot* slots;
    const double* rows;            // continuum rows of the batch, already offset: rows + rowBase * Ns is valid
    const double* bgChi;
    const double* bgEta;
    const double* bgSca;
    const double* n;
    const double* phi;
    const double* rho;
    const double* wphi;
    double* J;
    double* Iout;                  // [Nla, Nrays, Nx]
    double* depthChi;
    double* depthEta;
    double* depthI;
    double* zDown;                 // ZPlaneDecomposition outputs [Nla, Nrays, Nx] (or null), :372-384
    double* zUp;
    double2* cs;                   // [nLa, nRaysActive, Ns] batch buffer of (chi, S) pairs
    const double* I;
    const double* PsiStar;
    double* red;                   // [nGroups][Ntrans, 4, Ns] running sums of the iteration, one copy per group
    size_t redStride;              // doubles between the copies
    int32_t nGroups;               // wavelength groups of a batch that accumulate concurrently
    int32_t maxRowsLa;             // most continuum rows (program rows + one per continuum) any wavelength has
    const int32_t* slotTr;         // global transition of every slot record
    unsigned long long* dJbits;    // [Nla] max_k |1 - Jdag/J| as the bits of a non-negative double
};
hipError_t launch_gather2d(const Batch2dArgs& a, int maxL, hipStream_t stream);
hipError_t launch_rates2d(const Batch2dArgs& a, int maxL, int maxM, int maxP, hipStream_t stream);
hipError_t launch_iout2d(const Batch2dArgs& a, hipStream_t stream);
hipError_t launch_sum_groups(double* red, size_t n, size_t stride, int G, hipStream_t stream);
hipError_t launch_cont_range(const ContArgs* devArgs, const ContArgs& hostArgs, int contOff, int nCont, double* rowsBase,
                             hipStream_t stream);

// Ng acceleration of the populations (lwhip_pops.hip)
struct NgAtom
{
    int32_t len;        // Nlevel * Ns
    int32_t _pad;
    int64_t nOff;       // offset of the atom's populations in the n pool
    int64_t histOff;    // offset of its history [Norder + 2][len]
};
enum { LWHIP_NG_MAX_ORDER = 6 };
struct NgArgs
{
    const NgAtom* atoms;
    double* n;
    double* history;
    double* change;         // host-mapped [nAtoms][2]
    int32_t* status;
    int32_t Norder, storeSlot, oldSlot, doAccel, haveTwo;
    int32_t slots[LWHIP_NG_MAX_ORDER + 2]; // slots of count-1, count-2, ... count-Norder-2
};
hipError_t launch_ng(const NgArgs& a, int nAtoms, hipStream_t stream);
hipError_t launch_nr_post(const NrArgs& a, hipStream_t stream);
hipError_t launch_time_dep(int Nlevel, int Ns, int k0, int k1, double* n, const double* nOld, const double* Gamma, double dt,
                           int* status, hipStream_t stream);

// launch helpers implemented in lwhip_kernels.hip
// list / nBatch: column batches -- a device array of per-column argument blocks, one grid slice per column
hipError_t launch_reduce_sum(const ReduceArgs& a, hipStream_t stream, const ReduceArgs* list = nullptr, int nBatch = 0);
hipError_t launch_apply(const ApplyArgs& a, hipStream_t stream, const ApplyArgs* list = nullptr, int nBatch = 0);
hipError_t launch_prefill(double* Gamma, const double* C, double crsw, size_t n, hipStream_t stream);
// diagnosis (LWHIP_CHECK_ZERO=1): counts the non-zero words of p[0 .. n) into the host-mapped *counter
hipError_t launch_check_zero(const double* p, size_t n, int* counter, hipStream_t stream);
}

/*
 * lwhip.h -- C ABI of the MI355X (gfx950) implementation of Lightweaver's
 * `formal_sol_gamma_matrices` iteration (1D plane-parallel, unpolarised).
 *
 * This is the drop-in boundary.  Everything above it (the Lightweaver plugin
 * `fs_iteration_fns_provider`, the Python ctypes mirror) talks to the HIP
 * kernels only through the functions declared here: plain pointers and sizes,
 * no C++ types, no torch types.
 *
 * Each entry point names the reference interface it replaces (paths relative
 * to the Lightweaver source tree):
 *
 *   lwhip_formal_sol_gamma_matrices  <- FsIterationFns::fs_iter
 *        (Source/LwFormalInterface.hpp:86,118; trampoline Source/FormalScalar.cpp:678-681;
 *         scalar body Source/SimdFullIterationTemplates.hpp:588-719)
 *   lwhip_formal_sol                 <- FsIterationFns::simple_fs
 *        (Source/LwFormalInterface.hpp:87,119; Source/SimdFullIterationTemplates.hpp:721-781)
 *   lwhip_stat_equil                 <- FsIterationFns::stat_eq
 *        (Source/LwFormalInterface.hpp:91,122; Source/UpdatePopulations.cpp:7-47; Source/LuSolve.cpp:8-132)
 *   lwhip_compute_profiles           <- Transition::compute_phi / compute_wphi
 *        (Source/FormalScalar.cpp:28-68,106-134)
 *   lwhip_redistribute_prd           <- FsIterationFns::redistribute_prd
 *        (Source/LwFormalInterface.hpp:90,121; Source/PrdTemplates.hpp:175-351; Source/Prd.cpp:9-30,51-124,180-263,468-645)
 *   lwhip_create / lwhip_destroy     <- FsIterationFns::alloc_global_scratch / free_global_scratch
 *        (Source/LwFormalInterface.hpp:106-107,131-132; called from Source/ThreadStorage.cpp:480-493,538-566)
 *   lwhip_upload / lwhip_download    <- (none: the reference shares host memory; these move the
 *        borrowed numpy buffers of Source/LwMiddleLayer.pyx to/from HBM)
 *
 * The problem descriptor is a flat restatement of what `Context&` reaches
 * (Source/LwContext.hpp:20-46): Atmosphere (Source/LwAtmosphere.hpp:179-221), Spectrum and
 * Background (Source/LwMisc.hpp:85-110), Atom (Source/LwAtom.hpp:41-80) and Transition
 * (Source/LwTransition.hpp:21-69).  All arrays are C-contiguous fp64, depth `k` fastest,
 * exactly as the reference lays them out.
 *
 * All host pointers in the descriptor are BORROWED: the library reads them in
 * lwhip_create/lwhip_upload and writes them in lwhip_download, never otherwise.
 */
#ifndef LWHIP_H
#define LWHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LWHIP_ABI_VERSION 4

/* ---- enums (values mirror the reference where one exists) ---------------- */

/* TransitionType, Source/LwTransition.hpp:10-14 */
enum { LWHIP_LINE = 0, LWHIP_CONTINUUM = 1 };

/* RadiationBc, Source/LwAtmosphere.hpp:6-13 */
enum {
    LWHIP_BC_UNINITIALISED = 0,
    LWHIP_BC_ZERO = 1,
    LWHIP_BC_THERMALISED = 2,
    LWHIP_BC_PERIODIC = 3,
    LWHIP_BC_CALLABLE = 4
};

/* 1D formal solvers registered in Source/FormalInterface.cpp:30-32 */
enum {
    LWHIP_FS_LINEAR_1D = 0,  /* "piecewise_linear_1d"  */
    LWHIP_FS_BESSER_1D = 1,  /* "piecewise_besser_1d"  */
    LWHIP_FS_BEZIER3_1D = 2  /* "piecewise_bezier3_1d" (reference default, lightweaver/config.py:12) */
};

/* status codes */
enum {
    LWHIP_OK = 0,
    LWHIP_ERR_INVALID = 1,      /* bad descriptor / argument                       */
    LWHIP_ERR_UNSUPPORTED = 2,  /* valid in the reference, not built here (stated)   */
    LWHIP_ERR_DEVICE = 3,       /* HIP runtime failure or no gfx950 device          */
    LWHIP_ERR_SINGULAR = 4,     /* "Singular Matrix" of Source/LuSolve.cpp:22-23       */
    LWHIP_ERR_BUSY = 5          /* lwhip_destroy of a context whose tables are still borrowed (lwhip_create_like) */
};

/* array groups for lwhip_upload / lwhip_download (bit mask) */
enum {
    LWHIP_ATMOS      = 1 << 0,  /* height, temperature, muz, wmu, vlosMu, wavelength   (up)     */
    LWHIP_BACKGROUND = 1 << 1,  /* background chi, eta, sca                             (up)     */
    LWHIP_PROFILES   = 1 << 2,  /* phi, wphi of every line                              (up/down)*/
    LWHIP_POPS       = 1 << 3,  /* n of every atom                                      (up/down)*/
    LWHIP_NSTAR      = 1 << 4,  /* nStar, nTotal, vBroad of every atom, aDamp of lines   (up)     */
    LWHIP_J          = 1 << 5,  /* spect.J                                              (up/down)*/
    LWHIP_GAMMA      = 1 << 6,  /* Gamma of every active atom (pre-fill in, result out) (up/down)*/
    LWHIP_BC         = 1 << 7,  /* CALLABLE boundary data                               (up)     */
    LWHIP_RHOPRD     = 1 << 8,  /* rhoPrd of PRD lines                                  (up)     */
    LWHIP_I          = 1 << 9,  /* spect.I                                              (down)   */
    LWHIP_RATES      = 1 << 10, /* Rij, Rji of every transition                         (down)   */
    LWHIP_DEPTHDATA  = 1 << 11, /* depthData chi, eta, I (only when requested at create)(down)   */
    LWHIP_COLLISIONS = 1 << 12, /* C of every active atom                               (up)     */
    LWHIP_ALL_INPUTS = (1 << 0) | (1 << 1) | (1 << 2) | (1 << 3) | (1 << 4) | (1 << 5) | (1 << 6) |
                       (1 << 7) | (1 << 8) | (1 << 12),
    LWHIP_ALL_OUTPUTS = (1 << 5) | (1 << 6) | (1 << 9) | (1 << 10)
};

/* ---- descriptor ----------------------------------------------------------- */

/* One radiative transition; mirrors Transition (Source/LwTransition.hpp:21-69). */
typedef struct lwhip_transition {
    int32_t type;          /* LWHIP_LINE | LWHIP_CONTINUUM                                         */
    int32_t i, j;          /* lower, upper level                                                   */
    int32_t Nblue, Nred;   /* active for Nblue <= la < Nred of the global grid (is_active, :88-91)   */
    int32_t prd;           /* non-zero: rhoPrd is valid and multiplies gij (LwAtom.hpp:121-123)    */
    double Aji, Bji, Bij;  /* Einstein coefficients (lines)                                        */
    double lambda0;        /* [nm]                                                                 */
    double dopplerWidth;   /* c/lambda0 for lines, 1.0 for continua (LwMiddleLayer.pyx:1799,1815)  */
    const double* wavelength; /* [Nred-Nblue] = global wavelength[Nblue:Nred]                      */
    const double* alpha;   /* [Nred-Nblue] cross-section, continua only                            */
    double* phi;           /* [Nred-Nblue, Nrays, 2, Nspace] lines only; dir 0 = down, 1 = toObs   */
    double* wphi;          /* [Nspace] lines only                                                  */
    const double* aDamp;   /* [Nspace] lines only; needed by lwhip_compute_profiles                */
    double* rhoPrd;        /* [Nred-Nblue, Nspace] or NULL                                         */
    double* Rij;           /* [Nspace] out                                                         */
    double* Rji;           /* [Nspace] out                                                         */
    const double* Qelast;  /* [Nspace] elastic collision rate, PRD lines only (Prd.cpp:18)          */
} lwhip_transition;

/* One atom; mirrors Atom (Source/LwAtom.hpp:41-80). */
typedef struct lwhip_atom {
    int32_t Nlevel;
    int32_t Ntrans;
    int32_t detailed;      /* non-zero: member of ctx.detailedAtoms (rates only, no Gamma)         */
    int32_t _pad;
    double* n;             /* [Nlevel, Nspace] in; out of lwhip_stat_equil                         */
    const double* nStar;   /* [Nlevel, Nspace]                                                     */
    const double* nTotal;  /* [Nspace]                                                             */
    const double* vBroad;  /* [Nspace]                                                             */
    double* Gamma;         /* [Nlevel, Nlevel, Nspace] index (to, from, k); NULL if detailed       */
    const double* C;       /* [Nlevel, Nlevel, Nspace] collisional rates; may be NULL              */
    lwhip_transition* trans; /* [Ntrans]                                                           */
} lwhip_atom;

/* One z boundary; mirrors AtmosphericBoundaryCondition (Source/LwAtmosphere.hpp:17-43). */
typedef struct lwhip_boundary {
    int32_t type;          /* LWHIP_BC_*                                                           */
    int32_t Nmu;           /* second extent of bcData                                              */
    const int32_t* idxs;   /* [Nrays, 2] -> row of bcData, or -1 (CALLABLE only)                   */
    const double* bcData;  /* [Nlambda, Nmu] (third reference extent is 1 in 1D; [Nlambda, Nmu, Nz] for the x
                            * boundaries of a 2D grid) (CALLABLE only)                              */
} lwhip_boundary;

struct lwhip_grid2d; /* 2D geometry, defined below */

typedef struct lwhip_problem {
    int32_t abiVersion;    /* LWHIP_ABI_VERSION                                                    */
    int32_t Nspace;        /* depth points                                                         */
    int32_t Nrays;         /* mu quadrature points                                                 */
    int32_t Nlambda;       /* global wavelength grid                                               */
    int32_t Natom;         /* active atoms first, then detailed atoms                              */
    int32_t formalSolver;  /* LWHIP_FS_*                                                           */
    int32_t storeDepthData;/* non-zero: fill depthChi/depthEta/depthI (DepthData.fill)             */
    int32_t _pad;
    const double* height;      /* [Nspace] m, decreasing index 0 = top                              */
    const double* temperature; /* [Nspace] K                                                        */
    const double* vlosMu;      /* [Nrays, Nspace]; needed by lwhip_compute_profiles only            */
    const double* muz;         /* [Nrays]                                                           */
    const double* wmu;         /* [Nrays]                                                           */
    const double* wavelength;  /* [Nlambda] nm                                                      */
    lwhip_boundary zLowerBc;   /* bottom of the atmosphere (k = Nspace-1), feeds up-going rays      */
    lwhip_boundary zUpperBc;   /* top (k = 0), feeds down-going rays                                */
    const double* bgChi;       /* [Nlambda, Nspace]                                                 */
    const double* bgEta;       /* [Nlambda, Nspace]                                                 */
    const double* bgSca;       /* [Nlambda, Nspace]                                                 */
    double* J;                 /* [Nlambda, Nspace] in (J-dagger) / out                             */
    double* I;                 /* [Nlambda, Nrays] out, emergent intensity at k = 0                 */
    double* depthChi;          /* [Nlambda, Nrays, 2, Nspace] or NULL                               */
    double* depthEta;          /* [Nlambda, Nrays, 2, Nspace] or NULL                               */
    double* depthI;            /* [Nlambda, Nrays, 2, Nspace] or NULL                               */
    lwhip_atom* atoms;         /* [Natom]                                                           */
    /* 2D (tier 2): non-NULL selects the x-periodic 2D geometry.  Then Nspace = Nz * Nx (index k * Nx + j),
     * the formal solver is piecewise_besser_2d with interp_linear_2d whatever formalSolver says, `height` is
     * unused, the z boundaries come from the grid, and I is [Nlambda, Nrays, Nx] (the top row of every
     * column, Spectrum::I with Noutgoing = Nx).  Not supported on the device yet (LWHIP_ERR_UNSUPPORTED). */
    const struct lwhip_grid2d* grid2d;
} lwhip_problem;

/* Hybrid PRD (Leenaarts et al. 2012): what configure_hprd_coeffs (Source/Prd.cpp:697-946) leaves in Spectrum
 * (prdIdxs, hPrdIdxs, JRest, JCoeffs; Source/LwMisc.hpp:92-104) and in every PRD line (hPrdCoeffs,
 * Source/LwTransition.hpp:65), flattened.  With it a context
 *   - multiplies V_ji of a PRD line by rho interpolated to the ray's rest wavelength,
 *     rho = (1 - frac) rhoPrd(i0, k) + frac rhoPrd(i1, k)   (Transition::uv, Source/LwTransition.hpp:116-127)
 *     instead of by rhoPrd(lt, k) (Source/LwAtom.hpp:118-123),
 *   - scatters every ray's intensity into the rest-frame mean intensity,
 *     JRest(idx, k) += 0.5 wmu frac I(k)   (Source/SimdFullIterationTemplates.hpp:397-408),
 *   - takes J of the scattering integral from JRest (Source/Prd.cpp:384-389, 484-490),
 *   - visits hPrdIdxs in the PRD rates pass (Source/PrdTemplates.hpp:234-248).
 * All arrays are borrowed for the life of the context; JRest is written by lwhip_download(LWHIP_J). */
typedef struct lwhip_rho_coeff {   /* Prd::RhoInterpCoeffs, Source/LwMisc.hpp:50-57 */
    int32_t i0, i1;
    double frac;
} lwhip_rho_coeff;
typedef struct lwhip_j_coeff {     /* Prd::JInterpCoeffs, Source/LwMisc.hpp:58-64 */
    double frac;
    int32_t idx;                   /* row of JRest */
    int32_t _pad;
} lwhip_j_coeff;
typedef struct lwhip_hprd {
    int32_t NprdLambda;            /* spect.prdIdxs.size(): wavelengths where a PRD line is active = rows of JRest  */
    int32_t NhPrd;                 /* spect.hPrdIdxs.size(): wavelengths that scatter into them                     */
    int32_t Nlines;                /* PRD lines that carry hPrdCoeffs                                                */
    int32_t _pad;
    const int32_t* prdIdxs;        /* [NprdLambda] global wavelength indices, increasing                             */
    const int32_t* hPrdIdxs;       /* [NhPrd]                                                                        */
    double* JRest;                 /* [NprdLambda, Nspace] out                                                       */
    const int64_t* jCoeffOff;      /* [NhPrd * Nrays * 2 * Nspace + 1]: where JCoeffs(hPrdLa, mu, toObs, k) starts   */
    const lwhip_j_coeff* jCoeffs;  /* the entries, back to back                                                      */
    const int32_t* lineAtom;       /* [Nlines] index into prob->atoms                                                */
    const int32_t* lineTrans;      /* [Nlines] index into that atom's trans                                          */
    const lwhip_rho_coeff* const* rhoCoeffs; /* [Nlines] -> [Nred - Nblue, Nrays, 2, Nspace]                         */
} lwhip_hprd;

#define LWHIP_OPT_PRD_DETAILED (1 << 16)
/* Gamma and the rates summed in a fixed order (bit-reproducible run to run, as the reference's single-thread path and
 * its fixed-order thread reduction are, Source/ThreadStorage.cpp:343-396): per-workgroup partial slabs added in
 * workgroup order instead of fp64 atomics.  Served by the depth-across-lanes sweep with one wavefront per workgroup
 * (contexts it does not cover -- other solvers than Bezier3, column batches -- ignore the flag); J and I are
 * reproducible either way.  Costs a factor 2-4 in the sweep (one wavefront per workgroup): off by
 * default, also in the plugin (LWHIP_DETERMINISTIC=1 turns it on there and in any context). */
#define LWHIP_OPT_DETERMINISTIC (1 << 17)

typedef struct lwhip_options {
    int32_t device;        /* HIP device ordinal                                                   */
    int32_t laStart;       /* wavelength shard owned by this context: [laStart, laEnd)             */
    int32_t laEnd;         /* 0,0 = whole grid                                                     */
    int32_t flags;         /* bits 0-15: column-batch hint = number of contexts that will share the device
                            * through lwhip_batch_* (0: none; sizes the per-context chunking);
                            * bit 16 (LWHIP_OPT_PRD_DETAILED): the PRD calls include the PRD lines of
                            * detailed-static atoms -- ExtraParams "include_detailed_atoms" of
                            * redistribute_prd_lines (Source/PrdTemplates.hpp:25-29, 190-215), the default of
                            * LwContext.prd_redistribute (Source/LwMiddleLayer.pyx:3678-3680); rest reserved */
    int32_t worldSize;     /* number of wavelength shards of the job (0 or 1 = unsharded)          */
    int32_t worldRank;     /* this shard's ordinal: selects its (dJMax, idx) slot in the buffer    */
    void* stream;          /* hipStream_t to launch on, NULL = the library's own stream            */
    const lwhip_hprd* hprd;/* hybrid PRD tables (configure_hprd_coeffs was called on the Context), or NULL */
} lwhip_options;

typedef struct lwhip_context lwhip_context; /* opaque */

/* Result of one iteration; the members of IterationResult (Source/LwIterationResult.hpp:6-28)
 * that fs_iter fills (Source/SimdFullIterationTemplates.hpp:633-637). */
typedef struct lwhip_iter_result {
    int32_t updatedJ;
    int32_t dJMaxIdx;      /* wavelength index of the first maximum of dJ(la)                      */
    double dJMax;          /* max_la max_k |1 - Jdag/J|                                            */
} lwhip_iter_result;

/* ---- entry points ----------------------------------------------------------- */

/* Text of the last error raised on the calling thread ("" if none). */
const char* lwhip_last_error(void);

/* ---- the caller's J array as a direct output of the sweep (drop-in mode) -------------------------------------------------
 * lwhip_map_host_J(ctx, 1) page-locks the problem's J array ([Nlambda, Nspace], hipHostRegister) and from then on the sweep
 * stores every J it forms into the device's copy AND straight into that array (stores over PCIe, spread over the kernel's
 * duration): `lwhip_download(LWHIP_J)` then moves nothing -- the 6.7 MB copy of J at 10 240 wavelengths was the largest part
 * of a host-authoritative call.  The array holds the new J once the stream has been waited for (every download does).
 * 1D problems on the depth-across-lanes sweep, unsharded; LWHIP_ERR_UNSUPPORTED otherwise or when the registration fails (the
 * caller goes on with copies).  lwhip_map_host_J(ctx, 0) -- and lwhip_destroy -- release the registration; call it before
 * the array is freed or replaced.  Contract served: `spect.J` overwritten for all wavelengths when fs_iter returns
 * (Source/LwMiddleLayer.pyx:3198-3207, Source/SimdFullIterationTemplates.hpp:181-190). */
int lwhip_map_host_J(lwhip_context* ctx, int enable);

/* Exact change detection without reading J back: the 64-bit fingerprint lwhip_host_fingerprint(p, n) would give for the
 * DEVICE's current J if it were copied into an array at address `p` (n = Nla * Nspace doubles of the shard), computed on the
 * device (one pass over J in HBM, a few microseconds) -- so a host-authoritative caller can tell at its next call, from a
 * fingerprint of its own array, whether anybody rewrote `spect.J` in between, and skip the upload of J when nobody did.
 * The fingerprint is defined here: the array in blocks of LWHIP_FP_BLOCK doubles, per block a Fletcher-style pair of running
 * sums over 8 interleaved 64-bit lanes (lwhip_fp_block_sums), folded with lwhip_fp_mix in block order.  Any change of one
 * element changes it. */
int lwhip_fingerprint_J(lwhip_context* ctx, const void* p, uint64_t* out);
/* A copy of the device's J kept on the device / put back (stream-ordered, a few microseconds): with a mapped J array the sweep
 * overwrites the caller's copy of the J an iteration starts from, so a caller that may have to run the iteration a second time
 * on corrected inputs (the plugin's speculative run, lwhip_plugin.cpp fs_iter_hip) takes the starting J back from here. */
int lwhip_j_snapshot(lwhip_context* ctx);
int lwhip_j_restore(lwhip_context* ctx);
uint64_t lwhip_host_fingerprint(const double* p, size_t n);

enum { LWHIP_FP_BLOCK = 32768 };
static inline uint64_t lwhip_fp_mix(uint64_t h, uint64_t bits)
{
    h = (h ^ bits) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}
/* one block's fingerprint from its 16 sums (s1[8], s2[8]), its length and its offset in the array */
static inline uint64_t lwhip_fp_block_fold(const uint64_t* sums, size_t n, uint64_t offset)
{
    uint64_t h = lwhip_fp_mix(1469598103934665603ull ^ offset, (uint64_t)n);
    for (int l = 0; l < 8; ++l)
        h = lwhip_fp_mix(lwhip_fp_mix(h, sums[l]), sums[8 + l]);
    return h;
}
/* the array's fingerprint from its blocks' fingerprints */
static inline uint64_t lwhip_fp_array_fold(const uint64_t* blockHashes, size_t nBlocks, const void* p, size_t n)
{
    uint64_t h = lwhip_fp_mix((uint64_t)(uintptr_t)p, (uint64_t)n);
    for (size_t b = 0; b < nBlocks; ++b)
        h = lwhip_fp_mix(h, blockHashes[b]);
    return h;
}

/* Diagnosis (LWHIP_DEBUG set in the environment): the pinned host blocks of destroyed contexts are kept and filled with a
 * pattern instead of being given back to the runtime; this returns the number of their bytes that no longer hold it (0 = nothing
 * stored into host memory of a context that was gone) and prints the first few to stderr.  Without LWHIP_DEBUG there is no
 * quarantine and the answer is 0.  Also run when the process ends.  No counterpart in the reference. */
long lwhip_debug_check_quarantine(void);

/* Streams the library has created with hipStreamCreate in this process so far.  Contexts' own streams are POOLED per device
 * and never destroyed (a context that dies gives its stream to the next one): the HIP runtime's completion callback may store
 * into a stream's queue object after hipStreamDestroy has freed it (profiles/r06_stray_write.md), i.e. into whatever host
 * allocation reuses that block.  Making and closing contexts one after the other leaves this count at the number that were
 * alive at the same time (tests/test_abi.py).  No counterpart in the reference. */
long lwhip_debug_streams_created(void);

/* ABI version the library was built with. */
int lwhip_abi_version(void);

/* Number of visible gfx950 devices (0 if none / no driver). Never fails. */
int lwhip_device_count(void);

/* Allocate device state for `prob`, build the per-wavelength activity tables and upload
 * LWHIP_ALL_INPUTS.  Replaces alloc_global_scratch.  `opts` may be NULL. */
int lwhip_create(const lwhip_problem* prob, const lwhip_options* opts, lwhip_context** out);

/* The same for a problem of the SAME STRUCTURE as the one `like` was made for -- sizes, solver, wavelength shard and grid, and
 * of every transition its levels, wavelength range, coefficients, own grid and cross-sections; populations, atmosphere,
 * profiles, boundary data are free to differ: the columns of a 1.5D batch, lwhip_batch_create.  The new context borrows
 * `like`'s structure tables (per-wavelength activity tables, tile / workgroup / ray records: a few MB and half of the
 * creation time per context) instead of building device copies of its own.  The structures are compared by a fingerprint:
 * LWHIP_ERR_INVALID if they differ.  `like` cannot be destroyed before its borrowers (lwhip_destroy fails with
 * LWHIP_ERR_BUSY and leaves it intact).  No counterpart in the reference: every Context owns its tables
 * (update_deps, Source/LwMiddleLayer.pyx:3244-3288). */
int lwhip_create_like(const lwhip_problem* prob, const lwhip_options* opts, lwhip_context* like, lwhip_context** out);

/* Free everything lwhip_create allocated.  Replaces free_global_scratch. */
int lwhip_destroy(lwhip_context* ctx);

/* Copy the host arrays named by `mask` (LWHIP_* bits) from the descriptor given at create
 * into HBM / back out of it.  Synchronous with respect to the host buffers. */
int lwhip_upload(lwhip_context* ctx, uint32_t mask);
int lwhip_download(lwhip_context* ctx, uint32_t mask);

/* Gamma <- crsw * C on the device: the pre-fill LwContext.formal_sol_gamma_matrices performs on
 * the host before calling the core (Source/LwMiddleLayer.pyx:3198-3203). */
int lwhip_gamma_prefill_from_C(lwhip_context* ctx, double crsw);

/* lwhip_gamma_prefill_from_C(crsw) followed by lwhip_formal_sol_gamma_matrices in one call: the device-resident iteration of
 * LwContext.formal_sol_gamma_matrices (Source/LwMiddleLayer.pyx:3152-3210: Gamma <- crsw C, then the scheme's fs_iter). */
int lwhip_iterate_from_C(lwhip_context* ctx, int lambdaIterate, double crsw, lwhip_iter_result* res);

/* One formal_sol_gamma_matrices iteration over the context's wavelength shard:
 * J, I overwritten, Rij/Rji fresh integrals, Gamma += radiative terms, diagonal finalised.
 * Pre-condition (as in the reference): Gamma holds the collisional pre-fill.
 * `lambdaIterate` != 0 sets PsiStar = 0 (FsMode::PureLambdaIteration). */
int lwhip_formal_sol_gamma_matrices(lwhip_context* ctx, int lambdaIterate, lwhip_iter_result* res);

/* The same iteration split around the cross-GPU reduction (wavelength-sharded runs):
 *   lwhip_fs_partial   launches the sweep; leaves this shard's Gamma/R partial sums in the first
 *                      `nSum` doubles of the reduce buffer and its (dJMax, idx) in slot `worldRank`
 *                      of the `nGather` = 2*worldSize trailing doubles (all other slots zero);
 *   (caller all-reduces -- SUM -- the whole buffer of nSum + nGather doubles: one collective)
 *   lwhip_fs_finalise  adds the reduced sums into Gamma (which keeps its pre-fill, cf.
 *                      Source/ThreadStorage.cpp:155-156), finalises the diagonal, writes Rij/Rji. */
int lwhip_fs_partial(lwhip_context* ctx, int lambdaIterate);
int lwhip_fs_finalise(lwhip_context* ctx, lwhip_iter_result* res);
/* Device pointer and layout of the reduce buffer: [0,nSum) partial sums, [nSum,nSum+nGather) the
 * per-rank (dJMax, idx) slots. */
int lwhip_reduce_buffer(lwhip_context* ctx, void** devPtr, size_t* nSum, size_t* nGather);

/* The same exchange WITHOUT a collective call, for the ranks of one node (one process per GPU, or several contexts of one
 * process): every rank owns a window in device memory; lwhip_fs_partial then stores the rank's partial sums into its slot of
 * EVERY rank's window (peer-mapped stores: xGMI between GPUs) and raises its flag there, and lwhip_fs_finalise's launch waits for
 * the worldSize flags of its own window and adds the slots IN RANK ORDER -- so sharded Gamma / rates are the same bits on every
 * rank and from run to run (with the fixed-order mode for the sums inside a shard), which the summation tree of a library
 * all-reduce does not promise.  Replaces AtomStorageFactory::accumulate_Gamma / TransitionStorageFactory::accumulate_rates
 * across threads (Source/ThreadStorage.cpp:150-166, 334-396; called from Source/SimdFullIterationTemplates.hpp:675-703).
 *   lwhip_peer_window           allocates (once) and returns this rank's window;
 *   lwhip_peer_export           the window's inter-process handle (hipIpcMemHandle_t as 64 opaque bytes) -- the caller ships it
 *                               to the other ranks (e.g. torch.distributed.all_gather_object) ...
 *   lwhip_peer_attach           ... and hands in all worldSize handles, own rank's entry ignored: from now on fs_partial /
 *                               fs_finalise exchange through the windows and the caller must NOT all-reduce the reduce buffer;
 *   lwhip_peer_attach_pointers  the same for windows the process can address directly (contexts of ONE process: pointers
 *                               from lwhip_peer_window, peer access enabled by the caller where they live on other devices);
 *   lwhip_peer_detach           waits for the stream and goes back to the all-reduce contract.
 * All ranks must make the same sequence of fs_partial / fs_finalise calls while attached.  A rank whose peers never publish
 * (a dead process) does not hang the device: the launch gives up after ~2 s and lwhip_fs_finalise reports LWHIP_ERR_DEVICE.
 * The PRD calls (lwhip_prd_pack / _partial / _finalise) keep the all-reduce contract. */
int lwhip_peer_window(lwhip_context* ctx, void** devPtr, size_t* bytes);
int lwhip_peer_export(lwhip_context* ctx, void* handle64);
int lwhip_peer_attach(lwhip_context* ctx, const void* handles /* worldSize x 64 bytes */);
int lwhip_peer_attach_pointers(lwhip_context* ctx, void* const* windows /* worldSize device pointers */);
int lwhip_peer_detach(lwhip_context* ctx);
/* One exchange of a known per-rank pattern instead of the partial sums -- called by ALL attached ranks together, before the
 * exchange is relied on: *result = 0 every rank's slot arrived intact within timeoutMs, 1 a rank's flag never came, 2 a slot
 * holds something else (stores of a peer not visible to this device's kernels).  What a caller does on a non-zero result is
 * detach everywhere and fall back to the all-reduce (lightweaver_amd.distributed.ShardedIteration does).  The windows are
 * uncached / fine-grained device memory where the runtime offers it, so that a peer's stores are seen by a running kernel. */
int lwhip_peer_selftest(lwhip_context* ctx, int timeoutMs, int32_t* result);

/* formal_sol: chi/S/solve/I only, optionally up-going rays only (FsMode::FsOnly|UpOnly). */
int lwhip_formal_sol(lwhip_context* ctx, int upOnly);

/* Statistical equilibrium for atom `atom` (index into prob->atoms; -1 = every active atom):
 * per depth, eliminate the row of the largest population, solve by Crout LU with implicit
 * scaling, partial pivoting and one refinement pass; n overwritten on the device.
 * Returns LWHIP_ERR_SINGULAR where the reference throws "Singular Matrix". */
int lwhip_stat_equil(lwhip_context* ctx, int atom);
/* The same without waiting: the solve is queued on the context's stream and a singular matrix is
 * remembered; lwhip_check_status waits for the stream and returns (and clears) that condition.
 * For batches of contexts whose solves should overlap instead of serialising on a host wait each. */
int lwhip_stat_equil_async(lwhip_context* ctx, int atom);
/* lwhip_stat_equil that also reports, per active atom (arrays indexed by position among the active
 * atoms; entries of atoms not solved are left alone), the maximum relative population change
 * max |(n_new - n_old) / n_new| and the flattened [level, depth] index of its first occurrence: what
 * LwContext.stat_equil returns as dPops / dPopsMaxIdx (Ng::max_change, Source/Ng.hpp:138-156, with the
 * default Ng(0,0,0), i.e. no acceleration). */
int lwhip_stat_equil_report(lwhip_context* ctx, int atom, double* dPops, int32_t* dPopsMaxIdx);
int lwhip_check_status(lwhip_context* ctx);

/* PRD sub-iterations (redistribute_prd_lines): for every PRD line of an active atom, the
 * total depopulation + elastic rate, the angle-averaged scattering integral with Gouttebroze's
 * GII on a 0.15-Doppler-width fine grid (rho <- 1 + gamma (int J gII / int gII - Jbar)), then a
 * formal solution restricted to the PRD wavelengths that updates J and the PRD lines' Rij/Rji;
 * at most maxIter times, until max |d rho / rho| < tol.  rhoPrd, J, Rij/Rji stay on the device.
 * `dRho`/`dRhoMaxIdx` receive one entry per (sub-iteration, PRD line), `dJPrdMax`/`dJPrdMaxIdx`
 * one per sub-iteration; capacities maxIter * Nprd and maxIter (any may be NULL).
 * On a wavelength shard use the split form below (the scattering integral needs J over each PRD
 * line's whole grid, which one all-reduce provides). */
typedef struct lwhip_prd_result {
    int32_t NprdSubIter;
    int32_t Nprd;          /* number of PRD lines */
    double* dRho;
    int32_t* dRhoMaxIdx;
    double* dJPrdMax;
    int32_t* dJPrdMaxIdx;
} lwhip_prd_result;
int lwhip_redistribute_prd(lwhip_context* ctx, int maxIter, double tol, lwhip_prd_result* res);

/* One PRD sub-iteration across wavelength shards ("all-gather J rows of PRD lambda
 * before prd_scatter"), split around its two collectives:
 *   lwhip_prd_pack      this shard's J rows of every PRD line into the gather buffer (zero elsewhere);
 *                       devPtr/count describe it: all-reduce(SUM) it over the shards;
 *   lwhip_prd_partial   rho for this shard's emission wavelengths, max |d rho / rho| per line into this
 *                       shard's slots of the reduce tail, then the PRD rates pass + slab reduce;
 *                       all-reduce(SUM) the buffer of lwhip_reduce_buffer (nSum + nGather doubles);
 *   lwhip_prd_finalise  Rij/Rji of the PRD lines out; per-line (dRho, idx) [Nprd] and (dJMax, idx).
 * lwhip_redistribute_prd is this loop on one device. */
int lwhip_prd_pack(lwhip_context* ctx, void** devPtr, size_t* count);
int lwhip_prd_partial(lwhip_context* ctx);
int lwhip_prd_finalise(lwhip_context* ctx, double* dRho, int32_t* dRhoMaxIdx, double* dJMax, int32_t* dJMaxIdx);

/* Ng acceleration of the populations on the device (struct Ng, Source/Ng.hpp:16-156; driven by
 * LwContext.stat_equil -> rel_diff_ng_accelerate, Source/LwMiddleLayer.pyx:3318-3346).
 * lwhip_ng_configure(Norder, Nperiod, Ndelay): one history per active atom, seeded with the current
 *   populations (the reference builds its Ng objects when the Context is constructed); Norder = 0 keeps
 *   only what max_change needs.
 * lwhip_ng_accelerate: call after each population update: records the new populations, every Nperiod
 *   calls from the Ndelay-th on replaces them by the Ng extrapolation, and reports per active atom whether
 *   it accelerated, the max relative change between the last two recorded solutions and its flattened
 *   [level, depth] index. */
int lwhip_ng_configure(lwhip_context* ctx, int Norder, int Nperiod, int Ndelay);
int lwhip_ng_accelerate(lwhip_context* ctx, int32_t* accelerated, double* dPops, int32_t* dPopsMaxIdx);

/* time_dependent_update (FsIterationFns::time_dep_update, Source/LwFormalInterface.hpp:92,123;
 * time_dependent_update_impl, Source/UpdatePopulations.cpp:120-151): for atom `atom`, per depth
 * point solve (1 - dt Gamma_k) n_k = nOld_k with the solver of lwhip_stat_equil; n overwritten on
 * the device.  nOld: host [Nlevel, Nspace]. */
int lwhip_time_dep_update(lwhip_context* ctx, int atom, const double* nOld, double dt);

/* nr_post_update (FsIterationFns::nr_post_update, Source/LwFormalInterface.hpp:94,124;
 * nr_post_update_impl + F / Ftd, Source/UpdatePopulations.cpp:230-394): one Newton-Raphson step
 * of the coupled statistical-equilibrium (or backward-Euler, when nPrev is given) + charge-
 * conservation system of the listed atoms, per depth point: (sum Nlevel + 1) unknowns, the last
 * one the electron density.  n of the listed atoms is updated on the device, `ne` on the host.
 * Uses the device copies of Gamma and C (collisional rates). */
typedef struct lwhip_nr_args {
    int32_t Natoms;
    int32_t _pad;
    const int32_t* atoms;          /* [Natoms] indices into prob->atoms (active atoms), in equation order */
    const double* const* stages;   /* [Natoms] -> [Nlevel]: ionisation stage of each level (Atom::stages) */
    const double* const* dC;       /* [Natoms] -> [Nlevel, Nlevel, Nspace]: dC/dne, or NULL: no collisional term */
    const double* const* nPrev;    /* [Natoms] -> [Nlevel, Nspace]: previous time step, or NULL: not time dependent */
    const double* backgroundNe;    /* [Nspace] */
    double* ne;                    /* [Nspace] in: atmos.ne; out: ne + delta */
    double dt;                     /* time step (time-dependent form only) */
    double crsw;                   /* collisional-radiative switching value the Gamma pre-fill used */
} lwhip_nr_args;
int lwhip_nr_post_update(lwhip_context* ctx, const lwhip_nr_args* args);

/* spaceStart / spaceEnd of FsIterationFns::stat_eq, time_dep_update and nr_post_update
 * (Source/LwFormalInterface.hpp:90-100; the loops `for k in [spaceStart, spaceEnd)` of
 * Source/UpdatePopulations.cpp:22, :135, :316): the depth range the following lwhip_stat_equil* /
 * lwhip_time_dep_update / lwhip_nr_post_update calls of this context solve; points outside keep their
 * populations (and ne).  (-1, -1) = the whole atmosphere, the state of a new context.  Column batches
 * (lwhip_batch_stat_equil) always solve every point. */
int lwhip_set_depth_range(lwhip_context* ctx, int spaceStart, int spaceEnd);

/* Which wavelength index lwhip_formal_sol_gamma_matrices reports next to dJMax (lwhip_iter_result::dJMaxIdx):
 *   0 (default)  the first wavelength that attains dJMax -- what the reference's threaded schemes return
 *                (the per-thread maxima are merged with max_idx, Source/SimdFullIterationTemplates.hpp:700-715);
 *   1            the bookkeeping of the reference's single-thread branch, bit for bit: its loop calls
 *                `dJMax = max_idx(dJ, dJMax, maxIdx, la)` (Source/SimdFullIterationTemplates.hpp:627 with
 *                Source/Constants.hpp:114-125), which records `la` whenever the wavelength's dJ is BELOW the running
 *                maximum -- i.e. the last wavelength whose dJ is smaller than the largest one before it (0 if there is
 *                none).  The plugin selects it when the Context runs with Nthreads == 1, so a drop-in run reports the
 *                index the scalar scheme reports.  Unsharded 1D contexts; others keep mode 0. */
int lwhip_set_djmax_index_mode(lwhip_context* ctx, int mode);

/* ---- 2D short characteristics (tier 2): the formal solver primitive -------------------
 * piecewise_besser_2d with interp_linear_2d (Source/FormalScalar2d.cpp:740-1184, 209-255) on an x-periodic
 * Nz x Nx grid, given the intersection table the core builds (build_intersection_list, :1188-1327;
 * Atmosphere::intersections, Source/LwAtmosphere.hpp:145-173), flattened as below.  Not yet wired into
 * lwhip_formal_sol_gamma_matrices: this entry point solves chi, S -> I, Psi* for a batch of (ray,
 * direction) pairs. */
#define LWHIP_AXIS_NONE 0
#define LWHIP_AXIS_X 1
#define LWHIP_AXIS_Z 2
typedef struct lwhip_intersection {   /* IntersectionResult, Source/LwAtmosphere.hpp:93-140 */
    int32_t axis;                     /* InterpolationAxis (LWHIP_AXIS_*): 0 none (on a grid point), 1 interpolate along x, 2 along z */
    int32_t _pad;
    double fracZ, fracX;              /* fractional indices of the hit */
    double distance;                  /* path length to it */
} lwhip_intersection;

typedef struct lwhip_grid2d {
    int32_t Nx, Nz, Nrays, periodic;  /* 1: PERIODIC x boundaries; 0: CALLABLE ones (xLowerBc / xUpperBc below)    */
    int32_t zLowerBc, zUpperBc;       /* LWHIP_BC_ZERO / LWHIP_BC_THERMALISED / LWHIP_BC_CALLABLE (then lwhip_problem's
                                       * zLowerBc / zUpperBc carry idxs and bcData [Nlambda, Nmu, Nx])            */
    int32_t NlongChar, _pad;
    const double* x;                  /* [Nx] */
    const double* z;                  /* [Nz] */
    const double* mux;                /* [Nrays] */
    const double* muz;                /* [Nrays] */
    const double* temperature;        /* [Nz, Nx] (thermalised boundaries) */
    const lwhip_intersection* uw;     /* [Nrays, 2, Nz, Nx] upwind hit of every point */
    const lwhip_intersection* dw;     /* [Nrays, 2, Nz, Nx] downwind hit */
    const int32_t* longCharIdx;       /* [Nrays, 2, Nz, Nx] index of the point's long characteristic, or -1 */
    const int32_t* substepOff;        /* [NlongChar + 1] offsets into substeps */
    const lwhip_intersection* substeps; /* the sub-steps of every long characteristic, upwind-most first */
    /* periodic == 0 (Source/FormalScalar2d.cpp:806-852): the column a ray enters through is prescribed,
     * I(k, 0) = xLowerBc.bcData(la, idxs(mu, toObs), k) for mux > 0, I(k, Nx-1) from xUpperBc for mux < 0.
     * Both CALLABLE, idxs [Nrays, 2], bcData [Nlambda, Nmu, Nz]; no long characteristics.  NULL when periodic. */
    const struct lwhip_boundary* xLowerBc;
    const struct lwhip_boundary* xUpperBc;
} lwhip_grid2d;

/* nSolve problems: problem p is ray rays[p] (= 2 * mu + toObs), opacity chi + p * Nz * Nx, source S + ...;
 * writes I and PsiStar (= Psi / chi) of the same shape.  All pointers are host pointers. */
int lwhip_formal_solver_2d(int device, const lwhip_grid2d* grid, double wavelength, int nSolve, const int32_t* rays,
                           const double* chi, const double* S, double* I, double* PsiStar);

/* 1.5D column batches (BASELINE configs[3]): n structurally identical contexts -- same model atoms, wavelength grid,
 * solver and device; own atmospheres, profiles, populations -- advance together, one grid slice per column, on the
 * first column's stream (the others are moved onto it),
 * so a batch of short columns fills the device the way one long wavelength grid does.  Columns never exchange
 * radiation (the reference runs them as separate Contexts: README.md:9); results[i] receives column i's dJMax. */
typedef struct lwhip_batch lwhip_batch;
int lwhip_batch_create(lwhip_context* const* ctxs, int n, lwhip_batch** out);
void lwhip_batch_destroy(lwhip_batch* batch);
int lwhip_batch_formal_sol_gamma_matrices(lwhip_batch* batch, int lambdaIterate, double crsw,
                                          lwhip_iter_result* results /* [n] or NULL: nothing is read back */);
int lwhip_batch_stat_equil(lwhip_batch* batch);
/* lwhip_compute_profiles of every column, all their lines in one launch pair on the batch's stream (a launch pair per
 * line and column leaves the device empty: 7 680 pairs for 512 columns of 15 lines).  Columns whose atmosphere is
 * uploaded later (LWHIP_ATMOS) are brought up to date the same way by the next lwhip_batch_formal_sol_gamma_matrices. */
int lwhip_batch_compute_profiles(lwhip_batch* batch);

/* The intersection table of an x-periodic grid: build_intersection_list (Source/FormalScalar2d.cpp:1188-1327) with
 * dw_intersection_2d (:60-105), uw_intersection_2d (:107-152), uw_intersection_2d_frac_x (:166-206).  Host-side
 * geometry, once per atmosphere; reads Nx, Nz, Nrays, periodic, x, z, mux, muz of `grid`.  Call with uw == NULL to
 * learn the sizes (*nLongChar, *nSubsteps), then with uw, dw, longCharIdx [Nrays, 2, Nz, Nx], substepOff
 * [capLongChar + 1] and substeps [capSubsteps].  Bit-identical to the reference's table. */
int lwhip_build_intersections(const lwhip_grid2d* grid, lwhip_intersection* uw, lwhip_intersection* dw,
                              int32_t* longCharIdx, int32_t* substepOff, int32_t capLongChar,
                              lwhip_intersection* substeps, int64_t capSubsteps, int32_t* nLongChar,
                              int64_t* nSubsteps);

/* Voigt profiles phi and weights wphi of every line, on the device. */
int lwhip_compute_profiles(lwhip_context* ctx);

/* Block until all work queued on the context's stream has finished. */
int lwhip_synchronize(lwhip_context* ctx);

/* ZPlaneDecomposition (ExtraParams "ZPlaneDecomposition" with "ZPlaneDown" / "ZPlaneUp",
 * Source/SimdFullIterationTemplates.hpp:253-281, 351-384): from now on every formal solution also records the
 * intensity of the down rays in plane Nz - 2 (zPlaneDown) and of the up rays in plane 1 (zPlaneUp), host arrays
 * [Nlambda, Nrays] (1D) or [Nlambda, Nrays, Nx] (2D) filled by lwhip_download(LWHIP_I); NULL turns one off. */
int lwhip_set_zplane_outputs(lwhip_context* ctx, double* zPlaneDown, double* zPlaneUp);

/* Use `stream` (hipStream_t) for all subsequent launches; NULL = the library's own. */
int lwhip_set_stream(lwhip_context* ctx, void* stream);

/* Measurement support (bench.py): when enabled, launches of the sweep kernel (raymarch_kernel, the
 * dominant kernel of the iteration; not the pre-pass or the apply launch) are bracketed by HIP events on
 * the launch stream -- every launch for enable = 1, every n-th for enable = n > 1 (two event records
 * cost ~5 us of a 250 us step) -- and lwhip_sweep_time returns the mean duration (ms) and the number
 * of launches timed since the last reset. */
int lwhip_profile_enable(lwhip_context* ctx, int enable);
int lwhip_sweep_time(lwhip_context* ctx, double* meanMs, int* nLaunches);

/* Algorithmic bytes one iteration of this context moves (SURVEY.md 8d / DESIGN.md). */
int lwhip_algorithmic_bytes(lwhip_context* ctx, double* bytes);
/* Which sweep kernel serves this context (chosen at lwhip_create from its size, LWHIP_SWEEP overrides):
 * 0 the ray-column march (raymarch_kernel), 1 the depth-across-lanes sweep (lanesweep_kernel), 2 the 2D pipeline. */
int lwhip_sweep_kind(lwhip_context* ctx);

#ifdef __cplusplus
}
#endif
#endif /* LWHIP_H */

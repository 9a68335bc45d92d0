// lwhip_device.h -- device-side leaf numerics and address-space helpers shared by the gfx950 kernels.
#pragma once
#include "lwhip_internal.h"

#include <hip/hip_runtime.h>

// every kernel of the library is launched through this: under LWHIP_SCRATCH_POISON the scratch memory is poisoned first
#define LWHIP_LAUNCH(kernel, grid, block, lds, stream, ...)                                                                        \
    do                                                                                                                             \
    {                                                                                                                              \
        ::lwhip::dbg_scratch_poison(stream);                                                                                       \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                                         \
    } while (0)

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
#ifdef LWHIP_EXPERIMENT_NOATOMIC
    *addr += v; // timing experiment only (racy)
#else
    // ds_add_f64 on gfx950
    unsafeAtomicAdd(addr, v);
#endif
}

// Workgroup barrier that orders LDS traffic only: outstanding global loads (the software prefetch
// of the next wavelength) and stores (J, I, depth data) stay in flight across it, which a
// __syncthreads() -- s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier -- would drain.
DEVINL void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Address-space-qualified views of the pointers that live in the (device-memory) argument block:
// without them the loads compile to flat_load, which counts on lgkmcnt and so stalls every LDS
// barrier behind the HBM latency of the prefetches; tables read uniformly become scalar loads.
// Diagnostic build (-DLWHIP_DBG_POISON_LDS, tools/build_variant.py): every workgroup starts with its WHOLE LDS allocation --
// static and dynamic -- holding a NaN pattern, so that a read of an LDS entry nothing wrote shows up in the results instead of
// returning what the previous workgroup on the CU left there.  The allocation's size comes from the dispatch packet
// (hsa_kernel_dispatch_packet_t::group_segment_size, byte 28).  The shipped build compiles this to nothing.
DEVINL void dbg_poison_lds()
{
#ifdef LWHIP_DBG_POISON_LDS
    const unsigned bytes = ((const unsigned*)__builtin_amdgcn_dispatch_ptr())[7];
    const unsigned nT = blockDim.x * blockDim.y * blockDim.z;
    const unsigned t = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
#ifdef LWHIP_DBG_POISON_LDS_FINITE /* NaN is swallowed by the fmin / fmax of the Steffen derivative: a finite value is not */
    const unsigned long long pat = (unsigned long long)__double_as_longlong(1.0e-3 * (double)(1 + (t & 7)));
#else
    const unsigned long long pat = 0x7ff8dead0000beefull;
#endif
    for (unsigned o = t * 8u; o + 8u <= bytes; o += nT * 8u)
        asm volatile("ds_write_b64 %0, %1" ::"v"(o), "v"(pat) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
#endif
}

#define GLOBAL_AS __attribute__((address_space(1)))
#define CONST_AS __attribute__((address_space(4)))
#define GCD(p) ((const GLOBAL_AS double*)(p))
#define GD(p) ((GLOBAL_AS double*)(p))
#define CTAB(T, p) ((const CONST_AS T*)(p))
DEVINL const CONST_AS ContArgs* launder_args(const ContArgs* p)
{
    const CONST_AS ContArgs* q = (const CONST_AS ContArgs*)p;
    asm volatile("" : "+s"(q));
    return q;
}

template <typename T> DEVINL T ld_c(const CONST_AS T* p)
{
    T v;
    __builtin_memcpy(&v, p, sizeof(T));
    return v;
}

// 1/x to ~1 ulp: v_rcp_f64 + two Newton steps (5 VALU ops instead of the ~15 of an IEEE divide).
// Used where the reference divides by a quantity that is shared by several expressions; the
// results differ from a correctly rounded divide by at most a couple of ulp (tolerance 1e-9).
DEVINL double d_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    return r;
}

// exp(x) for -700 < x <= 0 (optical depths, Boltzmann factors): range reduction x = n ln2 + r,
// |r| <= ln2/2, degree-13 Taylor polynomial (truncation 4e-18), ldexp.  No overflow / NaN paths,
// about half the instructions of the library exp; agrees with it to ~1 ulp.
DEVINL double d_exp_neg(double x)
{
    const double n = rint(x * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// cent_deriv (Bezier.hpp:58-65) with the three reciprocals supplied
DEVINL double d_cent_deriv_r(double dsuw, double dsdw, double ruw, double rdw, double rsum,
                             double yuw, double y0, double ydw)
{
    const double S0 = (ydw - y0) * rdw;
    const double Suw = (y0 - yuw) * ruw;
    const double P0 = fabs((Suw * dsdw + S0 * dsuw) * rsum);
    return (copysign(1.0, S0) + copysign(1.0, Suw)) * fmin(fabs(Suw), fmin(fabs(S0), 0.5 * P0));
}

// Bezier3_coeffs (Bezier.hpp:81-127); the dt > 30 branch is the exact branch with edt = 0
DEVINL void d_bezier3_coeffs_r(double dt, double& alpha, double& beta, double& gamma, double& delta,
                               double& edt)
{
    const double dt2 = dt * dt;
    const double dt3 = dt2 * dt;
    if (dt < 5e-2)
    {
        edt = 1.0 - dt + 0.5 * dt2 - dt3 * (1.0 / 6.0);
        alpha = 0.25 * dt - 0.2 * dt2 + dt3 * (1.0 / 12.0);
        beta = 0.25 * dt - 0.05 * dt2 + dt3 * (1.0 / 120.0);
        gamma = 0.25 * dt - 0.15 * dt2 + 0.05 * dt3;
        delta = 0.25 * dt - 0.1 * dt2 + 0.025 * dt3;
    }
    else
    {
        edt = (dt > 30.0) ? 0.0 : exp(-dt);
        const double r3 = d_rcp(dt3);
        alpha = (6.0 - edt * (6.0 + 6.0 * dt + 3 * dt2 + dt3)) * r3;
        beta = (6.0 * edt - 6.0 + 6.0 * dt - 3.0 * dt2 + dt3) * r3;
        gamma = 3.0 * (2.0 * dt - 6.0 + edt * (6.0 + 4.0 * dt + dt2)) * r3;
        delta = 3.0 * (6.0 - 4.0 * dt + dt2 - 2.0 * edt * (3.0 + dt)) * r3;
    }
}


// ---- wave-wide inclusive scan of affine maps I -> E I + B with DPP (row shifts + row broadcasts) ----------
// Lane order = application order: after the scan lane L holds the composition of lanes 0..L.
// DPP moves are 32-bit; a double travels as two halves.  Lanes without a source keep the identity.
template <int CTRL, int ROW_MASK> DEVINL double dpp_f64(double identity, double v)
{
    const unsigned long long iv = __double_as_longlong(identity);
    const unsigned long long vv = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)iv, (int)(unsigned)vv, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(iv >> 32), (int)(unsigned)(vv >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
template <int CTRL, int ROW_MASK> DEVINL void affine_scan_step(double& E, double& B)
{
    const double Ep = dpp_f64<CTRL, ROW_MASK>(1.0, E);
    const double Bp = dpp_f64<CTRL, ROW_MASK>(0.0, B);
    B = fma(E, Bp, B);
    E = E * Ep;
}
DEVINL void affine_scan_wave(double& E, double& B)
{
    affine_scan_step<0x111, 0xf>(E, B); // row_shr:1
    affine_scan_step<0x112, 0xf>(E, B); // row_shr:2
    affine_scan_step<0x114, 0xf>(E, B); // row_shr:4
    affine_scan_step<0x118, 0xf>(E, B); // row_shr:8
    affine_scan_step<0x142, 0xa>(E, B); // row_bcast:15 into rows 1 and 3
    affine_scan_step<0x143, 0xc>(E, B); // row_bcast:31 into rows 2 and 3
}
// PRD: per line the max over (depth, slice) of the scattering integral's maxima of |d rho / rho| (lwhip_prd.hip), by ONE
// wavefront: prd_change_kernel on shards, an extra workgroup of the apply launch on one device (lwhip_kernels.hip)
DEVINL double prd_change_reduce(const PrdLineArgs& a, const int laneIn)
{
    const double* changeVal = a.changeVal;
    const int* changeIdx = a.changeIdx;
    const int nEntries = (a.laHi > a.laLo) ? a.Ns * a.slices : 0;
    double* tail = a.tail;
    const int world = a.world, rank = a.rank, q = a.q, Nprd = a.Nprd;
    // one wavefront: max value, smallest flattened index on ties (max_idx semantics: strictly greater
    // updates in index order, so the first occurrence wins; a zero maximum keeps idx 0).  The result
    // goes into this shard's slot of the line in the reduce tail; the other shards' slots are zeroed
    // (the all-reduce sums them).
    const int lane = laneIn;
    double best = 0.0;
    int idx = 0x7fffffff;
    for (int e = lane; e < nEntries; e += 64)
    {
        const double v = changeVal[e];
        const int i = changeIdx[e];
        if (i != 0x7fffffff && v > 0.0 && (v > best || (v == best && i < idx)))
        {
            best = v;
            idx = i;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        const double v = __shfl_xor(best, off);
        const int i = __shfl_xor(idx, off);
        if (v > best || (v == best && i < idx))
        {
            best = v;
            idx = i;
        }
    }
    for (int rk = lane; rk < world; rk += 64)
    {
        double* slot = tail + ((size_t)rk * Nprd + q) * 2;
        slot[0] = (rk == rank) ? best : 0.0;
        slot[1] = (rk == rank && best > 0.0) ? (double)idx : 0.0;
    }
    return best; // (every lane holds it)
}

}

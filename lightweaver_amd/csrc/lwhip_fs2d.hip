// lwhip_fs2d.hip -- 2D short characteristics: piecewise_besser_2d with interp_linear_2d
// (Source/FormalScalar2d.cpp:740-1184, 209-255) on an x-periodic grid, as a device primitive.
//
// One workgroup per (ray, direction) problem; thread p owns the p-th column of a z plane in the ray's
// processing order (x ascending for mu_x >= 0, descending otherwise).  Planes are sequential.  Inside a
// plane the reference walks the columns in order because a shallow ray's upwind hit lies on the x face
// shared with the column just computed: I(k, j) = edt * [(1 - f) I(zm, j') + f I(zp, j')] + sources, with one
// of zm / zp the plane being computed.  That is an affine recurrence I_p = a_p I_(p-1) + b_p along the
// processing order -- the same composition as the 1D sweep's depth recurrence -- so every thread evaluates
// its (a, b) independently and a workgroup-wide scan (DPP wavefront scans + one LDS hop) replaces the walk.
// The first column of a shallow ray is the periodic long characteristic, whose sub-steps reach back to the
// previous plane and never read the plane in progress, so its a is 0 and the recurrence has a known start.
// fp64; HBM traffic per solve is chi, S read ~3x (L1/L2-resident stencil neighbours) and I, Psi* written once;
// the intersection table (64 B per point) is shared by all wavelengths and stays in L2.
#include "lwhip_device.h"
#include "../../include/lwhip.h"

#include <vector>

namespace lwhip
{
namespace
{

// interp_linear_2d :209-255
DEVINL double f2_interp(const lwhip_intersection& loc, const double* p, int Nx)
{
    if (loc.axis == 1)
    {
        const int xm = (int)loc.fracX, z = (int)loc.fracZ;
        const double frac = loc.fracX - xm;
        return (1.0 - frac) * p[(size_t)z * Nx + xm] + frac * p[(size_t)z * Nx + xm + 1];
    }
    if (loc.axis == 2)
    {
        const int zm = (int)loc.fracZ, x = (int)loc.fracX;
        const double frac = loc.fracZ - zm;
        return (1.0 - frac) * p[(size_t)zm * Nx + x] + frac * p[(size_t)(zm + 1) * Nx + x];
    }
    return p[(size_t)((int)loc.fracZ) * Nx + (int)loc.fracX];
}

// besser_control_point :257-294
DEVINL double f2_control_point(double hM, double hP, double yM, double yO, double yP)
{
    const double deltaMO = yO - yM;
    const double dM = (yO - yM) / hM;
    const double dP = (yP - yO) / hP;
    if (dM * dP <= 0.0)
        return yO;
    double yOp = (hM * dP + hP * dM) / (hM + hP);
    double cM = yO - 0.5 * hM * yOp;
    double cP = yO + 0.5 * hP * yOp;
    double minYMO = yM, maxYMO = yO, minYOP = yO, maxYOP = yP;
    if (deltaMO < 0.0)
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

// besser_coeffs :715-738
DEVINL void f2_coeffs(double t, double& M, double& O, double& Cc, double& edt)
{
    if (t < 0.14)
    {
        M = (t * (t * (t * (t * (t * (t * ((140.0 - 18.0 * t) * t - 945.0) + 5400.0) - 25200.0) + 90720.0) - 226800.0) + 302400.0)) / 907200.0;
        O = (t * (t * (t * (t * (t * (t * ((10.0 - t) * t - 90.0) + 720.0) - 5040.0) + 30240.0) - 151200.0) + 604800.0)) / 1814400.0;
        Cc = (t * (t * (t * (t * (t * (t * ((35.0 - 4.0 * t) * t - 270.0) + 1800.0) - 10080.0) + 45360.0) - 151200.0) + 302400.0)) / 907200.0;
        edt = (t * (t * (t * (t * (t * (t * ((t / 40320.0 - 1.0 / 5040.0) * t + 1.0 / 720) - 1.0 / 120.0) + 1.0 / 24.0) - 1.0 / 6.0) + 1.0 / 2.0) - 1.0)) + 1.0;
    }
    else
    {
        const double t2 = t * t;
        const double e = exp(-t);
        M = (2.0 - e * (t2 + 2.0 * t + 2.0)) / t2;
        O = 1.0 - 2.0 * (e + t - 1.0) / t2;
        Cc = 2.0 * (t - 2.0 + e * (t + 2.0)) / t2;
        edt = e;
    }
}

// one BESSER step: I = edt Iuw + src; returns edt, src, psi and the chi control point
DEVINL void f2_besser(double dsUw, double dsDw, double chiUw, double chiLocal, double chiDw, double Suw, double SLocal,
                      double Sdw, double& edt, double& src, double& psi, double& chiC)
{
    chiC = f2_control_point(dsUw, dsDw, chiUw, chiLocal, chiDw);
    const double dtauUw = (1.0 / 3.0) * (chiUw + chiLocal + chiC) * dsUw;
    const double dtauDw = (0.5) * (chiLocal + chiDw) * dsDw;
    const double SC = f2_control_point(dtauUw, dtauDw, Suw, SLocal, Sdw);
    double M, O, Cc;
    f2_coeffs(dtauUw, M, O, Cc, edt);
    src = M * Suw + O * SLocal + Cc * SC;
    psi = O + Cc;
}

// the piecewise-linear step (no downwind point): I = e Iuw + src
DEVINL void f2_linear(double dtau, double Suw, double SLocal, double& e, double& src, double& psi)
{
    double w0, w1;
    d_w2(dtau, w0, w1);
    const double c1 = (Suw - SLocal) / dtau;
    e = 1.0 - w0;
    src = w0 * SLocal + w1 * c1;
    psi = w0 - w1 / dtau;
}

// long characteristic up to the point's upwind hit (:1003-1043): intensity there and the last chi control point
DEVINL double f2_long_char(const Fs2dArgs& a, int lc, const double* chi, const double* S, const double* I,
                           double chiHere, double SHere, double& chiC)
{
    const int Nx = a.Nx;
    const lwhip_intersection* st = a.substeps + a.substepOff[lc];
    const int n = a.substepOff[lc + 1] - a.substepOff[lc];
    double Iuw = f2_interp(st[0], I, Nx);
    double edt, src, psi;
    for (int s = 1; s < n - 1; ++s)
    {
        f2_besser(st[s - 1].distance, st[s + 1].distance, f2_interp(st[s - 1], chi, Nx), f2_interp(st[s], chi, Nx),
                  f2_interp(st[s + 1], chi, Nx), f2_interp(st[s - 1], S, Nx), f2_interp(st[s], S, Nx),
                  f2_interp(st[s + 1], S, Nx), edt, src, psi, chiC);
        Iuw = edt * Iuw + src;
    }
    const int s = n - 1;
    f2_besser(st[s - 1].distance, st[s].distance, f2_interp(st[s - 1], chi, Nx), f2_interp(st[s], chi, Nx), chiHere,
              f2_interp(st[s - 1], S, Nx), f2_interp(st[s], S, Nx), SHere, edt, src, psi, chiC);
    return edt * Iuw + src;
}

__global__ void __launch_bounds__(1024) fs2d_kernel(const Fs2dArgs a)
{
    __shared__ double sE[16], sB[16];
    const int Nx = a.Nx, Nz = a.Nz;
    // problem p: ray rays[p % nRayCycle]; wavelength wavs[p / nRayCycle] (2D iteration: a batch of wavelengths x
    // the active rays) or the single a.wav (primitive)
    const int ray = a.rays[blockIdx.x % a.nRayCycle];
    const double wavHere = a.wavs ? a.wavs[blockIdx.x / a.nRayCycle] : a.wav;
    const int mu = ray >> 1;
    const bool toObs = (ray & 1) != 0;
    const double mux = toObs ? a.mux[mu] : -a.mux[mu];
    const int dk = toObs ? -1 : 1;
    const int kStart = toObs ? Nz - 1 : 0;
    const int kEnd = toObs ? 0 : Nz - 1;
    const int dj = (mux < 0) ? -1 : 1;
    const int jStart = (mux < 0) ? Nx - 1 : 0;
    const int p = threadIdx.x;               // position in processing order
    const bool on = p < Nx;
    const int j = jStart + (on ? p : 0) * dj;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nWaves = (blockDim.x + 63) >> 6;
    const size_t plane0 = ((size_t)ray * Nz) * Nx;
    const lwhip_intersection* UW = a.uw + plane0;
    const lwhip_intersection* DW = a.dw + plane0;
    const int32_t* LC = a.longCharIdx + plane0;
    const double* chi = a.chi + (size_t)blockIdx.x * Nz * Nx;
    const double* S = a.S + (size_t)blockIdx.x * Nz * Nx;
    double* I = a.I + (size_t)blockIdx.x * Nz * Nx;
    double* Ps = a.PsiStar + (size_t)blockIdx.x * Nz * Nx;

    // boundary plane :876-948
    if (on)
    {
        const size_t o = (size_t)kStart * Nx + j;
        double I0 = 0.0;
        const int bc = toObs ? a.zLowerBc : a.zUpperBc;
        if (bc == LWHIP_BC_THERMALISED)
        {
            const lwhip_intersection dw = DW[o];
            const double chiDw = f2_interp(dw, chi, Nx);
            const double dtauDw = 0.5 * fabs(dw.distance) * (chi[o] + chiDw);
            const double Tdw = f2_interp(dw, a.temperature, Nx);
            if (toObs)
            {
                const double B0 = d_planck(Tdw, wavHere), B1 = d_planck(a.temperature[o], wavHere);
                I0 = B1 - (B0 - B1) / dtauDw;
            }
            else
            {
                const double B0 = d_planck(a.temperature[o], wavHere), B1 = d_planck(Tdw, wavHere);
                I0 = B0 - (B1 - B0) / dtauDw;
            }
        }
        I[o] = I0;
        Ps[o] = 0.0;
    }
    __syncthreads();

    for (int k = kStart + dk;; k += dk)
    {
        const bool last = (k == kEnd);
        // ---- every column's affine map I(k, j) = A * I(k, previous column) + B ----------------------------
        double A = 0.0, B = 0.0, psi = 0.0;
        size_t o = 0;
        if (on)
        {
            o = (size_t)k * Nx + j;
            const lwhip_intersection uw = UW[o];
            const lwhip_intersection dw = DW[o];
            const int lc = LC[o];
            const double chiL = chi[o], SL = S[o];
            double e, src;
            if (lc < 0)
            {
                const double chiUw = f2_interp(uw, chi, Nx);
                const double Suw = f2_interp(uw, S, Nx);
                if (last || dw.distance == 0.0)
                {
                    // linear step: the last plane (:1086-1101) uses the signed distance, the interior's
                    // zero-downwind branch (:961-975) its absolute value; both are non-negative here
                    const double dtau = 0.5 * (chiUw + chiL) * (last ? uw.distance : fabs(uw.distance));
                    f2_linear(dtau, Suw, SL, e, src, psi);
                }
                else
                {
                    double chiC;
                    f2_besser(uw.distance, dw.distance, chiUw, chiL, f2_interp(dw, chi, Nx), Suw, SL, f2_interp(dw, S, Nx),
                              e, src, psi, chiC);
                }
                // the upwind intensity: previous plane (known) and, for a hit on the x face, this plane's
                // previous column (the unknown of the recurrence)
                if (uw.axis == 2)
                {
                    const int zm = (int)uw.fracZ, x = (int)uw.fracX;
                    const double frac = uw.fracZ - zm;
                    const double wK = (zm == k) ? (1.0 - frac) : frac;         // weight of plane k
                    const double wP = (zm == k) ? frac : (1.0 - frac);
                    const int zOther = (zm == k) ? zm + 1 : zm;
                    A = e * wK;
                    B = e * (wP * I[(size_t)zOther * Nx + x]) + src;
                }
                else
                {
                    A = 0.0;
                    B = e * f2_interp(uw, I, Nx) + src;
                }
            }
            else
            {
                double chiC;
                const double Iuw = f2_long_char(a, lc, chi, S, I, chiL, SL, chiC);
                const lwhip_intersection st = a.substeps[a.substepOff[lc + 1] - 1]; // = the upwind hit
                const double chiUw = f2_interp(st, chi, Nx);
                const double Suw = f2_interp(st, S, Nx);
                if (last || dw.distance == 0.0)
                {
                    const double dtau = 0.5 * (chiUw + chiL) * (last ? uw.distance : st.distance);
                    f2_linear(dtau, last ? f2_interp(uw, S, Nx) : Suw, SL, e, src, psi);
                }
                else
                {
                    // :1060-1075: the chi control point of the last sub-step is re-used, not recomputed
                    const double chiDw = f2_interp(dw, chi, Nx);
                    const double dtauUw = (1.0 / 3.0) * (chiUw + chiL + chiC) * st.distance;
                    const double dtauDw = (0.5) * (chiL + chiDw) * dw.distance;
                    const double Sdw = f2_interp(dw, S, Nx);
                    const double SC = f2_control_point(dtauUw, dtauDw, Suw, SL, Sdw);
                    double M, O, Cc;
                    f2_coeffs(dtauUw, M, O, Cc, e);
                    src = M * Suw + O * SL + Cc * SC;
                    psi = O + Cc;
                }
                A = 0.0;
                B = e * Iuw + src;
            }
        }
        else
        {
            A = 1.0; // identity beyond the row
            B = 0.0;
        }
        // ---- inclusive scan of the maps along the processing order ------------------------------------------
        double E = A, Bs = B;
        affine_scan_wave(E, Bs);
        if (lane == 63)
        {
            sE[wave] = E;
            sB[wave] = Bs;
        }
        __syncthreads();
        if (wave > 0)
        {
            // compose with the waves before this one (at most 15 small steps; the first map has A = 0)
            double Ep = 1.0, Bp = 0.0;
            for (int w = 0; w < wave; ++w)
            {
                Bp = sE[w] * Bp + sB[w];
                Ep = sE[w] * Ep;
            }
            Bs = E * Bp + Bs;
            E = E * Ep;
        }
        (void)nWaves;
        if (on)
        {
            I[o] = Bs; // the first column's map has A = 0, so the composed offset is the intensity
            Ps[o] = psi / chi[o];
        }
        __syncthreads();
        if (last)
            break;
    }
}
}

hipError_t launch_fs2d(const Fs2dArgs& a, int nSolve, hipStream_t stream)
{
    const int threads = ((a.Nx + 63) / 64) * 64;
    hipLaunchKernelGGL(fs2d_kernel, dim3(nSolve), dim3(threads), 0, stream, a);
    return hipGetLastError();
}

extern "C" int lwhip_formal_solver_2d_impl(int device, const lwhip_grid2d* g, double wavelength, int nSolve,
                                           const int32_t* rays, const double* chi, const double* S, double* I,
                                           double* PsiStar, char* err, int errLen);
}

using namespace lwhip;

#define F2_TRY(x)                                                                                       \
    do                                                                                                  \
    {                                                                                                   \
        hipError_t e__ = (x);                                                                           \
        if (e__ != hipSuccess)                                                                          \
        {                                                                                               \
            std::snprintf(err, errLen, "%s: %s", #x, hipGetErrorString(e__));                           \
            for (void* q : allocs)                                                                      \
                (void)hipFree(q);                                                                       \
            return LWHIP_ERR_DEVICE;                                                                    \
        }                                                                                               \
    } while (0)

extern "C" int lwhip_formal_solver_2d_impl(int device, const lwhip_grid2d* g, double wavelength, int nSolve,
                                           const int32_t* rays, const double* chi, const double* S, double* I,
                                           double* PsiStar, char* err, int errLen)
{
    std::vector<void*> allocs;
    const size_t Ns = (size_t)g->Nx * g->Nz;
    const size_t nSt = (size_t)g->Nrays * 2 * Ns;
    const size_t nSub = (size_t)g->substepOff[g->NlongChar];
    F2_TRY(hipSetDevice(device));
    auto up = [&](const void* host, size_t bytes, void** dev) -> hipError_t {
        hipError_t e = hipMalloc(dev, bytes ? bytes : 8);
        if (e != hipSuccess)
            return e;
        allocs.push_back(*dev);
        return bytes ? hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    Fs2dArgs a{};
    a.Nx = g->Nx;
    a.Nz = g->Nz;
    a.Nrays = g->Nrays;
    a.zLowerBc = g->zLowerBc;
    a.zUpperBc = g->zUpperBc;
    a.wav = wavelength;
    a.wavs = nullptr;
    a.nRayCycle = nSolve;
    F2_TRY(up(g->mux, g->Nrays * sizeof(double), (void**)&a.mux));
    F2_TRY(up(g->temperature, Ns * sizeof(double), (void**)&a.temperature));
    F2_TRY(up(g->uw, nSt * sizeof(lwhip_intersection), (void**)&a.uw));
    F2_TRY(up(g->dw, nSt * sizeof(lwhip_intersection), (void**)&a.dw));
    F2_TRY(up(g->longCharIdx, nSt * sizeof(int32_t), (void**)&a.longCharIdx));
    F2_TRY(up(g->substepOff, ((size_t)g->NlongChar + 1) * sizeof(int32_t), (void**)&a.substepOff));
    F2_TRY(up(g->substeps, nSub * sizeof(lwhip_intersection), (void**)&a.substeps));
    F2_TRY(up(rays, (size_t)nSolve * sizeof(int32_t), (void**)&a.rays));
    F2_TRY(up(chi, (size_t)nSolve * Ns * sizeof(double), (void**)&a.chi));
    F2_TRY(up(S, (size_t)nSolve * Ns * sizeof(double), (void**)&a.S));
    F2_TRY(hipMalloc((void**)&a.I, (size_t)nSolve * Ns * sizeof(double)));
    allocs.push_back(a.I);
    F2_TRY(hipMalloc((void**)&a.PsiStar, (size_t)nSolve * Ns * sizeof(double)));
    allocs.push_back(a.PsiStar);
    F2_TRY(launch_fs2d(a, nSolve, nullptr));
    F2_TRY(hipDeviceSynchronize());
    F2_TRY(hipMemcpy(I, a.I, (size_t)nSolve * Ns * sizeof(double), hipMemcpyDeviceToHost));
    F2_TRY(hipMemcpy(PsiStar, a.PsiStar, (size_t)nSolve * Ns * sizeof(double), hipMemcpyDeviceToHost));
    for (void* q : allocs)
        (void)hipFree(q);
    return LWHIP_OK;
}

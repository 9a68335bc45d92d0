// lwhip_fs2d.hip -- 2D short characteristics: piecewise_besser_2d with interp_linear_2d
// (Source/FormalScalar2d.cpp:740-1184, 209-255) on an x-periodic grid, as a device primitive.
//
// The reference walks the columns of a z plane in the ray's processing order (x ascending for mu_x >= 0,
// descending otherwise) because a shallow ray's upwind hit lies on the x face shared with the column just
// computed: I(k, j) = edt * [(1 - f) I(zm, j') + f I(zp, j')] + sources, with one of zm / zp the plane being
// computed.  That is an affine recurrence I_p = a_p I_(p-1) + b_p along the processing order -- the same
// composition as the 1D sweep's depth recurrence -- and everything expensive in it (optical depths, control
// points, exp, Psi*) does not depend on any intensity.  So a solve is two passes:
//   fs2d_coef_kernel   one thread per (solve, plane, column): the point's map (A, c1, c2, src, where to read I)
//   fs2d_scan_kernel   one wavefront per solve: planes in sequence, a DPP scan of the maps per plane
// The first column of a shallow ray is the periodic long characteristic, whose sub-steps reach back to the
// previous plane and never read the plane in progress, so its a is 0 and the recurrence has a known start.
// fp64; HBM traffic per point and solve: chi, S read (neighbours through L1/L2), the 28 B map written and read,
// I, Psi* written once; the intersection table (64 B per point) is shared by all wavelengths of a tile.
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

// the same interpolation as two flat indices and weights, shared by every field read at one hit (chi and S):
// value = w0 p[i0] + w1 p[i1], bit-identical to f2_interp (for a hit on a grid point w1 = 0 and i1 = i0)
struct F2Loc
{
    int i0, i1;
    double w0, w1;
};
DEVINL F2Loc f2_loc(const lwhip_intersection& loc, int Nx)
{
    F2Loc l;
    const int z = (int)loc.fracZ, x = (int)loc.fracX;
    l.i0 = z * Nx + x;
    const double frac = (loc.axis == 1) ? loc.fracX - x : loc.fracZ - z;
    const bool two = loc.axis != 0;
    l.i1 = two ? l.i0 + ((loc.axis == 1) ? 1 : Nx) : l.i0;
    l.w0 = two ? 1.0 - frac : 1.0;
    l.w1 = two ? frac : 0.0;
    return l;
}
DEVINL double f2_at(const F2Loc& l, const double* p) { return l.w0 * p[l.i0] + l.w1 * p[l.i1]; }
// chi and S of a hit from the interleaved field: one 16-byte load per stencil point instead of two 8-byte ones (pass 1 is
// bound by the number of cache lines its loads touch)
DEVINL void f2_at2(const F2Loc& l, const double2* cs, double& chi, double& S)
{
    const double2 v0 = cs[l.i0], v1 = cs[l.i1];
    chi = l.w0 * v0.x + l.w1 * v1.x;
    S = l.w0 * v0.y + l.w1 * v1.y;
}

// besser_control_point :257-294.  Division-free: the caller supplies 1 / hM, 1 / hP and 1 / (hM + hP) (d_rcp:
// v_rcp_f64 + two Newton steps, ~1 ulp; an IEEE divide is ~25 instructions and the step has ten of them)
DEVINL double f2_control_point(double hM, double hP, double rhM, double rhP, double rSum, double yM, double yO, double yP)
{
    const double deltaMO = yO - yM;
    const double dM = (yO - yM) * rhM;
    const double dP = (yP - yO) * rhP;
    if (dM * dP <= 0.0)
        return yO;
    double yOp = (hM * dP + hP * dM) * rSum;
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
        yOp = (cP - yO) * (2.0 * rhP);
        cM = yO - 0.5 * hM * yOp;
    }
    return cM;
}

// besser_coeffs :715-738
DEVINL void f2_coeffs(double t, double& M, double& O, double& Cc, double& edt)
{
    if (t < 0.14)
    {
        M = (t * (t * (t * (t * (t * (t * ((140.0 - 18.0 * t) * t - 945.0) + 5400.0) - 25200.0) + 90720.0) - 226800.0) + 302400.0)) * (1.0 / 907200.0);
        O = (t * (t * (t * (t * (t * (t * ((10.0 - t) * t - 90.0) + 720.0) - 5040.0) + 30240.0) - 151200.0) + 604800.0)) * (1.0 / 1814400.0);
        Cc = (t * (t * (t * (t * (t * (t * ((35.0 - 4.0 * t) * t - 270.0) + 1800.0) - 10080.0) + 45360.0) - 151200.0) + 302400.0)) * (1.0 / 907200.0);
        edt = (t * (t * (t * (t * (t * (t * ((t * (1.0 / 40320.0) - 1.0 / 5040.0) * t + 1.0 / 720) - 1.0 / 120.0) + 1.0 / 24.0) - 1.0 / 6.0) + 1.0 / 2.0) - 1.0)) + 1.0;
    }
    else
    {
        const double t2 = t * t;
        const double rt2 = d_rcp(t2);
        const double e = (t > 700.0) ? 0.0 : d_exp_neg(-t);
        M = (2.0 - e * (t2 + 2.0 * t + 2.0)) * rt2;
        O = 1.0 - 2.0 * (e + t - 1.0) * rt2;
        Cc = 2.0 * (t - 2.0 + e * (t + 2.0)) * rt2;
        edt = e;
    }
}

// one BESSER step: I = edt Iuw + src; returns edt, src, psi and the chi control point
DEVINL void f2_besser(double dsUw, double dsDw, double chiUw, double chiLocal, double chiDw, double Suw, double SLocal,
                      double Sdw, double& edt, double& src, double& psi, double& chiC)
{
    chiC = f2_control_point(dsUw, dsDw, d_rcp(dsUw), d_rcp(dsDw), d_rcp(dsUw + dsDw), chiUw, chiLocal, chiDw);
    const double dtauUw = (1.0 / 3.0) * (chiUw + chiLocal + chiC) * dsUw;
    const double dtauDw = (0.5) * (chiLocal + chiDw) * dsDw;
    const double SC = f2_control_point(dtauUw, dtauDw, d_rcp(dtauUw), d_rcp(dtauDw), d_rcp(dtauUw + dtauDw), Suw, SLocal, Sdw);
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
    const double rdt = d_rcp(dtau);
    const double c1 = (Suw - SLocal) * rdt;
    e = 1.0 - w0;
    src = w0 * SLocal + w1 * c1;
    psi = w0 - w1 * rdt;
}

// long characteristic up to the point's upwind hit (:1003-1043).  The walk is affine in the intensity at its far
// end, I(hit) = Ecum * I(st[0]) + acc, so the pass that knows no intensities can finish it; also returns the last
// chi control point.
DEVINL void f2_long_char(const Fs2dArgs& a, int lc, const double2* cs, double chiHere, double SHere,
                         double& Ecum, double& acc, double& chiC)
{
    const int Nx = a.Nx;
    const lwhip_intersection* st = a.substeps + a.substepOff[lc];
    const int n = a.substepOff[lc + 1] - a.substepOff[lc];
    Ecum = 1.0;
    acc = 0.0;
    double edt, src, psi;
    // every sub-step is read three times (as the upwind, the local and the downwind point of consecutive steps):
    // carry its chi and S along instead of interpolating again
    F2Loc l0 = f2_loc(st[0], Nx), l1 = f2_loc(st[1], Nx);
    double chiA, SA, chiB, SB;
    f2_at2(l0, cs, chiA, SA);
    f2_at2(l1, cs, chiB, SB);
    for (int s = 1; s < n - 1; ++s)
    {
        const F2Loc l2 = f2_loc(st[s + 1], Nx);
        double chiC2, SC2;
        f2_at2(l2, cs, chiC2, SC2);
        f2_besser(st[s - 1].distance, st[s + 1].distance, chiA, chiB, chiC2, SA, SB, SC2, edt, src, psi, chiC);
        acc = edt * acc + src;
        Ecum = edt * Ecum;
        chiA = chiB;
        SA = SB;
        chiB = chiC2;
        SB = SC2;
    }
    const int s = n - 1;
    f2_besser(st[s - 1].distance, st[s].distance, chiA, chiB, chiHere, SA, SB, SHere, edt, src, psi, chiC);
    acc = edt * acc + src;
    Ecum = edt * Ecum;
}

// where the intensity of a hit on a z plane is read: weights of I(idx), I(idx + 1) and the encoded index
DEVINL void f2_hit_weights(const lwhip_intersection& loc, int Nx, int kPrev, double scale, double& c1, double& c2,
                           int& idx)
{
    const int z = (int)loc.fracZ;
    int x;
    if (loc.axis == 1)
    {
        x = (int)loc.fracX;
        const double frac = loc.fracX - x;
        c1 = scale * (1.0 - frac);
        c2 = scale * frac;
    }
    else
    {
        x = (int)loc.fracX;
        c1 = scale;
        c2 = 0.0;
    }
    idx = (z == kPrev) ? x : -(z * Nx + x) - 1;
}

// Pass 1, one thread per (solve, plane, column): everything of the step that does not need an intensity -- the
// optical depths, control points, BESSER weights, Psi*, the whole long characteristic -- reduced to the affine map
// of the point.  Fully parallel; this is where the flops and the chi, S stencil reads are.
// LC_INLINE: the long characteristics are walked here (the stand-alone solver primitive); otherwise their results come
// from fs2d_longchar_kernel's buffer, and without the walk the kernel needs 62 VGPRs: 8 wavefronts per SIMD, no spills.
template <bool LC_INLINE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LC_INLINE ? 6 : 8, LC_INLINE ? 6 : 8)))
fs2d_coef_kernel(const Fs2dArgs a)
{
    dbg_poison_lds();
    const int Nx = a.Nx, Nz = a.Nz;
    const int Ns = Nx * Nz;
    // workgroup order (blockIdx.x fastest): wavelength, then ray, then the 256-point tile -- neighbours in launch order
    // share the tile's intersection records (64 B per point, the largest read), so each XCD's L2 serves them after
    // the first.  A 3D grid, so that no index is recovered by an integer division (~40 scalar instructions each).
    const int laIdx = blockIdx.x;
    const int rayIdx = blockIdx.y;
    const int solve = laIdx * a.nRayCycle + rayIdx;
    const int t = blockIdx.z * blockDim.x + threadIdx.x;
    if (t >= Ns)
        return;
    // solve s: ray rays[s % nRayCycle]; wavelength wavs[s / nRayCycle] (2D iteration: a batch of wavelengths x
    // the active rays) or the single a.wav (primitive)
    // (uniform table reads through the constant address space: scalar loads, not a chain of vector-load latencies
    // at the start of every short-lived workgroup)
    const int ray = CTAB(int32_t, a.rays)[rayIdx];
    const double wavHere = a.wavs ? CTAB(double, a.wavs)[laIdx] : a.wav;
    const int mu = ray >> 1;
    const bool toObs = (ray & 1) != 0;
    const double mux = toObs ? CTAB(double, a.mux)[mu] : -CTAB(double, a.mux)[mu];
    const int dk = toObs ? -1 : 1;
    const int kStart = toObs ? Nz - 1 : 0;
    const int kEnd = toObs ? 0 : Nz - 1;
    // k = t / Nx through a float reciprocal and one correction (t < 2^23: Nspace is far below)
    int k = (int)(((float)t + 0.5f) * a.rNx);
    k -= (k * Nx > t) ? 1 : 0;
    k += ((k + 1) * Nx <= t) ? 1 : 0;
    const int p = t - k * Nx;                   // p: position in the ray's processing order
    const int j = (mux < 0) ? Nx - 1 - p : p;
    const size_t plane0 = ((size_t)ray * Nz) * Nx;
    const lwhip_intersection* UW = a.uw + plane0;
    const lwhip_intersection* DW = a.dw + plane0;
    const int32_t* LC = a.longCharIdx + plane0;
    const double2* cs = a.cs + (size_t)solve * Ns;
    double* Ps = a.PsiStar + (size_t)solve * Ns;
    double* cf = a.coef + (size_t)solve * 3 * Ns + (size_t)k * Nx + p;
    int32_t* ci = a.cidx + (size_t)solve * Ns + (size_t)k * Nx + p;
    const size_t o = (size_t)k * Nx + j;
    double A = 0.0, c1 = 0.0, c2 = 0.0, src = 0.0, psi = 0.0;
    int idx = 0;
    const bool edge = !a.periodic && mux != 0.0; // fixed x boundaries and an inclined ray

    if (edge && p == 0)
    {
        // the column the ray enters through is prescribed on every plane (:806-852): a constant map
        const int la = a.la0 + laIdx;
        const bool low = mux > 0.0;
        const int muIdx = (low ? a.xIdxLow : a.xIdxUp)[ray];
        const int Nmu = low ? a.xNmuLow : a.xNmuUp;
        src = (low ? a.xbcLow : a.xbcUp)[((size_t)la * Nmu + muIdx) * Nz + k];
        cf[0] = 0.0;
        cf[(size_t)Ns] = 0.0;
        cf[(size_t)2 * Ns] = src;
        *ci = 0;
        Ps[o] = 0.0;
        return;
    }

    if (k == kStart)
    {
        // boundary plane :876-948
        const int bc = toObs ? a.zLowerBc : a.zUpperBc;
        if (bc == LWHIP_BC_THERMALISED)
        {
            // (:884-898) the last column of a fixed-boundary grid has no downwind point: the preceding column's
            const lwhip_intersection dw = (edge && p == Nx - 1) ? DW[o - (mux < 0 ? -1 : 1)] : DW[o];
            double chiDw, SDwUnused;
            f2_at2(f2_loc(dw, Nx), cs, chiDw, SDwUnused);
            const double dtauDw = 0.5 * fabs(dw.distance) * (cs[o].x + chiDw);
            const double Tdw = f2_interp(dw, a.temperature, Nx);
            if (toObs)
            {
                const double B0 = d_planck(Tdw, wavHere), B1 = d_planck(a.temperature[o], wavHere);
                src = B1 - (B0 - B1) / dtauDw;
            }
            else
            {
                const double B0 = d_planck(a.temperature[o], wavHere), B1 = d_planck(Tdw, wavHere);
                src = B0 - (B1 - B0) / dtauDw;
            }
        }
        else if (bc == LWHIP_BC_CALLABLE)
        {
            // :930-938: the prescribed intensity of the plane the ray starts from
            const int la = a.la0 + laIdx;
            const int muIdx = (toObs ? a.zIdxLow : a.zIdxUp)[ray];
            const int Nmu = toObs ? a.zNmuLow : a.zNmuUp;
            src = (toObs ? a.zbcLow : a.zbcUp)[((size_t)la * Nmu + muIdx) * Nx + j];
        }
        cf[0] = 0.0;
        cf[(size_t)Ns] = 0.0;
        cf[(size_t)2 * Ns] = src;
        *ci = 0;
        Ps[o] = 0.0;
        return;
    }

    const bool last = (k == kEnd);
    const int kPrev = k - dk;
    lwhip_intersection uw, dw;
    if (a.uwS)
    {
        // fracZ = z + frac and fracX = x + frac reproduce the record bit for bit (frac was formed by the exact
        // subtraction of the integer part)
        auto unpack = [&](const int32_t* W, const double* F, lwhip_intersection& out) {
            const size_t r = plane0 + o;
            const int w = W[r];
            const double frac = F[r];
            out.distance = F[a.nRec + r];
            out.axis = (w >> 26) & 3;
            const int z = k + ((w >> 28) & 3) - 1;
            const int x = (w & 0x3ffffff) - z * Nx;
            out.fracZ = (double)z + (out.axis == 2 ? frac : 0.0);
            out.fracX = (double)x + (out.axis == 1 ? frac : 0.0);
        };
        unpack(a.uwA, a.uwS, uw);
        unpack(a.dwA, a.dwS, dw);
    }
    else
    {
        uw = UW[o];
        dw = DW[o];
    }
    const int lc = LC[o];
    const double2 csL = cs[o];
    const double chiL = csL.x, SL = csL.y;
    double e;
    if (lc < 0)
    {
        const F2Loc lu = f2_loc(uw, Nx);
        double chiUw, Suw;
        f2_at2(lu, cs, chiUw, Suw);
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
            const F2Loc ld = f2_loc(dw, Nx);
            double chiDw, Sdw;
            f2_at2(ld, cs, chiDw, Sdw);
            f2_besser(uw.distance, dw.distance, chiUw, chiL, chiDw, Suw, SL, Sdw, e, src, psi, chiC);
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
            c1 = e * wP;
            idx = (zOther == kPrev) ? x : -(zOther * Nx + x) - 1;
        }
        else
            f2_hit_weights(uw, Nx, kPrev, e, c1, c2, idx);
    }
    else
    {
        double chiC, Ecum, acc;
        if constexpr (LC_INLINE)
            f2_long_char(a, lc, cs, chiL, SL, Ecum, acc, chiC);
        else
        {
            const double* r = a.lcBuf + ((size_t)laIdx * a.NlongChar + lc) * 3;
            Ecum = r[0];
            acc = r[1];
            chiC = r[2];
        }
        const lwhip_intersection st = a.substeps[a.substepOff[lc + 1] - 1]; // = the upwind hit
        const F2Loc ls = f2_loc(st, Nx);
        double chiUw, Suw;
        f2_at2(ls, cs, chiUw, Suw);
        if (last || dw.distance == 0.0)
        {
            const double dtau = 0.5 * (chiUw + chiL) * (last ? uw.distance : st.distance);
            double chiUwEnd, SuwEnd = Suw;
            if (last)
                f2_at2(f2_loc(uw, Nx), cs, chiUwEnd, SuwEnd);
            f2_linear(dtau, SuwEnd, SL, e, src, psi);
        }
        else
        {
            // :1060-1075: the chi control point of the last sub-step is re-used, not recomputed
            const F2Loc ld = f2_loc(dw, Nx);
            double chiDw, Sdw;
            f2_at2(ld, cs, chiDw, Sdw);
            const double dtauUw = (1.0 / 3.0) * (chiUw + chiL + chiC) * st.distance;
            const double dtauDw = (0.5) * (chiL + chiDw) * dw.distance;
            const double SC = f2_control_point(dtauUw, dtauDw, d_rcp(dtauUw), d_rcp(dtauDw), d_rcp(dtauUw + dtauDw), Suw, SL, Sdw);
            double M, O, Cc;
            f2_coeffs(dtauUw, M, O, Cc, e);
            src = M * Suw + O * SL + Cc * SC;
            psi = O + Cc;
        }
        // I = e (Ecum I(far end) + acc) + src; the far end lies on a z plane (checked on the host)
        f2_hit_weights(a.substeps[a.substepOff[lc]], Nx, kPrev, e * Ecum, c1, c2, idx);
        src = e * acc + src;
    }
    // A (weight of this plane's previous column: hits on an x face) and c2 (second point of a hit on a z plane) are
    // never both non-zero: one plane holds whichever applies, bit 0 of the index word says which
    const bool faceHit = (lc < 0) && uw.axis == 2;
    cf[0] = faceHit ? A : c2;
    cf[(size_t)Ns] = c1;
    cf[(size_t)2 * Ns] = src;
    *ci = idx * 2 + (faceHit ? 1 : 0);
    Ps[o] = psi * d_rcp(chiL);
}

// Pass 0: the long characteristics, one thread per (wavelength, characteristic).  Inside pass 1 each of them sits alone
// in a wavefront of short steps (one lane walking 2 ... 16 BESSER sub-steps while 63 wait: 12 % of the wavefronts of
// the 256 x 82 grid took ~5 times as long as the rest); here they are packed 64 to a wavefront.
__global__ void __launch_bounds__(64) fs2d_longchar_kernel(const Fs2dArgs a)
{
    dbg_poison_lds();
    const int lc = blockIdx.x * blockDim.x + threadIdx.x;
    if (lc >= a.NlongChar)
        return;
    const int laIdx = blockIdx.y;
    const int ray = a.lcOwner[2 * lc], o = a.lcOwner[2 * lc + 1];
    const int rayIdx = a.lcUpOnly ? ((ray & 1) ? (ray >> 1) : -1) : ray;
    if (rayIdx < 0 || rayIdx >= a.nRayCycle)
        return;
    const size_t Ns = (size_t)a.Nx * a.Nz;
    const int solve = laIdx * a.nRayCycle + rayIdx;
    const double2* cs = a.cs + (size_t)solve * Ns;
    const double2 csO = cs[o];
    double Ecum, acc, chiC;
    f2_long_char(a, lc, cs, csO.x, csO.y, Ecum, acc, chiC);
    double* r = a.lcBuf + ((size_t)laIdx * a.NlongChar + lc) * 3;
    r[0] = Ecum;
    r[1] = acc;
    r[2] = chiC;
}

// Pass 2, one wavefront per solve: the planes in sequence, each lane D consecutive columns of the processing
// order.  A plane is: previous-plane intensities from LDS, the lane's D maps composed, one DPP scan over the lanes,
// the D intensities written.  No workgroup barrier (a one-wavefront group's barrier is a wait on its own LDS
// traffic), coefficients of the next plane prefetched while this one is scanned.
template <int D, bool FULL>
__global__ void __launch_bounds__(64) fs2d_scan_kernel(const Fs2dArgs a)
{
    dbg_poison_lds();
    extern __shared__ double prevI[]; // [Nx + 1], by x
    const int Nx = a.Nx, Nz = a.Nz;
    const size_t Ns = (size_t)Nx * Nz;
    const int solve = blockIdx.x;
    const int ray = CTAB(int32_t, a.rays)[solve % a.nRayCycle];
    const int mu = ray >> 1;
    const bool toObs = (ray & 1) != 0;
    const double mux = toObs ? CTAB(double, a.mux)[mu] : -CTAB(double, a.mux)[mu];
    const int dk = toObs ? -1 : 1;
    const int kStart = toObs ? Nz - 1 : 0;
    const bool rev = mux < 0;
    const int lane = threadIdx.x;
    const int p0 = lane * D;
    const double* cf = a.coef + (size_t)solve * 3 * Ns;
    const int32_t* ci = a.cidx + (size_t)solve * Ns;
    double* I = a.I + (size_t)solve * Ns;

    // a plane's maps of the lane.  FULL (Nx = 64 D): no guards, the lane's D values of each array are contiguous and
    // leave as wide loads
    struct Maps
    {
        double A[D], C1[D], C2[D], S[D];
        int ix[D];
    };
    auto load = [&](int k, Maps& m) {
        const size_t row = (size_t)k * Nx;
#pragma unroll
        for (int d = 0; d < D; ++d)
        {
            const int p = p0 + d;
            const bool on = FULL || p < Nx;
            const size_t q = row + (on ? p : 0);
            const double X = on ? cf[q] : 1.0;     // identity beyond the row
            const int w = on ? ci[q] : 1;
            m.A[d] = (w & 1) ? X : 0.0;
            m.C2[d] = (w & 1) ? 0.0 : X;
            m.C1[d] = on ? cf[Ns + q] : 0.0;
            m.S[d] = on ? cf[2 * Ns + q] : 0.0;
            m.ix[d] = w >> 1;
        }
    };
    // two planes ahead: a wavefront is alone on its SIMD, the loads of a plane take longer than its arithmetic
    Maps cur, nx1, nx2;
    load(kStart, cur);
    if (Nz > 1)
        load(kStart + dk, nx1);
    if (lane == 0)
        prevI[Nx] = 0.0;
    for (int step = 0; step < Nz; ++step)
    {
        const int k = kStart + step * dk;
        if (step + 2 < Nz)
            load(k + 2 * dk, nx2);
        // the lane's maps, with the previous plane's intensities folded into the offsets
        double b[D];
        double E = 1.0, B = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d)
        {
            double v = cur.S[d];
            if (step > 0 && (FULL || p0 + d < Nx))
            {
                const int idx = cur.ix[d];
                if (idx >= 0)
                {
                    v = fma(cur.C1[d], prevI[idx], v);
                    if (cur.C2[d] != 0.0)
                        v = fma(cur.C2[d], prevI[idx + 1], v);
                }
                else
                {
                    const size_t f = (size_t)(-(idx + 1));
                    v = fma(cur.C1[d], I[f], v);
                    if (cur.C2[d] != 0.0)
                        v = fma(cur.C2[d], I[f + 1], v);
                }
            }
            b[d] = v;
            B = fma(cur.A[d], B, v);
            E = cur.A[d] * E;
        }
        affine_scan_wave(E, B);
        // intensity entering the lane = the composed offset of the lane before (the row's first map has A = 0)
        double Iin = __shfl_up(B, 1);
        if (lane == 0)
            Iin = 0.0;
        __syncthreads(); // every lane has read the previous plane
        const size_t row = (size_t)k * Nx;
#pragma unroll
        for (int d = 0; d < D; ++d)
        {
            const int p = p0 + d;
            Iin = fma(cur.A[d], Iin, b[d]);
            if (FULL || p < Nx)
            {
                const int j = rev ? Nx - 1 - p : p;
                prevI[j] = Iin;
                I[row + j] = Iin;
            }
        }
        __syncthreads();
        cur = nx1;
        nx1 = nx2;
    }
}
}

template <int D> static void launch_scan(const Fs2dArgs& a, int nSolve, size_t lds, hipStream_t stream)
{
    if (a.Nx == 64 * D)
        LWHIP_LAUNCH((fs2d_scan_kernel<D, true>), dim3(nSolve), dim3(64), lds, stream, a);
    else
        LWHIP_LAUNCH((fs2d_scan_kernel<D, false>), dim3(nSolve), dim3(64), lds, stream, a);
}

bool fs2d_records_packed(const lwhip_intersection* rec, size_t n, int Nx, int Nz, std::vector<double>& s, std::vector<int32_t>& w)
{
    const size_t Ns = (size_t)Nx * Nz;
    if (Ns >= ((size_t)1 << 26))
        return false;
    s.assign(2 * n, 0.0);
    w.assign(n, 0);
    for (size_t i = 0; i < n; ++i)
    {
        const lwhip_intersection& r = rec[i];
        const int k = (int)((i % Ns) / Nx);
        const int z = (int)r.fracZ, x = (int)r.fracX;
        double frac = 0.0;
        if (r.axis == 1)
            frac = r.fracX - x;
        else if (r.axis == 2)
            frac = r.fracZ - z;
        const int dz = z - k + 1;
        // everything the packing drops must be recoverable: integral coordinates off the interpolation axis, a hit
        // within one plane of the point (records never read -- boundary planes -- may hold anything: they only have to
        // survive the round trip when they do fit)
        const bool fits = r.axis >= 0 && r.axis <= 2 && dz >= 0 && dz <= 2 && z >= 0 && z < Nz && x >= 0 && x < Nx
                          && (double)z + (r.axis == 2 ? frac : 0.0) == r.fracZ && (double)x + (r.axis == 1 ? frac : 0.0) == r.fracX;
        if (!fits)
        {
            if (r.distance == 0.0 && r.axis == 0) // the "no hit" record of a boundary point: never interpolated
            {
                w[i] = (int32_t)((size_t)k * Nx) | (1 << 28);
                continue;
            }
            return false;
        }
        w[i] = (int32_t)((size_t)z * Nx + x) | (r.axis << 26) | (dz << 28);
        s[i] = frac;
        s[n + i] = r.distance;
    }
    return true;
}

hipError_t launch_fs2d(const Fs2dArgs& a, int nSolve, hipStream_t stream)
{
    const int Ns = a.Nx * a.Nz;
    if (a.nSolve != nSolve || nSolve % a.nRayCycle != 0)
        return hipErrorInvalidValue;
    if (a.rNx != 1.0f / (float)a.Nx || Ns >= (1 << 23))
        return hipErrorInvalidValue;
    const dim3 grid1(nSolve / a.nRayCycle, a.nRayCycle, (Ns + 255) / 256);
    if (a.lcBuf && a.NlongChar > 0)
    {
        LWHIP_LAUNCH(fs2d_longchar_kernel, dim3((a.NlongChar + 63) / 64, nSolve / a.nRayCycle), dim3(64), 0, stream, a);
        LWHIP_LAUNCH(fs2d_coef_kernel<false>, grid1, dim3(256), 0, stream, a);
    }
    else
        LWHIP_LAUNCH(fs2d_coef_kernel<true>, grid1, dim3(256), 0, stream, a);
    const size_t lds = ((size_t)a.Nx + 1) * sizeof(double);
    const int D = (a.Nx + 63) / 64;
    if (D <= 1)
        launch_scan<1>(a, nSolve, lds, stream);
    else if (D <= 2)
        launch_scan<2>(a, nSolve, lds, stream);
    else if (D <= 4)
        launch_scan<4>(a, nSolve, lds, stream);
    else if (D <= 8)
        launch_scan<8>(a, nSolve, lds, stream);
    else if (D <= 16)
        launch_scan<16>(a, nSolve, lds, stream);
    else
        return hipErrorInvalidValue; // Nx <= 1024 is checked at the boundary
    return hipGetLastError();
}

// the far end of every long characteristic must lie on a z plane (axis X or none): pass 1 relies on it
bool fs2d_long_chars_ok(const lwhip_grid2d* g)
{
    for (int lc = 0; lc < g->NlongChar; ++lc)
        if (g->substeps[g->substepOff[lc]].axis == 2)
            return false;
    return true;
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
    a.rNx = 1.0f / (float)g->Nx;
    a.Nz = g->Nz;
    a.Nrays = g->Nrays;
    a.zLowerBc = g->zLowerBc;
    a.zUpperBc = g->zUpperBc;
    a.wav = wavelength;
    a.wavs = nullptr;
    a.nRayCycle = nSolve;
    a.nSolve = nSolve;
    F2_TRY(up(g->mux, g->Nrays * sizeof(double), (void**)&a.mux));
    F2_TRY(up(g->temperature, Ns * sizeof(double), (void**)&a.temperature));
    F2_TRY(up(g->uw, nSt * sizeof(lwhip_intersection), (void**)&a.uw));
    F2_TRY(up(g->dw, nSt * sizeof(lwhip_intersection), (void**)&a.dw));
    F2_TRY(up(g->longCharIdx, nSt * sizeof(int32_t), (void**)&a.longCharIdx));
    F2_TRY(up(g->substepOff, ((size_t)g->NlongChar + 1) * sizeof(int32_t), (void**)&a.substepOff));
    F2_TRY(up(g->substeps, nSub * sizeof(lwhip_intersection), (void**)&a.substeps));
    a.periodic = g->periodic ? 1 : 0;
    a.la0 = 0; // the primitive has one wavelength: row 0 of the boundary data
    if (!g->periodic)
    {
        a.xNmuLow = g->xLowerBc->Nmu;
        a.xNmuUp = g->xUpperBc->Nmu;
        F2_TRY(up(g->xLowerBc->bcData, (size_t)a.xNmuLow * g->Nz * sizeof(double), (void**)&a.xbcLow));
        F2_TRY(up(g->xUpperBc->bcData, (size_t)a.xNmuUp * g->Nz * sizeof(double), (void**)&a.xbcUp));
        F2_TRY(up(g->xLowerBc->idxs, (size_t)g->Nrays * 2 * sizeof(int32_t), (void**)&a.xIdxLow));
        F2_TRY(up(g->xUpperBc->idxs, (size_t)g->Nrays * 2 * sizeof(int32_t), (void**)&a.xIdxUp));
    }
    F2_TRY(up(rays, (size_t)nSolve * sizeof(int32_t), (void**)&a.rays));
    {
        std::vector<double> pairs((size_t)nSolve * Ns * 2); // the kernels read (chi, S) pairs
        for (size_t i = 0; i < (size_t)nSolve * Ns; ++i)
        {
            pairs[2 * i] = chi[i];
            pairs[2 * i + 1] = S[i];
        }
        F2_TRY(up(pairs.data(), pairs.size() * sizeof(double), (void**)&a.cs));
    }
    F2_TRY(hipMalloc((void**)&a.I, (size_t)nSolve * Ns * sizeof(double)));
    allocs.push_back(a.I);
    F2_TRY(hipMalloc((void**)&a.PsiStar, (size_t)nSolve * Ns * sizeof(double)));
    allocs.push_back(a.PsiStar);
    F2_TRY(hipMalloc((void**)&a.coef, (size_t)nSolve * 3 * Ns * sizeof(double)));
    allocs.push_back(a.coef);
    F2_TRY(hipMalloc((void**)&a.cidx, (size_t)nSolve * Ns * sizeof(int32_t)));
    allocs.push_back(a.cidx);
    F2_TRY(launch_fs2d(a, nSolve, nullptr));
    F2_TRY(hipDeviceSynchronize());
    F2_TRY(hipMemcpy(I, a.I, (size_t)nSolve * Ns * sizeof(double), hipMemcpyDeviceToHost));
    F2_TRY(hipMemcpy(PsiStar, a.PsiStar, (size_t)nSolve * Ns * sizeof(double), hipMemcpyDeviceToHost));
    for (void* q : allocs)
        (void)hipFree(q);
    return LWHIP_OK;
}

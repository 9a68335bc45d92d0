// lwhip_voigt.hip -- line profiles on the device: phi = H(a, v) / (sqrt(pi) vBroad) and
// wphi = 1 / sum phi w_lambda w_mu/2 (Transition::compute_phi_la / compute_wphi,
// Source/FormalScalar.cpp:28-51,106-134; voigt_H Source/LwMisc.hpp:13-19).
//
// H(a, v) = Re w(v + i a) is evaluated with the algorithm of the Faddeeva package the reference
// vendors (Source/Faddeeva.cc:680-972, S. G. Johnson, MIT): Laplace continued fraction for large
// |z|, Zaghloul & Ali's Algorithm 916 sums elsewhere.  exp(-a^2 n^2) comes from a table filled on
// the host; erfcx is the device math library's.
/* The Faddeeva-function algorithm restated in this file is that of the Faddeeva package
 * (http://ab-initio.mit.edu/Faddeeva), whose licence asks that its notice accompany substantial portions:
 *
 * Copyright (c) 2012 Massachusetts Institute of Technology
 *
 * Permission is hereby granted, free of charge, to any person obtaining
 * a copy of this software and associated documentation files (the
 * "Software"), to deal in the Software without restriction, including
 * without limitation the rights to use, copy, modify, merge, publish,
 * distribute, sublicense, and/or sell copies of the Software, and to
 * permit persons to whom the Software is furnished to do so, subject to
 * the following conditions:
 *
 * The above copyright notice and this permission notice shall be
 * included in all copies or substantial portions of the Software.
 *
 * THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND,
 * EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE WARRANTIES OF
 * MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND
 * NONINFRINGEMENT. IN NO EVENT SHALL THE AUTHORS OR COPYRIGHT HOLDERS BE
 * LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION
 * OF CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION
 * WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
 */
#include "lwhip_internal.h"
#include "lwhip_device.h"
#include "../../include/lwhip.h"

#include <hip/hip_runtime.h>
#include <atomic>

#include <cmath>

namespace lwhip
{
__constant__ double c_expa2n2[64];

__device__ __forceinline__ double v_sinc(double x, double sinx)
{
    return fabs(x) < 1e-4 ? 1 - (0.1666666666666666666667) * x * x : sinx / x;
}

__device__ __forceinline__ double v_sinh_taylor(double x)
{
    return x * (1 + (x * x) * (0.1666666666666666666667 + 0.00833333333333333333333 * (x * x)));
}

__device__ double d_voigt_H(double av, double v)
{
    const double a = 0.518321480430085929872;
    const double c = 0.329973702884629072537;
    const double a2 = 0.268657157075235951582;
    const double relerr = 2.2204460492503131e-16;
    const double x = fabs(v);
    const double y = av, ya = fabs(av);
    if (v == 0.0)
        return erfcx(y);
    if (y == 0.0)
        return exp(-x * x);

    double ret = 0.0;
    double sum1 = 0, sum2 = 0, sum3 = 0, sum5 = 0;
    if (ya > 7 || (x > 6 && (ya > 0.1 || (x > 8 && ya > 1e-10) || x > 28)))
    {
        const double ispi = 0.56418958354775628694807945156;
        const double xs = v;
        if (x + ya > 4000)
        {
            if (x + ya > 1e7)
            {
                if (x > ya)
                {
                    const double yax = ya / xs;
                    const double denom = ispi / (xs + yax * ya);
                    return denom * yax;
                }
                const double xya = xs / ya;
                return ispi / (xya * xs + ya);
            }
            const double dr = xs * xs - ya * ya - 0.5, di = 2 * xs * ya;
            const double denom = ispi / (dr * dr + di * di);
            return denom * (xs * di - ya * dr);
        }
        const double c0 = 3.9, c1 = 11.398, c2 = 0.08254, c3 = 0.1421, c4 = 0.2023;
        double nu = floor(c0 + c1 / (c2 * x + c3 * ya + c4));
        double wr = xs, wi = ya;
        for (nu = 0.5 * (nu - 1); nu > 0.4; nu -= 0.5)
        {
            const double denom = nu / (wr * wr + wi * wi);
            wr = xs - wr * denom;
            wi = ya + wi * denom;
        }
        const double denom = ispi / (wr * wr + wi * wi);
        return denom * wi;
    }
    else if (x < 10)
    {
        double prod2ax = 1, prodm2ax = 1;
        double expx2;
        if (x < 5e-4)
        {
            const double x2 = x * x;
            expx2 = 1 - x2 * (1 - 0.5 * x2);
            const double ax2 = 1.036642960860171859744 * x;
            const double exp2ax = 1 + ax2 * (1 + ax2 * (0.5 + 0.166666666666666666667 * ax2));
            const double expm2ax = 1 - ax2 * (1 - ax2 * (0.5 - 0.166666666666666666667 * ax2));
            for (int n = 1; n < 60; ++n)
            {
                const double coef = c_expa2n2[n - 1] * expx2 / (a2 * (n * n) + y * y);
                prod2ax *= exp2ax;
                prodm2ax *= expm2ax;
                sum1 += coef;
                sum2 += coef * prodm2ax;
                sum3 += coef * prod2ax;
                sum5 += coef * (2 * a) * n * v_sinh_taylor((2 * a) * n * x);
                if (coef * prod2ax < relerr * sum3)
                    break;
            }
        }
        else
        {
            expx2 = exp(-x * x);
            const double exp2ax = exp((2 * a) * x), expm2ax = 1 / exp2ax;
            for (int n = 1; n < 60; ++n)
            {
                const double coef = c_expa2n2[n - 1] * expx2 / (a2 * (n * n) + y * y);
                prod2ax *= exp2ax;
                prodm2ax *= expm2ax;
                sum1 += coef;
                sum2 += coef * prodm2ax;
                sum3 += coef * prod2ax;
                sum5 += (coef * prod2ax) * (a * n);
                if ((coef * prod2ax) * (a * n) < relerr * sum5)
                    break;
            }
        }
        const double expx2erfcxy = expx2 * erfcx(y);
        if (y > 5)
        {
            const double sinxy = sin(x * y);
            ret = (expx2erfcxy - c * y * sum1) * cos(2 * x * y) + (c * x * expx2) * sinxy * v_sinc(x * y, sinxy);
        }
        else
        {
            const double xs = v;
            const double sinxy = sin(xs * y);
            const double cos2xy = cos(2 * xs * y);
            const double coef1 = expx2erfcxy - c * y * sum1;
            const double coef2 = c * xs * expx2;
            ret = coef1 * cos2xy + coef2 * sinxy * v_sinc(xs * y, sinxy);
        }
    }
    else
    {
        ret = exp(-x * x);
        const double n0 = floor(x / a + 0.5);
        const double dx = a * n0 - x;
        sum3 = exp(-dx * dx) / (a2 * (n0 * n0) + y * y);
        sum5 = a * n0 * sum3;
        const double exp1 = exp(4 * a * dx);
        double exp1dn = 1;
        int dn;
        bool done = false;
        for (dn = 1; n0 - dn > 0; ++dn)
        {
            const double np = n0 + dn, nm = n0 - dn;
            double tp = exp(-(a * dn + dx) * (a * dn + dx));
            double tm = tp * (exp1dn *= exp1);
            tp /= (a2 * (np * np) + y * y);
            tm /= (a2 * (nm * nm) + y * y);
            sum3 += tp + tm;
            sum5 += a * (np * tp + nm * tm);
            if (a * (np * tp + nm * tm) < relerr * sum5)
            {
                done = true;
                break;
            }
        }
        while (!done)
        {
            const double np = n0 + dn++;
            const double tp = exp(-(a * dn + dx) * (a * dn + dx)) / (a2 * (np * np) + y * y);
            sum3 += tp;
            sum5 += a * np * tp;
            if (a * np * tp < relerr * sum5)
                break;
        }
    }
    return ret + (0.5 * c) * y * (sum2 + sum3);
}



// Both kernels take either one line's arguments by value or a device list of lines (every line of a context, or of all
// the columns of a batch whose atmosphere changed) indexed by the outermost grid dimension: a launch pair per line left
// the chip empty (11 workgroups) and cost a 512-column batch 7 680 launch pairs per atmosphere update.
#define LINE_ARGS(dimIdx) (list ? ld_c(CTAB(VoigtLineArgs, list) + (dimIdx)) : a0)

// phi[l, mu, dir, k] for the shard's rows of one line
__global__ void voigt_phi_kernel(const VoigtLineArgs a0, const VoigtLineArgs* __restrict__ list)
{
    dbg_poison_lds();
    const VoigtLineArgs a = LINE_ARGS(blockIdx.y);
    const double sqrtPi = 1.772453850905516027298167483341145182798;
    const size_t n = (size_t)a.nlt * a.Nrays * 2 * a.Ns;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n;
         idx += (size_t)gridDim.x * blockDim.x)
    {
        const int k = (int)(idx % a.Ns);
        size_t r = idx / a.Ns;
        const int d = (int)(r % 2);
        r /= 2;
        const int mu = (int)(r % a.Nrays);
        const int l = (int)(r / a.Nrays);
        const double vBase = (a.wave[a.ltStart + l] - a.lambda0) * CLight / a.lambda0;
        const double s = d ? 1.0 : -1.0;
        const double vb = a.vBroad[k];
        const double vk = (vBase + s * a.vlosMu[(size_t)mu * a.Ns + k]) / vb;
        a.phi[idx] = d_voigt_H(a.aDamp[k], vk) / (sqrtPi * vb);
    }
}

// wphi[k] = 1 / sum_{la, mu, dir} phi w_lambda(la) w_mu/2 over the line's FULL grid (independent of the wavelength
// shard).  The order of the additions is a function of the LINE alone, so that the same line gets the same bits from
// lwhip_compute_profiles, from a single-line launch and from lwhip_batch_compute_profiles: the grid is cut into
// C = clamp(ceil(nltFull / 2P), 1, wphiSlices) canonical slices of P = 1024 / KT interleaved parts (part q of slice y sums
// la = y P + q, + P C, ...), a slice's parts are added in part order and the slices in slice order -- whatever the launch
// looks like: a workgroup (x, y) owns KT consecutive points (all of them in 1D; tiles of a 2D domain) and the canonical
// slices y, y + gridDim.y, ...; with one workgroup per point tile (gridDim.y = 1: large launches, where a ticket per
// workgroup costs more than it saves) the slices never leave the workgroup, otherwise they meet in scratch[slice][k] and
// the last workgroup of a point tile to arrive (ticket) adds them.  Where the context holds the whole line (no wavelength
// shard) the profile just written by voigt_phi_kernel is read back instead of being evaluated a second time (the same
// bits: same function, same arguments).
__global__ void voigt_wphi_kernel(const VoigtLineArgs a0, const VoigtLineArgs* __restrict__ list, int KT)
{
    dbg_poison_lds();
    extern __shared__ double sPart[];
    __shared__ int sLast;
    const VoigtLineArgs a = LINE_ARGS(blockIdx.z);
    const double sqrtPi = 1.772453850905516027298167483341145182798;
    const int P = blockDim.x / KT;
    const int part = threadIdx.x / KT;
    const int kl = threadIdx.x - part * KT;
    const int k = blockIdx.x * KT + kl;
    const int PB = gridDim.y;
    const int C = max(1, min(max(a.wphiSlices, 1), (a.nltFull + 2 * P - 1) / (2 * P))); // the line's canonical slices
    const bool havePhi = a.nlt == a.nltFull && a.ltStart == 0;
    double total = 0.0;
    for (int y = blockIdx.y; y < C; y += PB)
    {
        double acc = 0.0;
        if (part < P && k < a.Ns)
        {
            const double vb = a.vBroad[k];
            const double ad = a.aDamp[k];
            for (int la = y * P + part; la < a.nltFull; la += P * C)
            {
                const double vBase = (a.wave[la] - a.lambda0) * CLight / a.lambda0;
                const double wla = a.wlam[la];
                for (int mu = 0; mu < a.Nrays; ++mu)
                {
                    const double wlamu = wla * 0.5 * a.wmu[mu];
                    const double vl = a.vlosMu[(size_t)mu * a.Ns + k];
                    for (int d = 0; d < 2; ++d)
                    {
                        double ph;
                        if (havePhi)
                            ph = a.phi[(((size_t)la * a.Nrays + mu) * 2 + d) * a.Ns + k];
                        else
                        {
                            const double vk = (vBase + (d ? vl : -vl)) / vb;
                            ph = d_voigt_H(ad, vk) / (sqrtPi * vb);
                        }
                        acc += ph * wlamu;
                    }
                }
            }
        }
        __syncthreads(); // (the previous slice's parts have been read)
        if (part < P)
            sPart[part * KT + kl] = acc;
        __syncthreads();
        if (threadIdx.x < KT && k < a.Ns)
        {
            double s = 0.0;
            for (int q = 0; q < P; ++q)
                s += sPart[q * KT + threadIdx.x];
            if (PB == 1)
                total += s; // slice order
            else
                a.wphiScratch[(size_t)y * a.Ns + k] = s;
        }
    }
    if (PB == 1)
    {
        if (threadIdx.x < KT && k < a.Ns)
            a.wphi[k] = 1.0 / total;
        return;
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const int old = atomicAdd(a.wphiTicket + blockIdx.x, 1);
        sLast = old == PB - 1;
        if (sLast)
            a.wphiTicket[blockIdx.x] = 0; // ready for the next launch on this stream
    }
    __syncthreads();
    if (!sLast)
        return;
    __threadfence();
    if (threadIdx.x < KT && k < a.Ns)
    {
        double s = 0.0;
        for (int y = 0; y < C; ++y)
            s += __builtin_nontemporal_load(a.wphiScratch + (size_t)y * a.Ns + k);
        a.wphi[k] = 1.0 / s;
    }
}
#undef LINE_ARGS

hipError_t voigt_init_tables()
{
    // the table is a __constant__ symbol: one copy per device, so the "already filled" flag is per device too
    static std::atomic<bool> done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        dev = -1;
    if (dev >= 0 && done[dev].load())
        return hipSuccess;
    double tab[64];
    const double a2 = 0.268657157075235951582;
    for (int n = 1; n <= 64; ++n)
        tab[n - 1] = std::exp(-a2 * (double)(n * n));
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_expa2n2), tab, sizeof(tab));
    if (e == hipSuccess && dev >= 0)
        done[dev].store(true);
    return e;
}

// The launch shape is common to the lines of a launch: point tiles of KT, P = 1024 / KT wavelength parts per workgroup
// (parts beyond a short line's grid add nothing), PB workgroups per point tile from the longest line; the ORDER of a
// line's additions does not depend on it (voigt_wphi_kernel).
static hipError_t launch_voigt(const VoigtLineArgs& a0, const VoigtLineArgs* list, int nLines, int Ns, size_t maxPhi,
                               int maxNltFull, bool sliced, int slices, hipStream_t stream)
{
    hipError_t e = voigt_init_tables();
    if (e != hipSuccess)
        return e;
    if (nLines <= 0)
        return hipSuccess;
    if (maxPhi > 0)
    {
        const int blocks = (int)std::min<size_t>((maxPhi + 255) / 256, 4096);
        LWHIP_LAUNCH(voigt_phi_kernel, dim3(blocks, nLines), dim3(256), 0, stream, a0, list);
    }
    // tiles of 8 points (1D columns: ~10 point tiles, each with up to 128 partial sums over a slice of the line's
    // wavelengths) or 64 (2D domains, which bring enough tiles); wavelength slices so that a thread sums ~2 wavelengths
    const int KT = Ns <= 1024 ? 8 : 64;
    const int P = 1024 / KT; // (fixed: a line's summation order must not depend on which other lines share the launch)
    // (slices only while the launch is small: the ticket costs every workgroup two device-scope fences, which with
    // the 170 000 workgroups of a 512-column batch was all of the kernel's 150 ms)
    int PB = 1;
    if (sliced && Ns <= 1024 && (size_t)nLines * ((Ns + KT - 1) / KT) <= 1024)
        PB = std::max(1, std::min(slices, (maxNltFull + 2 * P - 1) / (2 * P)));
    const int T = ((P * KT + 63) / 64) * 64;
    LWHIP_LAUNCH(voigt_wphi_kernel, dim3((Ns + KT - 1) / KT, PB, nLines), dim3(T), (size_t)P * KT * sizeof(double), stream,
                       a0, list, KT);
    return hipGetLastError();
}

hipError_t launch_voigt_line(const VoigtLineArgs& a, hipStream_t stream)
{
    return launch_voigt(a, nullptr, 1, a.Ns, (size_t)a.nlt * a.Nrays * 2 * a.Ns, a.nltFull, a.wphiScratch && a.wphiTicket,
                        a.wphiSlices, stream);
}

hipError_t launch_voigt_lines(const VoigtLineArgs* devList, const VoigtLineArgs* hostList, int nLines, hipStream_t stream)
{
    if (nLines <= 0)
        return hipSuccess;
    size_t maxPhi = 0;
    int maxFull = 1;
    bool sliced = true;
    for (int i = 0; i < nLines; ++i)
    {
        const VoigtLineArgs& a = hostList[i];
        if (a.Ns != hostList[0].Ns || a.wphiSlices != hostList[0].wphiSlices)
            return hipErrorInvalidValue;
        maxPhi = std::max(maxPhi, (size_t)a.nlt * a.Nrays * 2 * a.Ns);
        maxFull = std::max(maxFull, a.nltFull);
        sliced = sliced && a.wphiScratch && a.wphiTicket;
    }
    return launch_voigt(hostList[0], devList, nLines, hostList[0].Ns, maxPhi, maxFull, sliced, hostList[0].wphiSlices, stream);
}
}

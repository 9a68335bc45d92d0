// microbenchmark: fp64 issue rate of a lone wave per SIMD (dependent vs independent chains), rcp, exp,
// and cross-lane traffic -- sizing inputs for a thread-per-ray marching sweep.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <initializer_list>
__device__ __forceinline__ double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__global__ void __launch_bounds__(1024) k(double* out, long long* cyc, int iters, int mode)
{
    const int t = threadIdx.x;
    __shared__ double sm[1024];
    __shared__ double sm2[1024];
    sm[t & 1023] = (double)((t * 7 + 1) & 1023);
    sm2[t & 1023] = 0.0;
    double x = 1.0 + t * 1e-3;
    double y[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) y[q] = x + q * 0.1;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i)
    {
        if (mode == 0)
        {
#pragma unroll
            for (int j = 0; j < 64; ++j) x = fma(x, 0.9999, 1e-3);
        }
        else if (mode == 1)
        {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = fma(y[q], 0.9999, 1e-3);
        }
        else if (mode == 2)
        { // 64 raw v_rcp_f64, 8 chains
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = __builtin_amdgcn_rcp(y[q]) + 0.5;
        }
        else if (mode == 3)
        { // 64 refined reciprocals, 8 chains
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = rcp_nr(y[q]) + 0.5;
        }
        else if (mode == 4)
        { // 64 exp
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = exp(-y[q]) + 0.5;
        }
        else if (mode == 5)
        { // 64 IEEE divides
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = 1.0 / y[q] + 0.5;
        }
        else if (mode == 6)
        { // 64 shuffles (xor) of doubles
#pragma unroll
            for (int j = 0; j < 64; ++j) x += __shfl_xor(x, 1 + (j & 31));
        }
        else if (mode == 8)
        { // 2 chains
#pragma unroll
            for (int j = 0; j < 32; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) y[q] = fma(y[q], 0.9999, 1e-3);
        }
        else if (mode == 9)
        { // 4 chains
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) y[q] = fma(y[q], 0.9999, 1e-3);
        }
        else if (mode == 10)
        { // 64 double selects (2 v_cndmask_b32 each), 8 chains
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q)
                {
                    double a = y[q], b = y[(q + 1) & 7];
                    y[q] = (y[(q + 3) & 7] > 1.0) ? a : b;
                }
        }
        else if (mode == 11)
        { // 64 DPP wave_shr:1 of doubles (2 v_mov_dpp each)
#pragma unroll
            for (int j = 0; j < 64; ++j)
            {
                const unsigned long long vv = __double_as_longlong(x);
                const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)vv, 0x138, 0xf, 0xf, false);
                const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(vv >> 32), 0x138, 0xf, 0xf, false);
                x = __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo) + 1.0;
            }
        }
        else if (mode == 12)
        { // 64 ds_bpermute of doubles (2 each), dependent
#pragma unroll
            for (int j = 0; j < 64; ++j)
            {
                const unsigned long long vv = __double_as_longlong(x);
                const int lo = __builtin_amdgcn_ds_bpermute(((t + 1) & 63) << 2, (int)(unsigned)vv);
                const int hi = __builtin_amdgcn_ds_bpermute(((t + 1) & 63) << 2, (int)(unsigned)(vv >> 32));
                x = __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo) + 1.0;
            }
        }
        else if (mode == 13)
        { // 64 dependent LDS reads (pointer chase, b64)
            int idx = t;
#pragma unroll
            for (int j = 0; j < 64; ++j)
            {
                const double v = sm[idx];
                idx = (int)v;
            }
            x += idx;
        }
        else if (mode == 14)
        { // 64 ds_add_f64 without return
#pragma unroll
            for (int j = 0; j < 64; ++j)
                atomicAdd(&sm2[(t + j * 64) & 1023], x);
        }
        else if (mode == 15)
        { // 8 chains: 4 fma + 1 cndmask pair + 1 LDS read each (mixed stream)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q)
                {
                    const double l = sm[(t + q * 64 + j) & 1023];
                    y[q] = fma(y[q], 0.9999, l);
                }
        }
        else if (mode == 7)
        { // 64 int ops dependent
            int a = (int)x;
#pragma unroll
            for (int j = 0; j < 64; ++j) a = a * 3 + j;
            x += a;
        }
    }
    long long t1 = clock64();
    double s = x;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += y[q];
    out[blockIdx.x * blockDim.x + t] = s;
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8);
    const char* names[] = {"dep fma x64", "8 chains fma x64", "raw rcp x64", "rcp+2NR x64", "exp x64", "div x64", "shfl_xor f64 x64", "dep int mad x64", "2 chains fma", "4 chains fma", "select f64 x64", "dpp wave_shr f64", "bpermute f64 dep", "lds read dep", "ds_add_f64", "fma+lds 8 chains"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512, 768, 1024})
        for (int mode = 0; mode < 16; ++mode)
        {
            const int iters = 2000;
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, cyc, 10, mode);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, cyc, iters, mode);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            const double wavesPerSimd = threads / 256.0;
            // ns per op per wave (elapsed), and SIMD-time per wave-op assuming 2.4 GHz
            const double nsPerOp = ms * 1e6 / iters / 64;
            printf("waves/SIMD=%.2f %-18s ticks/op=%.2f  ns/op(elapsed)=%.2f  SIMD clk per wave-op @2.4GHz=%.2f\n", wavesPerSimd, names[mode],
                   s / 256 / iters / 64, nsPerOp, nsPerOp * 2.4 / (wavesPerSimd < 1 ? 1 : wavesPerSimd));
        }
    return 0;
}

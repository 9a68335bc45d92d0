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
    const char* names[] = {"dep fma x64", "8 chains fma x64", "raw rcp x64", "rcp+2NR x64", "exp x64", "div x64", "shfl_xor f64 x64", "dep int mad x64"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {64, 256, 512, 1024})
        for (int mode = 0; mode < 8; ++mode)
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

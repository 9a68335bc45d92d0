// microbenchmark: cost of an LDS exchange + workgroup barrier phase at 13 waves / CU
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) k(double* out, long long* cyc, int iters, int mode)
{
    extern __shared__ double lds[];
    const int t = threadIdx.x;
    double x = t * 0.5;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i)
    {
        lds[t] = x;
        if (mode == 0)
            __syncthreads();
        else
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
        x = x * 0.999 + lds[(t + 1) % blockDim.x];
        if (mode == 2)
        { // some f64 work: ~64 dependent fma
#pragma unroll
            for (int j = 0; j < 64; ++j)
                x = fma(x, 0.9999, 1e-3);
        }
        if (mode == 3)
        { // 8 independent chains x 8
            double y[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) y[q] = x + q;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = fma(y[q], 0.9999, 1e-3);
            x = y[0] + y[1] + y[2] + y[3] + y[4] + y[5] + y[6] + y[7];
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + t] = x;
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8);
    for (int threads : {832, 1024, 448, 256})
        for (int mode = 0; mode < 4; ++mode)
        {
            const int iters = 2000;
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 100 * 1024, 0, out, cyc, iters, mode);
            hipDeviceSynchronize();
            long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            printf("threads=%d mode=%d cycles/iter=%.1f\n", threads, mode, s / 256 / iters);
        }
    return 0;
}

// Does a multi-dword scalar load honour a 4-byte-aligned address on gfx950?  (hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
struct Q { int v[4]; };
template <int OFF> __global__ void k(const int* p, int* out)
{
    const __attribute__((address_space(4))) char* c = (const __attribute__((address_space(4))) char*)p;
    asm volatile("" : "+s"(c));
    const __attribute__((address_space(4))) int* q = (const __attribute__((address_space(4))) int*)(c + OFF);
    const int a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
    if (threadIdx.x == 0)
    {
        out[0] = a0;
        out[1] = a1;
        out[2] = a2;
        out[3] = a3;
    }
}
int main()
{
    int h[64];
    for (int i = 0; i < 64; ++i)
        h[i] = i;
    int *d, *o;
    hipMalloc(&d, sizeof(h));
    hipMalloc(&o, 16);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    int r[4];
#define RUN(OFF)                                                                                  \
    hipLaunchKernelGGL(k<OFF>, dim3(1), dim3(64), 0, 0, d, o);                                     \
    hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);                                                    \
    printf("offset %3d: got %d %d %d %d (want %d..)\n", OFF, r[0], r[1], r[2], r[3], OFF / 4);
    RUN(0) RUN(4) RUN(8) RUN(16) RUN(52) RUN(60) RUN(48)
    return 0;
}

#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned *out)
{
    __shared__ float pad[16000]; // 64 KB -> 2 blocks per CU
    pad[threadIdx.x] = 1.f;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // spin a bit so that blocks co-reside
    float x = pad[threadIdx.x];
    for (int i = 0; i < 20000; ++i) x = x * 1.0001f + 0.5f;
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = (unsigned)x; }
}
int main()
{
    const int blocks = 1024;
    unsigned *d; hipMalloc(&d, blocks * 8 * 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d);
    static unsigned h[1024 * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[16] = {0};
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) hist[h[(b * 4 + w) * 2] & 15]++;
    printf("wave_id histogram:"); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("\n");
    for (int b = 0; b < 4; ++b) for (int w = 0; w < 4; ++w) { unsigned v = h[(b * 4 + w) * 2]; printf("block %d wave %d: hw=%08x wave_id %u simd %u cu %u\n", b, w, v, v & 15, (v >> 4) & 3, (v >> 8) & 15); }
    return 0;
}

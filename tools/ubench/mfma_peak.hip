// Microbenchmark: sustained v_mfma_f32_16x16x4_f32 rate per CU as a function of waves per SIMD and of
// LDS operand traffic (the igemm inner loop without any global memory). Prints TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ f32x4 sm[2048];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x)
        sm[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        a[i] = sm[lane + 64 * i], b[i] = sm[1024 + lane + 64 * i];
    for (int it = 0; it < iters; ++it)
    {
        if (LDS)
        {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = sm[(lane + 64 * i + it) & 1023], b[i] = sm[1024 + ((lane + 64 * i + it) & 1023)];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[(i * 4 + j) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][c], a[i][c], acc[(i * 4 + j) % NACC], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i)
        s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC, bool LDS>
static void run(int wavesPerSimd, const char *name)
{
    const int iters = 20000;
    const int threads = 256; // 4 waves: one per SIMD
    const int blocks = 256 * wavesPerSimd;
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(threads), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 64 * 2048.0; // 64 MFMAs x 2048 flop per wave-iteration
    printf("%-28s waves/SIMD %d : %7.1f TFLOP/s (%.2f ms)\n", name, wavesPerSimd, flops / ms / 1e9, ms);
    hipFree(out);
}

int main()
{
    for (int w = 1; w <= 4; ++w)
        run<16, false>(w, "regs only, 16 acc");
    for (int w = 1; w <= 4; ++w)
        run<16, true>(w, "8 ds_read_b128 / 64 mfma");
    for (int w = 1; w <= 2; ++w)
        run<4, false>(w, "regs only, 4 acc");
    return 0;
}

// lds_overlap.hip — does a workgroup's LDS stay private when kernels with large static LDS allocations from two
// streams share a CU? Victim: fills its LDS (VB bytes) with a pattern using W-byte accesses, spins, verifies, many
// rounds. Aggressor: keeps rewriting its whole LDS (AB bytes) with 16-byte accesses and runs bf16 MFMAs on what it
// reads (the instruction mix of the exact-split kernels). Reports corrupted words seen by the victim.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int BYTES, int W>
__global__ __launch_bounds__(256) void victim(unsigned *errs, int rounds, int spin)
{
    __shared__ __attribute__((aligned(16))) unsigned buf[BYTES / 4];
    const int tid = threadIdx.x;
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r)
    {
        const unsigned tag = 0x5a000000u ^ (blockIdx.x * 131071u) ^ (unsigned)r;
        if (W == 4)
            for (int i = tid; i < BYTES / 4; i += 256)
                buf[i] = tag + i;
        else if (W == 8)
            for (int i = tid; i < BYTES / 8; i += 256)
                *reinterpret_cast<u32x2 *>(&buf[2 * (i ^ ((i >> 4) & 15))]) = u32x2{tag + 2 * i, tag + 2 * i + 1};
        else
            for (int i = tid; i < BYTES / 16; i += 256)
                *reinterpret_cast<u32x4 *>(&buf[4 * i]) = u32x4{tag + 4 * i, tag + 4 * i + 1, tag + 4 * i + 2, tag + 4 * i + 3};
        __syncthreads();
        for (int s = 0; s < spin; ++s)
            __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        if (W == 4)
            for (int i = tid; i < BYTES / 4; i += 256)
                bad += buf[i] != tag + i;
        else if (W == 8)
            for (int i = tid; i < BYTES / 8; i += 256)
            {
                const u32x2 v = *reinterpret_cast<u32x2 *>(&buf[2 * (i ^ ((i >> 4) & 15))]);
                bad += (v[0] != tag + 2 * i) + (v[1] != tag + 2 * i + 1);
            }
        else
            for (int i = tid; i < BYTES / 16; i += 256)
            {
                const u32x4 v = *reinterpret_cast<u32x4 *>(&buf[4 * i]);
                bad += (v[0] != tag + 4 * i) + (v[1] != tag + 4 * i + 1) + (v[2] != tag + 4 * i + 2) + (v[3] != tag + 4 * i + 3);
            }
        __syncthreads();
    }
    if (bad)
        atomicAdd(errs, bad);
}
template <int BYTES>
__global__ __launch_bounds__(256, 2) void aggressor(float *sink, int rounds)
{
    __shared__ u32x4 buf[BYTES / 16];
    const int tid = threadIdx.x;
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j)
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rounds; ++r)
    {
        for (int i = tid; i < BYTES / 16; i += 256)
        {
            // two 8-byte stores per 16-byte slot, like the plane stores of the exact-split kernels (ds_write_b64)
            u32x2 *q = reinterpret_cast<u32x2 *>(&buf[i]);
            q[(tid >> 3) & 1] = u32x2{0x3f803f80u, 0x3f803f80u + r};
            q[1 - ((tid >> 3) & 1)] = u32x2{0x3f803f80u, 0x3f803f80u};
        }
        __syncthreads();
        for (int i = tid; i + 256 < BYTES / 16; i += 512)
        {
            const bf16x8 a = __builtin_bit_cast(bf16x8, buf[i]), b = __builtin_bit_cast(bf16x8, buf[i + 256]);
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
    float t = 0.f;
    for (int j = 0; j < 8; ++j)
        t += acc[j][0] + acc[j][3];
    if (t == 12345.f)
        *sink = t;
}

template <int VB, int W, int AB>
static void run(const char *label)
{
    unsigned *d;
    (void)hipMalloc(&d, 8);
    (void)hipMemset(d, 0, 8);
    hipStream_t s1, s2;
    (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int it = 0; it < 4; ++it)
    {
        hipLaunchKernelGGL(aggressor<AB>, dim3(4096), dim3(256), 0, s2, reinterpret_cast<float *>(d + 1), 60);
        hipLaunchKernelGGL((victim<VB, W>), dim3(2048), dim3(256), 0, s1, d, 30, 30);
    }
    (void)hipStreamSynchronize(s1);
    (void)hipStreamSynchronize(s2);
    unsigned h[2] = {0, 0};
    (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("%s: victim LDS %d B, %d-byte accesses; aggressor LDS %d B (b128 + bf16 MFMA): corrupted words seen = %u\n", label, VB, W, AB, h[0]);
    (void)hipFree(d);
    (void)hipStreamDestroy(s1);
    (void)hipStreamDestroy(s2);
}

int main()
{
    run<65536, 4, 81920>("a");
    run<65536, 8, 81920>("b");
    run<65536, 16, 81920>("c");
    run<65568, 8, 81920>("d");
    run<65568, 8, 73728>("e");
    run<81920, 8, 81920>("f");
    run<81920, 8, 73728>("g");
    run<65568, 8, 65536>("h");
    printf("done\n");
    return 0;
}

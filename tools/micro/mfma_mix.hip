// mfma_mix.hip — what limits a wave64 loop of v_mfma_f32_16x16x32_bf16 on gfx950 when other work shares the SIMD?
// Each kernel runs ITER iterations of 80 MFMAs (8 independent accumulators, the dependency distance of the split GEMM)
// per wave, two workgroups of four waves per CU (two waves per SIMD), with per iteration:
//   NV  extra VALU instructions per MFMA (independent v_fma_f32 chains)      -> does VALU issue compete with the matrix pipe?
//   NL  ds_read_b128 per iteration (conflict-free), consumed by the MFMAs     -> LDS fragment traffic
//   NG  global_load_dwordx4 per iteration (L2-resident 64 MB window)          -> VMEM issue / return
//   BAR one s_barrier per iteration
// Prints fp32-equivalent TFLOP/s of a five-term split (MFMA flops / 5) and the fraction of 2516.6 / 5.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int NL, int NG, int BAR, int WPS, int NW = 0, int DEP = 0, int ND = 0, int BIG = 0, int MPG = 20>
__global__ __launch_bounds__(256, WPS) void mix(float *sink, const u32x4 *src, int iters, size_t window, unsigned strideA, unsigned strideB)
{
    __shared__ u32x4 lds[4096]; // 64 KB: [0, 2048) is read, [2048, 4096) is written
    const int tid = threadIdx.x;
    // bf16 pairs with random signs and mantissas, exponents near 1 (operands that toggle like real data)
    for (int i = tid; i < 2048; i += 256)
    {
        unsigned h = (i * 4u + blockIdx.x * 8191u) * 2654435761u;
        u32x4 t;
#pragma unroll
        for (int c = 0; c < 4; ++c)
        {
            h = h * 1664525u + 1013904223u;
            t[c] = ((h >> 7) & 0x807f807fu) | 0x3f003f00u;
        }
        lds[i] = t;
    }
    __syncthreads();
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 accB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            accB[j][e] = 0.f;
    bf16x8 f[2][8]; // fragment registers: 4 "weight" + 4 "activation" per set, two sets
#pragma unroll
    for (int j = 0; j < 8; ++j)
        f[0][j] = f[1][j] = __builtin_bit_cast(bf16x8, lds[tid + 256 * j]);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        v[j] = (float)tid * 1e-3f + j;
    const float ca = 0.999f, cb = 1e-3f;
    // 32-bit byte offsets into a power-of-two window (scalar base + vector offset: one v_add + one v_and per load)
    const unsigned mask = (unsigned)(window - 1) & ~15u;
    unsigned goff = ((blockIdx.x * 256u + tid) * 16u) & mask;
    u32x4 g[NG > 0 ? NG : 1], gprev[NG > 0 ? NG : 1];
    u32x4 gacc{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < (NG > 0 ? NG : 1); ++j)
        gprev[j] = u32x4{1u, 2u, 3u, 4u};
    for (int it = 0; it < iters; ++it)
    {
        if (ND > 0)
        {
            // direct global -> LDS loads (no VGPR round trip, no ds_write): lane l's 16 bytes land at base + 16 l
#pragma unroll
            for (int j = 0; j < ND; ++j)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const char *>(src) + ((goff + (8 + j) * 1048576u + j * 4096u) & mask),
                                                 (__attribute__((address_space(3))) void *)&lds[2048 + 64 * ((tid >> 6) + 4 * j) + 256 * (it & 1)], 16, 0, 0);
        }
        if (NG > 0)
        {
            if (strideA)
            {
                // the GEMM's pattern: 8 lanes read 128 contiguous bytes of one row, 32 rows per load, rows strideA / strideB bytes
                // apart; an "activation" row tile is shared by the 16 workgroups of a tile row, a "weight" tile by a tile column
                const unsigned rowA = (blockIdx.x >> 4) * 128u + (tid >> 3), rowB = (blockIdx.x & 15u) * 128u + (tid >> 3);
                const unsigned kb = (unsigned)it * 128u + (tid & 7u) * 16u;
#pragma unroll
                for (int j = 0; j < NG; ++j)
                {
                    const unsigned off = j < NG / 2 ? (rowA + 32u * j) * strideA + kb : (1u << 29) + (rowB + 32u * (j - NG / 2)) * strideB + (kb >> 1);
                    g[j] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(src) + (off & mask));
                }
            }
            else
            {
#pragma unroll
                for (int j = 0; j < NG; ++j)
                    g[j] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(src) + ((goff + j * 1048576u + j * 4096u) & mask));
                goff = (goff + 256u * 16u * 4099u) & mask;
            }
        }
        constexpr int LPG = NL > 0 ? NL / 4 : 0; // fragment reads per group of 20 MFMAs (at most 8), one group ahead
        if (NL > 0 && DEP)
        {
            // like the GEMM: the first group's fragments can only be read after the barrier that published the tile
#pragma unroll
            for (int j = 0; j < LPG; ++j)
                f[0][j] = __builtin_bit_cast(bf16x8, lds[(tid + 64 * j + it) & 2047]);
        }
#pragma unroll
        for (int grp = 0; grp < 4; ++grp)
        {
            const int cur = grp & 1, nxt = cur ^ 1;
            if (NW > 0 && grp < 2)
            {
                // staging stores of the next tile (data from the loads of the previous iteration), first half of the iteration
#pragma unroll
                for (int j = 0; j < NW / 2; ++j)
                    lds[2048 + ((tid + 256 * (grp * (NW / 2) + j)) & 2047)] = gprev[(grp * (NW / 2) + j) % (NG > 0 ? NG : 1)];
            }
            if (NL > 0 && !(DEP && grp == 3))
            {
#pragma unroll
                for (int j = 0; j < LPG; ++j)
                    f[nxt][j] = __builtin_bit_cast(bf16x8, lds[(tid + 64 * (grp * LPG + j) + it) & 2047]);
            }
#pragma unroll
            for (int m = 0; m < MPG; ++m) // MPG MFMAs per group, four groups per iteration
            {
                // volatile asm: exactly one MFMA, then its NV VALU companions, in this order, no packing
                if (BIG == 1)
                {
                    // the same flops from half as many instructions: 32x32x16 (32 pipe cycles each), 4 accumulators of 16 registers
                    if (m & 1)
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accB[(m >> 1) & 3]) : "v"(f[cur][m & 3]), "v"(f[cur][4 + ((m >> 2) & 3)]));
                }
                else if (BIG == 2)
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(f[cur][m & 3]), "v"(f[cur][4 + ((m >> 2) & 3)]));
                else
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(f[cur][m & 3]), "v"(f[cur][4 + ((m >> 2) & 3)]));
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(m * NV + k) & 7]) : "v"(ca), "v"(cb));
            }
        }
        if (NG > 0)
        {
#pragma unroll
            for (int j = 0; j < NG; ++j)
            {
                if (NW > 0)
                    gprev[j] = g[j];
                else
                    gacc ^= g[j];
            }
        }
        if (BAR)
            __syncthreads();
    }
    asm volatile("s_nop 15\n s_nop 15");
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3] + v[j] + accB[j & 3][j] + accB[j & 3][15 - j];
    s += (float)(gacc[0] ^ gacc[1] ^ gacc[2] ^ gacc[3]);
    if (s == 123.456f)
        sink[0] = s;
}

static int g_iters = 40000;
static unsigned g_strideA = 0, g_strideB = 0;
template <int NV, int NL, int NG, int BAR, int WPS, int NW = 0, int DEP = 0, int ND = 0, int BIG = 0, int MPG = 20>
static void run(const char *tag, float *sink, const u32x4 *src, size_t window, int cus)
{
    const int iters = g_iters, wgs = cus * WPS;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL((mix<NV, NL, NG, BAR, WPS, NW, DEP, ND, BIG, MPG>), dim3(wgs), dim3(256), 0, 0, sink, src, 200, window, g_strideA, g_strideB);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix<NV, NL, NG, BAR, WPS, NW, DEP, ND, BIG, MPG>), dim3(wgs), dim3(256), 0, 0, sink, src, iters, window, g_strideA, g_strideB);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * 4 * iters * (4 * MPG) * 16384.0;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("[%d MFMAs / iteration, %.3f us per iteration] %s%-44s NV=%d NL=%2d NG=%d BAR=%d NW=%2d DEP=%d ND=%d waves/SIMD=%d : %7.3f ms  %7.1f TFLOP/s bf16 = %5.1f fp32-equivalent (x/5) = %.3f of peak\n", 4 * MPG, ms * 1e3 / iters, BIG == 1 ? "[32x32x16] " : BIG == 2 ? "[f16] " : "", tag, NV, NL, NG, BAR, NW, DEP, ND, WPS, ms,
           tf, tf / 5, tf / 2516.6);
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    printf("%s: %d CUs, clock %d MHz\n", pr.gcnArchName, cus, pr.clockRate / 1000);
    float *sink;
    u32x4 *src;
    const size_t window = 1024ull << 20;
    hipMalloc(&sink, 4096);
    hipMalloc(&src, window);
    {
        std::vector<unsigned> h((size_t)64 << 18); // 64 MB of random bf16-like words, repeated
        unsigned x = 12345u;
        for (auto &w : h)
            x = x * 1664525u + 1013904223u, w = ((x >> 7) & 0x807f807fu) | 0x3f003f00u;
        for (size_t off = 0; off < window; off += h.size() * 4)
            hipMemcpy(reinterpret_cast<char *>(src) + off, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    const char *only = getenv("MIX_SET");
    const int set = only ? atoi(only) : 0;
    if (set == 0)
    {
        run<0, 0, 0, 0, 2>("MFMA only", sink, src, window, cus);
        run<0, 0, 0, 0, 1>("MFMA only, one wave per SIMD", sink, src, window, cus);
        run<1, 0, 0, 0, 2>("+1 VALU per MFMA", sink, src, window, cus);
        run<2, 0, 0, 0, 2>("+2 VALU per MFMA", sink, src, window, cus);
        run<3, 0, 0, 0, 2>("+3 VALU per MFMA", sink, src, window, cus);
        run<4, 0, 0, 0, 2>("+4 VALU per MFMA", sink, src, window, cus);
        run<0, 20, 0, 0, 2>("+20 ds_read_b128 per 80 MFMA", sink, src, window, cus);
        run<0, 0, 0, 1, 2>("+barrier per 80 MFMA", sink, src, window, cus);
    }
    if (set == 0 || set == 1)
    {
        for (size_t win : {(size_t)1 << 20, (size_t)64 << 20, (size_t)1024 << 20})
        {
            char tag[96];
            snprintf(tag, sizeof tag, "loads only, window %zu MB", win >> 20);
            run<0, 0, 4, 0, 2>(tag, sink, src, win, cus);
            run<0, 0, 8, 0, 2>(tag, sink, src, win, cus);
            snprintf(tag, sizeof tag, "VALU + 20 reads + barrier, window %zu MB", win >> 20);
            run<1, 20, 4, 1, 2>(tag, sink, src, win, cus);
            run<1, 20, 8, 1, 2>(tag, sink, src, win, cus);
            run<2, 20, 8, 1, 2>(tag, sink, src, win, cus);
        }
    }
    if (set == 0 || set == 2)
    {
        // towards the real loop: staging stores and the barrier -> fragment read dependence (all long runs)
        const size_t win = (size_t)16 << 20;
        run<0, 0, 0, 0, 2>("warm-up", sink, src, win, cus);
        run<0, 0, 0, 0, 2>("MFMA only", sink, src, win, cus);
        run<1, 0, 0, 0, 2>("1 VALU", sink, src, win, cus);
        run<1, 20, 0, 0, 2>("1 VALU + 20 reads", sink, src, win, cus);
        run<1, 20, 0, 1, 2>("1 VALU + 20 reads + barrier", sink, src, win, cus);
        run<1, 20, 0, 1, 2, 0, 1>("1 VALU + 20 reads after the barrier", sink, src, win, cus);
        run<1, 20, 8, 1, 2, 0, 0>("base: 1 VALU + 20 reads + 8 loads + barrier", sink, src, win, cus);
        run<1, 20, 8, 1, 2, 0, 1>("base, reads only after the barrier", sink, src, win, cus);
        run<1, 20, 8, 1, 2, 4, 1>("base + 4 ds_write_b128, reads after barrier", sink, src, win, cus);
        run<1, 20, 8, 1, 2, 10, 1>("base + 10 ds_write_b128, reads after barrier", sink, src, win, cus);
        run<1, 20, 8, 1, 2, 16, 1>("base + 16 ds_write_b128, reads after barrier", sink, src, win, cus);
        run<1, 16, 8, 1, 2, 4, 1>("16 reads + 4 writes (LIN direct kernel)", sink, src, win, cus);
        run<1, 20, 4, 1, 2, 10, 1>("4 loads, 10 writes, reads after barrier", sink, src, win, cus);
        run<2, 20, 8, 1, 2, 10, 1>("2 VALU, 8 loads, 10 writes, reads after barrier", sink, src, win, cus);
        run<1, 20, 8, 0, 2, 10, 0>("no barrier: 8 loads, 10 writes", sink, src, win, cus);
    }
    if (set == 3)
    {
        // weight planes by direct global -> LDS loads instead of load + ds_write
        const size_t win = (size_t)16 << 20;
        run<0, 0, 0, 0, 2>("warm-up", sink, src, win, cus);
        run<1, 20, 8, 1, 2, 10, 1>("staged GEMM: 8 loads, 10 writes", sink, src, win, cus);
        run<1, 20, 4, 1, 2, 6, 1, 4>("A staged (4 loads, 6 writes) + 4 direct-to-LDS", sink, src, win, cus);
        run<1, 16, 8, 1, 2, 4, 1>("LIN kernel: 8 loads, 4 writes, 16 reads", sink, src, win, cus);
        run<1, 16, 4, 1, 2, 0, 1, 4>("LIN kernel: 4 loads + 4 direct-to-LDS, 16 reads", sink, src, win, cus);
        run<1, 20, 8, 1, 2, 10, 1>("staged GEMM again", sink, src, win, cus);
        run<1, 16, 8, 1, 2, 4, 1>("LIN kernel again", sink, src, win, cus);
    }
    if (set == 4)
    {
        // does the address pattern of the loads matter? (rows a power of two apart vs a pseudo-random walk)
        const size_t win = (size_t)1024 << 20;
        g_iters = 64; // 64 K-tiles = K 2048, then the same rows again (run() repeats the kernel)
        run<0, 0, 0, 0, 2>("warm-up", sink, src, win, cus);
        g_iters = 20000;
        run<1, 16, 8, 1, 2, 4, 1>("LIN kernel, random walk", sink, src, win, cus);
        for (unsigned sa : {2048u, 8192u, 2048u + 128u, 8192u + 128u})
        {
            g_strideA = sa, g_strideB = sa / 2;
            char tag[96];
            snprintf(tag, sizeof tag, "LIN kernel, rows %u / %u bytes apart", g_strideA, g_strideB);
            run<1, 16, 8, 1, 2, 4, 1>(tag, sink, src, win, cus);
        }
        g_strideA = g_strideB = 0;
    }
    if (set == 5)
    {
        // half as many MFMA instructions for the same flops: v_mfma_f32_32x32x16_bf16 against 16x16x32
        const size_t win = (size_t)16 << 20;
        run<0, 0, 0, 0, 2>("warm-up", sink, src, win, cus);
        run<0, 0, 0, 0, 2>("MFMA only", sink, src, win, cus);
        run<0, 0, 0, 0, 2, 0, 0, 0, 1>("MFMA only", sink, src, win, cus);
        run<2, 0, 0, 0, 2>("2 VALU per 16x16x32", sink, src, win, cus);
        run<2, 0, 0, 0, 2, 0, 0, 0, 1>("2 VALU per 16x16x32", sink, src, win, cus);
        run<3, 0, 0, 0, 2>("3 VALU per 16x16x32", sink, src, win, cus);
        run<3, 0, 0, 0, 2, 0, 0, 0, 1>("3 VALU per 16x16x32", sink, src, win, cus);
        run<1, 20, 8, 1, 2, 10, 1>("staged GEMM: 1 VALU, 20 reads, 8 loads, 10 writes", sink, src, win, cus);
        run<1, 20, 8, 1, 2, 10, 1, 0, 1>("staged GEMM: 1 VALU, 20 reads, 8 loads, 10 writes", sink, src, win, cus);
        run<2, 20, 8, 1, 2, 10, 1>("conv GEMM: 2 VALU, 20 reads, 8 loads, 10 writes", sink, src, win, cus);
        run<2, 20, 8, 1, 2, 10, 1, 0, 1>("conv GEMM: 2 VALU, 20 reads, 8 loads, 10 writes", sink, src, win, cus);
        run<1, 16, 8, 1, 2, 4, 1>("LIN kernel: 16 reads, 4 writes", sink, src, win, cus);
        run<1, 16, 8, 1, 2, 4, 1, 0, 1>("LIN kernel: 16 reads, 4 writes", sink, src, win, cus);
    }
    if (set == 6)
    {
        // one 64 x 64 x 32 wave tile-step of the split GEMM per iteration, two ways of forming the fp32 product:
        // bf16 terms (a = 3 terms, w = 2 terms: 80 MFMAs, 20 fragment reads, 8 loads, 10 staging stores, 88 split operations)
        // fp16 terms (a = 3 terms, w = 1 exact term: 48 MFMAs, 16 reads, 6 loads, 8 stores, the same split work)
        const size_t win = (size_t)16 << 20;
        run<0, 0, 0, 0, 2>("warm-up", sink, src, win, cus);
        run<1, 20, 8, 1, 2, 10, 1, 0, 0, 20>("bf16 terms: 5 MFMAs per block", sink, src, win, cus);
        run<2, 16, 6, 1, 2, 8, 1, 0, 0, 12>("fp16 terms: 3 MFMAs per block", sink, src, win, cus);
        run<1, 20, 8, 1, 2, 10, 1, 0, 0, 20>("bf16 terms again", sink, src, win, cus);
        run<2, 16, 6, 1, 2, 8, 1, 0, 0, 12>("fp16 terms again", sink, src, win, cus);
        run<0, 0, 0, 0, 2, 0, 0, 0, 0, 12>("48 MFMAs only", sink, src, win, cus);
    }
    if (set == 8)
    {
        // does it matter WHERE the staged bytes come from? the linear-layer loop with its 8 loads per lane and K-tile served by the
        // CU's own L1 (16 KB window: every workgroup reads the same lines), by L2 (64 KB .. 1 MB), by the Infinity Cache (16 MB+)
        run<0, 0, 0, 0, 2>("warm-up", sink, src, (size_t)16 << 20, cus);
        for (size_t win : {(size_t)16 << 10, (size_t)64 << 10, (size_t)1 << 20, (size_t)16 << 20, (size_t)256 << 20})
        {
            char tag[96];
            snprintf(tag, sizeof tag, "LIN kernel (16 reads, 4 writes, 8 loads), window %zu KB", win >> 10);
            run<1, 16, 8, 1, 2, 4, 1>(tag, sink, src, win, cus);
            snprintf(tag, sizeof tag, "the same with 4 loads, window %zu KB", win >> 10);
            run<1, 16, 4, 1, 2, 4, 1>(tag, sink, src, win, cus);
        }
        run<1, 16, 0, 1, 2, 4, 1>("the same with no loads", sink, src, (size_t)16 << 20, cus);
    }
    if (set == 7)
    {
        // v_mfma_f32_16x16x32_f16 against ..._bf16: the same pipe rate? (random bit patterns as operands either way)
        const size_t win = (size_t)16 << 20;
        run<0, 0, 0, 0, 2>("warm-up", sink, src, win, cus);
        run<0, 0, 0, 0, 2>("MFMA only", sink, src, win, cus);
        run<0, 0, 0, 0, 2, 0, 0, 0, 2>("MFMA only", sink, src, win, cus);
        run<2, 16, 6, 1, 2, 8, 1, 0, 0, 12>("fp16-term tile-step with the bf16 instruction", sink, src, win, cus);
        run<2, 16, 6, 1, 2, 8, 1, 0, 2, 12>("fp16-term tile-step", sink, src, win, cus);
    }
    return 0;
}

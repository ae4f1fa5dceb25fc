// fft_mfma_repro.hip - standalone reproducer of the packed-fp32 erratum (round 4 knew it as "FFT frames next to bf16 MFMA waves").
//
// Victim: the PRODUCT's stft_kernel (this file includes csrc/fft.hip, so it is the same source text, compiled with the flags
// given on the command line), launched over and over on one stream. Aggressor: a loop of matrix instructions on registers on a
// second stream, nothing else (no global loads or stores in the loop, LDS only as a residency knob). Every victim launch is
// compared bit for bit with a reference taken while the GPU was otherwise idle; the report says how many launches differ,
// where (frame, bin, component) and in what pattern, per aggressor kind.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idemucs_cpp_amd/csrc [-fno-slp-vectorize | -DREPRO_VICTIM_LDS_PAD=70000 ...] \
//         -o tools/micro/fft_mfma_repro tools/micro/fft_mfma_repro.hip
//   fft_mfma_repro [rounds=24] [victim launches per round=12] [aggressor mask, default all]
//
// Aggressor kinds: 0 none, 1 v_mfma_f32_16x16x32_bf16 (the exact-split kernels' instruction), 2 v_mfma_f32_16x16x4_f32
// (the fp32 kernels'), 3 v_mfma_f32_32x32x16_bf16, 4 v_mfma_f32_16x16x32_f16, 5 VALU fma only, 6 = 1 with 80 KB of LDS per
// workgroup (at most one aggressor workgroup beside one victim workgroup per CU, the product's residency), 7 = 1 with
// zero operands (same instruction stream, least switching power).
#include "fft.hip"
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float rf32x4 __attribute__((ext_vector_type(4)));
typedef float rf32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 rbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rf16x8 __attribute__((ext_vector_type(8)));

#ifndef REPRO_VICTIM_LDS_PAD
#define REPRO_VICTIM_LDS_PAD 0
#endif

#define CK(x)                                                                                  \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
        {                                                                                      \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)

template <int KIND>
__global__ __launch_bounds__(256, 2) void aggressor_kernel(const unsigned *seed, float *sink, int iters)
{
    extern __shared__ unsigned dynlds[];
    const int tid = threadIdx.x;
    // operands: random bit patterns of moderate magnitude (sign, 7 exponent choices around 1, random significand)
    unsigned r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
        unsigned v = seed[(blockIdx.x * 256 + tid) * 8 + i];
        if (KIND == 7)
            v = 0;
        r[i] = v;
    }
    if (tid == 0 && iters < 0)
        dynlds[0] = r[0]; // (keeps the allocation referenced)
    rf32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        acc[i] = rf32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (KIND == 1 || KIND == 6 || KIND == 7)
    {
        // bf16 pairs: keep sign + 8 exponent bits in a sane range: 0x3f80 +- a few exponents
        rbf16x8 a, b;
        unsigned ua[4], ub[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            ua[i] = KIND == 7 ? 0u : ((r[i] & 0x807f807fu) | 0x3f003f00u);
            ub[i] = KIND == 7 ? 0u : ((r[4 + i] & 0x807f807fu) | 0x3f803f80u);
        }
        a = __builtin_bit_cast(rbf16x8, ua);
        b = __builtin_bit_cast(rbf16x8, ub);
        for (int it = 0; it < iters; ++it)
        {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] *= 0.0625f; // keeps the values bounded (VALU work like an epilogue's, 8 per 32 MFMAs)
        }
    }
    else if constexpr (KIND == 2)
    {
        const float a = __uint_as_float((r[0] & 0x807fffffu) | 0x3f000000u), b = __uint_as_float((r[1] & 0x807fffffu) | 0x3f800000u);
        for (int it = 0; it < iters; ++it)
        {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] *= 0.25f;
        }
    }
    else if constexpr (KIND == 3)
    {
        rbf16x8 a, b;
        unsigned ua[4], ub[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            ua[i] = (r[i] & 0x807f807fu) | 0x3f003f00u, ub[i] = (r[4 + i] & 0x807f807fu) | 0x3f803f80u;
        a = __builtin_bit_cast(rbf16x8, ua);
        b = __builtin_bit_cast(rbf16x8, ub);
        rf32x16 c0 = {}, c1 = {};
        for (int it = 0; it < iters; ++it)
        {
#pragma unroll
            for (int u = 0; u < 8; ++u)
            {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            }
            c0 *= 0.0625f;
            c1 *= 0.0625f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc[0][i] = c0[i] + c1[i + 4];
    }
    else if constexpr (KIND == 4)
    {
        rf16x8 a, b;
        unsigned ua[4], ub[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            ua[i] = (r[i] & 0x83ff83ffu) | 0x38003800u, ub[i] = (r[4 + i] & 0x83ff83ffu) | 0x3c003c00u;
        a = __builtin_bit_cast(rf16x8, ua);
        b = __builtin_bit_cast(rf16x8, ub);
        for (int it = 0; it < iters; ++it)
        {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] *= 0.0625f;
        }
    }
    else if constexpr (KIND == 5)
    {
        const float a = __uint_as_float((r[0] & 0x807fffffu) | 0x3f000000u), b = __uint_as_float((r[1] & 0x007fffffu) | 0x3e000000u);
        for (int it = 0; it < iters; ++it)
        {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[i][c] = fmaf(acc[i][c], a, b);
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        t += (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
    if (t == 123.456f)
        sink[0] = t;
}

// bitwise compare of float4 words; records the count and the first 64 mismatching word indices
__global__ void compare_kernel(const uint4 *got, const uint4 *ref, long n, unsigned *count, unsigned *first, unsigned *compMask)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    {
        const uint4 a = got[i], b = ref[i];
        const unsigned m = (a.x != b.x) | ((a.y != b.y) << 1) | ((a.z != b.z) << 2) | ((a.w != b.w) << 3);
        if (m)
        {
            const unsigned k = atomicAdd(count, 1u);
            if (k < 64)
                first[k] = (unsigned)i, compMask[k] = m;
        }
    }
}

template <int KIND>
static void launch_aggressor(int grid, int ldsBytes, const unsigned *seed, float *sink, int iters, hipStream_t s)
{
    if (ldsBytes > 48 * 1024)
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&aggressor_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes));
    hipLaunchKernelGGL(aggressor_kernel<KIND>, dim3(grid), dim3(256), ldsBytes, s, seed, sink, iters);
}
static void launch_aggressor_kind(int kind, int grid, const unsigned *seed, float *sink, int iters, hipStream_t s)
{
    switch (kind)
    {
    case 1: launch_aggressor<1>(grid, 32 * 1024, seed, sink, iters, s); break;
    case 2: launch_aggressor<2>(grid, 32 * 1024, seed, sink, iters / 2, s); break;
    case 3: launch_aggressor<3>(grid, 32 * 1024, seed, sink, iters, s); break;
    case 4: launch_aggressor<4>(grid, 32 * 1024, seed, sink, iters, s); break;
    case 5: launch_aggressor<5>(grid, 32 * 1024, seed, sink, iters / 4, s); break;
    case 6: launch_aggressor<6>(grid, 80 * 1024, seed, sink, iters, s); break;
    case 7: launch_aggressor<7>(grid, 32 * 1024, seed, sink, iters, s); break;
    default: break;
    }
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 24;
    const int vl = argc > 2 ? atoi(argv[2]) : 12;
    const unsigned mask = argc > 3 ? (unsigned)strtoul(argv[3], nullptr, 0) : 0xffu;
    const int B = 2, seg = 343980, T = 336, pad = 1536; // (the product's full segment: le = 336 frames, pad = hop/2*3)
    const long outWords = (long)B * T * 2048;             // float4 words per launch
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 0.1f);
    std::vector<float> mix((size_t)B * seg * 2), window(4096), tw(4096);
    for (auto &v : mix)
        v = nd(rng);
    for (int i = 0; i < 4096; ++i)
        window[i] = 0.5f * (1.f - cosf(2.f * 3.14159265358979f * i / 4096.f));
    for (int k = 0; k < 2048; ++k)
        tw[2 * k] = (float)cos(-2.0 * 3.14159265358979323846 * k / 4096.0), tw[2 * k + 1] = (float)sin(-2.0 * 3.14159265358979323846 * k / 4096.0);
    float *dMix, *dWin, *dTw, *dRef, *dOut, *dRs, *dRt, *dSink;
    unsigned *dSeed, *dCount, *dFirst, *dMask;
    CK(hipMalloc(&dMix, mix.size() * 4));
    CK(hipMalloc(&dWin, 4096 * 4));
    CK(hipMalloc(&dTw, 4096 * 4));
    CK(hipMalloc(&dRef, outWords * 16));
    CK(hipMalloc(&dOut, outWords * 16 * vl));
    CK(hipMalloc(&dRs, (size_t)B * T * 2 * 4 * (vl + 1)));
    CK(hipMalloc(&dRt, (size_t)B * T * 2 * 4 * (vl + 1)));
    CK(hipMalloc(&dSink, 64));
    const int aggGrid = 512 * 4;
    std::vector<unsigned> seed((size_t)aggGrid * 256 * 8);
    for (auto &v : seed)
        v = rng();
    CK(hipMalloc(&dSeed, seed.size() * 4));
    CK(hipMalloc(&dCount, 4));
    CK(hipMalloc(&dFirst, 64 * 4));
    CK(hipMalloc(&dMask, 64 * 4));
    CK(hipMemcpy(dMix, mix.data(), mix.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dWin, window.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dTw, tw.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dSeed, seed.data(), seed.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    if (REPRO_VICTIM_LDS_PAD > 0)
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&dmx::stft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, REPRO_VICTIM_LDS_PAD));
    // REPRO_CO=<code object>: the victim is that file's stft_kernel (tools/micro/pk_bisect.py: the product kernel's instruction
    // stream with chosen packed instructions rewritten as scalar pairs) instead of the one compiled into this binary
    hipFunction_t coFn = nullptr;
    if (const char *co = getenv("REPRO_CO"))
    {
        hipModule_t mod;
        CK(hipModuleLoad(&mod, co));
        CK(hipModuleGetFunction(&coFn, mod, "_ZN3dmx11stft_kernelENS_8StftArgsE"));
        printf("victim from %s\n", co);
    }
    auto victimK = [&](float *out, int slot, bool builtin) {
        dmx::StftArgs a{};
        a.mix = dMix, a.x = out, a.rowstat = dRs + (size_t)slot * B * T * 2, a.rowstatT = dRt + (size_t)slot * B * T * 2;
        a.B = B, a.T = T, a.seg = seg, a.pad = pad, a.window = dWin, a.twiddle = dTw;
        if (coFn && !builtin)
        {
            size_t sz = sizeof a;
            void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            CK(hipModuleLaunchKernel(coFn, a.T, a.B, 1, 256, 1, 1, REPRO_VICTIM_LDS_PAD, sv, nullptr, cfg));
        }
        else
            hipLaunchKernelGGL(dmx::stft_kernel, dim3(a.T, a.B), dim3(256), REPRO_VICTIM_LDS_PAD, sv, a);
    };
    auto victim = [&](float *out, int slot) { victimK(out, slot, false); };
    // reference on an idle GPU (always the built-in kernel: a rewritten stream must reproduce its bits), and the victim once
    victimK(dRef, vl, true);
    victim(dOut, 0);
    CK(hipDeviceSynchronize());
    {
        CK(hipMemset(dCount, 0, 4));
        hipLaunchKernelGGL(compare_kernel, dim3(1024), dim3(256), 0, sv, (const uint4 *)dOut, (const uint4 *)dRef, outWords, dCount, dFirst, dMask);
        unsigned c = 0;
        CK(hipMemcpy(&c, dCount, 4, hipMemcpyDeviceToHost));
        printf("idle self-check: %u differing words of %ld\n", c, outWords);
    }
    // calibrate the aggressor to about 1.5x the victim burst
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, sv));
    for (int i = 0; i < vl; ++i)
        victim(dOut + (size_t)i * outWords * 4, i);
    CK(hipEventRecord(e1, sv));
    CK(hipDeviceSynchronize());
    float msV = 0;
    CK(hipEventElapsedTime(&msV, e0, e1));
    printf("victim burst of %d launches alone: %.3f ms (%.1f us per launch, %d workgroups each), victim LDS pad %d\n", vl, msV, 1e3 * msV / vl, B * T,
           REPRO_VICTIM_LDS_PAD);
    const char *names[8] = {"none", "mfma_f32_16x16x32_bf16", "mfma_f32_16x16x4_f32", "mfma_f32_32x32x16_bf16", "mfma_f32_16x16x32_f16", "valu_fma",
                            "mfma bf16 + 80 KB LDS", "mfma bf16, zero operands"};
    for (int kind = 0; kind < 8; ++kind)
    {
        if (!(mask & (1u << kind)))
            continue;
        int iters = 2000;
        float msA = 0;
        if (kind)
        {
            for (int rep = 0; rep < 2; ++rep)
            {
                CK(hipEventRecord(e0, sa));
                launch_aggressor_kind(kind, aggGrid, dSeed, dSink, iters, sa);
                CK(hipEventRecord(e1, sa));
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&msA, e0, e1));
                if (rep == 0)
                    iters = (int)(iters * (2.5 * msV) / (msA > 1e-3f ? msA : 1e-3f)) + 64;
            }
        }
        long wrongLaunches = 0, wrongWords = 0, launches = 0;
        unsigned hist16 = 0, hist16aligned = 0, histComp[16] = {0};
        std::vector<std::string> samples;
        float msBoth = 0;
        for (int r = 0; r < rounds; ++r)
        {
            CK(hipEventRecord(e0, sv));
            if (kind)
                launch_aggressor_kind(kind, aggGrid, dSeed, dSink, iters, sa);
            for (int i = 0; i < vl; ++i)
                victim(dOut + (size_t)i * outWords * 4, i);
            CK(hipEventRecord(e1, sv));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            msBoth += ms;
            for (int i = 0; i < vl; ++i)
            {
                CK(hipMemset(dCount, 0, 4));
                hipLaunchKernelGGL(compare_kernel, dim3(1024), dim3(256), 0, sv, (const uint4 *)(dOut + (size_t)i * outWords * 4), (const uint4 *)dRef,
                                   outWords, dCount, dFirst, dMask);
                unsigned c = 0, first[64], cm[64];
                CK(hipMemcpy(&c, dCount, 4, hipMemcpyDeviceToHost));
                ++launches;
                if (!c)
                    continue;
                ++wrongLaunches;
                wrongWords += c;
                CK(hipMemcpy(first, dFirst, sizeof first, hipMemcpyDeviceToHost));
                CK(hipMemcpy(cm, dMask, sizeof cm, hipMemcpyDeviceToHost));
                const unsigned n = c < 64 ? c : 64;
                // pattern: are the wrong words one aligned run of 16 bins of one frame?
                unsigned lo = ~0u, hi = 0;
                for (unsigned k = 0; k < n; ++k)
                    lo = first[k] < lo ? first[k] : lo, hi = first[k] > hi ? first[k] : hi;
                if (c <= 16 && hi - lo < 16)
                {
                    ++hist16;
                    if ((lo & 15u) == 0 || ((2048u - (hi & 2047u)) & 15u) == 0)
                        ++hist16aligned;
                }
                for (unsigned k = 0; k < n; ++k)
                    ++histComp[cm[k] & 15];
                if (samples.size() < 3)
                {
                    // values of the first mismatching words: is the error a rounding difference or garbage?
                    for (unsigned k = 0; k < (n < 6 ? n : 6); ++k)
                    {
                        float g[4], w[4];
                        CK(hipMemcpy(g, dOut + ((size_t)i * outWords + first[k]) * 4, 16, hipMemcpyDeviceToHost));
                        CK(hipMemcpy(w, dRef + (size_t)first[k] * 4, 16, hipMemcpyDeviceToHost));
                        char vb[320];
                        snprintf(vb, sizeof vb, "      word %u (frame %u bin %u): got %.8g %.8g %.8g %.8g | ref %.8g %.8g %.8g %.8g", first[k], (first[k] / 2048) % T,
                                 first[k] % 2048, g[0], g[1], g[2], g[3], w[0], w[1], w[2], w[3]);
                        samples.push_back(vb);
                    }
                }
                if (samples.size() < 24)
                {
                    char buf[256];
                    snprintf(buf, sizeof buf, "round %d launch %d: %u words, frame %u bins %u..%u (b %u), component masks %x %x %x", r, i, c, (lo / 2048) % T,
                             lo % 2048, hi % 2048, lo / 2048 / T, cm[0], cm[n / 2], cm[n - 1]);
                    samples.push_back(buf);
                }
            }
        }
        printf("aggressor %d %-28s: %ld of %ld victim launches differ (%ld words); aggressor alone %.3f ms, together %.3f ms per round; "
               "single runs of <= 16 bins: %u (16-aligned: %u)\n",
               kind, names[kind], wrongLaunches, launches, wrongWords, msA, msBoth / rounds, hist16, hist16aligned);
        if (wrongLaunches)
        {
            printf("   component-mask histogram (bit0 re0, bit1 im0, bit2 re1, bit3 im1):");
            for (int m = 1; m < 16; ++m)
                if (histComp[m])
                    printf(" %x:%u", m, histComp[m]);
            printf("\n");
            for (auto &s : samples)
                printf("   %s\n", s.c_str());
        }
        fflush(stdout);
    }
    return 0;
}

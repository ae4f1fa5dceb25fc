// gemm_fp16x3.hip — standalone prototype (NOT product code) of the "next lever" of profiles/DESIGN_history_r1-r4.md 7.8: the linear-layer split
// GEMM of csrc/igemm_split.hip (igemm_split_lin_kernel: four waves stacked in M, activation fragments loaded straight into
// registers and split there, weight planes through LDS, 128 x 128 tile, K-tile 32) with
//   ARITH 0: bf16 terms, a = a1 + a2 + a3, w = w1 + w2, five v_mfma_f32_16x16x32_bf16 per block (the product's arithmetic)
//   ARITH 1: fp16 terms, a = h1 + h2 + h3, w one exact fp16 term, three v_mfma_f32_16x16x32_f16 per block, ONE weight plane
// Y[M][N] = X[M][K] W[N][K]^T. Checks sampled outputs against a double-precision sum over the quantised operands and
// times both. Usage: gemm_fp16x3 [M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

template <int ARITH>
__device__ __forceinline__ void split3(float x0, float x1, unsigned &h1, unsigned &h2, unsigned &h3)
{
    if (ARITH == 0)
    {
        h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
        const float r0 = x0 - __uint_as_float(h1 << 16), r1 = x1 - __uint_as_float(h1 & 0xffff0000u);
        h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
        const float s0 = r0 - __uint_as_float(h2 << 16), s1 = r1 - __uint_as_float(h2 & 0xffff0000u);
        h3 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
    }
    else
    {
        const f16x2 a = __builtin_convertvector(f32x2{x0, x1}, f16x2);
        const f32x2 r = f32x2{x0, x1} - __builtin_convertvector(a, f32x2);
        const f16x2 b = __builtin_convertvector(r, f16x2);
        const f32x2 q = r - __builtin_convertvector(b, f32x2);
        const f16x2 c = __builtin_convertvector(q, f16x2);
        h1 = __builtin_bit_cast(unsigned, a), h2 = __builtin_bit_cast(unsigned, b), h3 = __builtin_bit_cast(unsigned, c);
    }
}

template <int ARITH>
__device__ __forceinline__ f32x4 mma(u32x4 w, u32x4 a, f32x4 c)
{
    if (ARITH == 0)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
}

struct Args
{
    const float *X;            // [M][K]
    const unsigned short *Wp;  // [NBP][N][K] 16-bit planes
    float *Y;                  // [M][N]
    int M, N, K, tilesM, tilesN;
};

template <int ARITH>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const Args p)
{
    constexpr int KT = 32, WMF = 2, WNF = 8, BM = 128, BN = 128;
    constexpr int NBP = ARITH ? 1 : 2;       // weight planes
    constexpr int BR = NBP * BN * 4 / 256;   // 16-byte chunks per thread and K-tile
    __shared__ u32x4 Bp0[NBP][BN][4], Bp1[NBP][BN][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int tileN = blockIdx.x % p.tilesN, tileM = blockIdx.x / p.tilesN;
    const int m0 = tileM * BM, n0 = tileN * BN;
    unsigned aOff[WMF], bOff[BR];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
        aOff[i] = (unsigned)(min(m0 + wave * (WMF * 16) + i * 16 + l15, p.M - 1)) * (unsigned)p.K * 4u + kq * 32;
    const int bOct = tid & 3;
    auto bRowOf = [&](int i) { return 64 * (i / NBP) + 2 * (tid >> 3) + ((tid >> 2) & 1); };
#pragma unroll
    for (int i = 0; i < BR; ++i)
        bOff[i] = (((unsigned)(i % NBP) * (unsigned)p.N + (unsigned)min(n0 + bRowOf(i), p.N - 1)) * (unsigned)p.K + (unsigned)bOct * 8u) * 2u;
    f32x4 aRaw[2][WMF][2];
    u32x4 bReg[2][BR], aPl[2][WMF][3];
    auto issue = [&](int set) {
#pragma unroll
        for (int i = 0; i < WMF; ++i)
        {
            const char *src = reinterpret_cast<const char *>(p.X) + aOff[i];
            aRaw[set][i][0] = *reinterpret_cast<const f32x4 *>(src);
            aRaw[set][i][1] = *reinterpret_cast<const f32x4 *>(src + 16);
            aOff[i] += KT * 4;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
        {
            bReg[set][i] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.Wp) + bOff[i]);
            bOff[i] += KT * 2;
        }
    };
    auto split_block = [&](int set, int i) {
        const f32x4 lo = aRaw[set][i][0], hi = aRaw[set][i][1];
        unsigned h1[4], h2[4], h3[4];
        split3<ARITH>(lo[0], lo[1], h1[0], h2[0], h3[0]);
        split3<ARITH>(lo[2], lo[3], h1[1], h2[1], h3[1]);
        split3<ARITH>(hi[0], hi[1], h1[2], h2[2], h3[2]);
        split3<ARITH>(hi[2], hi[3], h1[3], h2[3], h3[3]);
        u32x4 q1{h1[0], h1[1], h1[2], h1[3]}, q2{h2[0], h2[1], h2[2], h2[3]}, q3{h3[0], h3[1], h3[2], h3[3]};
        asm volatile("" : "+v"(q1), "+v"(q2), "+v"(q3));
        aPl[set][i][0] = q1, aPl[set][i][1] = q2, aPl[set][i][2] = q3;
    };
    auto store_B = [&](int set, int buf, int b0, int b1) {
        u32x4(*Bp)[BN][4] = buf ? Bp1 : Bp0;
#pragma unroll
        for (int i = 0; i < BR; ++i)
            if (i >= b0 && i < b1)
            {
                const int row = bRowOf(i);
                Bp[i % NBP][row][bOct ^ swz(row)] = bReg[set][i];
            }
    };
    f32x4 acc[WMF][WNF];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
        for (int j = 0; j < WNF; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = p.K / KT;
    issue(0);
    issue(1);
#pragma unroll
    for (int i = 0; i < WMF; ++i)
        split_block(0, i);
    store_B(0, 0, 0, BR);
    __syncthreads();
    const int fslot = kq ^ swz(l15);
    auto iteration = [&](auto parTag) {
        constexpr int PAR = decltype(parTag)::value;
        u32x4(*Bp)[BN][4] = PAR ? Bp1 : Bp0;
        issue(PAR); // tile kt + 2
        u32x4 b1[2][4], b2[2][4];
        auto read_half = [&](int h) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                const int r = (h * 4 + j) * 16 + l15;
                b1[h][j] = Bp[0][r][fslot];
                if (NBP == 2)
                    b2[h][j] = Bp[NBP - 1][r][fslot];
            }
        };
        auto term = [&](int h, u32x4(&b)[4], int plane) {
#pragma unroll
            for (int i = 0; i < WMF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][h * 4 + j] = mma<ARITH>(b[j], aPl[PAR][i][plane], acc[i][h * 4 + j]);
        };
        read_half(0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
            term(h, b1[h], 2); // smallest first
            if (h == 0)
                read_half(1);
            if (ARITH == 0)
                term(h, b2[h], 1);
            split_block(PAR ^ 1, h); // tile kt + 1 (WMF = 2 blocks, one per half)
            store_B(PAR ^ 1, PAR ^ 1, h * BR / 2, (h + 1) * BR / 2);
            if (ARITH == 0)
                term(h, b2[h], 0);
            term(h, b1[h], 1);
            term(h, b1[h], 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2)
    {
        iteration(std::integral_constant<int, 0>{});
        if (kt + 1 < nk)
            iteration(std::integral_constant<int, 1>{});
    }
    // accumulators hold C^T: lane (l15, kq) owns row 16 i + l15, columns 16 j + 4 kq .. + 3
#pragma unroll
    for (int i = 0; i < WMF; ++i)
    {
        const int m = m0 + wave * (WMF * 16) + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < WNF; ++j)
        {
            const int n = n0 + j * 16 + 4 * kq;
            if (m < p.M && n < p.N)
                *reinterpret_cast<f32x4 *>(p.Y + (size_t)m * p.N + n) = acc[i][j];
        }
    }
}

static float bf16_rn(float x)
{
    unsigned u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    float r;
    memcpy(&r, &u, 4);
    return r;
}
static unsigned short f16_bits(float x)
{
    const _Float16 h = (_Float16)x;
    unsigned short b;
    memcpy(&b, &h, 2);
    return b;
}

int main(int argc, char **argv)
{
    const int M = argc > 3 ? atoi(argv[1]) : 32768, N = argc > 3 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 512;
    std::vector<float> X((size_t)M * K + 256), W((size_t)N * K);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((float)(s >> 8) * (1.0f / 16777216.0f)) * 2.0f - 1.0f; };
    for (auto &v : X)
        v = rnd() * (rnd() > 0.9f ? 8.0f : 1.0f) * (rnd() > 0.8f ? 1e-3f : 1.0f); // a few large, a few tiny values
    for (auto &v : W)
        v = (float)(_Float16)(rnd() * 0.08f); // fp16 numbers, like the weight files
    std::vector<unsigned short> Wb((size_t)2 * N * K + 512), Wh((size_t)N * K + 512);
    for (size_t i = 0; i < (size_t)N * K; ++i)
    {
        const float w1 = bf16_rn(W[i]), w2 = W[i] - w1;
        unsigned u1, u2;
        memcpy(&u1, &w1, 4), memcpy(&u2, &w2, 4);
        if (bf16_rn(w2) != w2)
            printf("weight %zu is not two bf16 terms\n", i);
        Wb[i] = (unsigned short)(u1 >> 16), Wb[(size_t)N * K + i] = (unsigned short)(u2 >> 16), Wh[i] = f16_bits(W[i]);
    }
    float *dX, *dY;
    unsigned short *dWb, *dWh;
    hipMalloc(&dX, X.size() * 4), hipMalloc(&dY, (size_t)M * N * 4), hipMalloc(&dWb, Wb.size() * 2), hipMalloc(&dWh, Wh.size() * 2);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dWb, Wb.data(), Wb.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dWh, Wh.data(), Wh.size() * 2, hipMemcpyHostToDevice);
    Args a{dX, dWb, dY, M, N, K, (M + 127) / 128, (N + 127) / 128};
    std::vector<float> Y0((size_t)M * N), Y1((size_t)M * N);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int arith = 0; arith < 2; ++arith)
    {
        a.Wp = arith ? dWh : dWb;
        const dim3 grid((unsigned)(a.tilesM * a.tilesN));
        auto launch = [&]() {
            if (arith)
                hipLaunchKernelGGL(gemm_kernel<1>, grid, dim3(256), 0, 0, a);
            else
                hipLaunchKernelGGL(gemm_kernel<0>, grid, dim3(256), 0, 0, a);
        };
        hipMemset(dY, 0xff, (size_t)M * N * 4);
        launch();
        hipDeviceSynchronize();
        hipMemcpy((arith ? Y1 : Y0).data(), dY, (size_t)M * N * 4, hipMemcpyDeviceToHost);
        for (int w = 0; w < 3; ++w)
            launch();
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r)
            launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 20;
        const double tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12;
        // sampled check against a double-precision sum over the quantised activation (fp16 terms: what the three terms hold)
        const std::vector<float> &Y = arith ? Y1 : Y0;
        double worst = 0, worstq = 0;
        unsigned t = 99u;
        for (int c = 0; c < 4096; ++c)
        {
            t = t * 1664525u + 1013904223u;
            const int m = (int)((t >> 8) % (unsigned)M);
            t = t * 1664525u + 1013904223u;
            const int n = (int)((t >> 8) % (unsigned)N);
            double ref = 0, refq = 0, mag = 0;
            for (int k = 0; k < K; ++k)
            {
                const float x = X[(size_t)m * K + k];
                float q = x;
                if (arith)
                {
                    const float h1 = (float)(_Float16)x, h2 = (float)(_Float16)(x - h1), h3 = (float)(_Float16)(x - h1 - h2);
                    q = (h1 + h2) + h3;
                }
                ref += (double)x * W[(size_t)n * K + k], refq += (double)q * W[(size_t)n * K + k], mag += fabs((double)x * W[(size_t)n * K + k]);
            }
            worst = fmax(worst, fabs(Y[(size_t)m * N + n] - ref) / mag), worstq = fmax(worstq, fabs(Y[(size_t)m * N + n] - refq) / mag);
        }
        printf("%s: %d x %d x %d  %.3f ms  %.1f TFLOP/s fp32-equivalent   worst |y - exact| / sum|terms| %.2e   vs the quantised operands %.2e\n",
               arith ? "fp16 terms (3 MFMAs, one weight plane)" : "bf16 terms (5 MFMAs, two weight planes)", M, N, K, ms, tf, worst, worstq);
    }
    double d = 0, mx = 0;
    for (size_t i = 0; i < Y0.size(); ++i)
        d = fmax(d, fabs((double)Y0[i] - Y1[i])), mx = fmax(mx, fabs((double)Y0[i]));
    printf("max |bf16-term result - fp16-term result| = %.3e of max |y| = %.3e (%.2e relative)\n", d, mx, d / mx);
    return 0;
}

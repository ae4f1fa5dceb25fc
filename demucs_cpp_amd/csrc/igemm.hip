// igemm.hip — fp32 implicit-GEMM convolution / linear kernel for gfx950 (CDNA4).
//
// One kernel family covers every conv and linear layer of HTDemucs
// (/root/reference/src/conv.hpp:13-524 conv1d/conv2d/conv*_tr + fused GELU,
//  src/layers.cpp:426-440,488-511 transformer linears, src/encdec.cpp rewrite convs):
// activations are channels-last, so an im2col row is a few contiguous runs of memory
// and is gathered on the fly while staging the A tile into LDS - the 75%-zero / 9x
// im2col matrices of the reference never exist.
//
// Math: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: bit-exact fmaf chain, 157 TF
// peak). Each lane reads ONE ds_read_b128 per fragment = 4 consecutive k of its row and
// feeds 4 MFMAs; MFMA j pairs k-slot h = lane>>4 with real k = 4h + j for both
// operands, so the 16x16x4 k-slots never need a shuffle. LDS layout [kq][row] float4
// makes every 16-lane ds_read_b128 group hit 16 distinct 16-byte slots (conflict free).
//
// Tile = (WAVES_M*WMF*16) x (WAVES_N*WNF*16) x 16, 256 threads, register-prefetched
// double-buffered LDS (one barrier per K-tile).
// Epilogues (plan.h): bias/GELU/residual, LayerScale+residual, GLU on paired fragments,
// GroupNorm+GLU+LayerScale+residual (DConv tail), statistics-only, transposed-conv
// scatter with crop; optional per-row (sum, sumsq) partials for GroupNorm statistics.
#include "kernels.h"

namespace dmx
{

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + __expf(-v)); }

template <int WAVES_M, int WAVES_N, int WMF, int WNF>
__global__ __launch_bounds__(256) void igemm_kernel(const GemmArgs p)
{
    constexpr int BM = WAVES_M * WMF * 16;
    constexpr int BN = WAVES_N * WNF * 16;
    constexpr int AL = (BM * 4 + 255) / 256; // float4 A loads per thread per K-tile
    constexpr int BL = (BN * 4 + 255) / 256;
    static_assert(WAVES_M * WAVES_N == 4, "256 threads");

    __shared__ float4 As[2][4][BM];
    __shared__ float4 Bs[2][4][BN];
    __shared__ int4 rowinfo[BM];          // b, p1, p0, group (-1: row >= M)
    __shared__ float2 rsum[BM][WAVES_N];  // cross-wave row statistics

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const i64 m0 = (i64)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    for (int r = tid; r < BM; r += 256)
    {
        i64 m = m0 + r;
        int4 ri;
        if (m < p.M)
        {
            int p0 = (int)(m % p.P0);
            i64 t = m / p.P0;
            int p1 = (int)(t % p.P1);
            int b = (int)(t / p.P1);
            ri = make_int4(b, p1, p0, b * p.G0 + (p.G0 > 1 ? p0 : 0));
        }
        else
            ri = make_int4(0, 0, 0, -1);
        rowinfo[r] = ri;
    }
    __syncthreads();

    // ---- per-thread A gather state
    const i64 rowLen = (i64)p.L0 * p.Cin;
    const float *aBase[AL];
    int aIn1[AL], aE0[AL], aGrp[AL], aB[AL];
    bool aValid[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i)
    {
        int idx = tid + i * 256;
        int row = idx >> 2;
        aValid[i] = false;
        aBase[i] = p.X;
        aIn1[i] = aE0[i] = aGrp[i] = aB[i] = 0;
        if (row < BM)
        {
            int4 ri = rowinfo[row];
            aValid[i] = ri.w >= 0;
            aBase[i] = p.X + (i64)ri.x * p.xBS;
            aIn1[i] = ri.y * p.stride1 - p.pad1;
            aE0[i] = (ri.z * p.stride0 - p.pad0) * p.Cin;
            aGrp[i] = ri.w;
            aB[i] = ri.x;
        }
    }

    float4 aReg[AL], bReg[BL];
    const int nk = p.Kp >> 4;

    auto load_tiles = [&](int kt) {
        const int kbase = kt << 4;
        int s1 = 0, offb = kbase;
        if (p.S1 > 1)
        {
            s1 = kbase / p.seg0;
            offb = kbase - s1 * p.seg0;
        }
#pragma unroll
        for (int i = 0; i < AL; ++i)
        {
            int idx = tid + i * 256;
            int kq = idx & 3;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int k = kbase + kq * 4;
            int in1 = aIn1[i] + s1 * p.dil1;
            if ((idx >> 2) < BM && aValid[i] && k < p.K && in1 >= 0 && in1 < p.L1)
            {
                i64 e = (i64)aE0[i] + offb + kq * 4;
                const float *src = aBase[i] + (i64)in1 * rowLen + e;
                bool m0v, m1v, m2v, m3v;
                if (e >= 0 && e + 3 < rowLen && ((reinterpret_cast<uintptr_t>(src) & 15) == 0))
                {
                    v = *reinterpret_cast<const float4 *>(src);
                    m0v = m1v = m2v = m3v = true;
                }
                else
                {
                    m0v = e >= 0 && e < rowLen;
                    m1v = e + 1 >= 0 && e + 1 < rowLen;
                    m2v = e + 2 >= 0 && e + 2 < rowLen;
                    m3v = e + 3 >= 0 && e + 3 < rowLen;
                    if (m0v) v.x = src[0];
                    if (m1v) v.y = src[1];
                    if (m2v) v.z = src[2];
                    if (m3v) v.w = src[3];
                }
                if (p.pro == PRO_AFFINE)
                {
                    const float mean = p.proStats[aB[i] * 4], sc = p.proStats[aB[i] * 4 + 1];
                    if (m0v) v.x = (v.x - mean) * sc;
                    if (m1v) v.y = (v.y - mean) * sc;
                    if (m2v) v.z = (v.z - mean) * sc;
                    if (m3v) v.w = (v.w - mean) * sc;
                }
                else if (p.pro == PRO_GN_GELU)
                {
                    const float mean = p.proStats[aGrp[i] * 4], sc = p.proStats[aGrp[i] * 4 + 1];
                    const float4 gw = *reinterpret_cast<const float4 *>(p.proW + k);
                    const float4 gb = *reinterpret_cast<const float4 *>(p.proB + k);
                    v.x = m0v ? gelu_f((v.x - mean) * sc * gw.x + gb.x) : 0.f;
                    v.y = m1v ? gelu_f((v.y - mean) * sc * gw.y + gb.y) : 0.f;
                    v.z = m2v ? gelu_f((v.z - mean) * sc * gw.z + gb.z) : 0.f;
                    v.w = m3v ? gelu_f((v.w - mean) * sc * gw.w + gb.w) : 0.f;
                }
            }
            aReg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BL; ++i)
        {
            int idx = tid + i * 256;
            int n = n0 + (idx >> 2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < BN * 4 && n < p.Np)
                v = *reinterpret_cast<const float4 *>(p.Wt + (i64)n * p.Kp + kbase + (idx & 3) * 4);
            bReg[i] = v;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AL; ++i)
        {
            int idx = tid + i * 256;
            if (idx < BM * 4)
                As[buf][idx & 3][idx >> 2] = aReg[i];
        }
#pragma unroll
        for (int i = 0; i < BL; ++i)
        {
            int idx = tid + i * 256;
            if (idx < BN * 4)
                Bs[buf][idx & 3][idx >> 2] = bReg[i];
        }
    };

    f32x4 acc[WMF][WNF];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
        for (int j = 0; j < WNF; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    int cur = 0;
    const int l15 = lane & 15, kq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt)
    {
        const bool next = kt + 1 < nk;
        if (next)
            load_tiles(kt + 1);
        float4 a[WMF], b[WNF];
#pragma unroll
        for (int i = 0; i < WMF; ++i)
            a[i] = As[cur][kq][wm * (WMF * 16) + i * 16 + l15];
#pragma unroll
        for (int j = 0; j < WNF; ++j)
            b[j] = Bs[cur][kq][wn * (WNF * 16) + j * 16 + l15];
#pragma unroll
        for (int i = 0; i < WMF; ++i)
#pragma unroll
            for (int j = 0; j < WNF; ++j)
            {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
            }
        if (next)
            store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ------------------------------------------------------------------ epilogue
    // C layout of v_mfma_f32_16x16x4_f32: col = lane&15, row = (lane>>4)*4 + reg
    const bool wantStats = p.rowstat != nullptr;
    const int colBase = n0 + wn * (WNF * 16) + l15;
    float biasv[WNF];
#pragma unroll
    for (int j = 0; j < WNF; ++j)
    {
        int n = colBase + j * 16;
        biasv[j] = n < p.N ? p.bias[n] : 0.f;
    }

    int trR[WNF], trC[WNF]; // EPI_TRCONV: column n -> (phase r, channel co)
#pragma unroll
    for (int j = 0; j < WNF; ++j)
    {
        trR[j] = trC[j] = 0;
        if (p.epi == EPI_TRCONV)
        {
            int n = colBase + j * 16;
            trR[j] = n / p.Cout;
            trC[j] = n - trR[j] * p.Cout;
        }
    }
#pragma unroll
    for (int i = 0; i < WMF; ++i)
    {
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const int rl = wm * (WMF * 16) + i * 16 + kq * 4 + r;
            const int4 ri = rowinfo[rl];
            const bool rowOk = ri.w >= 0;
            const i64 m = m0 + rl;
            float s = 0.f, ss = 0.f;
            if (p.epi == EPI_LINEAR || p.epi == EPI_SCALE_RES || p.epi == EPI_STATS_ONLY)
            {
#pragma unroll
                for (int j = 0; j < WNF; ++j)
                {
                    const int n = colBase + j * 16;
                    float v = acc[i][j][r] + biasv[j];
                    if (rowOk && n < p.N)
                    {
                        if (p.epi == EPI_LINEAR)
                        {
                            if (p.act)
                                v = gelu_f(v);
                            const i64 o = m * p.ldy + n;
                            if (p.res)
                                v += p.res[o];
                            p.Y[o] = v;
                        }
                        else if (p.epi == EPI_SCALE_RES)
                        {
                            const i64 o = m * p.ldy + n;
                            v = p.res[o] + v * p.scale[n];
                            p.Y[o] = v;
                        }
                        s += v;
                        ss += v * v;
                    }
                }
            }
            else if (p.epi == EPI_GLU || p.epi == EPI_GN_GLU_SCALE_RES)
            {
                if constexpr (WNF % 2 == 0)
                {
#pragma unroll
                    for (int j = 0; j < WNF; j += 2)
                    {
                        const int na = colBase + j * 16, nb = na + 16;
                        if (rowOk && nb < p.N)
                        {
                            float a = acc[i][j][r] + biasv[j];
                            float g = acc[i][j + 1][r] + biasv[j + 1];
                            const int c = (na >> 5) * 16 + (na & 15);
                            const i64 o = m * p.ldy + c;
                            float v;
                            if (p.epi == EPI_GN_GLU_SCALE_RES)
                            {
                                const float mean = p.epiStats[ri.w * 4], sc = p.epiStats[ri.w * 4 + 1];
                                a = (a - mean) * sc * p.epiW[na] + p.epiB[na];
                                g = (g - mean) * sc * p.epiW[nb] + p.epiB[nb];
                                v = p.res[o] + p.scale[c] * (a * sigmoid_f(g));
                            }
                            else
                            {
                                v = a * sigmoid_f(g);
                                if (p.table)
                                    v += p.tableScale * p.table[(i64)ri.z * (p.N >> 1) + c];
                            }
                            p.Y[o] = v;
                        }
                    }
                }
            }
            else // EPI_TRCONV
            {
#pragma unroll
                for (int j = 0; j < WNF; ++j)
                {
                    const int n = colBase + j * 16;
                    if (rowOk && n < p.N)
                    {
                        const int rr = trR[j], co = trC[j];
                        const int jj = 4 * ri.z + rr - 2;
                        if (jj >= 0 && jj < p.Lout)
                        {
                            float v = acc[i][j][r] + biasv[j];
                            if (p.act)
                                v = gelu_f(v);
                            const i64 o = (i64)ri.x * p.yBS + ((i64)ri.y * p.Lout + jj) * p.ldy + co;
                            if (p.res)
                                v += p.res[o];
                            p.Y[o] = v;
                        }
                    }
                }
            }
            if (wantStats)
            {
                // reduce over the 16 lanes that share this row (same lane>>4)
#pragma unroll
                for (int off = 1; off < 16; off <<= 1)
                {
                    s += __shfl_xor(s, off);
                    ss += __shfl_xor(ss, off);
                }
                if (l15 == 0)
                    rsum[rl][wn] = make_float2(s, ss);
            }
        }
    }
    if (wantStats)
    {
        __syncthreads();
        for (int r = tid; r < BM; r += 256)
        {
            i64 m = m0 + r;
            if (m < p.M)
            {
                float s = 0.f, ss = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES_N; ++w)
                {
                    s += rsum[r][w].x;
                    ss += rsum[r][w].y;
                }
                float *dst = p.rowstat + (m * p.NB + blockIdx.y) * 2;
                dst[0] = s;
                dst[1] = ss;
            }
        }
    }
}

template <int WM_, int WN_, int MF, int NF>
static void launch_cfg(const GemmArgs &a, hipStream_t s)
{
    constexpr int BM = WM_ * MF * 16, BN = WN_ * NF * 16;
    dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)((a.N + BN - 1) / BN));
    hipLaunchKernelGGL((igemm_kernel<WM_, WN_, MF, NF>), grid, dim3(256), 0, s, a);
}

void launch_igemm(int cfg, const GemmArgs &a, hipStream_t s)
{
    switch (cfg)
    {
    case 0: launch_cfg<2, 2, 4, 4>(a, s); break; // 128x128
    case 1: launch_cfg<2, 2, 2, 2>(a, s); break; // 64x64
    case 2: launch_cfg<4, 1, 2, 6>(a, s); break; // 128x96
    case 3: launch_cfg<4, 1, 2, 3>(a, s); break; // 128x48
    case 4: launch_cfg<4, 1, 4, 1>(a, s); break; // 256x16
    case 5: launch_cfg<4, 1, 2, 2>(a, s); break; // 128x32
    case 6: launch_cfg<4, 1, 2, 4>(a, s); break; // 128x64
    default: abort();
    }
}

} // namespace dmx

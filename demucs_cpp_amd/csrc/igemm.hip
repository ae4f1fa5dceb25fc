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
// feeds 4 MFMAs; MFMA c pairs k-slot h = lane>>4 with real k = 4h + c for both
// operands, so the 16x16x4 k-slots never need a shuffle. LDS layout [kq][row] float4
// makes every 16-lane ds_read_b128 group hit 16 distinct 16-byte slots (conflict free).
//
// Tile = (WAVES_M*WMF*16) x (WAVES_N*WNF*16) x (16*KS), 256 threads. The staging loads
// of tile t+1 are issued branch-free (clamped addresses, validity kept as a bit mask)
// BEFORE the MFMA block of tile t and only touched (prologue transform, zero fill,
// ds_write) AFTER it, so global-load latency hides behind the matrix pipe; LDS is double
// buffered, one barrier per K-tile. Prologue and epilogue kinds are template parameters
// (one small specialised kernel per (tile, prologue, epilogue) used by the plan).
//
// Alignment contract (checked by the host, dmx_ctx_create): every float4 staging chunk is
// either entirely inside or entirely outside the valid input, and 16-byte aligned:
// Cin*L0, Cin*stride0, Cin*pad0, seg0, K, xBatchStride are multiples of 4 elements.
#include "kernels.h"

namespace dmx
{

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + __expf(-v)); }
__device__ __forceinline__ float f4c(const float4 &v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

template <int WAVES_M, int WAVES_N, int WMF, int WNF, int KS, int PRO, int EPI>
__global__ __launch_bounds__(256) void igemm_kernel(const GemmArgs p)
{
    constexpr int BM = WAVES_M * WMF * 16;
    constexpr int BN = WAVES_N * WNF * 16;
    constexpr int AR = BM / 64;              // A rows staged per thread (row = tid/4 + i*64)
    constexpr int BR = (BN + 63) / 64;       // B rows staged per thread
    static_assert(WAVES_M * WAVES_N == 4, "256 threads");
    static_assert(BM % 64 == 0, "BM multiple of 64");

    __shared__ float4 As[2][4 * KS][BM];
    __shared__ float4 Bs[2][4 * KS][BN];
    __shared__ int4 rowinfo[BM];         // b, p1, p0, group (-1: row >= M)
    __shared__ float2 rsum[BM][WAVES_N]; // cross-wave row statistics

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const i64 m0 = (i64)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    for (int r = tid; r < BM; r += 256)
    {
        const i64 m = m0 + r;
        int4 ri = make_int4(0, 0, 0, -1);
        if (m < p.M)
        {
            const int p0 = (int)(m % p.P0);
            const i64 t = m / p.P0;
            const int p1 = (int)(t % p.P1);
            const int b = (int)(t / p.P1);
            ri = make_int4(b, p1, p0, b * p.G0 + (p.G0 > 1 ? p0 : 0));
        }
        rowinfo[r] = ri;
    }
    __syncthreads();

    // ---- per-thread staging state: AR rows of A (same k-quad), BR rows of B
    const int skq = tid & 3, srow = tid >> 2;
    const i64 rowLen = (i64)p.L0 * p.Cin;
    const float *aBase[AR];
    int aIn1[AR], aE0[AR];
    bool aRowOk[AR];
    float aMean[AR], aScale[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i)
    {
        const int4 ri = rowinfo[srow + i * 64];
        aRowOk[i] = ri.w >= 0;
        aBase[i] = p.X + (i64)ri.x * p.xBS;
        aIn1[i] = ri.y * p.stride1 - p.pad1;
        aE0[i] = (ri.z * p.stride0 - p.pad0) * p.Cin + skq * 4;
        aMean[i] = 0.f, aScale[i] = 1.f;
        if (PRO == PRO_AFFINE && aRowOk[i])
        {
            aMean[i] = p.proStats[ri.x * 4];
            aScale[i] = p.proStats[ri.x * 4 + 1];
        }
        if (PRO == PRO_GN_GELU && aRowOk[i])
        {
            aMean[i] = p.proStats[ri.w * 4];
            aScale[i] = p.proStats[ri.w * 4 + 1];
        }
    }
    const float *bBase[BR];
    bool bRowOk[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i)
    {
        const int rl = srow + i * 64;
        const int n = n0 + rl;
        bRowOk[i] = rl < BN && n < p.Np;
        bBase[i] = p.Wt + (i64)(bRowOk[i] ? n : 0) * p.Kp + skq * 4;
    }

    float4 aReg[KS][AR], bReg[KS][BR], gW[KS], gB[KS];
    unsigned aMask = 0, bMask = 0;
    const int nk16 = p.Kp >> 4;
    const int nk = (nk16 + KS - 1) / KS;

    // issue the global loads of K-tile kt: no branches, no use of the loaded values
    auto load_tiles = [&](int kt) {
        aMask = 0;
        bMask = 0;
#pragma unroll
        for (int ch = 0; ch < KS; ++ch)
        {
            const int kbase = (kt * KS + ch) << 4;
            const bool chOk = kbase < p.Kp;
            int s1 = 0, offb = kbase;
            if (p.S1 > 1)
            {
                s1 = kbase / p.seg0; // scalar
                offb = kbase - s1 * p.seg0;
            }
            const int k = kbase + skq * 4;
#pragma unroll
            for (int i = 0; i < AR; ++i)
            {
                const int in1 = aIn1[i] + s1 * p.dil1;
                const i64 e = (i64)aE0[i] + offb;
                const bool ok = chOk && aRowOk[i] && k < p.K && in1 >= 0 && in1 < p.L1 && e >= 0 && e < rowLen;
                const float *src = ok ? aBase[i] + (i64)in1 * rowLen + e : p.X;
                aReg[ch][i] = *reinterpret_cast<const float4 *>(src);
                aMask |= (ok ? 1u : 0u) << (ch * AR + i);
            }
#pragma unroll
            for (int i = 0; i < BR; ++i)
            {
                const bool ok = chOk && bRowOk[i];
                bReg[ch][i] = *reinterpret_cast<const float4 *>(ok ? bBase[i] + kbase : p.Wt);
                bMask |= (ok ? 1u : 0u) << (ch * BR + i);
            }
            if (PRO == PRO_GN_GELU)
            {
                const int kk = (chOk && k < p.K) ? k : 0;
                gW[ch] = *reinterpret_cast<const float4 *>(p.proW + kk);
                gB[ch] = *reinterpret_cast<const float4 *>(p.proB + kk);
            }
        }
    };
    // prologue transform + zero fill + LDS write of the staged tile
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int ch = 0; ch < KS; ++ch)
        {
#pragma unroll
            for (int i = 0; i < AR; ++i)
            {
                float4 v = aReg[ch][i];
                const bool ok = (aMask >> (ch * AR + i)) & 1u;
                if (PRO == PRO_AFFINE)
                {
                    v.x = (v.x - aMean[i]) * aScale[i];
                    v.y = (v.y - aMean[i]) * aScale[i];
                    v.z = (v.z - aMean[i]) * aScale[i];
                    v.w = (v.w - aMean[i]) * aScale[i];
                }
                if (PRO == PRO_GN_GELU)
                {
                    v.x = gelu_f((v.x - aMean[i]) * aScale[i] * gW[ch].x + gB[ch].x);
                    v.y = gelu_f((v.y - aMean[i]) * aScale[i] * gW[ch].y + gB[ch].y);
                    v.z = gelu_f((v.z - aMean[i]) * aScale[i] * gW[ch].z + gB[ch].z);
                    v.w = gelu_f((v.w - aMean[i]) * aScale[i] * gW[ch].w + gB[ch].w);
                }
                if (!ok)
                    v = make_float4(0.f, 0.f, 0.f, 0.f);
                As[buf][ch * 4 + skq][srow + i * 64] = v;
            }
#pragma unroll
            for (int i = 0; i < BR; ++i)
            {
                const int rl = srow + i * 64;
                float4 v = bReg[ch][i];
                if (!((bMask >> (ch * BR + i)) & 1u))
                    v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rl < BN)
                    Bs[buf][ch * 4 + skq][rl] = v;
            }
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
#pragma unroll
        for (int ch = 0; ch < KS; ++ch)
        {
            float4 a[WMF], b[WNF];
#pragma unroll
            for (int i = 0; i < WMF; ++i)
                a[i] = As[cur][ch * 4 + kq][wm * (WMF * 16) + i * 16 + l15];
#pragma unroll
            for (int j = 0; j < WNF; ++j)
                b[j] = Bs[cur][ch * 4 + kq][wn * (WNF * 16) + j * 16 + l15];
            // k sub-step outermost: consecutive MFMAs hit DIFFERENT accumulators (the 16x16x4 f32
            // MFMA has a 40-cycle dependent latency vs a 32-cycle issue interval)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < WMF; ++i)
#pragma unroll
                    for (int j = 0; j < WNF; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(a[i], c), f4c(b[j], c), acc[i][j], 0, 0, 0);
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
    int trR[WNF], trC[WNF]; // EPI_TRCONV: column n -> (phase r, channel co)
#pragma unroll
    for (int j = 0; j < WNF; ++j)
    {
        const int n = colBase + j * 16;
        biasv[j] = n < p.N ? p.bias[n] : 0.f;
        trR[j] = trC[j] = 0;
        if (EPI == EPI_TRCONV)
        {
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
            if (EPI == EPI_LINEAR || EPI == EPI_SCALE_RES || EPI == EPI_STATS_ONLY)
            {
#pragma unroll
                for (int j = 0; j < WNF; ++j)
                {
                    const int n = colBase + j * 16;
                    float v = acc[i][j][r] + biasv[j];
                    if (rowOk && n < p.N)
                    {
                        if (EPI == EPI_LINEAR)
                        {
                            if (p.act)
                                v = gelu_f(v);
                            const i64 o = m * p.ldy + n;
                            if (p.res)
                                v += p.res[o];
                            p.Y[o] = v;
                        }
                        else if (EPI == EPI_SCALE_RES)
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
            else if (EPI == EPI_GLU || EPI == EPI_GN_GLU_SCALE_RES)
            {
                if constexpr (WNF % 2 == 0)
                {
                    float mean = 0.f, sc = 1.f;
                    if (EPI == EPI_GN_GLU_SCALE_RES && rowOk)
                    {
                        mean = p.epiStats[ri.w * 4];
                        sc = p.epiStats[ri.w * 4 + 1];
                    }
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
                            if (EPI == EPI_GN_GLU_SCALE_RES)
                            {
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
                    const int jj = 4 * ri.z + trR[j] - 2;
                    if (rowOk && n < p.N && jj >= 0 && jj < p.Lout)
                    {
                        float v = acc[i][j][r] + biasv[j];
                        if (p.act)
                            v = gelu_f(v);
                        const i64 o = (i64)ri.x * p.yBS + ((i64)ri.y * p.Lout + jj) * p.ldy + trC[j];
                        if (p.res)
                            v += p.res[o];
                        p.Y[o] = v;
                    }
                }
            }
            if (EPI == EPI_LINEAR || EPI == EPI_SCALE_RES || EPI == EPI_STATS_ONLY)
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
    if (EPI == EPI_LINEAR || EPI == EPI_SCALE_RES || EPI == EPI_STATS_ONLY)
        if (wantStats)
        {
            __syncthreads();
            for (int r = tid; r < BM; r += 256)
            {
                const i64 m = m0 + r;
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

template <int WM_, int WN_, int MF, int NF, int KS, int PRO, int EPI>
static void launch_one(const GemmArgs &a, hipStream_t s)
{
    constexpr int BM = WM_ * MF * 16, BN = WN_ * NF * 16;
    dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)((a.N + BN - 1) / BN));
    hipLaunchKernelGGL((igemm_kernel<WM_, WN_, MF, NF, KS, PRO, EPI>), grid, dim3(256), 0, s, a);
}

// Instantiated (tile, prologue, epilogue) combinations = exactly what plan.cpp emits for the
// 4- and 6-source models at any segment length (enumerated with tests/cpu_interp.cpp
// interp_combos). key = cfg*100 + pro*10 + epi.
int launch_igemm(int cfg, const GemmArgs &a, hipStream_t s, bool dry)
{
#define DMX_CASE(cfgid, WM_, WN_, MF, NF, KS, PRO, EPI) \
    case (cfgid * 100 + PRO * 10 + EPI):                \
        if (!dry)                                       \
            launch_one<WM_, WN_, MF, NF, KS, PRO, EPI>(a, s); \
        return 0;
    switch (cfg * 100 + a.pro * 10 + a.epi)
    {
        // cfg 0: 128x128, cfg 7: 64x128 (same column decomposition)
        DMX_CASE(0, 2, 2, 4, 4, 2, PRO_NONE, EPI_LINEAR)
        DMX_CASE(0, 2, 2, 4, 4, 2, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(0, 2, 2, 4, 4, 2, PRO_NONE, EPI_GLU)
        DMX_CASE(0, 2, 2, 4, 4, 2, PRO_NONE, EPI_TRCONV)
        DMX_CASE(0, 2, 2, 4, 4, 2, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES)
        DMX_CASE(0, 2, 2, 4, 4, 2, PRO_GN_GELU, EPI_STATS_ONLY)
        DMX_CASE(7, 2, 2, 2, 4, 2, PRO_NONE, EPI_LINEAR)
        DMX_CASE(7, 2, 2, 2, 4, 2, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(7, 2, 2, 2, 4, 2, PRO_NONE, EPI_GLU)
        DMX_CASE(7, 2, 2, 2, 4, 2, PRO_NONE, EPI_TRCONV)
        DMX_CASE(7, 2, 2, 2, 4, 2, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES)
        DMX_CASE(7, 2, 2, 2, 4, 2, PRO_GN_GELU, EPI_STATS_ONLY)
        // cfg 2: 128x96
        DMX_CASE(2, 4, 1, 2, 6, 1, PRO_NONE, EPI_LINEAR)
        DMX_CASE(2, 4, 1, 2, 6, 1, PRO_NONE, EPI_GLU)
        DMX_CASE(2, 4, 1, 2, 6, 1, PRO_NONE, EPI_TRCONV)
        DMX_CASE(2, 4, 1, 2, 6, 1, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES)
        DMX_CASE(2, 4, 1, 2, 6, 1, PRO_GN_GELU, EPI_STATS_ONLY)
        // cfg 3: 128x48
        DMX_CASE(3, 4, 1, 2, 3, 1, PRO_NONE, EPI_LINEAR)
        DMX_CASE(3, 4, 1, 2, 3, 1, PRO_NONE, EPI_TRCONV)
        DMX_CASE(3, 4, 1, 2, 3, 1, PRO_AFFINE, EPI_LINEAR)
        // cfg 4: 256x16, cfg 5: 128x32, cfg 6: 128x64
        DMX_CASE(4, 4, 1, 4, 1, 1, PRO_NONE, EPI_LINEAR)
        DMX_CASE(5, 4, 1, 2, 2, 1, PRO_NONE, EPI_LINEAR)
        DMX_CASE(5, 4, 1, 2, 2, 1, PRO_NONE, EPI_TRCONV)
        DMX_CASE(6, 4, 1, 2, 4, 1, PRO_NONE, EPI_TRCONV)
        DMX_CASE(6, 4, 1, 2, 4, 1, PRO_NONE, EPI_LINEAR)
    default:
        return -1;
    }
#undef DMX_CASE
}

} // namespace dmx

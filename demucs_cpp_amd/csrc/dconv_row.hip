// dconv_row.hip — the frequency branch's DConv residual branch with the (C, T) row RESIDENT on the CU (gfx950).
//
// /root/reference/src/layers.cpp:152-375 (apply_dconv) is called by the frequency encoders / decoders with the freq bins
// as the batch (src/encdec.cpp:43-45,203-207): both GroupNorms of a layer take their statistics over ONE (channels, T)
// row of one bin. The op chain K1 -> r1 -> K2 -> r2 -> K3 (plan.cpp Builder::dconv, dgemm.hip) makes three passes over the
// tensor per layer - six per DConv - and is bound by exactly those bytes (61 GB per 42-segment step, 4.5 TB/s). Here ONE
// workgroup owns one row (segment b, bin f) for the whole DConv:
//   * x[b][.][f][.] (T x C floats: 64.5 KB at C = 48, 129 KB at C = 96) is read ONCE from the channels-last tensor straight
//     into registers in the MFMA operand layout (lane (row = lane & 15, h = lane >> 4) holds channels 16 j + 4 h + {0..3} of
//     time step t0 + row), 3 fragments of 16 time steps per wave, 7 waves at T = 336;
//   * K1 (Conv1d C -> C/8, k3, dilation d) runs as ONE product per fragment against the three taps' weights side by side
//     (columns s HR + j: 18 or 36 wide instead of three 16-wide tap products): P[t][s HR + j] = W_s[j] . x[t]; the taps
//     meet in LDS: h[t][j] = b[j] + P[t-d][j] + P[t][HR + j] + P[t+d][2 HR + j];
//   * GroupNorm(1, C/8) statistics in the workgroup (fp64 partials, fixed order), GELU, hn in place in LDS;
//   * GroupNorm(1, 2C) statistics of y = W2 hn + b2 through the factor [L; u; v] of the packed model (plan.h EPI_STATS_FACT):
//     one 16-column product per fragment instead of 2C columns;
//   * K3: y pair by pair (GLU halves are interleaved per 16 columns) -> GroupNorm -> GLU -> LayerScale -> added to the x
//     registers: the accumulator layout of the swapped-operand MFMA IS the operand layout x is held in;
//   * second layer (dilation 2) on the same registers, then x is written ONCE.
// Two passes over the tensor instead of six, no statistics launches, no hidden tensor in HBM. Every reduction has a fixed
// order inside the workgroup, so results do not depend on batch, sharding or stream schedule.
// Semantics: plan.h OP_DCONV_ROW, executable in tests/cpu_interp.cpp.
#include "kernels.h"

namespace dmx
{

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace
{
__device__ __forceinline__ float rsigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// geometry shared by the kernel and its launcher
template <int C, int HR>
struct RowGeo
{
    static constexpr int HP = (HR + 3) / 4 * 4;   // hidden width padded to the packed model's C8p
    static constexpr int RPL = HP / 4;            // K3 / k2f: k = RPL * h + c, c < RPL (dgemm.hip's remainder layout)
    static constexpr int NP = 3 * HR;             // tap-product columns
    static constexpr int PS = (NP + 3) / 4 * 4;   // P row stride (floats)
    static constexpr int NPF = (NP + 15) / 16;    // column fragments of the tap product
    static constexpr int WS = C + 8;              // Wp row stride: conflict-free ds_read_b128 of the MFMA A operand
    // LDS image (floats)
    static constexpr int oWp = 0;                       // [NP][WS]
    static constexpr int oW3 = oWp + NP * WS;           // [RPL][2C][4]
    static constexpr int oLf = oW3 + RPL * 2 * C * 4;   // [RPL][16][4]
    static constexpr int oCst = oLf + RPL * 16 * 4;     // k2 bias | gn2 w | gn2 b (2C each, packed order) | LayerScale (C) | k1 bias, gn1 w, gn1 b, k2f bias (16 each)
    static constexpr int oRed = oCst + 7 * C + 64;      // 2 slots x 16 waves x 2 doubles = 128 floats
    static constexpr int oP = oRed + 128;               // [T][PS]; the centre tap's slots become h, then hn
    static size_t lds_bytes(int T) { return (size_t)(oP + (size_t)T * PS) * sizeof(float); }
};

// sum of (s, q) over the workgroup in a fixed order: lanes by xor-shuffle, waves in index order
__device__ __forceinline__ void block_sum2(double &s, double &q, double *red, int w, int lane, int nw)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        s += __shfl_xor(s, off);
        q += __shfl_xor(q, off);
    }
    if (lane == 0)
    {
        red[2 * w] = s;
        red[2 * w + 1] = q;
    }
    __syncthreads();
    double S = 0.0, Q = 0.0;
    for (int i = 0; i < nw; ++i)
    {
        S += red[2 * i];
        Q += red[2 * i + 1];
    }
    s = S, q = Q;
}
} // namespace

template <int C, int HR, int FPW, int MINW>
__global__ __launch_bounds__(1024, MINW) void dconv_row_kernel(const DconvRowArgs p)
{
    using G = RowGeo<C, HR>;
    constexpr int HP = G::HP, RPL = G::RPL, NP = G::NP, PS = G::PS, NPF = G::NPF, WS = G::WS, NJ = C / 16;
    extern __shared__ float lds[];
    float *Wp = lds + G::oWp, *W3 = lds + G::oW3, *Lf = lds + G::oLf, *cst = lds + G::oCst, *P = lds + G::oP;
    double *red = reinterpret_cast<double *>(lds + G::oRed);
    float *cB2 = cst, *cGw = cst + 2 * C, *cGb = cst + 4 * C, *cSc = cst + 6 * C, *cK1b = cst + 7 * C, *cG1w = cK1b + 16, *cG1b = cK1b + 32,
          *cFb = cK1b + 48;

    const int tid0 = threadIdx.x, nthr = blockDim.x, nw = nthr >> 6;
    const int T = p.T, F = p.F;
    // XCD-aware row map: workgroup ids are dealt round-robin to the 8 XCDs; XCD x walks the contiguous rows
    // [x * rowsPerXcd, (x + 1) * rowsPerXcd) in dispatch order, so the bins that share a 128-byte line (C = 48: a bin is
    // 192 bytes per time step) are resident on one XCD at the same time
    const int rows = p.B * F;
    const int row = (int)(blockIdx.x & 7) * p.rowsPerXcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= p.rowsPerXcd || row >= rows)
        return;
    const int b = row / F, f = row - b * F;
    float *xrow = p.x + ((i64)b * T * F + f) * C + 4 * ((tid0 & 63) >> 4); // + t * F * C + 16 j

    // ---- x -> registers (MFMA B-operand order), once
    f32x4 xr[FPW][NJ];
    bool tOk[FPW];
#pragma unroll
    for (int i = 0; i < FPW; ++i)
    {
        const int t = ((tid0 >> 6) * FPW + i) * 16 + (tid0 & 15);
        tOk[i] = t < T;
        const float *src = tOk[i] ? xrow + (i64)t * F * C : p.zero;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            xr[i][j] = *reinterpret_cast<const f32x4 *>(tOk[i] ? src + 16 * j : src);
    }

#pragma unroll 1
    for (int layer = 0; layer < 2; ++layer)
    {
        const int d = layer + 1;
        // the LDS addresses below are functions of the thread id alone: hidden from loop-invariant code motion, which would
        // otherwise keep (and spill) some forty of them across the layer loop beside the row's own registers
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int w = tid >> 6, lane = tid & 63, l15 = lane & 15, h = lane >> 4;
        // ---- this layer's weights and constants -> LDS
        {
            const float *k1w = p.k1w[layer], *k2w = p.k2w[layer], *k2fw = p.k2fw[layer];
            for (int i = tid; i < NP * (C / 4); i += nthr) // Wp[s HR + j][k] = k1.Wt[j][s C + k]  (k1.Wt: [16][3C])
            {
                const int n = i / (C / 4), k4 = i - n * (C / 4), s = n / HR, j = n - s * HR;
                *reinterpret_cast<f32x4 *>(Wp + n * WS + 4 * k4) = *reinterpret_cast<const f32x4 *>(k1w + (i64)j * (3 * C) + s * C + 4 * k4);
            }
            for (int i = tid; i < 2 * C + 16; i += nthr) // planes [c][row][h] = W[row][RPL h + c]  (k2.Wt: [2C][16]; k2f.Wt: [16][16])
            {
                const bool fact = i >= 2 * C;
                const int r = fact ? i - 2 * C : i;
                const float *src = (fact ? k2fw : k2w) + (i64)r * 16;
                float *dst = fact ? Lf : W3;
                const int nrow = fact ? 16 : 2 * C;
                // (element-wise: a register-level regrouping into 16-byte stores compiles to v_pk_mov_b32 with half routing,
                // an instruction class this library keeps out of its code objects - tests/test_isa_rules.py)
#pragma unroll
                for (int q = 0; q < HP / 4; ++q)
                {
                    const f32x4 u = *reinterpret_cast<const f32x4 *>(src + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                    {
                        const int k = 4 * q + e, hh = k / RPL, c = k - hh * RPL;
                        dst[(c * nrow + r) * 4 + hh] = u[e];
                    }
                }
            }
            for (int i = tid; i < 2 * C; i += nthr)
            {
                cB2[i] = p.k2b[layer][i];
                cGw[i] = p.gn2w[layer][i];
                cGb[i] = p.gn2b[layer][i];
            }
            for (int i = tid; i < C; i += nthr)
                cSc[i] = p.scale[layer][i];
            if (tid < 16)
            {
                cK1b[tid] = p.k1b[layer][tid];
                cG1w[tid] = p.gn1w[layer][tid];
                cG1b[tid] = p.gn1b[layer][tid];
                cFb[tid] = p.k2fb[layer][tid];
            }
        }
        __syncthreads();

        // ---- K1 as the tap product P[t][s HR + j] = W_s[j] . x[t]
#pragma unroll
        for (int pf = 0; pf < NPF; ++pf)
        {
            f32x4 acc[FPW];
#pragma unroll
            for (int i = 0; i < FPW; ++i)
                acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int wrow = min(16 * pf + l15, NP - 1); // rows >= NP feed columns nobody reads
#pragma unroll
            for (int j = 0; j < NJ; ++j)
            {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(Wp + wrow * WS + 16 * j + 4 * h);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < FPW; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c], xr[i][j][c], acc[i], 0, 0, 0);
            }
            if (16 * pf + 4 * h < PS)
#pragma unroll
                for (int i = 0; i < FPW; ++i)
                    if (tOk[i])
                        *reinterpret_cast<f32x4 *>(P + ((w * FPW + i) * 16 + l15) * PS + 16 * pf + 4 * h) = acc[i];
        }
        __syncthreads();

        // ---- taps meet: h[t][j], kept in the centre tap's slot P[t][HR + j] (read by this item only); GroupNorm(1, C/8)
        // statistics over the row (HR * T values). Rolled loops: an unrolled GELU per item costs more registers than the row.
        double s1 = 0.0, q1 = 0.0;
#pragma unroll 1
        for (int idx = tid; idx < T * HR; idx += nthr)
        {
            const int t = idx / HR, j = idx - t * HR;
            float v = cK1b[j] + P[t * PS + HR + j];
            if (t - d >= 0)
                v += P[(t - d) * PS + j];
            if (t + d < T)
                v += P[(t + d) * PS + 2 * HR + j];
            P[t * PS + HR + j] = v;
            s1 += (double)v;
            q1 += (double)v * (double)v;
        }
        block_sum2(s1, q1, red + (0 * 32), w, lane, nw);
        float mean1, rstd1;
        {
            const double n1 = (double)HR * T, mean = s1 / n1;
            double var = (q1 - n1 * mean * mean) / (n1 - 1.0); // unbiased (Q3)
            var = var < 0.0 ? 0.0 : var;
            mean1 = (float)mean;
            rstd1 = (float)(1.0 / sqrt(var + (double)p.eps));
        }
        // hn = gelu(gn(h)) in place. K3 / k2f read their operand k = RPL h + c at P[t][HR + k]; for HR <= k < HP that is a
        // tap-2 product (finite) and meets a zero weight column
#pragma unroll 1
        for (int idx = tid; idx < T * HR; idx += nthr)
        {
            const int t = idx / HR, j = idx - t * HR;
            P[t * PS + HR + j] = dmx_gelu((P[t * PS + HR + j] - mean1) * rstd1 * cG1w[j] + cG1b[j]);
        }
        __syncthreads();

        // ---- GroupNorm(1, 2C) statistics of y = W2 hn + b2 through the factor (EPI_STATS_FACT): z = [L; u; v] hn + bias
        float hb[FPW][RPL];
#pragma unroll
        for (int i = 0; i < FPW; ++i)
        {
            const int t = min((w * FPW + i) * 16 + l15, T - 1);
#pragma unroll
            for (int c = 0; c < RPL; ++c)
                hb[i][c] = P[t * PS + HR + RPL * h + c];
        }
        double s2 = 0.0, q2 = 0.0;
        {
            float la[RPL];
#pragma unroll
            for (int c = 0; c < RPL; ++c)
                la[c] = Lf[(c * 16 + l15) * 4 + h];
            const f32x4 fb = *reinterpret_cast<const f32x4 *>(cFb + 4 * h);
#pragma unroll
            for (int i = 0; i < FPW; ++i)
            {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < RPL; ++c)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(la[c], hb[i][c], acc, 0, 0, 0);
                float s = 0.f, ss = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                {
                    const int nn = 4 * h + r;
                    const float z = acc[r] + fb[r];
                    ss += nn < HP ? z * z : (nn == HP + 1 ? 2.0f * z : 0.f);
                    s += nn == HP ? z : 0.f;
                }
                if (tOk[i])
                {
                    s2 += (double)s;
                    q2 += (double)ss;
                }
            }
        }
        block_sum2(s2, q2, red + (1 * 32), w, lane, nw);
        float mean2, rstd2;
        {
            const double n2 = 2.0 * C * T, mean = s2 / n2;
            double var = (q2 - n2 * mean * mean) / (n2 - 1.0);
            var = var < 0.0 ? 0.0 : var;
            mean2 = (float)mean;
            rstd2 = (float)(1.0 / sqrt(var + (double)p.eps));
        }

        // ---- K3: y pair by pair -> GroupNorm -> GLU -> LayerScale -> += into the x registers. The GroupNorm affine of a
        // column is folded with its statistics: gn(acc + b2) = acc * (rstd gw) + ((b2 - mean) (rstd gw) + gb). The value half of
        // a pair is finished for all fragments before the gate half's constants are loaded (register budget: the row itself
        // holds 72 of the 128 registers at C = 96).
#pragma unroll
        for (int pp = 0; pp < NJ; ++pp)
        {
            f32x4 av[FPW];
#pragma unroll
            for (int half = 0; half < 2; ++half)
            {
                const int n0 = 32 * pp + 16 * half + 4 * h;
                float wk[RPL];
#pragma unroll
                for (int c = 0; c < RPL; ++c)
                    wk[c] = W3[(c * 2 * C + 32 * pp + 16 * half + l15) * 4 + h];
                f32x4 a1 = *reinterpret_cast<const f32x4 *>(cGw + n0);
                f32x4 a0 = *reinterpret_cast<const f32x4 *>(cB2 + n0);
                const f32x4 gb = *reinterpret_cast<const f32x4 *>(cGb + n0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                {
                    a1[r] = rstd2 * a1[r];
                    a0[r] = fmaf(a0[r] - mean2, a1[r], gb[r]);
                }
                f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f};
                if (half)
                    sv = *reinterpret_cast<const f32x4 *>(cSc + 16 * pp + 4 * h);
#pragma unroll
                for (int i = 0; i < FPW; ++i)
                {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < RPL; ++c)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[c], hb[i][c], acc, 0, 0, 0);
                    if (!half)
                    {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            av[i][r] = fmaf(acc[r], a1[r], a0[r]);
                    }
                    else
                    {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            xr[i][pp][r] += sv[r] * (av[i][r] * rsigmoid(fmaf(acc[r], a1[r], a0[r])));
                    }
                }
            }
        }
        __syncthreads(); // the next layer overwrites the weight images and P
    }

    // ---- x back, once
#pragma unroll
    for (int i = 0; i < FPW; ++i)
        if (tOk[i])
        {
            float *dst = xrow + (i64)(((tid0 >> 6) * FPW + i) * 16 + (tid0 & 15)) * F * C;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                *reinterpret_cast<f32x4 *>(dst + 16 * j) = xr[i][j];
        }
}

// C = 48 / 96 with hidden C/8 (HTDemucs v4) and C = 48 with hidden C/4 (Demucs v3); T up to 16 waves x 3 fragments and the
// row's LDS image within 160 KB. Returns 0, or -1 when no kernel exists for the shape (the plan keeps the K1/K2/K3 chain).
int launch_dconv_row(const DconvRowArgs &a, hipStream_t s, bool dry)
{
    constexpr int FPW = 3;
    const int nfrag = (a.T + 15) / 16, nw = (nfrag + FPW - 1) / FPW;
    if (a.T < 2 || nw > 16 || (i64)a.B * a.T * a.F * a.C >= (1ll << 31))
        return -1;
    auto go = [&](auto kern, size_t smem) -> int {
        if (smem > 160 * 1024)
            return -1;
        if (dry)
            return 0;
        DconvRowArgs k = a;
        const int rows = a.B * a.F;
        k.rowsPerXcd = (rows + 7) / 8;
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3(8 * k.rowsPerXcd), dim3(64 * nw), smem, s, k);
        return 0;
    };
    if (a.C == 48 && a.hid == 6)
        return go(dconv_row_kernel<48, 6, FPW, 6>, RowGeo<48, 6>::lds_bytes(a.T));
    if (a.C == 96 && a.hid == 12)
        return go(dconv_row_kernel<96, 12, FPW, 4>, RowGeo<96, 12>::lds_bytes(a.T));
    if (a.C == 48 && a.hid == 12)
        return go(dconv_row_kernel<48, 12, FPW, 6>, RowGeo<48, 12>::lds_bytes(a.T));
    return -1;
}

} // namespace dmx

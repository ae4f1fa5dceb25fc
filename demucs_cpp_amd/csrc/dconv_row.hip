// dconv_row.hip — the frequency branch's DConv residual branch with the (C, T) row RESIDENT on the CU (gfx950).
//
// /root/reference/src/layers.cpp:152-375 (apply_dconv) is called by the frequency encoders / decoders with the freq bins
// as the batch (src/encdec.cpp:43-45,203-207): both GroupNorms of a layer take their statistics over ONE (channels, T)
// row of one bin. The op chain K1 -> r1 -> K2 -> r2 -> K3 (plan.cpp Builder::dconv, dgemm.hip) makes three passes over the
// tensor per layer - six per DConv - and is bound by exactly those bytes (61 GB per 42-segment step, 4.5 TB/s). Here a
// workgroup owns one row (segment b, bin f) at a time for the whole DConv and walks rows persistently:
//   * x[b][.][f][.] (T x C floats: 64.5 KB at C = 48, 129 KB at C = 96) is read ONCE from the channels-last tensor straight
//     into registers in the MFMA operand layout (lane (row = lane & 15, h = lane >> 4) holds channels 16 j + 4 h + {0..3} of
//     time step t0 + row), 3 fragments of 16 time steps per wave, 7 waves at T = 336;
//   * K1 (Conv1d C -> C/8, k3, dilation d) runs as ONE product per fragment against the three taps' weights side by side,
//     its columns ordered so that lane (row, h) ends up with, for ITS time step, the centre tap's hidden units RPL h + c
//     (c < RPL = hidden / 4) and the same units of the two outer taps; the outer taps' products go to LDS (P0[t][j],
//     P2[t][j]) and come back from t - d / t + d: h[t][RPL h + c] lane-local, in exactly the operand order of K3;
//   * GroupNorm(1, C/8) statistics in the workgroup (fp64 partials, fixed order), GELU in registers: hn never leaves them;
//   * GroupNorm(1, 2C) statistics of y = W2 hn + b2 through the factor [L; u; v] of the packed model (plan.h EPI_STATS_FACT):
//     one 16-column product per fragment instead of 2C columns;
//   * K3: y pair by pair (GLU halves are interleaved per 16 columns) -> GroupNorm -> GLU -> LayerScale -> added to the x
//     registers: the accumulator layout of the swapped-operand MFMA IS the operand layout x is held in;
//   * second layer (dilation 2) on the same registers, then x is written ONCE.
// The weights of a layer come as ONE image in the LDS layout (model_pack.cpp "rowimg"); where both layers' images fit beside
// three workgroups' scratch (C = 48) they are staged once per workgroup, otherwise (C = 96) the K3-side parts of both layers
// are resident and the K1-side part of the next layer is fetched by LDS-DMA while the current layer computes: no global load
// sits between two phases of a row (a CU's vector-memory queue is in order: a small weight fetch would wait behind the
// neighbour workgroup's whole row).
// Two passes over the tensor instead of six, no statistics launches, no hidden tensor in HBM. Every reduction has a fixed
// order inside the workgroup, so results do not depend on batch, sharding or stream schedule.
// Semantics: plan.h OP_DCONV_ROW, executable in tests/cpu_interp.cpp.
#include "igemm_common.h"

namespace dmx
{

namespace
{
// geometry shared by the kernel and its launcher; plan.cpp dconv_row_image_floats / dconv_row_lds_bytes restate it for
// model_pack.cpp and the plan builder (checked by the launcher)
template <int C, int HR>
struct RowGeo
{
    static constexpr int HP = (HR + 3) / 4 * 4;   // hidden width padded to the packed model's C8p
    static constexpr int RPL = HP / 4;            // K3 / k2f: k = RPL * h + c, c < RPL (dgemm.hip's remainder layout)
    static constexpr int NQ = 3 * RPL;            // tap-product values per lane and time step: slot q = o RPL + c, o = centre | tap 0 | tap 2
    static constexpr int NPF = (NQ + 3) / 4;      // column fragments of the tap product: slot q is column 16 (q / 4) + 4 h + q % 4
    static constexpr int NP = 16 * NPF;           // image rows (columns of the product), unused ones zero
    static constexpr int WS = C + 8;              // Wp row stride: conflict-free ds_read_b128 of the MFMA A operand
    // weight image of one layer (floats): [K1 side: Wp | K3 side: W3 planes, Lf planes, constants], each part whole 1 KB
    // pieces (LDS-DMA moves 1 KB per wave instruction)
    static constexpr int nWp = (NP * WS + 255) / 256 * 256;
    static constexpr int oW3 = 0;                            // offsets inside the K3-side part: [RPL][2C][4],
    static constexpr int oLf = oW3 + RPL * 2 * C * 4;        // [RPL][16][4],
    static constexpr int oCst = oLf + RPL * 16 * 4;          // k2 bias | gn2 w | gn2 b (2C each, packed order) | LayerScale (C) | k1 bias, gn1 w, gn1 b, k2f bias (16 each)
    static constexpr int nWk = (oCst + 7 * C + 64 + 255) / 256 * 256;
    // both layers' K1-side parts resident? (three workgroups per CU beside them at C = 48; not at C = 96)
    static constexpr bool RES = C == 48;
    // LDS image (floats)
    static constexpr int lWp = 0;                            // [RES ? 2 : 1][nWp]
    static constexpr int lWk = lWp + (RES ? 2 : 1) * nWp;    // [2][nWk]
    static constexpr int lRed = lWk + 2 * nWk;               // 2 slots x 16 waves x 2 doubles = 128 floats
    static constexpr int lP = lRed + 128;                    // P0 [T][HP], P2 [T][HP]
    static size_t lds_bytes(int T) { return (size_t)(lP + (size_t)2 * T * HP) * sizeof(float); }
};

// v + (v of the lane n places to the left in its 16-lane row, 0 beyond the row's start): one step of a row prefix sum on the
// VALU's DPP path (a double moves as two dwords); no LDS traffic, unlike ds_bpermute shuffles
template <int N>
__device__ __forceinline__ double row_shr_add(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + N, 0xf, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v = row_shr_add<1>(v);
    v = row_shr_add<2>(v);
    v = row_shr_add<4>(v);
    v = row_shr_add<8>(v); // lane 15 of every row: the row's sum
    auto lane_of = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
    return ((lane_of(15) + lane_of(31)) + lane_of(47)) + lane_of(63);
}
// sum of (s, q) over the workgroup in a fixed order: lanes by the prefix steps above, waves in index order
__device__ __forceinline__ void block_sum2(double &s, double &q, double *red, int w, int lane, int nw)
{
    s = wave_sum(s);
    q = wave_sum(q);
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
// {mean, 1 / sqrt(var + eps)} of n values with sum s and sum of squares q; unbiased variance (Q3). The reciprocals of n and
// n - 1 come from the host: no fp64 division or square root on the critical path of every thread
__device__ __forceinline__ void finish_stats(double s, double q, double n, double invN, double invNm1, float eps, float &mean, float &rstd)
{
    const double m = s * invN;
    double var = (q - n * m * m) * invNm1;
    var = var < 0.0 ? 0.0 : var;
    const float v = (float)var + eps;
    float y = __builtin_amdgcn_rsqf(v);
    y = y * (1.5f - 0.5f * v * y * y); // one Newton step on the 1-ulp hardware estimate
    mean = (float)m;
    rstd = y;
}
} // namespace

template <int C, int HR, int FPW, int MINW>
__global__ __launch_bounds__(1024, MINW) void dconv_row_kernel(const DconvRowArgs p)
{
    using G = RowGeo<C, HR>;
    constexpr int HP = G::HP, RPL = G::RPL, NPF = G::NPF, WS = G::WS, NJ = C / 16;
    constexpr bool RES = G::RES;
    extern __shared__ float lds[];
    double *red = reinterpret_cast<double *>(lds + G::lRed);

    const int tid0 = threadIdx.x, nthr = blockDim.x, nw = nthr >> 6;
    const int T = p.T, F = p.F;
    float *P0 = lds + G::lP, *P2 = P0 + T * HP;

    // ---- weight images -> LDS. The K3-side parts of both layers (and, RES, the K1-side parts) once per workgroup
    {
        const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6), ln = tid0 & 63;
        for (int l = 0; l < 2; ++l)
        {
            for (int pc = wv; pc < G::nWk / 256; pc += nw)
                load_to_lds_b128(p.img[l] + G::nWp + pc * 256 + 4 * ln, reinterpret_cast<float4 *>(lds + G::lWk + l * G::nWk + pc * 256));
            if (RES || l == 0)
                for (int pc = wv; pc < G::nWp / 256; pc += nw)
                    load_to_lds_b128(p.img[l] + pc * 256 + 4 * ln, reinterpret_cast<float4 *>(lds + G::lWp + (RES ? l : 0) * G::nWp + pc * 256));
        }
    }

    if (RES)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // own pieces, then everybody's
        __syncthreads();
    }

    // persistent walk, XCD-aware: workgroup ids are dealt round-robin to the 8 XCDs; XCD x owns the contiguous rows
    // [x * rowsPerXcd, (x + 1) * rowsPerXcd) and its workgroups take them in dispatch order, so the bins that share a
    // 128-byte line (C = 48: a bin is 192 bytes per time step) are resident on one XCD at the same time
    const int rows = p.B * F, slots = (int)(gridDim.x >> 3);
    const int xcdLo = (int)(blockIdx.x & 7) * p.rowsPerXcd, xcdHi = min(xcdLo + p.rowsPerXcd, rows);
#pragma unroll 1
    for (int row = xcdLo + (int)(blockIdx.x >> 3); row < xcdHi; row += slots)
    {
        const int b = row / F, f = row - b * F;
        // the row's base is uniform (scalar registers); a lane adds a 32-bit byte offset (t F C + 4 h floats: < 2^31 bytes, checked
        // by the launcher) - one address register per fragment instead of a 64-bit pointer pair. Time steps beyond T re-read the
        // last valid one: finite values that never reach the statistics or memory (every use is guarded by tOk).
        char *xrow = reinterpret_cast<char *>(p.x + ((i64)b * T * F + f) * C);

        // ---- x -> registers (MFMA B-operand order), once. (The offsets are functions of the thread id alone; computed from a
        // laundered copy here and again at the store, they are a few VALU instructions per row instead of 64-bit pairs kept -
        // and spilled - across the whole row: scratch reloads evicted by the streaming rows were as many HBM bytes as x itself.)
        f32x4 xr[FPW][NJ];
        bool tOk[FPW];
        int tidl = tid0;
        asm volatile("" : "+v"(tidl));
#pragma unroll
        for (int i = 0; i < FPW; ++i)
        {
            const int t = ((tidl >> 6) * FPW + i) * 16 + (tidl & 15);
            tOk[i] = t < T;
            const unsigned off = ((unsigned)min(t, T - 1) * (unsigned)(F * C) + 4u * ((tidl & 63) >> 4)) * 4u;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                xr[i][j] = *reinterpret_cast<const f32x4 *>(xrow + off + 64 * j);
        }

#pragma unroll 1
        for (int layer = 0; layer < 2; ++layer)
        {
            const int d = layer + 1;
            // the LDS addresses below are functions of the thread id alone: hidden from loop-invariant code motion, which
            // would otherwise keep (and spill) dozens of them across the loops beside the row's own registers
            int tid = tid0;
            asm volatile("" : "+v"(tid));
            const int w = tid >> 6, lane = tid & 63, l15 = lane & 15, h = lane >> 4;
            const float *Wp = lds + G::lWp + (RES ? layer : 0) * G::nWp;
            const float *Wk = lds + G::lWk + layer * G::nWk;
            const float *W3 = Wk + G::oW3, *Lf = Wk + G::oLf, *cB2 = Wk + G::oCst, *cGw = cB2 + 2 * C, *cGb = cB2 + 4 * C, *cSc = cB2 + 6 * C,
                        *cK1b = cB2 + 7 * C, *cG1w = cK1b + 16, *cG1b = cK1b + 32, *cFb = cK1b + 48;
            if (!RES)
            {
                // this layer's K1-side image (first step: the K3 sides too) has landed: own pieces, then everybody's. (With the
                // images resident no barrier is needed here: this step's first LDS writes - P0 / P2 - come after the two
                // reduction barriers of the previous step, behind its last reads of them.)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }

            // ---- K1 as the tap product: column 16 (q / 4) + 4 h + q % 4 of fragment t holds slot q = o RPL + c of lane (t, h):
            // o = 0 the centre tap's hidden unit RPL h + c, o = 1 tap 0's, o = 2 tap 2's (image rows in that order)
            float v[FPW][NPF * 4];
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf)
            {
                f32x4 acc[FPW];
#pragma unroll
                for (int i = 0; i < FPW; ++i)
                    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(Wp + (16 * pf + l15) * WS + 16 * j + 4 * h);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int i = 0; i < FPW; ++i)
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c], xr[i][j][c], acc[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < FPW; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[i][4 * pf + r] = acc[i][r];
            }
            // outer taps -> LDS: P0[t][RPL h + c], P2[t][RPL h + c]
#pragma unroll
            for (int i = 0; i < FPW; ++i)
                if (tOk[i])
                {
                    const int t = (w * FPW + i) * 16 + l15;
#pragma unroll
                    for (int c = 0; c < RPL; ++c)
                    {
                        P0[t * HP + RPL * h + c] = v[i][RPL + c];
                        P2[t * HP + RPL * h + c] = v[i][2 * RPL + c];
                    }
                }
            __syncthreads();
            if (!RES)
            {
                // the K1-side image of the NEXT layer step (the other layer) by LDS-DMA behind this layer's remaining phases;
                // it is waited for at the top of that step
                const int wv = __builtin_amdgcn_readfirstlane(w);
                for (int pc = wv; pc < G::nWp / 256; pc += nw)
                    load_to_lds_b128(p.img[1 - layer] + pc * 256 + 4 * lane, reinterpret_cast<float4 *>(lds + G::lWp + pc * 256));
            }

            // ---- taps meet: h[t][RPL h + c] = centre + bias + P0[t - d] + P2[t + d]; GroupNorm(1, C/8) statistics over the row
            // (pad units j >= HR have zero weights and bias: h = 0 there, the count is HR * T)
            float hb[FPW][RPL];
            double s1 = 0.0, q1 = 0.0;
#pragma unroll
            for (int i = 0; i < FPW; ++i)
            {
                const int t = (w * FPW + i) * 16 + l15;
                const bool lo = tOk[i] && t - d >= 0, hi = tOk[i] && t + d < T;
                float s = 0.f, ss = 0.f;
#pragma unroll
                for (int c = 0; c < RPL; ++c)
                {
                    float a = 0.f, e = 0.f;
                    if (lo)
                        a = P0[(t - d) * HP + RPL * h + c];
                    if (hi)
                        e = P2[(t + d) * HP + RPL * h + c];
                    const float u = tOk[i] ? (v[i][c] + cK1b[RPL * h + c]) + (a + e) : 0.f;
                    hb[i][c] = u;
                    s += u;
                    ss += u * u;
                }
                s1 += (double)s;
                q1 += (double)ss;
            }
            block_sum2(s1, q1, red + (0 * 32), w, lane, nw);
            float mean1, rstd1;
            finish_stats(s1, q1, p.n1, p.invN1, p.invN1m1, p.eps, mean1, rstd1);
            // hn = gelu(gn(h)) in registers = the MFMA operand of K3 / k2f (k = RPL h + c); rows beyond T stay 0
#pragma unroll
            for (int c = 0; c < RPL; ++c)
            {
                const float gw = cG1w[RPL * h + c], gb = cG1b[RPL * h + c];
#pragma unroll
                for (int i = 0; i < FPW; ++i)
                    hb[i][c] = tOk[i] ? dmx_gelu((hb[i][c] - mean1) * rstd1 * gw + gb) : 0.f;
            }

            // ---- GroupNorm(1, 2C) statistics of y = W2 hn + b2 through the factor (EPI_STATS_FACT): z = [L; u; v] hn + bias
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
            finish_stats(s2, q2, p.n2, p.invN2, p.invN2m1, p.eps, mean2, rstd2);

            // ---- K3: y pair by pair -> GroupNorm -> GLU -> LayerScale -> += into the x registers. The GroupNorm affine of a
            // column is folded with its statistics: gn(acc + b2) = acc * (rstd gw) + ((b2 - mean) (rstd gw) + gb). The value half
            // of a pair is finished for all fragments before the gate half's constants are loaded (register budget: the row
            // itself holds 72 of the 128 registers at C = 96).
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
                    // value half: LayerScale folded in; gate half: -log2(e) folded in, so that sigmoid(g) = 1 / (1 + exp2(acc a1 + a0))
                    f32x4 fs;
                    if (half)
                        fs = f32x4{-1.44269504088896341f, -1.44269504088896341f, -1.44269504088896341f, -1.44269504088896341f};
                    else
                        fs = *reinterpret_cast<const f32x4 *>(cSc + 16 * pp + 4 * h);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                    {
                        a1[r] = rstd2 * a1[r];
                        a0[r] = fmaf(a0[r] - mean2, a1[r], gb[r]) * fs[r];
                        a1[r] *= fs[r];
                    }
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
                                xr[i][pp][r] = fmaf(av[i][r], __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(acc[r], a1[r], a0[r]))), xr[i][pp][r]);
                        }
                    }
                }
            }
            // (no barrier here: the next LDS writes - the next K1's P0 / P2 - come after both reduction barriers of this step)
        }

        // ---- x back, once
        int tids = tid0;
        asm volatile("" : "+v"(tids));
#pragma unroll
        for (int i = 0; i < FPW; ++i)
            if (tOk[i])
            {
                const unsigned off = ((unsigned)(((tids >> 6) * FPW + i) * 16 + (tids & 15)) * (unsigned)(F * C) + 4u * ((tids & 63) >> 4)) * 4u;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    *reinterpret_cast<f32x4 *>(xrow + off + 64 * j) = xr[i][j];
            }
    }
}

// C = 48 / 96 with hidden C/8 (HTDemucs v4) and C = 48 with hidden C/4 (Demucs v3); T up to 16 waves x 3 fragments and the
// workgroup's LDS image within 160 KB. Returns 0, or -1 when no kernel exists for the shape (the plan keeps the K1/K2/K3 chain).
int launch_dconv_row(const DconvRowArgs &a, hipStream_t s, bool dry)
{
    constexpr int FPW = 3;
    const int nfrag = (a.T + 15) / 16, nw = (nfrag + FPW - 1) / FPW;
    if (a.T < 2 || nw > 16 || (i64)a.B * a.T * a.F * a.C >= (1ll << 31) || (i64)a.T * a.F * a.C * 4 >= (1ll << 32)) // (32-bit byte offsets inside a segment)
        return -1;
    auto go = [&](auto kern, size_t smem, i64 imgFloats) -> int {
        if (smem > 160 * 1024 || smem != dconv_row_lds_bytes(a.C, a.hid, a.T) || imgFloats != dconv_row_image_floats(a.C, a.hid))
            return -1;
        if (dry)
            return 0;
        DconvRowArgs k = a;
        const int rows = a.B * a.F;
        k.rowsPerXcd = (rows + 7) / 8;
        k.n1 = (double)a.hid * a.T, k.invN1 = 1.0 / k.n1, k.invN1m1 = 1.0 / (k.n1 - 1.0);
        k.n2 = 2.0 * a.C * a.T, k.invN2 = 1.0 / k.n2, k.invN2m1 = 1.0 / (k.n2 - 1.0);
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        // persistent: as many workgroups as the device keeps resident (registers, LDS), at most one per row
        int dev = 0, cus = 256, perCu = 1;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kern, 64 * nw, smem) != hipSuccess || perCu < 1)
            perCu = 1;
        int slots = (cus * perCu + 7) / 8; // workgroups per XCD
        if (slots > k.rowsPerXcd)
            slots = k.rowsPerXcd;
        hipLaunchKernelGGL(kern, dim3(8 * slots), dim3(64 * nw), smem, s, k);
        return 0;
    };
    if (a.C == 48 && a.hid == 6)
        return go(dconv_row_kernel<48, 6, FPW, 6>, RowGeo<48, 6>::lds_bytes(a.T), RowGeo<48, 6>::nWp + RowGeo<48, 6>::nWk);
    if (a.C == 96 && a.hid == 12)
        return go(dconv_row_kernel<96, 12, FPW, 4>, RowGeo<96, 12>::lds_bytes(a.T), RowGeo<96, 12>::nWp + RowGeo<96, 12>::nWk);
    if (a.C == 48 && a.hid == 12)
        return go(dconv_row_kernel<48, 12, FPW, 6>, RowGeo<48, 12>::lds_bytes(a.T), RowGeo<48, 12>::nWp + RowGeo<48, 12>::nWk);
    return -1;
}

} // namespace dmx

// attention.hip — fused multi-head attention (flash style, fp32 MFMA) for gfx950.
//
// Restates the per-head block of /root/reference/src/layers.cpp:442-482:
//   softmax_rows(Q_h K_h^T / sqrt(d_h)) V_h  with max-subtracted exp,
// heads = contiguous column blocks of width d_h = C/8 (:445-450). The reference
// materialises a (Tq x Tk) score matrix per head (29 MB at 2688^2); here scores live only
// in registers: online softmax over 64-key tiles.
//
// MFMA mapping (v_mfma_f32_16x16x4_f32, exact fp32): we compute the TRANSPOSED products
//   S^T = K Q^T   (A = K tile from LDS, B = Q from registers)
//   O^T = V^T P^T (A = V^T tile from LDS, B = P^T)
// because the C/D layout of S^T (row = key = 4*(lane>>4)+reg, col = query = lane&15) is
// exactly the B-operand layout the second product needs (k-slot lane>>4, MFMA index reg
// <-> key 4*(lane>>4)+reg), so P never leaves registers and never needs a shuffle; the
// softmax row reduction is in-lane + two xor-shuffles (16, 32), and every lane's O^T
// registers belong to one query, so the online rescale is one scalar per lane.
//
// Workgroup = 4 waves x 16 queries; K/V tiles of 64 keys are register-prefetched and
// double-buffered in LDS as [quad][row] float4 (conflict-free ds_read_b128).
#include "kernels.h"

namespace dmx
{

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int HS>
__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs p)
{
    constexpr int DQ = HS / 4;  // dim quads
    constexpr int DF = HS / 16; // dim fragments
    constexpr int KT = 64;      // keys per tile
    constexpr int KL = (KT * DQ + 255) / 256;      // K float4 loads per thread per tile
    constexpr int VL = ((KT / 4) * HS + 255) / 256; // V (key-quad, dim) items per thread
    __shared__ float4 Ks[2][DQ][KT];      // Ks[dq][key]  = K[key][4dq..4dq+3]
    __shared__ float4 Vs[2][KT / 4][HS];  // Vs[kq][dim]  = V[4kq..4kq+3][dim]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, h4 = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const float *Q = p.q + (i64)b * p.qB + head * HS;
    const float *K = p.k + (i64)b * p.kB + head * HS;
    const float *V = p.v + (i64)b * p.vB + head * HS;

    // Q fragment: lane holds Q[q0 + l15][16kk + 4h4 .. +3]
    float4 qf[DF];
    {
        const int qr = q0 + l15;
#pragma unroll
        for (int kk = 0; kk < DF; ++kk)
            qf[kk] = qr < p.Tq ? *reinterpret_cast<const float4 *>(Q + (i64)qr * p.ldq + 16 * kk + 4 * h4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    float4 kreg[KL];
    float vreg[VL][4];
    auto load_tile = [&](int t0) {
#pragma unroll
        for (int i = 0; i < KL; ++i)
        {
            const int idx = tid + i * 256;
            const int key = idx / DQ, dq = idx - key * DQ;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < KT * DQ && t0 + key < p.Tk)
                v = *reinterpret_cast<const float4 *>(K + (i64)(t0 + key) * p.ldk + 4 * dq);
            kreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < VL; ++i)
        {
            const int idx = tid + i * 256;
            const int kq = idx / HS, dim = idx - kq * HS;
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                const int key = t0 + 4 * kq + j;
                vreg[i][j] = (idx < (KT / 4) * HS && key < p.Tk) ? V[(i64)key * p.ldv + dim] : 0.f;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < KL; ++i)
        {
            const int idx = tid + i * 256;
            const int key = idx / DQ, dq = idx - key * DQ;
            if (idx < KT * DQ)
                Ks[buf][dq][key ^ (2 * (dq & 3))] = kreg[i]; // XOR swizzle (see igemm.hip)
        }
#pragma unroll
        for (int i = 0; i < VL; ++i)
        {
            const int idx = tid + i * 256;
            const int kq = idx / HS, dim = idx - kq * HS;
            if (idx < (KT / 4) * HS)
                Vs[buf][kq][dim] = make_float4(vreg[i][0], vreg[i][1], vreg[i][2], vreg[i][3]);
        }
    };

    f32x4 o[DF];
#pragma unroll
    for (int d = 0; d < DF; ++d)
        o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;

    const int nt = (p.Tk + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nt; ++t)
    {
        const bool next = t + 1 < nt;
        if (next)
            load_tile((t + 1) * KT);
        // ---- S^T = K Q^T : 4 key fragments x 16 queries
        f32x4 sT[4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
            sT[kf] = f32x4{0.f, 0.f, 0.f, 0.f};
        // dim step outermost, key fragment innermost: 4 independent accumulator chains
#pragma unroll
        for (int kk = 0; kk < DF; ++kk)
        {
            float4 kv[4];
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
                kv[kf] = Ks[cur][4 * kk + h4][(16 * kf + l15) ^ (2 * h4)];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
                {
                    const float av = c == 0 ? kv[kf].x : (c == 1 ? kv[kf].y : (c == 2 ? kv[kf].z : kv[kf].w));
                    const float bv = c == 0 ? qf[kk].x : (c == 1 ? qf[kk].y : (c == 2 ? qf[kk].z : qf[kk].w));
                    sT[kf] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, sT[kf], 0, 0, 0);
                }
        }
        // ---- online softmax for query l15; lane holds keys 16kf + 4h4 + r
        float tmax = -INFINITY;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int key = t * KT + 16 * kf + 4 * h4 + r;
                float s = sT[kf][r] * p.scale;
                s = key < p.Tk ? s : -INFINITY;
                sT[kf][r] = s;
                tmax = fmaxf(tmax, s);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = __expf(mrun - mnew); // first tile: exp(-inf) = 0
        float psum = 0.f;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const float pv = __expf(sT[kf][r] - mnew);
                sT[kf][r] = pv;
                psum += pv;
            }
        lrun = lrun * alpha + psum;
        mrun = mnew;
#pragma unroll
        for (int d = 0; d < DF; ++d)
        {
            o[d][0] *= alpha;
            o[d][1] *= alpha;
            o[d][2] *= alpha;
            o[d][3] *= alpha;
        }
        // ---- O^T += V^T P^T  (key fragment outermost, dim fragment innermost: DF independent chains)
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
        {
            float4 vv[DF];
#pragma unroll
            for (int d = 0; d < DF; ++d)
                vv[d] = Vs[cur][4 * kf + h4][16 * d + l15];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < DF; ++d)
                {
                    const float av = c == 0 ? vv[d].x : (c == 1 ? vv[d].y : (c == 2 ? vv[d].z : vv[d].w));
                    o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sT[kf][c], o[d], 0, 0, 0);
                }
        }
        if (next)
            store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // total row sum over the 4 lanes of this query
    lrun += __shfl_xor(lrun, 16);
    lrun += __shfl_xor(lrun, 32);
    const float inv = 1.0f / lrun;
    const int qr = q0 + l15;
    if (qr < p.Tq)
    {
        float *O = p.o + (i64)b * p.oB + (i64)qr * p.ldo + head * HS;
#pragma unroll
        for (int d = 0; d < DF; ++d) // lane holds dims 16d + 4h4 + r of query l15
            *reinterpret_cast<float4 *>(O + 16 * d + 4 * h4) =
                make_float4(o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv);
    }
}

void launch_attention(const AttnArgs &a, hipStream_t s)
{
    dim3 grid((a.Tq + 63) / 64, a.H, a.B);
    if (a.hs == 64)
        hipLaunchKernelGGL(attention_kernel<64>, grid, dim3(256), 0, s, a);
    else if (a.hs == 48)
        hipLaunchKernelGGL(attention_kernel<48>, grid, dim3(256), 0, s, a);
    else
        abort();
}

} // namespace dmx

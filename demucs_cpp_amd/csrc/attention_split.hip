// attention_split.hip — the flash attention of attention.hip with its two fp32 products formed on the bf16 matrix pipe
// from EXACT operand splits (contexts created with DMX_GEMM_BF16X3; include/demucs_hip.h).
//
// Reference block: softmax_rows(Q_h K_h^T / sqrt(d_h)) V_h per head, /root/reference/src/layers.cpp:442-482.
// Both products have ACTIVATIONS on both sides, so both operands are three-term splits (igemm_common.h split3_pk:
// x = x1 + x2 + x3 exactly, |x2| <= 2^-8 |x|, |x3| <= 2^-16 |x|):
//   a b = a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1   (+ a2 b3 + a3 b2 + a3 b3, dropped: <= 2^-23 |a b|)
// six v_mfma_f32_16x16x32_bf16 per 16x16x32 block (96 matrix-pipe cycles) instead of eight v_mfma_f32_16x16x4_f32 (256):
// every partial product is exact in fp32, accumulation is fp32, the softmax between the products is the fp32 code of
// attention_common.h. Issue order per accumulator, smallest terms first: a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1 - the
// same for every query whatever the workgroup shape or batch, so results are bit-identical across batching and sharding.
//
// MFMA mapping (transposed products, as in attention.hip): S^T = K Q^T (A = K fragments from LDS: row = key, k = 8
// consecutive head dims per lane; B = Q from registers), O^T = V^T P^T (A = V^T fragments from LDS: row = dim, k = keys;
// B = P^T straight from the S^T accumulators). The C/D layout of S^T gives lane (l15, h4) the scores of keys
// 16 kf + 4 h4 + r of its query; a 32-key MFMA step s takes kf = 2s, 2s + 1, so k-slot j of lane h4 IS key
// 32 s + 16 (j >> 2) + 4 h4 + (j & 3): P needs no shuffle, and the V^T image is laid out in exactly that key order.
//
// LDS images, per split plane (3 each), a row = 64 bf16 = 128 B = 8 slots of 16 B:
//   K  [key 64][slot = dim / 8]  slot ^= (key >> 1) & 7                      two buffers (the next tile is staged while
//                                                                            this one is multiplied)
//   V^T [dim][slot = 4 s + h4]   slot ^= (dim.1, dim.3, dim.2) as a 3-bit number   one buffer, 16-B slot = keys
//                                32 s + 4 h4 + {0..3} and 32 s + 16 + 4 h4 + {0..3}
// both swizzles make every ds_read_b128 lane group ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md) hit 16 distinct 16-byte
// slots, the K stores write whole rows, the V^T stores (8 bytes = 4 keys of one dim and plane per lane) are conflict-free
// for the thread -> (key quad, dim quad) map used below. 72 KB per workgroup: two workgroups per CU.
// Staging splits happen in the staging threads (fp32 from global memory -> 3 bf16 planes); head dim 48 (htdemucs-6s) is
// padded to 64 with zeros in Q and K (a half-empty second k-step costs what a full one costs on this pipe).
//
// PL = true (round 5): K and V arrive ALREADY split, as the bf16 planes the K / V projections' epilogues wrote (plan.h EPI_KPL /
// EPI_VT: the same split3_pk of the same fp32 values, so the operand bits are those of the PL = false path), K row-major
// [plane][token][D], V^T tile by tile in exactly the LDS image order. Both go global -> LDS directly (global_load_lds_dwordx4,
// 1 KB per wave instruction, the XOR swizzle applied to the SOURCE slot): no staging registers, no split and no ds_write in
// the tile loop - the K / V splits were 1.6 of the 16.7 ms this kernel took per 42-segment step (ablation, profiles/r05_*),
// redone by every query tile. Whole 64-key tiles only (plan.cpp planes_ok), so no masked tile exists in this form.
#include "attention_common.h"
#include "igemm_common.h"
#include <cstdlib>
#include <type_traits>

namespace dmx
{

__device__ __forceinline__ int swzK(int key) { return (key >> 1) & 7; }
__device__ __forceinline__ int swzV(int dim) { return ((dim >> 1) & 1) | (((dim >> 3) & 1) << 1) | (((dim >> 2) & 1) << 2); }

template <int HS, int QF, bool PL>
__global__ __launch_bounds__(256, 2) void attention_split_kernel(const AttnArgs p)
{
    constexpr int DF = HS / 16;  // dim fragments of O
    constexpr int KT = 64;       // keys per tile
    constexpr int NS = 8;        // 16-byte slots per image row
    constexpr int KSL = HS / 8;  // slots of a key row that hold data (8 | 6)
    constexpr int KCH = (KT * KSL + 255) / 256; // K chunks (8 dims of one key) per thread and tile
    constexpr int DQ = HS / 4;   // dim quads
    static_assert(HS == 64 || HS == 48, "head dims of htdemucs (512 / 8, 384 / 8)");

    __shared__ u32x4 Kp0[3][KT][NS], Kp1[3][KT][NS]; // [plane][key][slot]
    __shared__ u32x4 Vp[3][HS][NS];                  // [plane][dim][slot]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, h4 = lane >> 4;
    unsigned qt, bh;
    if (!att_tile_of_block(p, qt, bh))
        return;
    const int b = (int)(bh / (unsigned)p.H), head = (int)(bh - (unsigned)b * (unsigned)p.H);
    const int q0 = (int)qt * (64 * QF) + wave * (16 * QF);
    const float *Q = p.q + (i64)b * p.qB + head * HS;
    const float *K = p.k + (i64)b * p.kB + head * HS;
    const float *V = p.v + (i64)b * p.vB + head * HS;

    // (HS = 48) slots 6, 7 of every K row are padding: they meet zero Q dims, but 0 x garbage could be NaN
    if constexpr (KSL < NS)
    {
        for (int i = tid; i < 2 * 3 * KT * (NS - KSL); i += 256)
        {
            const int sl = KSL + i % (NS - KSL), key = (i / (NS - KSL)) % KT, pl = (i / ((NS - KSL) * KT)) % 3, bf = i / ((NS - KSL) * KT * 3);
            (bf ? Kp1 : Kp0)[pl][key][sl ^ swzK(key)] = u32x4{0u, 0u, 0u, 0u};
        }
    }

    // ---- Q fragments, three planes: lane holds Q[q0 + 16 f + l15][32 kk + 8 h4 .. +7] pre-multiplied by scale * log2(e)
    // (one fp32 rounding, as in attention.hip: the scores leave the MFMAs in the exp2 domain), then split exactly
    const float qs = p.scale * kLog2e;
    u32x4 qp[3][QF][2];
#pragma unroll
    for (int f = 0; f < QF; ++f)
    {
        const int qr = min(q0 + 16 * f + l15, p.Tq - 1); // rows >= Tq: duplicates, never stored
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
        {
            const int d0 = 32 * kk + 8 * h4;
            f32x4 v0 = f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (d0 < HS)
            {
                v0 = *reinterpret_cast<const f32x4 *>(Q + (i64)qr * p.ldq + d0);
                v1 = *reinterpret_cast<const f32x4 *>(Q + (i64)qr * p.ldq + d0 + 4);
            }
            unsigned h1[4], h2[4], h3[4];
            split3_pk(v0[0] * qs, v0[1] * qs, h1[0], h2[0], h3[0]);
            split3_pk(v0[2] * qs, v0[3] * qs, h1[1], h2[1], h3[1]);
            split3_pk(v1[0] * qs, v1[1] * qs, h1[2], h2[2], h3[2]);
            split3_pk(v1[2] * qs, v1[3] * qs, h1[3], h2[3], h3[3]);
            qp[0][f][kk] = u32x4{h1[0], h1[1], h1[2], h1[3]};
            qp[1][f][kk] = u32x4{h2[0], h2[1], h2[2], h2[3]};
            qp[2][f][kk] = u32x4{h3[0], h3[1], h3[2], h3[3]};
        }
    }

    // ---- staging. Keys beyond Tk re-read the last valid row (their scores are masked to -inf in the last tile, so
    // P = 0 meets finite V values).
    // K: thread -> chunks c = tid + 256 i: (key = c / KSL, slot = c % KSL), 8 consecutive dims (two float4).
    // V: thread -> (key quad kq4 = 8 s + 4 half + h4v, dim quad dqv): the same 4 dims of 4 consecutive keys; per dim the 4
    //    keys are one 8-byte piece of the V^T image. Lane bits: h4v = tid & 3, half = (tid >> 2) & 1, dqv bit 0 = (tid >> 3) & 1,
    //    dqv >> 1 = (tid >> 4) & 7, s = tid >> 7: the 16 lanes of a ds_write_b64 group then differ in (h4v, half, dqv bit 0),
    //    i.e. in 16 distinct 8-byte pieces of the 256-byte bank row.
    f32x4 kreg[KCH][2], vreg[4];
    const int vh4 = tid & 3, vhalf = (tid >> 2) & 1, vdq = ((tid >> 3) & 1) + 2 * ((tid >> 4) & 7), vs = tid >> 7;
    const int vkq4 = 8 * vs + 4 * vhalf + vh4;
    const bool vOk = vdq < DQ;
    auto load_k = [&](int t0) {
#pragma unroll
        for (int i = 0; i < KCH; ++i)
        {
            const int c = min(tid + 256 * i, KT * KSL - 1);
            const int key = c / KSL, sl = c - key * KSL;
            const i64 r = min(t0 + key, p.Tk - 1);
            kreg[i][0] = *reinterpret_cast<const f32x4 *>(K + r * p.ldk + 8 * sl);
            kreg[i][1] = *reinterpret_cast<const f32x4 *>(K + r * p.ldk + 8 * sl + 4);
        }
    };
    auto load_v = [&](int t0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const i64 r = min(t0 + 4 * vkq4 + j, p.Tk - 1);
            vreg[j] = *reinterpret_cast<const f32x4 *>(V + r * p.ldv + 4 * (vOk ? vdq : 0));
        }
    };
    auto store_k = [&](int buf) {
#pragma unroll
        for (int i = 0; i < KCH; ++i)
        {
            const int c = tid + 256 * i;
            if (KT * KSL % 256 != 0 && c >= KT * KSL)
                continue;
            const int key = c / KSL, sl = c - key * KSL;
            unsigned h1[4], h2[4], h3[4];
            split3_pk(kreg[i][0][0], kreg[i][0][1], h1[0], h2[0], h3[0]);
            split3_pk(kreg[i][0][2], kreg[i][0][3], h1[1], h2[1], h3[1]);
            split3_pk(kreg[i][1][0], kreg[i][1][1], h1[2], h2[2], h3[2]);
            split3_pk(kreg[i][1][2], kreg[i][1][3], h1[3], h2[3], h3[3]);
            u32x4(*Kp)[KT][NS] = buf ? Kp1 : Kp0;
            const int sw = sl ^ swzK(key);
            Kp[0][key][sw] = u32x4{h1[0], h1[1], h1[2], h1[3]};
            Kp[1][key][sw] = u32x4{h2[0], h2[1], h2[2], h2[3]};
            Kp[2][key][sw] = u32x4{h3[0], h3[1], h3[2], h3[3]};
        }
    };
    auto store_v = [&]() {
        if (!vOk)
            return;
#pragma unroll
        for (int c = 0; c < 4; ++c) // dim 4 vdq + c: keys 4 vkq4 .. +3
        {
            unsigned a1, a2, a3, b1, b2, b3;
            split3_pk(vreg[0][c], vreg[1][c], a1, a2, a3);
            split3_pk(vreg[2][c], vreg[3][c], b1, b2, b3);
            const int dim = 4 * vdq + c;
            const int sw = (4 * vs + vh4) ^ swzV(dim);
            *(reinterpret_cast<u32x2 *>(&Vp[0][dim][sw]) + vhalf) = u32x2{a1, b1};
            *(reinterpret_cast<u32x2 *>(&Vp[1][dim][sw]) + vhalf) = u32x2{a2, b2};
            *(reinterpret_cast<u32x2 *>(&Vp[2][dim][sw]) + vhalf) = u32x2{a3, b3};
        }
    };

    // ---- PL: direct global -> LDS staging of the pre-split planes. One wave instruction moves 1 KB = 8 image rows; lane l lands
    // at (row 8 c + l / 8, slot position l % 8) and therefore FETCHES source slot (l % 8) ^ swizzle(row).
    const int drow = lane >> 3, dpos = lane & 7;
    const int D = p.H * HS;
    const int wv = __builtin_amdgcn_readfirstlane(wave); // (uniform: the chunk tests below are scalar branches)
    auto dma_k = [&](int t0, int buf) {
        u32x4(*Kp)[KT][NS] = buf ? Kp1 : Kp0;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i)
            {
                const int c = 2 * wv + i, key = 8 * c + drow;
                const int src = dpos ^ swzK(key);
                const i64 r = (i64)b * p.Tk + min(t0 + key, p.Tk - 1);
                const unsigned short *g = p.kpl + (i64)pl * p.kvPlane + r * D + head * HS + 8 * src;
                if (KSL == NS || src < KSL) // (HS = 48: the two padding slots of a row stay zero)
                    load_to_lds_b128(reinterpret_cast<const float *>(g), reinterpret_cast<float4 *>(&Kp[pl][8 * c][0]));
            }
    };
    auto dma_v = [&](int t) {
        const int nt = p.Tk >> 6;
        const i64 tile = ((i64)b * p.H + head) * nt + min(t, nt - 1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i)
            {
                const int c = wv + 4 * i; // 8 (HS = 64) or 6 (HS = 48) chunks of 8 dims
                if (c < HS / 8)
                {
                    const int dim = 8 * c + drow;
                    const unsigned short *g = p.vt + (i64)pl * p.kvPlane + (tile * HS + dim) * 64 + 8 * (dpos ^ swzV(dim));
                    load_to_lds_b128(reinterpret_cast<const float *>(g), reinterpret_cast<float4 *>(&Vp[pl][8 * c][0]));
                }
            }
    };

    f32x4 o[QF][DF];
    float mrun[QF], lrun[QF], mcur[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f)
    {
        mrun[f] = -INFINITY;
        lrun[f] = 0.f;
        mcur[f] = 0.f;
#pragma unroll
        for (int d = 0; d < DF; ++d)
            o[f][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int ksw = swzK(l15), vsw = swzV(l15); // swizzle terms of this lane's fragment rows (fragment bases are multiples of 16)

    // S^T = K Q^T of the tile in Kp<buf>: 4 key fragments x QF query fragments, two 32-dim steps
    auto scores = [&](int buf, f32x4 (*sT)[4]) {
        u32x4(*Kp)[KT][NS] = buf ? Kp1 : Kp0;
#pragma unroll
        for (int f = 0; f < QF; ++f)
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
                sT[f][kf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int kf0 = 0; kf0 < 4; kf0 += 2) // two key fragments at a time: 2 QF independent accumulators per term
            {
                bf16x8 ka[2][3];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        ka[u][pl] = __builtin_bit_cast(bf16x8, Kp[pl][16 * (kf0 + u) + l15][(4 * kk + h4) ^ ksw]);
                // term order: k3 q1, k1 q3, k2 q2, k2 q1, k1 q2, k1 q1
                constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int f = 0; f < QF; ++f)
                            sT[f][kf0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[u][TA[tm]], __builtin_bit_cast(bf16x8, qp[TB[tm]][f][kk]),
                                                                                      sT[f][kf0 + u], 0, 0, 0);
            }
    };
    // O^T += V^T P^T: per 32-key step s the P planes of both query fragments, then the dim fragments two at a time
    auto pvprod = [&](f32x4 (*sT)[4]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
        {
            bf16x8 pp[QF][3];
#pragma unroll
            for (int f = 0; f < QF; ++f)
            {
                unsigned h1[4], h2[4], h3[4];
                split3_pk(sT[f][2 * s][0], sT[f][2 * s][1], h1[0], h2[0], h3[0]);
                split3_pk(sT[f][2 * s][2], sT[f][2 * s][3], h1[1], h2[1], h3[1]);
                split3_pk(sT[f][2 * s + 1][0], sT[f][2 * s + 1][1], h1[2], h2[2], h3[2]);
                split3_pk(sT[f][2 * s + 1][2], sT[f][2 * s + 1][3], h1[3], h2[3], h3[3]);
                pp[f][0] = __builtin_bit_cast(bf16x8, u32x4{h1[0], h1[1], h1[2], h1[3]});
                pp[f][1] = __builtin_bit_cast(bf16x8, u32x4{h2[0], h2[1], h2[2], h2[3]});
                pp[f][2] = __builtin_bit_cast(bf16x8, u32x4{h3[0], h3[1], h3[2], h3[3]});
            }
#pragma unroll
            for (int d0 = 0; d0 < DF; d0 += 2)
            {
                constexpr int ND = 2;
                bf16x8 va[ND][3];
#pragma unroll
                for (int u = 0; u < ND; ++u)
                    if (d0 + u < DF)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            va[u][pl] = __builtin_bit_cast(bf16x8, Vp[pl][16 * (d0 + u) + l15][(4 * s + h4) ^ vsw]);
                constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0}; // v3 p1, v1 p3, v2 p2, v2 p1, v1 p2, v1 p1
#pragma unroll
                for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                    for (int u = 0; u < ND; ++u)
                        if (d0 + u < DF)
#pragma unroll
                            for (int f = 0; f < QF; ++f)
                                o[f][d0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[u][TA[tm]], pp[f][TB[tm]], o[f][d0 + u], 0, 0, 0);
            }
        }
    };

    const int nt = (p.Tk + KT - 1) / KT;
    const bool partial = (p.Tk % KT) != 0;
    if constexpr (PL)
    {
        // prologue: K(0) -> Kp0; then per tile t (PAR = t & 1):
        //   request V(t) -> V^T image (every wave is past P V of tile t-1: barrier X) and K(t+1) -> K buffer PAR ^ 1 (last read by
        //   S(t-1)); S(t) from K buffer PAR; softmax; wait for this wave's V(t) pieces (the K(t+1) pieces may stay in flight);
        //   barrier Z; O += V(t)^T P^T; wait for the K(t+1) pieces; barrier X.
        dma_k(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        auto tilep = [&](int t, auto parTag) {
            constexpr int PAR = decltype(parTag)::value;
            dma_v(t);
            // the vmcnt(6) below counts on the V pieces being ISSUED before the six K pieces: both are LDS-DMA builtins into
            // different arrays, which the scheduler may otherwise legally reorder: pinned
            __builtin_amdgcn_sched_barrier(0);
            dma_k((t + 1) * KT, PAR ^ 1); // beyond the end: a clamped re-read, never used
            f32x4 sT[QF][4];
            scores(PAR, sT);
#pragma unroll
            for (int f = 0; f < QF; ++f)
            {
                att_softmax_pre<DF, false, false>(sT[f], o[f], mrun[f], lrun[f], mcur[f], t, h4, p.Tk, 0, 0.f);
                att_softmax_post(sT[f], lrun[f], mcur[f]);
            }
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // the six K(t+1) pieces were issued last: everything before them has landed
            __syncthreads();
            pvprod(sT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        };
        int t = 0;
        for (; t + 1 < nt; t += 2)
        {
            tilep(t, std::integral_constant<int, 0>{});
            tilep(t + 1, std::integral_constant<int, 1>{});
        }
        if (t < nt)
            tilep(t, std::integral_constant<int, 0>{});
    }
    else
    {
    // ---- prologue: K(0) -> Kp0, K(1) and V(0) in the staging registers
    load_k(0);
    load_v(0);
    store_k(0);
    load_k(KT);
    __syncthreads();
    // ---- one key tile. PAR = t & 1 = the K buffer of tile t (compile-time: two copies of the body).
    //   V(t) registers -> V^T image (free: every wave passed the barrier behind P V of tile t-1); request V(t+1)
    //   S(t) from K buffer PAR; softmax
    //   barrier Z: V(t) is complete for every wave
    //   O += V(t)^T P^T
    //   K(t+1) registers -> K buffer PAR ^ 1 (last read by S(t-1)); request K(t+2)
    //   barrier X: K(t+1) complete, nobody reads V(t) any more
    auto tile = [&](int t, auto parTag, auto maskTag) {
        constexpr int PAR = decltype(parTag)::value;
        constexpr bool MASK = decltype(maskTag)::value;
        store_v();
        load_v((t + 1) * KT); // beyond the end: clamped re-read, never used
        f32x4 sT[QF][4];
        scores(PAR, sT);
#pragma unroll
        for (int f = 0; f < QF; ++f)
        {
            att_softmax_pre<DF, MASK, false>(sT[f], o[f], mrun[f], lrun[f], mcur[f], t, h4, p.Tk, 0, 0.f);
            att_softmax_post(sT[f], lrun[f], mcur[f]);
        }
        __syncthreads();
        pvprod(sT);
        store_k(PAR ^ 1);
        load_k((t + 2) * KT);
        __syncthreads();
    };
    const std::integral_constant<int, 0> even{};
    const std::integral_constant<int, 1> odd{};
    int t = 0;
    for (; t + 2 < nt; t += 2)
    {
        tile(t, even, std::false_type{});
        tile(t + 1, odd, std::false_type{});
    }
    if (t + 1 < nt) // two tiles left: t (full) and t + 1 (last)
    {
        tile(t, even, std::false_type{});
        if (partial)
            tile(t + 1, odd, std::true_type{});
        else
            tile(t + 1, odd, std::false_type{});
    }
    else if (partial) // one tile left
        tile(t, even, std::true_type{});
    else
        tile(t, even, std::false_type{});
    } // !PL

#pragma unroll
    for (int f = 0; f < QF; ++f)
    {
        // total row sum over the 4 lanes of this query
        const float l = quad_lanes_sum(lrun[f]);
        const float inv = 1.0f / l;
        const int qr = q0 + 16 * f + l15;
        if (qr < p.Tq)
        {
            float *O = p.o + (i64)b * p.oB + (i64)qr * p.ldo + head * HS;
#pragma unroll
            for (int d = 0; d < DF; ++d) // lane holds dims 16d + 4h4 + r of its query
                *reinterpret_cast<float4 *>(O + 16 * d + 4 * h4) =
                    make_float4(o[f][d][0] * inv, o[f][d][1] * inv, o[f][d][2] * inv, o[f][d][3] * inv);
        }
    }
}

// -1: no split kernel for this head dim (the caller falls back to the fp32 MFMA kernel)
int launch_attention_split(const AttnArgs &a0, hipStream_t s, bool dry)
{
    if (a0.hs != 64 && a0.hs != 48)
        return -1;
    if (dry)
        return 0;
    constexpr int xcdMap = 1; // all query tiles of one (batch, head) on one XCD
    AttnArgs a = a0;
    a.xcdMap = xcdMap;
    const bool big = att_use_big_shape(a); // 128- or 64-query workgroups (attention_common.h): same arithmetic either way
    a.nQt = (unsigned)(big ? (a.Tq + 127) / 128 : (a.Tq + 63) / 64);
    const unsigned nbh = (unsigned)(a.B * a.H);
    const dim3 grid(xcdMap ? 8u * ((nbh + 7u) / 8u) * a.nQt : nbh * a.nQt);
    const bool planes = a.kpl && a.vt && a.Tk % 64 == 0;
#define DMX_ATT_LAUNCH(HS_, QF_)                                                                            \
    do                                                                                                      \
    {                                                                                                       \
        if (planes)                                                                                         \
            hipLaunchKernelGGL((attention_split_kernel<HS_, QF_, true>), grid, dim3(256), 0, s, a);          \
        else                                                                                                \
            hipLaunchKernelGGL((attention_split_kernel<HS_, QF_, false>), grid, dim3(256), 0, s, a);         \
    } while (0)
    if (big)
    {
        if (a.hs == 64)
            DMX_ATT_LAUNCH(64, 2);
        else
            DMX_ATT_LAUNCH(48, 2);
    }
    else
    {
        if (a.hs == 64)
            DMX_ATT_LAUNCH(64, 1);
        else
            DMX_ATT_LAUNCH(48, 1);
    }
#undef DMX_ATT_LAUNCH
    return 0;
}

} // namespace dmx

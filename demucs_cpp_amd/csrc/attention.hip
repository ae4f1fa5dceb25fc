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
//   O^T = V^T P^T (A = V^T from LDS, B = P^T)
// because the C/D layout of S^T (row = key = 4*(lane>>4)+reg, col = query = lane&15) is
// exactly the B-operand layout the second product needs (k-slot lane>>4, MFMA index reg
// <-> key 4*(lane>>4)+reg), so P never leaves registers and never needs a shuffle; the
// softmax row reduction is in-lane + two xor-shuffles (16, 32), and every lane's O^T
// registers belong to one query, so the online rescale is one scalar per lane.
//
// Workgroup = 4 waves x (16*QF) queries. K/V tiles of 64 keys are fetched with float4 loads one
// tile ahead into registers and double-buffered in LDS:
//   K image  [dim-quad q][key ^ 2(q&3)] float4  -> conflict-free ds_read_b128 of A fragments
//   V image  [key-quad][dim'] float4 (TRANSPOSED while staging: a thread loads the same dim-quad of 4
//            consecutive keys and writes, per dim, the 4 keys as one float4 - the 4x4 transpose is register
//            renaming) -> V^T A-fragment (dim = lane&15, keys 4*(lane>>4)+c) = ONE ds_read_b128 (was 4
//            ds_read_b32). dim' = dim ^ ((dim >> 3) & 3) keeps the 8-lane write groups conflict-free; reads of
//            16 consecutive dims are conflict-free under any permutation inside aligned groups of 4.
// Online softmax: row maxima / sums cross the 4 lanes of a query with v_permlane16_swap / v_permlane32_swap (VALU)
// instead of two LDS round trips (ds_bpermute) in the middle of the dependency chain.
#include "attention_common.h"
#include <cstdlib>
#include <type_traits>

namespace dmx
{

__device__ __forceinline__ float a4c(const float4 &v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// global -> LDS load of 16 bytes per lane (global_load_lds_dwordx4): lane l writes lds_base + 16 l; lds_base is
// wave-uniform (M0). The builtin exists in the device pass only. (Semantics: tools/micro/lds_dma.hip.)
__device__ __forceinline__ void load_to_lds_b128(const float *gptr, float4 *lds_base)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(gptr, (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
#else
    (void)gptr;
    (void)lds_base;
#endif
}

// LOC: LocalState attention of Demucs v3 (/root/reference/src/layers.cpp:533-721): the score of (query s, key t) gets the
// decay penalty sum_n -(n+1) |t-s| / 2 * sigmoid(d_n(s)) / 2 = -|t-s| g(s) (one scalar per query and head), the diagonal
// is set to -100 (not masked out), then the same softmax over the keys and the same weighted sum of the content.
template <int HS, int QF, bool LOC = false>
__global__ __launch_bounds__(256, HS > 64 ? 1 : 2) void attention_kernel(const AttnArgs p)
{
    constexpr int DQ = HS / 4;  // dim quads
    constexpr int DF = HS / 16; // dim fragments
    constexpr int KT = 64;      // keys per tile
    constexpr int NL = (KT * DQ) / 256; // float4 loads per thread per tile (K and V each)
    // K image: d_h = 64: row-major [key][dim-quad ^ (key & 15)] (a key = 256 B), filled by DIRECT global -> LDS loads
    // (global_load_lds_dwordx4: no staging registers - the software pipeline below needs them for the scores of
    // two tiles); d_h = 48 (rows of 192 B do not tile a wave's 1 KB): [dim-quad][key ^ 2(q&3)] through registers.
    // Buffers are distinct objects selected at compile time (the compiler can then tell a direct load into one
    // buffer from the fragment reads of the other).
    constexpr bool KDIRECT = HS == 64;
    __shared__ float4 Ks0[KDIRECT ? KT * DQ : 1], Ks1[KDIRECT ? KT * DQ : 1];
    __shared__ float4 Kq0[KDIRECT ? 1 : DQ][KDIRECT ? 1 : KT], Kq1[KDIRECT ? 1 : DQ][KDIRECT ? 1 : KT];
    __shared__ float4 Vt0[KT / 4][HS], Vt1[KT / 4][HS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, h4 = lane >> 4;
    unsigned qt, bh;
    if (!att_tile_of_block(p, qt, bh)) // XCD-aware workgroup -> (query tile, batch * head) map (attention_common.h)
        return;
    const int b = (int)(bh / (unsigned)p.H), head = (int)(bh - (unsigned)b * (unsigned)p.H);
    const int q0 = (int)qt * (64 * QF) + wave * (16 * QF);
    const float *Q = p.q + (i64)b * p.qB + head * HS;
    const float *K = p.k + (i64)b * p.kB + head * HS;
    const float *V = p.v + (i64)b * p.vB + head * HS;

    // Q fragments: lane holds Q[q0 + 16 f + l15][16kk + 4h4 .. +3]
    // pre-multiplied by scale * log2(e): the scores leave the MFMAs in the exp2 domain (softmax is
    // invariant; for d_h = 64 the scale 1/8 is exact, so only the log2(e) factor adds one rounding)
    const float qs = p.scale * kLog2e;
    float4 qf[QF][DF];
#pragma unroll
    for (int f = 0; f < QF; ++f)
    {
        const int qr = q0 + 16 * f + l15;
#pragma unroll
        for (int kk = 0; kk < DF; ++kk)
        {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(Q + (i64)min(qr, p.Tq - 1) * p.ldq + 16 * kk + 4 * h4);
            qf[f][kk] = make_float4(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs); // rows >= Tq: duplicates, never stored
        }
    }
    // LOC: decay slope of this lane's query in the exp2 domain, and the query's own index (the diagonal)
    float gq[QF];
    int qrow[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f)
    {
        gq[f] = 0.f;
        qrow[f] = q0 + 16 * f + l15;
        if constexpr (LOC)
        {
            const float *dl = p.decay + (i64)b * p.dB + (i64)min(qrow[f], p.Tq - 1) * p.ldd + head * 4;
            float g = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n)
                g += ((float)(n + 1) * 0.5f) * (0.5f / (1.0f + __expf(-dl[n])));
            gq[f] = g * kLog2e;
        }
    }

    // staging: branch-free loads into native vectors; keys beyond Tk re-read the last valid row (their
    // scores are masked to -inf in the last tile, so P = 0 meets finite V values).
    // K: thread -> (key, dim-quad), NL float4 per tile. V: thread -> (key-quad, dim-quad): the same dim-quad of
    // 4 consecutive keys (NV = 4 float4 per pass), transposed on the way to LDS.
    static_assert((KT * DQ) % 256 == 0, "whole staging passes");
    constexpr int NP = (KT / 4 * DQ + 255) / 256; // V passes per tile (1 for d_h = 64 and 48)
    f32x4 kreg[NL], vreg[NP][4];
    f32x4 o[QF][DF];
    float mrun[QF], lrun[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f)
    {
        mrun[f] = -INFINITY;
        lrun[f] = 0.f;
#pragma unroll
        for (int d = 0; d < DF; ++d)
            o[f][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int nt = (p.Tk + KT - 1) / KT;
    const bool partial = (p.Tk % KT) != 0;
    // ---- software pipeline over key tiles: the scores of tile t+1 are computed (MFMA) while the softmax of tile t
    // runs (VALU) in the same wave - the two have no data dependence, so the matrix pipe does not wait for the
    // exp/max/sum chain. LDS therefore holds K one tile AHEAD of V:
    //   iteration t:  reads K(t+1) [Ks[(t+1)&1]] and V(t) [Vt[t&1]];  stages K(t+2) -> Ks[t&1], V(t+1) -> Vt[(t+1)&1]
    // (K(t) was read in iteration t-1, V(t-1) in iteration t-1: both buffers are free). The arithmetic and its
    // order - tile by tile, deferred maximum per 16-query fragment - are unchanged.
    // K(t0 ..) -> registers, or (KDIRECT) straight into Ks<buf>: thread (key = tid/16 + 16 i, slot = tid & 15)
    // fetches dim-quad slot ^ (key & 15); a wave fills 4 keys = 1 KB contiguous from the wave-uniform base
    auto load_k = [&](int t0, int buf) {
#pragma unroll
        for (int i = 0; i < NL; ++i)
        {
            const int idx = tid + i * 256;
            const int key = idx / DQ, dq = idx - key * DQ;
            const i64 r = min(t0 + key, p.Tk - 1);
            if constexpr (KDIRECT)
                load_to_lds_b128(K + r * p.ldk + 4 * (dq ^ (key & 15)), &(buf ? Ks1 : Ks0)[(4 * wave + 16 * i) * DQ]);
            else
                kreg[i] = *reinterpret_cast<const f32x4 *>(K + r * p.ldk + 4 * dq);
        }
    };
    auto load_v = [&](int t0) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
        {
            const int idx = tid + i * 256;
            const int kq4 = min(idx / DQ, KT / 4 - 1), dq = idx - (idx / DQ) * DQ;
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                const i64 r = min(t0 + 4 * kq4 + j, p.Tk - 1);
                vreg[i][j] = *reinterpret_cast<const f32x4 *>(V + r * p.ldv + 4 * dq);
            }
        }
    };
    auto store_k = [&](int buf) {
        if constexpr (!KDIRECT)
        {
#pragma unroll
            for (int i = 0; i < NL; ++i)
            {
                const int idx = tid + i * 256;
                const int key = idx / DQ, dq = idx - key * DQ;
                *reinterpret_cast<f32x4 *>(&(buf ? Kq1 : Kq0)[dq][key ^ (2 * (dq & 3))]) = kreg[i];
            }
        }
    };
    auto store_v = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
        {
            const int idx = tid + i * 256;
            const int kq4 = idx / DQ, dq = idx - kq4 * DQ;
            if ((KT / 4 * DQ) % 256 == 0 || kq4 < KT / 4)
            {
#pragma unroll
                for (int c = 0; c < 4; ++c) // dim 4 dq + c: keys 4 kq4 .. +3
                    *reinterpret_cast<f32x4 *>(&(buf ? Vt1 : Vt0)[kq4][(4 * dq + c) ^ ((dq >> 1) & 3)]) =
                        f32x4{vreg[i][0][c], vreg[i][1][c], vreg[i][2][c], vreg[i][3][c]};
            }
        }
    };
    // S^T = K Q^T of the tile in Ks[buf] (4 key fragments x QF query fragments); dim step outermost, consecutive
    // MFMAs hit different accumulators
    // (kk0 .. kk1: dim chunks of 16 - the pipelined step issues them in pieces between the parts of the softmax)
    auto scores_part = [&](int buf, f32x4 (*sT)[4], int kk0, int kk1) {
        if (kk0 == 0)
        {
#pragma unroll
            for (int f = 0; f < QF; ++f)
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
                    sT[f][kf] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kk = kk0; kk < kk1; ++kk)
        {
            float4 kv[4];
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
            {
                if constexpr (KDIRECT)
                    kv[kf] = (buf ? Ks1 : Ks0)[(16 * kf + l15) * DQ + ((4 * kk + h4) ^ l15)];
                else
                    kv[kf] = (buf ? Kq1 : Kq0)[4 * kk + h4][(16 * kf + l15) ^ (2 * h4)];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int f = 0; f < QF; ++f)
                        sT[f][kf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4c(kv[kf], c), a4c(qf[f][kk], c), sT[f][kf], 0, 0, 0);
        }
    };
    // online softmax of tile t for query (f, l15); lane holds keys 16kf + 4h4 + r (attention_common.h). MASK: the (only)
    // tile of a Tk that is not a multiple of 64
    float mcur[QF]; // maximum the current tile of fragment f is exponentiated against (set by softmax_pre)
    auto softmax_pre = [&](int t, int f, f32x4 (*sT)[4], auto maskTag) {
        constexpr bool MASK = decltype(maskTag)::value;
        att_softmax_pre<DF, MASK, LOC>(sT[f], o[f], mrun[f], lrun[f], mcur[f], t, h4, p.Tk, qrow[f], gq[f]);
    };
    auto softmax_post = [&](int f, f32x4 (*sT)[4]) {
        att_softmax_post(sT[f], lrun[f], mcur[f]);
    };
    // O_f^T += V^T P_f^T : A = V^T[dim = 16d + l15][key = 16kf + 4h4 + c] = one float4 of the V image
    auto pvprod = [&](int buf, f32x4 (*sT)[4]) {
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
        {
            f32x4 vv[DF];
#pragma unroll
            for (int d = 0; d < DF; ++d)
            {
                vv[d] = *reinterpret_cast<const f32x4 *>(&(buf ? Vt1 : Vt0)[4 * kf + h4][(16 * d + l15) ^ (((16 * d + l15) >> 3) & 3)]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < DF; ++d)
#pragma unroll
                    for (int f = 0; f < QF; ++f)
                        o[f][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[d][c], sT[f][kf][c], o[f][d], 0, 0, 0);
        }
    };

    // prologue: K(0), V(0) -> LDS; K(1) -> LDS; S(0)
    load_k(0, 0);
    load_v(0);
    store_k(0);
    store_v(0);
    load_k(KT, 1);
    store_k(1);
    if constexpr (KDIRECT)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's direct loads have landed
    __syncthreads();
    f32x4 sA[QF][4], sB[QF][4];
    scores_part(0, sA, 0, DF);
    // Step 0 puts K(2) into the buffer of K(0) - KDIRECT requests it at once, the register-staged form writes it at the END of the
    // step: every wave must have read K(0) first. (Until round 6 only KDIRECT had this barrier: "a wave cannot fall a whole step
    // behind" does not hold when eight contexts share the GPU - a late wave of a head-dim-48 launch then took part of K(2) for K(0)
    // in its first tile: errors of 1e-6 .. 4e-5 in whole segments, caught by the eight-logical-device test of the 6-source model
    // in fp32 mode, profiles/r06_experiments/attention48_prologue_race.txt.)
    __syncthreads();
    // one pipelined step (PAR = t & 1, compile-time): softmax + PV of tile t from sCur, scores of tile t+1 into sNext
    auto step = [&](int t, auto parTag, f32x4 (*sCur)[4], f32x4 (*sNext)[4]) {
        constexpr int PAR = decltype(parTag)::value;
        load_k((t + 2) * KT, PAR); // beyond the end: clamped re-read, never used. KDIRECT: lands in the buffer of K(t), read last in step t-1
        if constexpr (!KDIRECT || QF == 1)
            load_v((t + 1) * KT);
        // The deferred-maximum test is a (wave-uniform) branch, and the scheduler interleaves only inside a basic
        // block: the MFMAs of the next tile's scores are therefore issued in pieces, one in front of each part
        // of the softmax, so that every block holds matrix work next to its VALU chain.
        constexpr int H = DF / 2 > 0 ? DF / 2 : 1;
        scores_part(PAR ^ 1, sNext, 0, H);
        softmax_pre(t, 0, sCur, std::false_type{});
        if constexpr (QF == 2)
        {
            scores_part(PAR ^ 1, sNext, H, H + (DF - H) / 2);
            softmax_post(0, sCur);
            softmax_pre(t, 1, sCur, std::false_type{});
            scores_part(PAR ^ 1, sNext, H + (DF - H) / 2, DF);
            softmax_post(1, sCur);
        }
        else
        {
            scores_part(PAR ^ 1, sNext, H, DF);
            softmax_post(0, sCur);
        }
        if constexpr (KDIRECT && QF == 2)
        {
            // 128-query shape: the scores of two tiles, O and Q fill the register file; V(t+1) is requested only
            // now (its 16 staging registers are not live during the phase above) and has the 128 MFMAs of the
            // PV product to arrive
            __builtin_amdgcn_sched_barrier(0);
            load_v((t + 1) * KT);
            __builtin_amdgcn_sched_barrier(0);
        }
        pvprod(PAR, sCur);
        store_k(PAR);
        store_v(PAR ^ 1);
        if constexpr (KDIRECT)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    auto last = [&](int t, auto parTag, f32x4 (*sCur)[4]) {
        constexpr int PAR = decltype(parTag)::value;
#pragma unroll
        for (int f = 0; f < QF; ++f)
        {
            if (partial)
                softmax_pre(t, f, sCur, std::true_type{});
            else
                softmax_pre(t, f, sCur, std::false_type{});
            softmax_post(f, sCur);
        }
        pvprod(PAR, sCur);
    };
    const std::integral_constant<int, 0> even{};
    const std::integral_constant<int, 1> odd{};
    int t = 0;
    for (; t + 2 < nt; t += 2)
    {
        step(t, even, sA, sB);
        step(t + 1, odd, sB, sA);
    }
    if (t + 1 < nt) // two tiles left: t (full) and t+1 (last)
    {
        step(t, even, sA, sB);
        last(t + 1, odd, sB);
    }
    else // one tile left
        last(t, even, sA);
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

void launch_attention(const AttnArgs &a0, hipStream_t s)
{
    constexpr int xcdMap = 1; // all query tiles of one (batch, head) on one XCD
    AttnArgs a = a0;
    a.xcdMap = xcdMap;
    const bool big = att_use_big_shape(a); // 128- or 64-query workgroups (attention_common.h)
    if (a.hs != 64 && a.hs != 48)
        abort();
    a.nQt = (unsigned)(big ? (a.Tq + 127) / 128 : (a.Tq + 63) / 64);
    const unsigned nbh = (unsigned)(a.B * a.H);
    const dim3 grid(xcdMap ? 8u * ((nbh + 7u) / 8u) * a.nQt : nbh * a.nQt);
    if (big)
    {
        if (a.hs == 64)
            hipLaunchKernelGGL((attention_kernel<64, 2>), grid, dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((attention_kernel<48, 2>), grid, dim3(256), 0, s, a);
    }
    else
    {
        if (a.hs == 64)
            hipLaunchKernelGGL((attention_kernel<64, 1>), grid, dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((attention_kernel<48, 1>), grid, dim3(256), 0, s, a);
    }
}

int launch_attention_local(const AttnArgs &a0, hipStream_t s)
{
    constexpr int xcdMap = 1; // all query tiles of one (batch, head) on one XCD
    AttnArgs a = a0;
    a.xcdMap = xcdMap;
    if (!a.decay || (a.hs != 48 && a.hs != 96))
        return -1;
    a.nQt = (unsigned)((a.Tq + 63) / 64); // 64-query workgroups: T = 336 / 168, a handful of key tiles per query tile
    const unsigned nbh = (unsigned)(a.B * a.H);
    const dim3 grid(xcdMap ? 8u * ((nbh + 7u) / 8u) * a.nQt : nbh * a.nQt);
    if (a.hs == 48)
        hipLaunchKernelGGL((attention_kernel<48, 1, true>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((attention_kernel<96, 1, true>), grid, dim3(256), 0, s, a);
    return 0;
}

} // namespace dmx

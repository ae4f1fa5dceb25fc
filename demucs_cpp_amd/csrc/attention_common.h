// attention_common.h — the online softmax shared by the two flash-attention kernels (attention.hip: fp32 MFMA;
// attention_split.hip: exact bf16 operand splits). Restates the per-head block of /root/reference/src/layers.cpp:442-482
// (max-subtracted exp, row sum, division) in the tiled form: scores arrive as S^T fragments in the exp2 domain (Q is
// pre-multiplied by scale * log2 e), lane (l15, h4) of a wave holds, for the query 16 f + l15 of a fragment, the scores of
// keys 16 kf + 4 h4 + r (kf < 4, r < 4) of the current 64-key tile.
#pragma once
#include "kernels.h"

namespace dmx
{

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); } // v_max3_f32
static constexpr float kDeferLog2 = 16.0f; // exp2-domain threshold of the deferred running maximum
static constexpr float kLog2e = 1.44269504088896340736f;

// Reductions over the 4 lanes {l, l^16, l^32, l^48} that hold one query's keys, on the VALU:
// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of the second,
// v_permlane32_swap the upper half of the first with the lower half of the second; applied to two copies of x
// they leave (x[row^1] | x) resp. (x[half^1] | x) side by side.
__device__ __forceinline__ float quad_lanes_max(float x)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float quad_lanes_sum(float x)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// First half of the softmax of one 16-query fragment on tile t: decay penalty / masking, tile maximum, and the DEFERRED
// raise of the running maximum with the rescale of O. MASK: the (only) tile of a Tk that is not a multiple of 64.
// LOC (Demucs v3 LocalState, /root/reference/src/layers.cpp:533-721): score(query s, key t) -= |t - s| g(s), diagonal = -100.
// The running maximum is raised only when some score of this 16-query fragment exceeds it by more than 2^16
// (wave-uniform per fragment, hence the same decision in the 64- and the 128-query workgroup shape; always taken on
// the first tile, mrun = -inf). Otherwise the tile is exponentiated against the old maximum - in fp32 a common factor
// <= 2^16 on P and on the row sum costs no accuracy - and the rescale of O (one exp + 17 multiplies per fragment) is
// skipped. (Measured alternative, kept out: the rescale without a branch, alpha = 1 when the maximum stays, makes the
// pipelined step ONE basic block so that the scheduler can spread the next tile's score MFMAs over all the
// exponentials - but the always-executed rescale and the longer live ranges cost more than the overlap gains:
// 118.3 vs 123.8 TFLOP/s for the 64-query shape, spills in the loop for the 128-query shape.)
template <int DF, bool MASK, bool LOC>
__device__ __forceinline__ void att_softmax_pre(f32x4 (&sT)[4], f32x4 (&o)[DF], float &mrun, float &lrun, float &mcur, int t, int h4, int Tk,
                                                int qrow, float gq)
{
    constexpr int KT = 64;
    if constexpr (LOC)
    {
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int key = t * KT + 16 * kf + 4 * h4 + r;
                const float dist = fabsf((float)(key - qrow));
                sT[kf][r] = key == qrow ? -100.0f * kLog2e : sT[kf][r] - dist * gq;
            }
    }
    if constexpr (MASK)
    {
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (t * KT + 16 * kf + 4 * h4 + r >= Tk)
                    sT[kf][r] = -INFINITY;
    }
    float tmax = max3f(sT[0][0], sT[0][1], sT[0][2]);
    tmax = max3f(tmax, sT[0][3], sT[1][0]);
    float tmx2 = max3f(sT[1][1], sT[1][2], sT[1][3]);
    tmx2 = max3f(tmx2, sT[2][0], sT[2][1]);
    float tmx3 = max3f(sT[2][2], sT[2][3], sT[3][0]);
    tmx3 = max3f(tmx3, sT[3][1], sT[3][2]);
    tmax = max3f(tmax, tmx2, fmaxf(tmx3, sT[3][3]));
    tmax = quad_lanes_max(tmax);
    float mnew = mrun;
    if (__builtin_amdgcn_ballot_w64(tmax > mnew + kDeferLog2) != 0ull)
    {
        mnew = fmaxf(mnew, tmax);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew); // first tile: exp2(-inf) = 0
        lrun *= alpha;
        mrun = mnew;
#pragma unroll
        for (int d = 0; d < DF; ++d)
        {
            o[d][0] *= alpha;
            o[d][1] *= alpha;
            o[d][2] *= alpha;
            o[d][3] *= alpha;
        }
    }
    mcur = mnew;
}
// second half: P = exp2(S - m) in place, row-sum partial of this lane added to lrun
__device__ __forceinline__ void att_softmax_post(f32x4 (&sT)[4], float &lrun, float mcur)
{
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const float pv = __builtin_amdgcn_exp2f(sT[kf][r] - mcur); // bare v_exp_f32: underflow to 0 is the right answer
            sT[kf][r] = pv;
            ps[kf] += pv;
        }
    lrun += (ps[0] + ps[1]) + (ps[2] + ps[3]);
}

// workgroup -> (query tile, batch*head). Workgroup w runs on XCD w % 8: with the XCD-aware map every
// query tile of one (batch, head) lands on the same XCD, so its K and V (2 x Tk x d_h x 4 B, 1.4 MB)
// are fetched once and re-read from that XCD's L2 instead of once per XCD. Returns false for a workgroup without work.
__device__ __forceinline__ bool att_tile_of_block(const AttnArgs &p, unsigned &qt, unsigned &bh)
{
    if (p.xcdMap)
    {
        const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const unsigned g = j / p.nQt;
        qt = j - g * p.nQt;
        bh = g * 8u + xcd;
        return bh < (unsigned)(p.B * p.H);
    }
    bh = blockIdx.x / p.nQt;
    qt = blockIdx.x - bh * p.nQt;
    return true;
}

// 32 queries per wave (128 per workgroup) when that still leaves >= 2 rounds of workgroups per CU; the key-tile order,
// hence every rounding, is the same for both shapes. (rounds of the 512 resident workgroups) x (relative duration of
// one workgroup): the 64-query shape does ~0.6x the work of the 128-query one per workgroup (K/V LDS reads amortised
// over half the queries); e.g. Tq = 1344 at batch 12 is 1056 big workgroups = 2.06 rounds -> 3, or 2016 small = 3.94 ->
// 4 x 0.6. Below two full rounds of 128-query workgroups the launch is latency-bound, not matrix-bound, and the 64-query
// shape (twice the waves for the same work) wins whatever the round count says: measured at 1 / 2 / 4 segments
// 1.12 / 1.78 / 3.17 ms (64) vs 1.39 / 2.05 / 3.32 ms (128) for the ten attention launches of a plan run.
inline bool att_use_big_shape(const AttnArgs &a)
{
    const long wg128 = (long)((a.Tq + 127) / 128) * a.H * a.B, wg64 = (long)((a.Tq + 63) / 64) * a.H * a.B;
    const double costBig = (double)((wg128 + 511) / 512), costSmall = 0.6 * (double)((wg64 + 511) / 512);
    return wg128 >= 1024 && costBig <= costSmall;
}

} // namespace dmx

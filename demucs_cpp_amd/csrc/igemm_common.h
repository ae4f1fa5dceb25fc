// igemm_common.h — device code shared by the implicit-GEMM kernels: igemm.hip (fp32 MFMA) and igemm_split.hip (exact
// bf16 operand splits). One text for everything the two families have in common:
//   * workgroup -> tile map (XCD-aware) and row decomposition m -> (b, p1, p0, group);
//   * StageWalk: the K walk of the staging threads over an implicit im2col row (conv addressing with one validity bit
//     per tap, or the linear-layer fast path), i.e. where the next float4 of an A row / a weight row comes from;
//   * the prologue transform of a staged A chunk (z-norm, GroupNorm + GELU);
//   * the epilogue (bias, GELU, residual, LayerScale, GLU, GroupNorm + GLU, transposed-conv scatter, row statistics):
//     the accumulator layout of the 16x16 MFMAs does not depend on the operand type, so the epilogue is the same
//     function for both families;
//   * the three-term bf16 split of an activation (igemm_split.hip, attention.hip, dmx_debug_split_activations).
// Reference semantics of the layers: /root/reference/src/conv.hpp:13-524, src/layers.cpp:9-531 (cited per op in plan.cpp).
#pragma once
#include "kernels.h"

namespace dmx
{

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float gelu_f(float v) { return dmx_gelu(v); }
// 1 / (1 + e^-v) with v_rcp_f32 (1 ulp): the IEEE division expands to ~10 VALU instructions, and a GLU
// epilogue evaluates one sigmoid per output
__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float f4c(const f32x4 &v, int c) { return v[c]; }
// `ok ? *ptr : zero` as written selects between a global pointer and a private temporary and loads
// through a FLAT pointer (plus a scratch slot); select the address against the zero page instead
__device__ __forceinline__ float4 ld4z(const float *ptr, bool ok, const float *zero)
{
    const f32x4 v = *reinterpret_cast<const f32x4 *>(ok ? ptr : zero);
    return make_float4(v[0], v[1], v[2], v[3]);
}

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

// ---------------------------------------------------------------------------------------------- exact bf16 splits
// x = a1 + a2 + a3 with a1 = bf16(x), a2 = bf16(x - a1), a3 = bf16(x - a1 - a2), every conversion round-to-nearest-even
// (v_cvt_pk_bf16_f32). Each remainder is exact in fp32 and at most half an ulp of the term before it, so
// |a2| <= 2^-8 |x|, |a3| <= 2^-16 |x|, and the 3 x 8 significand bits plus the two remainder signs cover all 24 bits of
// an fp32: the sum is EXACT for 2^-109 <= |x| <= 0x7f7f7fff (3.3895e38). Outside (measured on the device,
// tests/test_gpu_parity.py::test_activation_split_is_exact_and_bounded_on_the_device): below 2^-109 a remainder can be a
// denormal, which the conversion flushes - the sum is then within 2^-125 of x, less than the smallest normal fp32; in the
// last half-ulp of bf16's range bf16(x) rounds to inf, and for x = +-inf / NaN a1 = x, a2 = a3 = NaN - an operand out of
// range can only make a product non-finite, as a non-finite operand would in the fp32 kernels. Values are returned as
// PAIRS packed for the MFMA operand registers: low 16 bits = the term of x0, high 16 bits = the term of x1.
__device__ __forceinline__ unsigned bf16_pk_rn(float x0, float x1)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
}
__device__ __forceinline__ void split3_pk(float x0, float x1, unsigned &h1, unsigned &h2, unsigned &h3)
{
    h1 = bf16_pk_rn(x0, x1);
    const float r0 = x0 - __uint_as_float(h1 << 16), r1 = x1 - __uint_as_float(h1 & 0xffff0000u);
    h2 = bf16_pk_rn(r0, r1);
    const float s0 = r0 - __uint_as_float(h2 << 16), s1 = r1 - __uint_as_float(h2 & 0xffff0000u);
    h3 = bf16_pk_rn(s0, s1);
}

// ---------------------------------------------------------------------------------------------- fp16 terms (DMX_GEMM_FP16X3, opt-in)
// x = h1 + h2 + h3 with h1 = fp16(x), h2 = fp16(x - h1), h3 = fp16(x - h1 - h2), conversions round-to-nearest-even
// (v_cvt_pk_f16_f32; fp16 subnormals kept). 11 + 11 + 2 significand bits: EXACT for 0.5 <= |x| <= 65504; below 0.5 the last
// bits fall under fp16's subnormal spacing 2^-24 and the sum is within 2^-25 of x; above 65504 h1 is inf (measured per binade
// on the device: tools/micro/split_fp16.hip, tests). The kernels only ever split x = a 2^s with the power of two s chosen per
// ROW so that the row's largest |a 2^s| lies in [2^14, 2^15): no element can overflow, an element loses bits only when it is
// more than 2^15 times smaller than the largest of its row, and then at most 2^-25 absolute in scaled units, i.e. 2^-39 of
// the row's maximum. The weights of the model files ARE fp16 numbers, so a weight is ONE exact term and a product term needs
// three MFMAs (h3 w, h2 w, h1 w) instead of five. Packed like split3_pk: low 16 bits = the term of x0, high = of x1.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split3h_pk(float x0, float x1, unsigned &h1, unsigned &h2, unsigned &h3)
{
    const f16x2_t a = __builtin_convertvector(f32x2{x0, x1}, f16x2_t);
    const f32x2 af = __builtin_convertvector(a, f32x2);
    const float r0 = x0 - af[0], r1 = x1 - af[1];
    const f16x2_t b = __builtin_convertvector(f32x2{r0, r1}, f16x2_t);
    const f32x2 bf = __builtin_convertvector(b, f32x2);
    const float q0 = r0 - bf[0], q1 = r1 - bf[1];
    const f16x2_t c = __builtin_convertvector(f32x2{q0, q1}, f16x2_t);
    h1 = __builtin_bit_cast(unsigned, a), h2 = __builtin_bit_cast(unsigned, b), h3 = __builtin_bit_cast(unsigned, c);
}
// the row scale of the fp16-term kernels from the row's largest magnitude: {2^s, 2^-s} with s = 14 - floor(log2 max), so the
// scaled row's largest magnitude lies in [2^14, 2^15) - whatever finite fp32 number it was, the fp16 terms cannot overflow
// (s >= -113 for every finite maximum). s is capped at 126 (2^-s must stay a normal number): rows whose largest magnitude is
// below 2^-112 are scaled by 2^126 and lose at most 2^-25 of the SCALED value = 2^-151 absolute, less than fp32's smallest
// denormal step. Rows of zeros: 2^126. inf / NaN maximum: 2^-114, the products stay non-finite.
__device__ __forceinline__ float2 rowscale_of(float amax)
{
    int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127; // floor(log2 amax) for normal numbers; -127 for denormals and zero
    int sexp = 14 - e;
    sexp = sexp > 126 ? 126 : sexp; // (>= -114 by construction)
    return make_float2(__uint_as_float((unsigned)(127 + sexp) << 23), __uint_as_float((unsigned)(127 - sexp) << 23));
}

// ---------------------------------------------------------------------------------------------- tiles and rows
// workgroup -> tile. Workgroup b is dispatched to XCD b % 8 (observed; used for speed only). With the
// XCD-aware map all column tiles of a row tile run on the same XCD right after one another, so the
// A row block is fetched from HBM / Infinity Cache once and re-read from that XCD's 4 MB L2, and the
// 64 workgroups resident on an XCD form a (few row tiles) x (all column tiles) patch that shares both
// operands' k-slices. Row tiles are dealt round-robin to the XCDs (balanced to one tile).
// Returns false for a workgroup without a tile (the whole workgroup leaves, before any barrier).
__device__ __forceinline__ bool tile_of_block(const GemmArgs &p, unsigned &tileM, unsigned &tileN)
{
    if (p.xcdMap)
    {
        const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const unsigned mi = j / p.tilesN;
        tileN = j - mi * p.tilesN;
        tileM = mi * 8u + xcd;
        return tileM < p.tilesM;
    }
    tileN = blockIdx.x / p.tilesM;
    tileM = blockIdx.x - tileN * p.tilesM;
    return true;
}
// row m -> (b, p1, p0, group), .w = -1 beyond M. Magic-number divisions (kernels.h FastDiv): three 64-bit software
// divides per row were a visible part of the per-tile prologue.
__device__ __forceinline__ int4 row_info(const GemmArgs &p, i64 m)
{
    int4 ri = make_int4(0, 0, 0, -1);
    if (m < p.M)
    {
        const unsigned mu = (unsigned)m;
        const unsigned t = p.dP0.magic ? (__umulhi(mu, p.dP0.magic) >> p.dP0.shift) : (mu >> p.dP0.shift);
        const int p0 = (int)(mu - t * (unsigned)p.P0);
        const unsigned b = p.dP1.magic ? (__umulhi(t, p.dP1.magic) >> p.dP1.shift) : (t >> p.dP1.shift);
        const int p1 = (int)(t - b * (unsigned)p.P1);
        ri = make_int4((int)b, p1, p0, (int)b * p.G0 + (p.G0 > 1 ? p0 : 0));
    }
    return ri;
}

// ---------------------------------------------------------------------------------------------- staging walk
// Per-thread state of the staging: AR rows of A and BR rows of B, all at k-quad `slaneK` of the K-tile (KT = floats per
// K-tile). Sequential K walk, one tile = KT consecutive k; this lane stages k = kl .. kl+3. (s1, offb) = conv tap along
// axis 1 / offset inside its contiguous run, advanced per lane without division.
// LIN: "linear layer" addressing - one contiguous run of K floats per row (S1 == 1, no padding, K a multiple of the
// K-tile): the staging addresses of a row just advance by one K-tile per iteration, no per-tile bounds checks, tap
// bookkeeping or pointer selects (transformer linears, 1x1 rewrites).
// General (conv) addressing: validity of a staged chunk depends only on (row, tap), tap c = k / Cin =
// s1 * (seg0 / Cin) + (tap along axis 0) - padding starts and ends at whole taps - so each row carries ONE
// bit per tap, computed once per tile; the K walk then tests a bit instead of re-deriving four range
// checks per row and K-tile (measured: the address arithmetic of the general path cost the 3x3 rewrites 10 %).
// Out-of-range chunks point at the zero page: PRO_NONE needs no masking at all.
template <int AR, int BR, int KT, int PRO, bool LIN>
struct StageWalk
{
    const GemmArgs &p;
    const int slaneK;
    int rowLenI;
    const float *aRow[AR]; // X + b*xBS + in1_0*rowLen + e0  (tap s1 = 0, k = 0)
    bool aRowOk[AR];
    unsigned aTapMask[AR];
    float aMean[AR], aScale[AR];
    const float *bRow[BR];
    bool bRowOk[BR];
    int kl, s1, offb;
    int tapC, tapOff; // tap index kl / Cin and offset inside the tap
    const float *addrA[AR], *addrB[BR], *addrG;
    unsigned maskNext;
    i64 stepA[AR], stepB[BR]; // LIN: per-row advance (0 for rows that stay on the zero page)
    bool linInit;
    int segOffCur;
    unsigned tapBit;

    __device__ __forceinline__ StageWalk(const GemmArgs &p_, int slaneK_) : p(p_), slaneK(slaneK_) {}

    // A row i of this thread is tile row rowA(i) (rowinfo: tile row -> int4), B row i is weight row n0 + rowB(i)
    template <typename RowInfo, typename RowA, typename RowB>
    __device__ __forceinline__ void init(RowInfo rowinfo, RowA rowA, RowB rowB, int n0, int BN)
    {
        const i64 rowLen = (i64)p.L0 * p.Cin;
        rowLenI = (int)rowLen;
        const int taps0 = p.seg0 / p.Cin;
#pragma unroll
        for (int i = 0; i < AR; ++i)
        {
            const int4 ri = rowinfo(rowA(i));
            aRowOk[i] = ri.w >= 0;
            const int in1_0 = ri.y * p.stride1 - p.pad1;
            const int e0 = (ri.z * p.stride0 - p.pad0) * p.Cin;
            aRow[i] = p.X + (i64)ri.x * p.xBS + (i64)in1_0 * rowLen + e0;
            aTapMask[i] = 0;
            if (!LIN && aRowOk[i])
            {
                unsigned m0bits = 0; // taps along axis 0 whose chunk lies inside the row
                for (int t0 = 0; t0 < taps0; ++t0)
                {
                    const int e = e0 + t0 * p.Cin;
                    m0bits |= (e >= 0 && e < rowLenI ? 1u : 0u) << t0;
                }
                for (int s = 0; s < p.S1; ++s)
                {
                    const int in1 = in1_0 + s * p.dil1;
                    if (in1 >= 0 && in1 < p.L1)
                        aTapMask[i] |= m0bits << (s * taps0);
                }
            }
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
#pragma unroll
        for (int i = 0; i < BR; ++i)
        {
            const int rl = rowB(i);
            const int n = n0 + rl;
            bRowOk[i] = rl < BN && n < p.Np;
            bRow[i] = p.Wt + (i64)(bRowOk[i] ? n : 0) * p.Kp;
        }
        kl = slaneK * 4, s1 = 0, offb = slaneK * 4;
        if (p.S1 > 1)
            while (offb >= p.seg0)
            {
                offb -= p.seg0;
                ++s1;
            }
        tapC = 0, tapOff = slaneK * 4;
        if (!LIN)
            while (tapOff >= p.Cin)
            {
                tapOff -= p.Cin;
                ++tapC;
            }
        addrG = p.zero;
        maskNext = 0;
        linInit = false;
        segOffCur = 0;
        tapBit = 0;
    }

    // addresses of the next tile to fetch, in two halves (the interleaved loops slot them between MFMA groups):
    // addrs_A = validity + A row addresses, addrs_B = B row addresses + advance of the K walk.
    // general addressing of the A rows; half = 0 / 1: first / second half of the rows (the tile-wide
    // quantities are set up with the first half), 2: all rows
    __device__ __forceinline__ void addrs_A_general(int half)
    {
        if (half != 1)
        {
            maskNext = 0;
            tapBit = tapC < 32 ? 1u << tapC : 0u; // taps beyond K (k >= K) have no bit in any row mask
            segOffCur = s1 * p.dil1 * rowLenI + offb; // fits int32 for every layer of the model
            if (PRO == PRO_GN_GELU)
                addrG = p.proW + (kl < p.K ? kl : 0);
        }
#pragma unroll
        for (int i = 0; i < AR; ++i)
            if (half == 2 || (i < (AR + 1) / 2) == (half == 0))
            {
                const bool ok = (aTapMask[i] & tapBit) != 0u;
                addrA[i] = ok ? aRow[i] + segOffCur : p.zero;
                maskNext |= (ok ? 1u : 0u) << i;
            }
    }
    __device__ __forceinline__ void addrs_A()
    {
        if (LIN)
        {
            if (!linInit)
            {
                maskNext = 0;
#pragma unroll
                for (int i = 0; i < AR; ++i)
                {
                    addrA[i] = aRowOk[i] ? aRow[i] + slaneK * 4 : p.zero;
                    stepA[i] = aRowOk[i] ? KT : 0;
                    maskNext |= (aRowOk[i] ? 1u : 0u) << i;
                }
                if (PRO == PRO_GN_GELU)
                    addrG = p.proW + slaneK * 4;
                return;
            }
#pragma unroll
            for (int i = 0; i < AR; ++i)
                addrA[i] += stepA[i];
            if (PRO == PRO_GN_GELU)
                addrG += KT;
            return;
        }
        addrs_A_general(2);
    }
    __device__ __forceinline__ void addrs_B()
    {
        if (LIN)
        {
            if (!linInit)
            {
                linInit = true;
#pragma unroll
                for (int i = 0; i < BR; ++i)
                {
                    addrB[i] = bRowOk[i] ? bRow[i] + slaneK * 4 : p.zero;
                    stepB[i] = bRowOk[i] ? KT : 0;
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < BR; ++i)
                addrB[i] += stepB[i];
            return;
        }
        // B rows advance like a linear layer's. k >= Kp (second half of the last K-tile when Kp is an odd
        // multiple of 16) reads the next weight row / the zeroed tail of the blob: those k meet A chunks of
        // the zero page (no tap bit), and 0 x finite adds exactly 0.
        if (!linInit)
        {
            linInit = true;
#pragma unroll
            for (int i = 0; i < BR; ++i)
            {
                addrB[i] = bRowOk[i] ? bRow[i] + slaneK * 4 : p.zero;
                stepB[i] = bRowOk[i] ? KT : 0;
            }
        }
        else
        {
#pragma unroll
            for (int i = 0; i < BR; ++i)
                addrB[i] += stepB[i];
        }
        kl += KT;
        offb += KT;
        if (p.S1 > 1 && offb >= p.seg0)
        {
            offb -= p.seg0;
            ++s1;
        }
        tapOff += KT;
        while (tapOff >= p.Cin)
        {
            tapOff -= p.Cin;
            ++tapC;
        }
    }
    __device__ __forceinline__ void compute_addrs()
    {
        addrs_A();
        addrs_B();
    }
    // the same work in pieces for the interleaved loop (piece c goes behind MFMA group c of k-chunk 0)
    __device__ __forceinline__ void addr_piece(int c)
    {
        if (LIN)
        {
            if (c == 0)
                addrs_A();
            if (c == 2)
                addrs_B();
            return;
        }
        if (c < 2)
            addrs_A_general(c);
        if (c == 2)
            addrs_B();
    }
    // prologue transform of a staged chunk of A row i (+ zero fill where a transform would make padding non-zero);
    // gW / gB: the GroupNorm affine of the chunk's four k (PRO_GN_GELU), ok: the chunk lies inside the input
    __device__ __forceinline__ f32x4 transform(f32x4 v, int i, bool ok, const f32x4 &gW, const f32x4 &gB) const
    {
        if (PRO != PRO_NONE)
        {
            if (PRO == PRO_AFFINE)
            {
                v.x = (v.x - aMean[i]) * aScale[i];
                v.y = (v.y - aMean[i]) * aScale[i];
                v.z = (v.z - aMean[i]) * aScale[i];
                v.w = (v.w - aMean[i]) * aScale[i];
            }
            if (PRO == PRO_GN_GELU)
            {
                v.x = gelu_f((v.x - aMean[i]) * aScale[i] * gW.x + gB.x);
                v.y = gelu_f((v.y - aMean[i]) * aScale[i] * gW.y + gB.y);
                v.z = gelu_f((v.z - aMean[i]) * aScale[i] * gW.z + gB.z);
                v.w = gelu_f((v.w - aMean[i]) * aScale[i] * gW.w + gB.w);
            }
            if (!ok)
                v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return v;
    }
};

// ---------------------------------------------------------------------------------------------- epilogue
// The MFMAs are issued with the operands swapped (weights as A, activations as B), so each accumulator holds C^T:
// lane (l15, kq) owns row m = tile row 16 i + l15 and the 4 CONSECUTIVE channels n = 16 j + 4 kq + {0..3} -> one
// float4 global access per fragment, one row-info lookup per row fragment, 2-step cross-lane reduction for the row
// statistics. rowinfo: tile row -> int4 (b, p1, p0, group); rsum: BM x (WAVES_N * SSEG) scratch for the cross-wave row
// statistics (LDS; may alias a staging image that is no longer read - the caller synchronises before the call).
// SSEG: a wave's WNF column fragments are summed into the row statistics as SSEG separate runs of WNF / SSEG fragments,
// each reduced on its own - a 1 x 8-fragment wave with SSEG = 2 adds in exactly the order of two 4-fragment waves, so
// the statistics (and everything normalised with them) do not depend on which of the two wave layouts ran.
template <int WAVES_N, int WMF, int WNF, int EPI, int NT, int SSEG = 1, typename RowInfo>
__device__ __forceinline__ void igemm_epilogue(const GemmArgs &p, f32x4 (&acc)[WMF][WNF], RowInfo rowinfo, float2 (*rsum)[WAVES_N * SSEG], i64 m0,
                                               int n0, unsigned tileN, int wm, int wn, int BM)
{
    static_assert(WNF % SSEG == 0, "whole fragments per statistics run");
    const int tid = threadIdx.x, lane = tid & 63;
    const int l15 = lane & 15, kq = lane >> 4;
    if constexpr (EPI == EPI_VT)
    {
        // The caller issued its MFMAs the OTHER way round (activations as A, weights as B): the accumulator holds C, lane
        // (l15, kq) owns channel n = 16 j + l15 and the 4 CONSECUTIVE tokens 16 i + 4 kq + {0..3} of row block i - one 8-byte piece
        // (4 keys of one dim) of the attention kernel's V^T tile image (attention_split.hip): tile = 64 keys, 16-byte slot
        // 4 s + h4 of a dim's row = keys 32 s + 4 h4 + {0..3} (first half) and 32 s + 16 + 4 h4 + {0..3} (second half): the two
        // halves are the SAME lane's pieces of row blocks i and i + 1 (i even), so a wave tile of 32 rows stores whole slots.
        // Tokens are rows m = b kvT + t with kvT a multiple of 64: the tokens of a 32-row block share b and the tile.
        const int nt = p.kvT >> 6;
        constexpr int IS = (WMF % 2 == 0) ? 2 : 1;
#pragma unroll
        for (int i = 0; i < WMF; i += IS)
        {
            const i64 m = m0 + wm * (WMF * 16) + i * 16 + 4 * kq; // first of this lane's four tokens of row block i
            if (m >= p.M)
                continue;
            const unsigned b = (unsigned)m / (unsigned)p.kvT, t = (unsigned)m - b * (unsigned)p.kvT;
            const unsigned tt = t & 63u, slot = 4u * (tt >> 5) + ((tt >> 2) & 3u), half = (tt >> 4) & 1u;
#pragma unroll
            for (int j = 0; j < WNF; ++j)
            {
                const int n = n0 + wn * (WNF * 16) + j * 16 + l15;
                if (n >= p.N)
                    continue;
                const float bv = p.bias[n];
                const unsigned head = (unsigned)n / (unsigned)p.kvHs, dim = (unsigned)n - head * (unsigned)p.kvHs;
                unsigned short *dst = p.kvPl + ((((i64)b * p.kvH + head) * nt + (t >> 6)) * p.kvHs + dim) * 64 + slot * 8;
                unsigned h1[2 * IS], h2[2 * IS], h3[2 * IS];
#pragma unroll
                for (int u = 0; u < IS; ++u)
                {
                    split3_pk(acc[i + u][j][0] + bv, acc[i + u][j][1] + bv, h1[2 * u], h2[2 * u], h3[2 * u]);
                    split3_pk(acc[i + u][j][2] + bv, acc[i + u][j][3] + bv, h1[2 * u + 1], h2[2 * u + 1], h3[2 * u + 1]);
                }
                if constexpr (IS == 2)
                {
                    *reinterpret_cast<u32x4 *>(dst) = u32x4{h1[0], h1[1], h1[2], h1[3]};
                    *reinterpret_cast<u32x4 *>(dst + p.kvPlane) = u32x4{h2[0], h2[1], h2[2], h2[3]};
                    *reinterpret_cast<u32x4 *>(dst + 2 * p.kvPlane) = u32x4{h3[0], h3[1], h3[2], h3[3]};
                }
                else
                {
                    *reinterpret_cast<u32x2 *>(dst + half * 4) = u32x2{h1[0], h1[1]};
                    *reinterpret_cast<u32x2 *>(dst + half * 4 + p.kvPlane) = u32x2{h2[0], h2[1]};
                    *reinterpret_cast<u32x2 *>(dst + half * 4 + 2 * p.kvPlane) = u32x2{h3[0], h3[1]};
                }
            }
        }
        return;
    }
    if constexpr (EPI == EPI_KPL)
    {
        // v = acc + bias. Column fragments below kvCol0 (the q third of a self-attention projection): fp32 [row][n], as
        // EPI_LINEAR. From kvCol0 on: the three bf16 planes of the exact split, [plane][row][n - kvCol0] - the K operand image
        // of attention_split.hip. A lane owns 4 channels (8 bytes of a plane); fragments are taken in pairs and the 16-lane rows
        // kq = 0 / 1 (2 / 3) exchange halves (v_permlane16_swap) so that a lane stores 8 consecutive channels = 16 bytes of ONE
        // fragment: row kq even -> fragment j, channels 8 (kq >> 1) .. + 7; row kq odd -> fragment j + 1, same channels.
        static_assert(WNF % 2 == 0, "fragment pairs");
#pragma unroll
        for (int i = 0; i < WMF; ++i)
        {
            const int rl = wm * (WMF * 16) + i * 16 + l15;
            const i64 m = m0 + rl;
            const bool rowOk = m < p.M;
#pragma unroll
            for (int j = 0; j < WNF; j += 2)
            {
                const int nb = n0 + wn * (WNF * 16) + j * 16; // first channel of the pair (a multiple of 32)
                float4 v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                    const int n = nb + 16 * u + 4 * kq;
                    const float4 bs = ld4z(p.bias + n, n < p.N, p.zero);
                    v[u] = make_float4(acc[i][j + u][0] + bs.x, acc[i][j + u][1] + bs.y, acc[i][j + u][2] + bs.z, acc[i][j + u][3] + bs.w);
                }
                if (nb < p.kvCol0) // (uniform per fragment pair)
                {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (rowOk && nb + 16 * u + 4 * kq < p.N)
                            *reinterpret_cast<float4 *>(p.Y + m * p.ldy + nb + 16 * u + 4 * kq) = v[u];
                    continue;
                }
                unsigned h[3][2][2]; // [plane][fragment of the pair][dword]
#pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                    split3_pk(v[u].x, v[u].y, h[0][u][0], h[1][u][0], h[2][u][0]);
                    split3_pk(v[u].z, v[u].w, h[0][u][1], h[1][u][1], h[2][u][1]);
                }
                const int nn = nb + 16 * (kq & 1) + 8 * (kq >> 1);
                unsigned short *dst = p.kvPl + m * (i64)(p.kvH * p.kvHs) + (nn - p.kvCol0);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                {
                    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(h[pl][0][0], h[pl][1][0], false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(h[pl][0][1], h[pl][1][1], false, false);
                    if (rowOk && nn < p.N)
                        *reinterpret_cast<u32x4 *>(dst + pl * p.kvPlane) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
            }
        }
        return;
    }
    const bool wantStats = p.rowstat != nullptr;
    const int colBase = n0 + wn * (WNF * 16) + 4 * kq;
    float4 biasv[WNF], scalev[WNF], gnWv[WNF], gnBv[WNF];
    int trR[WNF], trC[WNF]; // EPI_TRCONV: column n -> (phase r, channel co); Cout % 4 == 0
#pragma unroll
    for (int j = 0; j < WNF; ++j)
    {
        const int n = colBase + j * 16;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        biasv[j] = ld4z(p.bias + n, n < p.N, p.zero);
        scalev[j] = gnWv[j] = gnBv[j] = z;
        trR[j] = trC[j] = 0;
        if (EPI == EPI_TRCONV)
        {
            trR[j] = n / p.Cout;
            trC[j] = n - trR[j] * p.Cout;
        }
        if (EPI == EPI_SCALE_RES && n < p.N)
            scalev[j] = *reinterpret_cast<const float4 *>(p.scale + n);
        if (EPI == EPI_GN_GLU_SCALE_RES && n < p.N)
        {
            gnWv[j] = *reinterpret_cast<const float4 *>(p.epiW + n);
            gnBv[j] = *reinterpret_cast<const float4 *>(p.epiB + n);
            if ((j & 1) == 0)
                scalev[j] = *reinterpret_cast<const float4 *>(p.scale + (n >> 5) * 16 + (n & 15));
        }
    }

    // ---- pass 1: EVERY global load of the epilogue (residuals / tables / GroupNorm statistics of all WMF row blocks), then one
    // wait. On gfx9 stores count in vmcnt like loads and the counter retires in order: a load issued after a store cannot be
    // waited for without waiting for the store's acknowledgement, and the compiler, which loses the exact count across the
    // row / column guards, waits with vmcnt(0). With the residual loads of row block i + 1 behind the stores of row block i
    // (the first form of this function) every store of a tile was followed by a full L2 round trip before the next one was
    // issued: 16 stores took 8 us of a 34 us linear-layer tile (tools/gpu_wg_timeline.py). res may alias Y element-wise
    // (in-place updates): every element is read here before the same lane overwrites it below.
    // Row blocks are taken in groups of G (at most 16 fragments of operands in registers: the 256-row tiles would spill):
    // one wait - and one store round trip - per group.
    constexpr int G = (16 / WNF) < 1 ? 1 : ((16 / WNF) > WMF ? WMF : (16 / WNF));
    int4 riA[G];
    float4 resA[G][WNF];
    float meanA[G], scA[G];
    i64 offsA[G][WNF]; // (EPI_TRCONV only; dead otherwise)
#pragma unroll
    for (int g0 = 0; g0 < WMF; g0 += G)
    {
#pragma unroll
    for (int ig = 0; ig < G; ++ig)
    {
        const int i = g0 + ig;
        if (i >= WMF)
            break;
        const int rl = wm * (WMF * 16) + i * 16 + l15;
        const int4 ri = rowinfo(rl);
        riA[ig] = ri;
        const bool rowOk = ri.w >= 0;
        const i64 m = m0 + rl;
        meanA[ig] = 0.f, scA[ig] = 1.f;
#pragma unroll
        for (int j = 0; j < WNF; ++j)
            resA[ig][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI == EPI_LINEAR || EPI == EPI_SCALE_RES)
        {
            if ((EPI == EPI_LINEAR && p.res) || EPI == EPI_SCALE_RES)
            {
#pragma unroll
                for (int j = 0; j < WNF; ++j)
                {
                    const int n = colBase + j * 16;
                    if (rowOk && n < p.N)
                        resA[ig][j] = *reinterpret_cast<const float4 *>(p.res + m * p.ldy + n);
                }
            }
        }
        else if (EPI == EPI_GLU || EPI == EPI_GN_GLU_SCALE_RES)
        {
            if (EPI == EPI_GN_GLU_SCALE_RES && rowOk)
            {
                meanA[ig] = p.epiStats[ri.w * 4];
                scA[ig] = p.epiStats[ri.w * 4 + 1];
            }
#pragma unroll
            for (int j = 0; j < WNF; j += 2)
            {
                const int na = colBase + j * 16;
                const int c = (na >> 5) * 16 + (na & 15);
                if (rowOk && na + 16 < p.N)
                {
                    if (EPI == EPI_GN_GLU_SCALE_RES)
                        resA[ig][j] = *reinterpret_cast<const float4 *>(p.res + m * p.ldy + c);
                    else if (p.table)
                    {
                        const float4 tv = *reinterpret_cast<const float4 *>(p.table + (i64)ri.z * (p.N >> 1) + c);
                        resA[ig][j] = make_float4(p.tableScale * tv.x, p.tableScale * tv.y, p.tableScale * tv.z, p.tableScale * tv.w);
                    }
                }
            }
        }
        else if (EPI == EPI_TRCONV)
        {
#pragma unroll
            for (int j = 0; j < WNF; ++j)
            {
                const int n = colBase + j * 16;
                const int jj = p.trS * ri.z + trR[j] - p.trOff;
                const i64 off = (rowOk && n < p.N && jj >= 0 && jj < p.Lout) ? (i64)ri.x * p.yBS + ((i64)ri.y * p.Lout + jj) * p.ldy + trC[j] : -1;
                offsA[ig][j] = off;
                if (p.res && off >= 0)
                    resA[ig][j] = *reinterpret_cast<const float4 *>(p.res + off);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): nothing below loads from global memory

    // ---- pass 2: arithmetic and stores (no global loads: the stores go out back to back)
#pragma unroll
    for (int ig = 0; ig < G; ++ig)
    {
        const int i = g0 + ig;
        if (i >= WMF)
            break;
        const int rl = wm * (WMF * 16) + i * 16 + l15;
        const int4 ri = riA[ig];
        const bool rowOk = ri.w >= 0;
        const i64 m = m0 + rl;
        float s = 0.f, ss = 0.f;
        float sSeg[SSEG], ssSeg[SSEG]; // (SSEG > 1) finished runs
        float4(&resv)[WNF] = resA[ig];
        if (EPI == EPI_LINEAR || EPI == EPI_SCALE_RES || EPI == EPI_STATS_ONLY || EPI == EPI_STATS_FACT)
        {
#pragma unroll
            for (int j = 0; j < WNF; ++j)
            {
                const int n = colBase + j * 16;
                if (rowOk && n < p.N)
                {
                    float4 v = make_float4(acc[i][j][0] + biasv[j].x, acc[i][j][1] + biasv[j].y, acc[i][j][2] + biasv[j].z,
                                           acc[i][j][3] + biasv[j].w);
                    if (EPI == EPI_LINEAR)
                    {
                        if (p.act)
                            v = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
                        v.x += resv[j].x, v.y += resv[j].y, v.z += resv[j].z, v.w += resv[j].w;
                        *reinterpret_cast<float4 *>(p.Y + m * p.ldy + n) = v;
                    }
                    else if (EPI == EPI_SCALE_RES)
                    {
                        v = make_float4(resv[j].x + v.x * scalev[j].x, resv[j].y + v.y * scalev[j].y, resv[j].z + v.z * scalev[j].z,
                                        resv[j].w + v.w * scalev[j].w);
                        *reinterpret_cast<float4 *>(p.Y + m * p.ldy + n) = v;
                    }
                    if (EPI == EPI_STATS_FACT)
                    {
                        // factorised statistics (plan.h): columns < hid are L a (squares), column hid is the row
                        // sum of the full product, column hid+1 half of the remaining second-moment terms
                        const int hid = p.Cout;
                        const float vr[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                        {
                            const int nn = n + r;
                            ss += nn < hid ? vr[r] * vr[r] : (nn == hid + 1 ? 2.0f * vr[r] : 0.f);
                            s += nn == hid ? vr[r] : 0.f;
                        }
                    }
                    else
                    {
                        s += (v.x + v.y) + (v.z + v.w);
                        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                    }
                }
                if (SSEG > 1 && (j + 1) % (WNF / SSEG) == 0)
                {
                    sSeg[j / (WNF / SSEG)] = s, ssSeg[j / (WNF / SSEG)] = ss;
                    s = ss = 0.f;
                }
            }
            if (wantStats)
            {
#pragma unroll
                for (int q = 0; q < SSEG; ++q)
                {
                    if (SSEG > 1)
                        s = sSeg[q], ss = ssSeg[q];
                    s += __shfl_xor(s, 16);
                    ss += __shfl_xor(ss, 16);
                    s += __shfl_xor(s, 32);
                    ss += __shfl_xor(ss, 32);
                    if (kq == 0)
                    {
                        rsum[rl][wn * SSEG + q].x = s; // member-wise: a whole-struct store goes through a private-memory temporary
                        rsum[rl][wn * SSEG + q].y = ss;
                    }
                }
            }
        }
        else if (EPI == EPI_GLU || EPI == EPI_GN_GLU_SCALE_RES)
        {
            if constexpr (WNF % 2 == 0)
            {
                const float mean = meanA[ig], sc = scA[ig];
#pragma unroll
                for (int j = 0; j < WNF; j += 2)
                {
                    const int na = colBase + j * 16, nb = na + 16;
                    if (rowOk && nb < p.N)
                    {
                        const int c = (na >> 5) * 16 + (na & 15);
                        const float av[4] = {acc[i][j][0] + biasv[j].x, acc[i][j][1] + biasv[j].y, acc[i][j][2] + biasv[j].z,
                                             acc[i][j][3] + biasv[j].w};
                        const float gv[4] = {acc[i][j + 1][0] + biasv[j + 1].x, acc[i][j + 1][1] + biasv[j + 1].y,
                                             acc[i][j + 1][2] + biasv[j + 1].z, acc[i][j + 1][3] + biasv[j + 1].w};
                        const float rv[4] = {resv[j].x, resv[j].y, resv[j].z, resv[j].w};
                        float ov[4];
                        if (EPI == EPI_GN_GLU_SCALE_RES)
                        {
                            const float gw[4] = {gnWv[j].x, gnWv[j].y, gnWv[j].z, gnWv[j].w};
                            const float gb[4] = {gnBv[j].x, gnBv[j].y, gnBv[j].z, gnBv[j].w};
                            const float hw[4] = {gnWv[j + 1].x, gnWv[j + 1].y, gnWv[j + 1].z, gnWv[j + 1].w};
                            const float hb[4] = {gnBv[j + 1].x, gnBv[j + 1].y, gnBv[j + 1].z, gnBv[j + 1].w};
                            const float sv[4] = {scalev[j].x, scalev[j].y, scalev[j].z, scalev[j].w};
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                            {
                                const float a = (av[r] - mean) * sc * gw[r] + gb[r];
                                const float g = (gv[r] - mean) * sc * hw[r] + hb[r];
                                ov[r] = rv[r] + sv[r] * (a * sigmoid_f(g));
                            }
                        }
                        else
                        {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                ov[r] = av[r] * sigmoid_f(gv[r]) + rv[r];
                        }
                        *reinterpret_cast<float4 *>(p.Y + m * p.ldy + c) = make_float4(ov[0], ov[1], ov[2], ov[3]);
                    }
                }
            }
        }
        else // EPI_TRCONV
        {
            i64(&offs)[WNF] = offsA[ig];
#pragma unroll
            for (int j = 0; j < WNF; ++j)
                if (offs[j] >= 0)
                {
                    float4 v = make_float4(acc[i][j][0] + biasv[j].x, acc[i][j][1] + biasv[j].y, acc[i][j][2] + biasv[j].z,
                                           acc[i][j][3] + biasv[j].w);
                    if (p.act)
                        v = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
                    v.x += resv[j].x, v.y += resv[j].y, v.z += resv[j].z, v.w += resv[j].w;
                    *reinterpret_cast<float4 *>(p.Y + offs[j]) = v;
                }
        }
    }
    } // groups of row blocks
    if (EPI == EPI_LINEAR || EPI == EPI_SCALE_RES || EPI == EPI_STATS_ONLY || EPI == EPI_STATS_FACT)
        if (wantStats) // (EPI_KPL / EPI_VT ops carry no row statistics)
        {
            __syncthreads();
            for (int r = tid; r < BM; r += NT)
            {
                const i64 m = m0 + r;
                if (m < p.M)
                {
                    float s = 0.f, ss = 0.f;
#pragma unroll
                    for (int w = 0; w < WAVES_N * SSEG; ++w)
                    {
                        s += rsum[r][w].x;
                        ss += rsum[r][w].y;
                    }
                    float *dst = p.rowstat + (m * p.NB + tileN) * 2;
                    dst[0] = s;
                    dst[1] = ss;
                }
            }
        }
}

// "linear layer" addressing applies (StageWalk LIN): one contiguous run of K floats per row, K a multiple of the K-tile
inline bool gemm_is_linear(const GemmArgs &a, int pro, int epi, int ktile)
{
    return pro == PRO_NONE && (epi == EPI_LINEAR || epi == EPI_SCALE_RES || epi == EPI_GLU || epi == EPI_KPL || epi == EPI_VT) && a.S1 == 1 && a.pad0 == 0 &&
           a.seg0 == a.K && a.K == a.Kp && a.K % ktile == 0 && a.Np % 4 == 0 &&
           (i64)(a.P0 - 1) * a.stride0 * a.Cin + a.seg0 <= (i64)a.L0 * a.Cin && a.P1 == a.L1 && a.stride1 == 1 && a.pad1 == 0;
}

} // namespace dmx

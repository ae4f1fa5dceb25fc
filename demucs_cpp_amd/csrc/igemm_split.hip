// igemm_split.hip — EXPERIMENT (opt-in, env DMX_GEMM=bf16x3): the implicit-GEMM of igemm.hip with the fp32 products
// formed on the bf16 matrix pipe from EXACT operand splits, fp32 accumulation.
//
//   activation a (fp32)            = a1 + a2 + a3   three bf16 terms by truncation (8 + 8 + 8 significand bits: exact)
//   weight     w (fp16 in the file) = w1 + w2        two bf16 terms (11 significand bits <= 8 + 8: exact; checked per op
//                                                    on the host, ops whose weights are not fp16-exact keep the fp32 kernel)
//   a w = a1 w1 + a1 w2 + a2 w1 + a2 w2 + a3 w1  (+ a3 w2, dropped: <= 2^-24 |a w|, half an fp32 ulp of the product)
//
// Every bf16 x bf16 product is exact in fp32 and v_mfma_f32_16x16x32_bf16 accumulates in fp32, so the result differs
// from the fp32 kernel's k-ordered fmaf chain only by the order of the fp32 additions and by the dropped term: fp32
// arithmetic at 5 x 16 = 80 matrix-pipe cycles per 16x16x32 block instead of 8 x 32 = 256 (v_mfma_f32_16x16x4_f32).
// NOT the default path: bench.py's headline and every parity claim are on the fp32 MFMA kernels; this file exists to
// measure what the exact-split route is worth.
//
// Staging: A rows are fetched as fp32 (prologue transforms unchanged), split by the staging thread and written as three
// bf16 planes; B comes from two bf16 planes prepared at model upload (api.cpp). LDS image per plane: [row][4 octets of
// 8 bf16 = 16 B], octet slot XOR-swizzled by g(row & 15) = {0,2,3,1}[(row >> 2) & 3]: each of ds_read_b128's four lane
// groups ({0-3,12-15,20-27}, ...) then hits 16 distinct 16-byte slots. Everything else (row decomposition, conv
// addressing with one validity bit per tap, prologues, epilogues, XCD-aware tile map, row statistics) is the text of
// igemm.hip's kernel; the accumulator layout of the 16x16 MFMAs does not depend on the operand type.
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

namespace dmx
{

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float gelu_f(float v) { return dmx_gelu(v); }
__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float4 ld4z(const float *ptr, bool ok, const float *zero)
{
    const f32x4 v = *reinterpret_cast<const f32x4 *>(ok ? ptr : zero);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ int swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; } // {0,2,3,1}[(row >> 2) & 3]

template <int WAVES_M, int WAVES_N, int WMF, int WNF, int PRO, int EPI, bool LIN>
__global__ __launch_bounds__(256, 2) void igemm_split_kernel(const GemmArgs p)
{
    constexpr int KS = 2; // a K-tile is 32 k = one 16x16x32 MFMA deep
    constexpr int BM = WAVES_M * WMF * 16;
    constexpr int BN = WAVES_N * WNF * 16;
    constexpr int LPR = 8;                   // lanes per staged row
    constexpr int RP = 256 / LPR;            // rows staged per pass
    constexpr int RPB = 1 << 20;             // (no swizzle on the global side: this lane fetches k-quad `slane`)
    constexpr int AR = BM / RP;
    constexpr int BR = (BN + RP - 1) / RP;
    constexpr int BRP = BR * RP;
    static_assert(WAVES_M * WAVES_N == 4, "256 threads");
    static_assert(BM % RP == 0, "BM multiple of the staging pass");

    __shared__ u32x4 Ap0[3][BM][4], Ap1[3][BM][4];   // [plane][row][octet slot]
    __shared__ u32x4 Bp0[2][BRP][4], Bp1[2][BRP][4];
    // (row info is recomputed where it is needed and the row-statistics scratch aliases the A image after the K loop:
    // the two staging images are exactly 80 KB for the 128x128 tile, two workgroups = the CU's 160 KB)
    float2(*rsum)[WAVES_N] = reinterpret_cast<float2(*)[WAVES_N]>(&Ap0[0][0][0]);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // workgroup -> tile. Workgroup b is dispatched to XCD b % 8 (observed; used for speed only). With the
    // XCD-aware map all column tiles of a row tile run on the same XCD right after one another, so the
    // A row block is fetched from HBM / Infinity Cache once and re-read from that XCD's 4 MB L2, and the
    // 64 workgroups resident on an XCD form a (few row tiles) x (all column tiles) patch that shares both
    // operands' k-slices. Row tiles are dealt round-robin to the XCDs (balanced to one tile).
    unsigned tileM, tileN;
    if (p.xcdMap)
    {
        const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const unsigned mi = j / p.tilesN;
        tileN = j - mi * p.tilesN;
        tileM = mi * 8u + xcd;
        if (tileM >= p.tilesM)
            return; // whole workgroup, before any barrier
    }
    else
    {
        tileN = blockIdx.x / p.tilesM;
        tileM = blockIdx.x - tileN * p.tilesM;
    }
    const i64 m0 = (i64)tileM * BM;
    const int n0 = (int)tileN * BN;

    // row r of the tile -> (b, p1, p0, group), -1 in .w beyond M: magic-number divisions (kernels.h FastDiv)
    auto rowinfo_of = [&](int r) -> int4 {
        const i64 m = m0 + r;
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
    };

    // ---- per-thread staging state: AR rows of A and BR rows of B, all at k-quad `slane`
    const int slane = tid % LPR, srow = tid / LPR;
    const int slaneK = slane ^ ((srow / RPB) % LPR); // k-quad this lane fetches (LDS slot `slane` of its rows)
    const i64 rowLen = (i64)p.L0 * p.Cin;
    const int rowLenI = (int)rowLen;
    const float *aRow[AR]; // X + b*xBS + in1_0*rowLen + e0  (tap s1 = 0, k = 0)
    bool aRowOk[AR];
    // General (conv) addressing: validity of a staged chunk depends only on (row, tap), tap c = k / Cin =
    // s1 * (seg0 / Cin) + (tap along axis 0) - padding starts and ends at whole taps - so each row carries ONE
    // bit per tap, computed here once per tile; the K walk then tests a bit instead of re-deriving four range
    // checks per row and K-tile (measured: the address arithmetic of the general path cost the 3x3 rewrites 10 %).
    unsigned aTapMask[AR];
    float aMean[AR], aScale[AR];
    const int taps0 = p.seg0 / p.Cin;
#pragma unroll
    for (int i = 0; i < AR; ++i)
    {
        const int4 ri = rowinfo_of(srow + i * RP);
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
    const float *bRow[BR];
    bool bRowOk[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i)
    {
        const int rl = srow + i * RP;
        const int n = n0 + rl;
        bRowOk[i] = rl < BN && n < p.Np;
        bRow[i] = p.Wt + (i64)(bRowOk[i] ? n : 0) * p.Kp;
    }


    // two staging register sets: a tile is requested a whole iteration before it is written to LDS
    f32x4 aRegS[2][AR], gWS[2], gBS[2];
    u32x4 bRegS[2][BR];
    unsigned maskHeldS[2] = {0u, 0u};
    const int nk = (p.Kp + 31) >> 5;
    // Sequential K walk, one tile = 16*KS consecutive k; this lane stages k = kl .. kl+3.
    // (s1, offb) = conv tap along axis 1 / offset inside its contiguous run, advanced per
    // lane without division. Addresses of the NEXT tile are computed one iteration ahead
    // (after the MFMA block), so the loop body starts with nothing but the global loads:
    //   loads(t+1) ; MFMA(t) ; transform+ds_write(t+1) ; addresses(t+2) ; barrier
    // Out-of-range chunks point at the zero page: PRO_NONE needs no masking at all.
    int kl = slaneK * 4, s1 = 0, offb = slaneK * 4;
    if (p.S1 > 1)
        while (offb >= p.seg0)
        {
            offb -= p.seg0;
            ++s1;
        }
    int tapC = 0, tapOff = slaneK * 4; // tap index kl / Cin and offset inside the tap
    if (!LIN)
        while (tapOff >= p.Cin)
        {
            tapOff -= p.Cin;
            ++tapC;
        }
    const float *addrA[AR], *addrB[BR], *addrG = p.zero;
    unsigned maskNext = 0;
    i64 stepA[AR], stepB[BR]; // LIN: per-row advance (0 for rows that stay on the zero page)
    bool linInit = false;
    // addresses of the next tile to fetch, in two halves (the interleaved loop slots them between MFMA groups):
    // addrs_A = validity + A row addresses, addrs_B = B row addresses + advance of the K walk
    int segOffCur = 0;
    unsigned tapBit = 0;
    // general addressing of the A rows; half = 0 / 1: first / second half of the rows (the tile-wide
    // quantities are set up with the first half), 2: all rows
    auto addrs_A_general = [&](int half) {
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
    };
    auto addrs_A = [&]() {
        if (LIN)
        {
            if (!linInit)
            {
                maskNext = 0;
#pragma unroll
                for (int i = 0; i < AR; ++i)
                {
                    addrA[i] = aRowOk[i] ? aRow[i] + slaneK * 4 : p.zero;
                    stepA[i] = aRowOk[i] ? 16 * KS : 0;
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
                addrG += 16 * KS;
            return;
        }
        addrs_A_general(2);
    };
    auto addrs_B = [&]() {
        if (LIN)
        {
            if (!linInit)
            {
                linInit = true;
#pragma unroll
                for (int i = 0; i < BR; ++i)
                {
                    addrB[i] = bRowOk[i] ? bRow[i] + slaneK * 4 : p.zero;
                    stepB[i] = bRowOk[i] ? 16 * KS : 0;
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
                stepB[i] = bRowOk[i] ? 16 * KS : 0;
            }
        }
        else
        {
#pragma unroll
            for (int i = 0; i < BR; ++i)
                addrB[i] += stepB[i];
        }
        kl += 16 * KS;
        offb += 16 * KS;
        if (p.S1 > 1 && offb >= p.seg0)
        {
            offb -= p.seg0;
            ++s1;
        }
        tapOff += 16 * KS;
        while (tapOff >= p.Cin)
        {
            tapOff -= p.Cin;
            ++tapC;
        }
    };
    auto compute_addrs = [&]() {
        addrs_A();
        addrs_B();
    };
    // B planes: lanes 0-3 of a row fetch the octets of plane 1, lanes 4-7 those of plane 2
    const int bPlane = slane >> 2, bOct = slane & 3;
    const unsigned short *bPtr[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i)
        bPtr[i] = bRowOk[i] ? (bPlane ? p.Wb2 : p.Wb1) + (bRow[i] - p.Wt) + bOct * 8 : reinterpret_cast<const unsigned short *>(p.zero);
    auto issue_loads = [&](auto setTag) {
        constexpr int SET = decltype(setTag)::value;
#pragma unroll
        for (int i = 0; i < AR; ++i)
            aRegS[SET][i] = *reinterpret_cast<const f32x4 *>(addrA[i]);
#pragma unroll
        for (int i = 0; i < BR; ++i)
        {
            bRegS[SET][i] = *reinterpret_cast<const u32x4 *>(bPtr[i]);
            bPtr[i] += bRowOk[i] ? 32 : 0;
        }
        if (PRO == PRO_GN_GELU)
        {
            gWS[SET] = *reinterpret_cast<const f32x4 *>(addrG);
            gBS[SET] = *reinterpret_cast<const f32x4 *>(addrG + (p.proB - p.proW));
        }
        maskHeldS[SET] = maskNext;
    };
    auto store_tiles = [&](auto setTag, int buf, int a0, int a1e, int b0, int b1e) {
        constexpr int SET = decltype(setTag)::value;
        const f32x4 gW = gWS[SET], gB = gBS[SET];
        u32x4(*Ap)[BM][4] = buf ? Ap1 : Ap0;
        u32x4(*Bp)[BRP][4] = buf ? Bp1 : Bp0;
#pragma unroll
        for (int i = 0; i < AR; ++i)
        {
            if (i < a0 || i >= a1e)
                continue;
            f32x4 v = aRegS[SET][i];
            if (PRO != PRO_NONE)
            {
                const bool ok = (maskHeldS[SET] >> i) & 1u;
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
            // exact three-way split by truncation: h1 = top 16 bits, r = v - h1 (exact), h2 = top 16 bits of r, ...
            unsigned h1[4], h2[4], h3[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                const float x = v[c];
                h1[c] = __float_as_uint(x) & 0xffff0000u;
                const float r = x - __uint_as_float(h1[c]);
                h2[c] = __float_as_uint(r) & 0xffff0000u;
                const float r2 = r - __uint_as_float(h2[c]);
                h3[c] = __float_as_uint(r2) & 0xffff0000u;
            }
            const int row = srow + i * RP;
            const int slot = (slane >> 1) ^ swz(row);
            // element k = 4 slane + c sits at bits [16 (k & 7), +16) of the octet's 128 bits
            u32x2 *d1 = reinterpret_cast<u32x2 *>(&Ap[0][row][slot]) + (slane & 1);
            u32x2 *d2 = reinterpret_cast<u32x2 *>(&Ap[1][row][slot]) + (slane & 1);
            u32x2 *d3 = reinterpret_cast<u32x2 *>(&Ap[2][row][slot]) + (slane & 1);
            *d1 = u32x2{(h1[0] >> 16) | h1[1], (h1[2] >> 16) | h1[3]};
            *d2 = u32x2{(h2[0] >> 16) | h2[1], (h2[2] >> 16) | h2[3]};
            *d3 = u32x2{(h3[0] >> 16) | h3[1], (h3[2] >> 16) | h3[3]};
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
        {
            if (i < b0 || i >= b1e)
                continue;
            const int row = srow + i * RP;
            Bp[bPlane][row][bOct ^ swz(row)] = bRegS[SET][i];
        }
    };

    f32x4 acc[WMF][WNF];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
        for (int j = 0; j < WNF; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- pipeline (one barrier per K-tile, two staging register sets): iteration kt multiplies tile kt row fragment by row
    // fragment; at its start it requests tile kt+2 into the register set whose contents went to LDS one iteration ago, and
    // in its first row steps it writes tile kt+1 (requested a whole iteration earlier) into the other LDS image. A fragments
    // are read one row step ahead.
    const std::integral_constant<int, 0> set0{};
    const std::integral_constant<int, 1> set1{};
    compute_addrs();
    issue_loads(set0);
    compute_addrs();
    issue_loads(set1);
    compute_addrs(); // tile 2
    store_tiles(set0, 0, 0, AR, 0, BR);
    __syncthreads();
    const int l15 = lane & 15, kq = lane >> 4;
    const int fslot = kq ^ swz(l15);
    constexpr int SH = (WMF + 1) / 2; // row steps that store
    auto iteration = [&](auto parTag) {
        constexpr int PAR = decltype(parTag)::value; // tile kt lives in LDS image PAR and came from register set PAR
        u32x4(*Ap)[BM][4] = PAR ? Ap1 : Ap0;
        u32x4(*Bp)[BRP][4] = PAR ? Bp1 : Bp0;
        issue_loads(parTag); // tile kt+2 (zero page beyond the end: no branch)
        bf16x8 b1[WNF], b2[WNF], a1, a2, a3, n1, n2, n3;
#pragma unroll
        for (int j = 0; j < WNF; ++j)
        {
            const int r = wn * (WNF * 16) + j * 16 + l15;
            b1[j] = __builtin_bit_cast(bf16x8, Bp[0][r][fslot]);
            b2[j] = __builtin_bit_cast(bf16x8, Bp[1][r][fslot]);
        }
        {
            const int r = wm * (WMF * 16) + l15;
            a1 = __builtin_bit_cast(bf16x8, Ap[0][r][fslot]);
            a2 = __builtin_bit_cast(bf16x8, Ap[1][r][fslot]);
            a3 = __builtin_bit_cast(bf16x8, Ap[2][r][fslot]);
        }
#pragma unroll
        for (int i = 0; i < WMF; ++i)
        {
            if (i + 1 < WMF)
            {
                const int r = wm * (WMF * 16) + (i + 1) * 16 + l15;
                n1 = __builtin_bit_cast(bf16x8, Ap[0][r][fslot]);
                n2 = __builtin_bit_cast(bf16x8, Ap[1][r][fslot]);
                n3 = __builtin_bit_cast(bf16x8, Ap[2][r][fslot]);
            }
            // smallest terms first; operands swapped (weights as A, activations as B): the accumulator holds C^T
#pragma unroll
            for (int j = 0; j < WNF; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a3, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < WNF; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2[j], a2, acc[i][j], 0, 0, 0);
            if (i < SH) // tile kt+1: the other register set -> the other image
                store_tiles(std::integral_constant<int, PAR ^ 1>{}, PAR ^ 1, i * AR / SH, (i + 1) * AR / SH, i * BR / SH, (i + 1) * BR / SH);
#pragma unroll
            for (int j = 0; j < WNF; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2[j], a1, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < WNF; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a2, acc[i][j], 0, 0, 0);
            if (i == WMF - 1)
                compute_addrs(); // addresses of tile kt+3
#pragma unroll
            for (int j = 0; j < WNF; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a1, acc[i][j], 0, 0, 0);
            a1 = n1, a2 = n2, a3 = n3;
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2)
    {
        iteration(set0);
        if (kt + 1 < nk)
            iteration(set1);
    }
    __syncthreads(); // (rsum aliases the A image)

    // ------------------------------------------------------------------ epilogue
    // The MFMAs were issued with the operands swapped (weights as A, activations as B), so each
    // accumulator holds C^T: lane (l15, kq) owns row m = tile row 16 i + l15 and the 4 CONSECUTIVE
    // channels n = 16 j + 4 kq + {0..3} -> one float4 global access per fragment, one row-info
    // lookup per row fragment, 2-step cross-lane reduction for the row statistics.
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

#pragma unroll
    for (int i = 0; i < WMF; ++i)
    {
        const int rl = wm * (WMF * 16) + i * 16 + l15;
        const int4 ri = rowinfo_of(rl);
        const bool rowOk = ri.w >= 0;
        const i64 m = m0 + rl;
        float s = 0.f, ss = 0.f;
        // residual operands of the whole row are loaded FIRST (independent loads in flight), then
        // combined and stored: res may alias Y element-wise (in-place updates), every element is read
        // before the same lane overwrites it.
        float4 resv[WNF];
#pragma unroll
        for (int j = 0; j < WNF; ++j)
            resv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI == EPI_LINEAR || EPI == EPI_SCALE_RES || EPI == EPI_STATS_ONLY || EPI == EPI_STATS_FACT)
        {
            if ((EPI == EPI_LINEAR && p.res) || EPI == EPI_SCALE_RES)
            {
#pragma unroll
                for (int j = 0; j < WNF; ++j)
                {
                    const int n = colBase + j * 16;
                    if (rowOk && n < p.N)
                        resv[j] = *reinterpret_cast<const float4 *>(p.res + m * p.ldy + n);
                }
            }
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
            }
            if (wantStats)
            {
                s += __shfl_xor(s, 16);
                ss += __shfl_xor(ss, 16);
                s += __shfl_xor(s, 32);
                ss += __shfl_xor(ss, 32);
                if (kq == 0)
                {
                    rsum[rl][wn].x = s; // member-wise: a whole-struct store goes through a private-memory temporary
                    rsum[rl][wn].y = ss;
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
                    const int na = colBase + j * 16;
                    const int c = (na >> 5) * 16 + (na & 15);
                    if (rowOk && na + 16 < p.N)
                    {
                        if (EPI == EPI_GN_GLU_SCALE_RES)
                            resv[j] = *reinterpret_cast<const float4 *>(p.res + m * p.ldy + c);
                        else if (p.table)
                        {
                            const float4 tv = *reinterpret_cast<const float4 *>(p.table + (i64)ri.z * (p.N >> 1) + c);
                            resv[j] = make_float4(p.tableScale * tv.x, p.tableScale * tv.y, p.tableScale * tv.z, p.tableScale * tv.w);
                        }
                    }
                }
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
            i64 offs[WNF];
#pragma unroll
            for (int j = 0; j < WNF; ++j)
            {
                const int n = colBase + j * 16;
                const int jj = p.trS * ri.z + trR[j] - p.trOff;
                offs[j] = (rowOk && n < p.N && jj >= 0 && jj < p.Lout) ? (i64)ri.x * p.yBS + ((i64)ri.y * p.Lout + jj) * p.ldy + trC[j] : -1;
            }
            if (p.res)
            {
#pragma unroll
                for (int j = 0; j < WNF; ++j)
                    if (offs[j] >= 0)
                        resv[j] = *reinterpret_cast<const float4 *>(p.res + offs[j]);
            }
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
    if (EPI == EPI_LINEAR || EPI == EPI_SCALE_RES || EPI == EPI_STATS_ONLY || EPI == EPI_STATS_FACT)
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
                    float *dst = p.rowstat + (m * p.NB + tileN) * 2;
                    dst[0] = s;
                    dst[1] = ss;
                }
            }
        }
}


template <int WM_, int WN_, int MF, int NF, int PRO, int EPI>
static void launch_split_one(const GemmArgs &a0, hipStream_t s)
{
    constexpr int BM = WM_ * MF * 16, BN = WN_ * NF * 16;
    GemmArgs a = a0;
    a.tilesM = (unsigned)((a.M + BM - 1) / BM);
    a.tilesN = (unsigned)((a.N + BN - 1) / BN);
    a.xcdMap = 1;
    a.dP0 = make_fastdiv((unsigned)a.P0), a.dP1 = make_fastdiv((unsigned)a.P1);
    const unsigned blocks = 8u * ((a.tilesM + 7u) / 8u) * a.tilesN;
    const bool lin = PRO == PRO_NONE && (EPI == EPI_LINEAR || EPI == EPI_SCALE_RES || EPI == EPI_GLU) && a.S1 == 1 && a.pad0 == 0 &&
                     a.seg0 == a.K && a.K == a.Kp && a.K % 32 == 0 && a.Np % 4 == 0 &&
                     (i64)(a.P0 - 1) * a.stride0 * a.Cin + a.seg0 <= (i64)a.L0 * a.Cin && a.P1 == a.L1 && a.stride1 == 1 && a.pad1 == 0;
    if constexpr (PRO == PRO_NONE && (EPI == EPI_LINEAR || EPI == EPI_SCALE_RES || EPI == EPI_GLU))
    {
        if (lin)
        {
            hipLaunchKernelGGL((igemm_split_kernel<WM_, WN_, MF, NF, PRO, EPI, true>), dim3(blocks), dim3(256), 0, s, a);
            return;
        }
    }
    hipLaunchKernelGGL((igemm_split_kernel<WM_, WN_, MF, NF, PRO, EPI, false>), dim3(blocks), dim3(256), 0, s, a);
}

// The MFMA-bound tile families only (plan.h kTileCfgs): 0 / 7 / 15 (2x2 waves, 4 column fragments), 9 / 16 (2 column
// fragments), 2 / 10 (4x1 waves, 6 column fragments). Returns -1 for anything else: the op keeps its fp32 kernel.
int launch_igemm_split(int cfg, const GemmArgs &a, hipStream_t s, bool dry)
{
    if (a.M >= (1ll << 31) - 256 || !a.Wb1 || !a.Wb2)
        return -1;
#define DMX_CASE(cfgid, WM_, WN_, MF, NF, PRO, EPI)          \
    case (cfgid * 100 + PRO * 10 + EPI):                    \
        if (!dry)                                           \
            launch_split_one<WM_, WN_, MF, NF, PRO, EPI>(a, s); \
        return 0;
    switch (cfg * 100 + a.pro * 10 + a.epi)
    {
        DMX_CASE(0, 2, 2, 4, 4, PRO_NONE, EPI_LINEAR)
        DMX_CASE(0, 2, 2, 4, 4, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(0, 2, 2, 4, 4, PRO_NONE, EPI_GLU)
        DMX_CASE(0, 2, 2, 4, 4, PRO_NONE, EPI_TRCONV)
        DMX_CASE(7, 2, 2, 2, 4, PRO_NONE, EPI_LINEAR)
        DMX_CASE(7, 2, 2, 2, 4, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(7, 2, 2, 2, 4, PRO_NONE, EPI_GLU)
        DMX_CASE(7, 2, 2, 2, 4, PRO_NONE, EPI_TRCONV)
        DMX_CASE(2, 4, 1, 2, 6, PRO_NONE, EPI_LINEAR)
        DMX_CASE(2, 4, 1, 2, 6, PRO_NONE, EPI_GLU)
        DMX_CASE(2, 4, 1, 2, 6, PRO_NONE, EPI_TRCONV)
        // half / quarter-height siblings (few segments in flight): same column decomposition, same bits as their parents
        DMX_CASE(15, 2, 2, 1, 4, PRO_NONE, EPI_LINEAR)
        DMX_CASE(15, 2, 2, 1, 4, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(15, 2, 2, 1, 4, PRO_NONE, EPI_GLU)
        DMX_CASE(15, 2, 2, 1, 4, PRO_NONE, EPI_TRCONV)
        DMX_CASE(9, 2, 2, 2, 2, PRO_NONE, EPI_LINEAR)
        DMX_CASE(9, 2, 2, 2, 2, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(9, 2, 2, 2, 2, PRO_NONE, EPI_GLU)
        DMX_CASE(9, 2, 2, 2, 2, PRO_NONE, EPI_TRCONV)
        DMX_CASE(16, 2, 2, 1, 2, PRO_NONE, EPI_LINEAR)
        DMX_CASE(16, 2, 2, 1, 2, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(16, 2, 2, 1, 2, PRO_NONE, EPI_GLU)
        DMX_CASE(16, 2, 2, 1, 2, PRO_NONE, EPI_TRCONV)
        DMX_CASE(10, 4, 1, 1, 6, PRO_NONE, EPI_LINEAR)
        DMX_CASE(10, 4, 1, 1, 6, PRO_NONE, EPI_GLU)
        DMX_CASE(10, 4, 1, 1, 6, PRO_NONE, EPI_TRCONV)
    default:
        return -1;
    }
#undef DMX_CASE
}

} // namespace dmx

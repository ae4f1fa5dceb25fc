// igemm_split.hip — the implicit GEMM of igemm.hip with the fp32 products formed on the bf16 matrix pipe from EXACT
// operand splits, fp32 accumulation (contexts created with DMX_GEMM_BF16X3, include/demucs_hip.h).
//
//   activation a (fp32)            = a1 + a2 + a3   three bf16 terms, round-to-nearest splits (igemm_common.h split3_pk):
//                                                    exact for 2^-109 <= |a| < 3.39e38 (igemm_common.h), |a2| <= 2^-8 |a|, |a3| <= 2^-16 |a|
//   weight     w (fp16 in the file) = w1 + w2        two bf16 terms (11 significand bits <= 8 + 8 + sign: exact, |w2| <= 2^-8 |w|;
//                                                    checked per op on the host, ops whose weights are not exact keep the
//                                                    fp32 kernel)
//   a w = a1 w1 + a1 w2 + a2 w1 + a2 w2 + a3 w1  (+ a3 w2, dropped: <= 2^-24 |a w|, half an ulp of the fp32 product)
//
// Every bf16 x bf16 product is exact in fp32 and v_mfma_f32_16x16x32_bf16 accumulates in fp32, so the result differs
// from the fp32 kernel's k-ordered fmaf chain only by the order of the fp32 additions and by the dropped term: fp32
// arithmetic at 5 x 16 = 80 matrix-pipe cycles per 16x16x32 block instead of 8 x 32 = 256 (v_mfma_f32_16x16x4_f32).
// Term order per 32-deep k-step and accumulator, smallest first: a3 w1, a2 w2, a1 w2, a2 w1, a1 w1 - the same in every
// tile shape, so results do not depend on batching or sharding (tested bitwise, like the fp32 family).
//
// Staging: A rows are fetched as fp32 (prologue transforms unchanged), split by the staging thread and written as three
// bf16 planes; B comes from two bf16 planes prepared at model upload (api.cpp). LDS image per plane: [row][4 octets of
// 8 bf16 = 16 B], octet slot XOR-swizzled by g(row & 15) = {0,2,3,1}[(row >> 2) & 3]: each of ds_read_b128's four lane
// groups ({0-3,12-15,20-27}, ...) then hits 16 distinct 16-byte slots. Row decomposition, conv addressing, prologues,
// epilogues, tile map and row statistics are igemm_common.h - one text with igemm.hip.
#include "igemm_common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace dmx
{

__device__ __forceinline__ int swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; } // {0,2,3,1}[(row >> 2) & 3]

template <int WAVES_M, int WAVES_N, int WMF, int WNF, int PRO, int EPI, bool LIN>
__global__ __launch_bounds__(256, 2) void igemm_split_kernel(const GemmArgs p)
{
    constexpr int KT = 32; // a K-tile is 32 k = one 16x16x32 MFMA deep
    constexpr int BM = WAVES_M * WMF * 16;
    constexpr int BN = WAVES_N * WNF * 16;
    constexpr int LPR = 8;                   // lanes per staged row (one float4 of A / one 16-byte octet of a B plane each)
    constexpr int RP = 256 / LPR;            // rows staged per pass
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
    unsigned tileM, tileN;
    if (!tile_of_block(p, tileM, tileN))
        return; // whole workgroup, before any barrier
    const i64 m0 = (i64)tileM * BM;
    const int n0 = (int)tileN * BN;
    auto rowinfo_of = [&](int r) -> int4 { return row_info(p, m0 + r); };

    // ---- per-thread staging state: AR rows of A and BR rows of B, all at k-quad `slane` (no swizzle on the global side)
    const int slane = tid % LPR, srow = tid / LPR;
    // B chunks (16 bytes = one octet of one plane of one weight row) of a thread. An 8-lane ds_write_b128 group must stay
    // inside ONE plane: rows are 64 B, so the two planes of a row are 8 KB apart = the same banks (the first form of this
    // kernel let lanes 0-3 write plane 1 and lanes 4-7 plane 2 of one row: every B store two-way conflicted,
    // SQ_LDS_BANK_CONFLICT = 17 % of the LDS cycles). BN a multiple of 64: chunk i = plane i & 1 of row
    // 64 (i >> 1) + 2 (tid >> 3) + ((tid >> 2) & 1) - a group writes two whole rows of one plane, 128 contiguous bytes.
    // Other widths (96) keep the row-per-8-lanes form.
    // Widths that are a multiple of 32 but not of 64 (the 96-wide family; round 5): the same idea with the row pairs of both
    // planes numbered through - 8-lane group g = tid / 8 + 32 i writes row pair g % (BN / 2) of plane g / (BN / 2) (round 4 kept
    // the row-per-8-lanes form there: SQ_LDS_BANK_CONFLICT 14.8 % of the 128x96 class's LDS cycles). 48-wide tiles keep it.
    constexpr bool BCF = BN % 64 == 0;
    constexpr bool BCP = !BCF && BN % 32 == 0;
    const int bOct = (BCF || BCP) ? (tid & 3) : (slane & 3);
    auto bGrp = [&](int i) { return (tid >> 3) + 32 * i; };
    auto bRowOf = [&](int i) {
        return BCF ? 64 * (i >> 1) + 2 * (tid >> 3) + ((tid >> 2) & 1) : BCP ? 2 * (bGrp(i) % (BN / 2)) + ((tid >> 2) & 1) : srow + i * RP;
    };
    auto bPlaneOf = [&](int i) { return BCF ? (i & 1) : BCP ? bGrp(i) / (BN / 2) : (slane >> 2); };
    StageWalk<AR, BR, KT, PRO, LIN> w(p, slane);
    w.init(rowinfo_of, [&](int i) { return srow + i * RP; }, bRowOf, n0, BN);

    // two staging register sets: a tile is requested a whole iteration before it is written to LDS
    f32x4 aRegS[2][AR], gWS[2], gBS[2];
    u32x4 bRegS[2][BR];
    unsigned maskHeldS[2] = {0u, 0u};
    const int nk = (p.Kp + 31) >> 5;
    const unsigned short *bPtr[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i)
        bPtr[i] = w.bRowOk[i] ? (bPlaneOf(i) ? p.Wb2 : p.Wb1) + (w.bRow[i] - p.Wt) + bOct * 8 : reinterpret_cast<const unsigned short *>(p.zero);
    // LIN: a row is one contiguous run of K floats, so a staging address is (uniform base) + (32-bit byte offset that
    // advances by one K-tile): the loads take the scalar-base form (global_load ... v_off, s[base]) - one address register
    // per lane instead of a 64-bit pointer pair, one 32-bit add per row and tile. Rows beyond M / columns beyond Np read
    // the last valid row instead of the zero page: their accumulators are never stored (the launcher takes this path
    // only when every offset fits 32 bits).
    unsigned aOff[AR], bOff[BR];
    if constexpr (LIN)
    {
        const i64 rowLen = (i64)p.L0 * p.Cin;
#pragma unroll
        for (int i = 0; i < AR; ++i)
        {
            const i64 m = min(m0 + srow + i * RP, p.M - 1);
            const int4 ri = row_info(p, m);
            aOff[i] = (unsigned)(((i64)ri.x * p.xBS + (i64)ri.y * rowLen + (i64)ri.z * p.stride0 * p.Cin) * 4 + slane * 16);
        }
        const unsigned planeDelta = (unsigned)(p.Wb2 - p.Wb1);
#pragma unroll
        for (int i = 0; i < BR; ++i)
        {
            const int n = min(n0 + bRowOf(i), p.Np - 1);
            bOff[i] = ((unsigned)n * (unsigned)p.Kp + (unsigned)bOct * 8u + (bPlaneOf(i) ? planeDelta : 0u)) * 2u;
        }
    }
    auto issue_loads = [&](auto setTag) {
        constexpr int SET = decltype(setTag)::value;
        if constexpr (LIN)
        {
#pragma unroll
            for (int i = 0; i < AR; ++i)
            {
                aRegS[SET][i] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(p.X) + aOff[i]);
                aOff[i] += KT * 4;
            }
#pragma unroll
            for (int i = 0; i < BR; ++i)
            {
                bRegS[SET][i] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.Wb1) + bOff[i]);
                bOff[i] += KT * 2;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < AR; ++i)
            aRegS[SET][i] = *reinterpret_cast<const f32x4 *>(w.addrA[i]);
#pragma unroll
        for (int i = 0; i < BR; ++i)
        {
            bRegS[SET][i] = *reinterpret_cast<const u32x4 *>(bPtr[i]);
            bPtr[i] += w.bRowOk[i] ? KT : 0;
        }
        if (PRO == PRO_GN_GELU)
        {
            gWS[SET] = *reinterpret_cast<const f32x4 *>(w.addrG);
            gBS[SET] = *reinterpret_cast<const f32x4 *>(w.addrG + (p.proB - p.proW));
        }
        maskHeldS[SET] = w.maskNext;
    };
    auto store_tiles = [&](auto setTag, int buf, int a0, int a1e, int b0, int b1e) {
        constexpr int SET = decltype(setTag)::value;
        u32x4(*Ap)[BM][4] = buf ? Ap1 : Ap0;
        u32x4(*Bp)[BRP][4] = buf ? Bp1 : Bp0;
#pragma unroll
        for (int i = 0; i < AR; ++i)
        {
            if (i < a0 || i >= a1e)
                continue;
            const f32x4 v = w.transform(aRegS[SET][i], i, (maskHeldS[SET] >> i) & 1u, gWS[SET], gBS[SET]);
            // exact three-way split (igemm_common.h): element k = 4 slane + c sits at bits [16 (k & 7), +16) of its octet
            unsigned h1[2], h2[2], h3[2];
            split3_pk(v[0], v[1], h1[0], h2[0], h3[0]);
            split3_pk(v[2], v[3], h1[1], h2[1], h3[1]);
            const int row = srow + i * RP;
            const int slot = (slane >> 1) ^ swz(row);
            *(reinterpret_cast<u32x2 *>(&Ap[0][row][slot]) + (slane & 1)) = u32x2{h1[0], h1[1]};
            *(reinterpret_cast<u32x2 *>(&Ap[1][row][slot]) + (slane & 1)) = u32x2{h2[0], h2[1]};
            *(reinterpret_cast<u32x2 *>(&Ap[2][row][slot]) + (slane & 1)) = u32x2{h3[0], h3[1]};
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
        {
            if (i < b0 || i >= b1e)
                continue;
            const int row = bRowOf(i);
            Bp[bPlaneOf(i)][row][bOct ^ swz(row)] = bRegS[SET][i];
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
    if (!LIN)
        w.compute_addrs();
    issue_loads(set0);
    if (!LIN)
        w.compute_addrs();
    issue_loads(set1);
    if (!LIN)
        w.compute_addrs(); // tile 2
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
            if (!LIN && i == WMF - 1)
                w.compute_addrs(); // addresses of tile kt+3
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

    igemm_epilogue<WAVES_N, WMF, WNF, EPI, 256>(p, acc, rowinfo_of, rsum, m0, n0, tileN, wm, wn, BM);
}

// ---- linear layers with the activation fragments loaded straight into registers (no LDS image of A) -------------------
//
// A token-major operand ([row][K] floats, the transformer's layers) already has the shape of the matrix pipe's operand:
// lane (l15, kq) of a 16x16x32 block needs row l15, k = 8 kq .. 8 kq + 7 = 32 contiguous bytes. Four waves stacked in M
// (each WMF x 16 rows by the tile's whole width) fetch their own rows as two 16-byte loads per block and K-tile, split
// them in registers and never write them to LDS; only the two weight planes are staged (16 KB per K-tile, read by all
// four waves). Per 128x128 K-tile the workgroup moves 64 KB of fragment reads + 16 KB of plane stores through LDS
// instead of 80 + 40 KB (the 2 x 2-wave kernel above: 94 of the CU's 128 B/clk at the matrix pipe's full rate), no
// activation byte is fetched twice, and each element is still split exactly once.
// The MFMA operands are the same 8-element groups in the same order as in the kernel above, so the results are the
// same bits (the batch / shard invariance tests run small batches on the 2 x 2 kernels and large ones on this one).
// (Two experiments on this kernel were measured in round 5 and removed again - three staging register sets, persistent
// workgroups with cross-tile prefetch: profiles/r05_experiments_split_gemm.md.)
// ARITH 0: bf16 terms (a = a1 + a2 + a3, w = w1 + w2: five MFMAs per block, two weight planes), the arithmetic of every other
// split kernel. ARITH 1 (contexts of DMX_GEMM_FP16X3, opt-in): fp16 terms - the weights ARE fp16 numbers, so a weight is one
// exact term (p.Wb1 = the fp16 plane of the blob); an activation row is first multiplied by its power-of-two scale
// p.rowScale[m] (launch_rowscale: the row's largest magnitude lands in [2^14, 2^15), so nothing can overflow) and split into
// three fp16 terms (igemm_common.h split3h_pk); three v_mfma_f32_16x16x32_f16 per block, one weight plane through LDS, and the
// accumulators are multiplied by 2^-s of their row before the epilogue. Bounded, not exact (the contract: igemm_common.h).
template <int WMF, int WNF, int EPI, int ARITH = 0>
__global__ __launch_bounds__(256, 2) void igemm_split_lin_kernel(const GemmArgs p)
{
    constexpr int KT = 32;
    constexpr int BM = 4 * WMF * 16, BN = WNF * 16;
    constexpr int NBP = ARITH ? 1 : 2;  // weight planes
    constexpr int BR = NBP * BN / 64;   // 16-byte chunks of the weight planes per thread and K-tile
    static_assert(BN % 64 == 0, "two rows of one plane per 8-lane store group");
    __shared__ u32x4 Bp0[NBP][BN][4], Bp1[NBP][BN][4]; // [plane][column][octet slot]
    float2(*rsum)[2] = reinterpret_cast<float2(*)[2]>(&Bp0[0][0][0]); // after the K loop: two runs of four fragments per row

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    unsigned tileM, tileN;
    if (!tile_of_block(p, tileM, tileN))
        return;
    const i64 m0 = (i64)tileM * BM;
    const int n0 = (int)tileN * BN;
    auto rowinfo_of = [&](int r) -> int4 { return row_info(p, m0 + r); };

    // activation rows of this lane: block i = rows wave * WMF * 16 + 16 i + l15, k-octet kq (rows beyond M read the last
    // valid row; their accumulators are never stored)
    unsigned aOff[WMF], bOff[BR];
    float aScale[WMF], aInv[WMF]; // (ARITH 1) 2^s / 2^-s of this lane's row of block i
    {
        const i64 rowLen = (i64)p.L0 * p.Cin;
#pragma unroll
        for (int i = 0; i < WMF; ++i)
        {
            const i64 m = min(m0 + wave * (WMF * 16) + i * 16 + l15, p.M - 1);
            const int4 ri = row_info(p, m);
            aOff[i] = (unsigned)(((i64)ri.x * p.xBS + (i64)ri.y * rowLen + (i64)ri.z * p.stride0 * p.Cin) * 4 + kq * 32);
            aScale[i] = aInv[i] = 1.0f;
            if constexpr (ARITH == 1)
            {
                const float2 sc = *reinterpret_cast<const float2 *>(p.rowScale + 2 * m);
                aScale[i] = sc.x, aInv[i] = sc.y;
            }
        }
    }
    // weight chunks: chunk i = plane i % NBP of column 64 (i / NBP) + 2 (tid >> 3) + ((tid >> 2) & 1), octet tid & 3
    const int bOct = tid & 3;
    auto bRowOf = [&](int i) { return 64 * (i / NBP) + 2 * (tid >> 3) + ((tid >> 2) & 1); };
    {
        const unsigned planeDelta = (unsigned)(p.Wb2 - p.Wb1);
#pragma unroll
        for (int i = 0; i < BR; ++i)
        {
            const int n = min(n0 + bRowOf(i), p.Np - 1);
            bOff[i] = ((unsigned)n * (unsigned)p.Kp + (unsigned)bOct * 8u + ((i % NBP) ? planeDelta : 0u)) * 2u;
        }
    }

    f32x4 aRaw[2][WMF][2];
    u32x4 bReg[2][BR];
    u32x4 aPl[2][WMF][3]; // three 16-bit planes of 8 k each (bf16 or fp16 terms)
    auto issue_loads = [&](auto setTag) {
        constexpr int SET = decltype(setTag)::value;
#pragma unroll
        for (int i = 0; i < WMF; ++i)
        {
            const char *src = reinterpret_cast<const char *>(p.X) + aOff[i];
            aRaw[SET][i][0] = *reinterpret_cast<const f32x4 *>(src);
            aRaw[SET][i][1] = *reinterpret_cast<const f32x4 *>(src + 16);
            aOff[i] += KT * 4;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
        {
            bReg[SET][i] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.Wb1) + bOff[i]);
            bOff[i] += KT * 2;
        }
    };
    auto split_block = [&](auto setTag, auto dstTag, int i) {
        constexpr int SET = decltype(setTag)::value, DST = decltype(dstTag)::value;
        const f32x4 lo = aRaw[SET][i][0], hi = aRaw[SET][i][1];
        unsigned h1[4], h2[4], h3[4];
        if (ARITH)
        {
            const float sc = aScale[i]; // a power of two: exact
            split3h_pk(lo[0] * sc, lo[1] * sc, h1[0], h2[0], h3[0]);
            split3h_pk(lo[2] * sc, lo[3] * sc, h1[1], h2[1], h3[1]);
            split3h_pk(hi[0] * sc, hi[1] * sc, h1[2], h2[2], h3[2]);
            split3h_pk(hi[2] * sc, hi[3] * sc, h1[3], h2[3], h3[3]);
        }
        else
        {
            split3_pk(lo[0], lo[1], h1[0], h2[0], h3[0]);
            split3_pk(lo[2], lo[3], h1[1], h2[1], h3[1]);
            split3_pk(hi[0], hi[1], h1[2], h2[2], h3[2]);
            split3_pk(hi[2], hi[3], h1[3], h2[3], h3[3]);
        }
        u32x4 q1{h1[0], h1[1], h1[2], h1[3]}, q2{h2[0], h2[1], h2[2], h2[3]}, q3{h3[0], h3[1], h3[2], h3[3]};
        // (the planes are computed HERE, between the MFMA groups: without this the compiler sinks the split of the odd
        // tiles into the next iteration's head, in front of its first fragment reads)
        asm volatile("" : "+v"(q1), "+v"(q2), "+v"(q3));
        aPl[DST][i][0] = q1;
        aPl[DST][i][1] = q2;
        aPl[DST][i][2] = q3;
    };
    auto store_B = [&](auto setTag, int buf, int b0, int b1e) {
        constexpr int SET = decltype(setTag)::value;
        u32x4(*Bp)[BN][4] = buf ? Bp1 : Bp0;
#pragma unroll
        for (int i = 0; i < BR; ++i)
        {
            if (i < b0 || i >= b1e)
                continue;
            const int row = bRowOf(i);
            Bp[i % NBP][row][bOct ^ swz(row)] = bReg[SET][i];
        }
    };

    f32x4 acc[WMF][WNF];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
        for (int j = 0; j < WNF; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // pipeline: iteration kt multiplies tile kt (weight image PAR, activation planes aPl[PAR]); at its start it requests
    // tile kt + 2 into register set PAR, and between its MFMA groups it splits the activations of tile kt + 1 (set PAR ^ 1,
    // requested a whole iteration earlier) and writes that tile's weight chunks into the other image. One barrier per tile.
    const std::integral_constant<int, 0> set0{};
    const std::integral_constant<int, 1> set1{};
    const int nk = (p.Kp + 31) >> 5;
    issue_loads(set0);
    issue_loads(set1);
#pragma unroll
    for (int i = 0; i < WMF; ++i)
        split_block(set0, set0, i);
    store_B(set0, 0, 0, BR);
    __syncthreads();
    const int fslot = kq ^ swz(l15);
    constexpr int NH = WNF / 4; // column fragments are processed four at a time
    u32x4 b1[NH][4], b2[NH][4]; // weight fragments (b2: the second bf16 plane, ARITH 0 only)
    // PAR = kt & 1: the weight image, activation planes and staging register set of tile kt
    auto iteration = [&](auto parTag) {
        constexpr int PAR = decltype(parTag)::value;
        const std::integral_constant<int, PAR ^ 1> other{};
        const std::integral_constant<int, PAR ^ 1> nextSet{};
        u32x4(*Bp)[BN][4] = PAR ? Bp1 : Bp0;
        issue_loads(parTag); // tile kt + 2
        auto read_half = [&](int h) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                const int r = (h * 4 + j) * 16 + l15;
                b1[h][j] = Bp[0][r][fslot];
                if (NBP == 2)
                    b2[h][j] = Bp[NBP - 1][r][fslot];
            }
        };
        // one term of the product for the four column fragments of half h, all row blocks: 4 WMF independent accumulators
        auto term = [&](int h, u32x4(&b)[4], int plane) {
#pragma unroll
            for (int i = 0; i < WMF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    // EPI_VT: the transposed product (activations as the matrix pipe's A operand): the accumulator holds C, a lane
                    // owns 4 consecutive TOKENS of one channel - the V^T pieces of the attention kernel (igemm_common.h)
                    if constexpr (ARITH == 1 && EPI == EPI_VT)
                        acc[i][h * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, aPl[PAR][i][plane]), __builtin_bit_cast(f16x8, b[j]), acc[i][h * 4 + j], 0, 0, 0);
                    else if constexpr (ARITH == 1)
                        acc[i][h * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b[j]), __builtin_bit_cast(f16x8, aPl[PAR][i][plane]), acc[i][h * 4 + j], 0, 0, 0);
                    else if constexpr (EPI == EPI_VT)
                        acc[i][h * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aPl[PAR][i][plane]), __builtin_bit_cast(bf16x8, b[j]), acc[i][h * 4 + j], 0, 0, 0);
                    else
                        acc[i][h * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[j]), __builtin_bit_cast(bf16x8, aPl[PAR][i][plane]), acc[i][h * 4 + j], 0, 0, 0);
                }
        };
        read_half(0);
#pragma unroll
        for (int h = 0; h < NH; ++h)
        {
            // smallest terms first, as in igemm_split_kernel: a3 w1, a2 w2, a1 w2, a2 w1, a1 w1 (fp16 terms: h3 w, h2 w, h1 w)
            term(h, b1[h], 2);
            if (h + 1 < NH)
                read_half(h + 1); // the next half's fragments are in flight during this half's remaining MFMAs
            if (!ARITH)
                term(h, b2[h], 1);
            // tile kt + 1: activations of set PAR ^ 1 -> planes, weight chunks -> the other image (spread over the halves)
#pragma unroll
            for (int i = 0; i < WMF; ++i)
                if ((i * NH) / WMF == h)
                    split_block(nextSet, other, i);
            store_B(nextSet, PAR ^ 1, h * BR / NH, (h + 1) * BR / NH);
            if (!ARITH)
                term(h, b2[h], 0);
            term(h, b1[h], 1);
            term(h, b1[h], 0);
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
    __syncthreads(); // (rsum aliases the weight image)
    if constexpr (ARITH == 1)
    {
        // back to the scale of the activations: acc = 2^s (a . w) -> a . w, a power-of-two factor per ROW (exact unless the product
        // itself leaves fp32's range)
#pragma unroll
        for (int i = 0; i < WMF; ++i)
        {
            if constexpr (EPI == EPI_VT) // transposed accumulators: this lane holds tokens 16 i + 4 kq + {0..3} of its channel
            {
                float inv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    inv[r] = p.rowScale[2 * min(m0 + wave * (WMF * 16) + i * 16 + 4 * kq + r, p.M - 1) + 1];
#pragma unroll
                for (int j = 0; j < WNF; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[i][j][r] *= inv[r];
            }
            else
            {
#pragma unroll
                for (int j = 0; j < WNF; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[i][j][r] *= aInv[i];
            }
        }
    }
    igemm_epilogue<1, WMF, WNF, EPI, 256, 2>(p, acc, rowinfo_of, rsum, m0, n0, tileN, wave, 0, BM);
}

// ---- the linear-layer kernel on a 128 x 256 tile (round 6) -------------------------------------------------------------------
//
// igemm_split_lin_kernel spends, per 80 MFMAs of a wave, two 32-byte activation fetches and 88 VALU instructions of split work;
// an ablation that fetched and split only every other K-tile (tools/micro/lin_variants.py linhalf: half of both per MFMA,
// everything else as it is) took the 58 linear-layer ops of a 42-segment step from 24.7 to 21.6 ms. Twice the tile WIDTH is
// the form of that which keeps two workgroups per CU: four waves stacked in M as before (32 rows each), 16 column fragments
// per wave (128 accumulator registers), so a wave's activation planes meet twice as many weight columns. What pays for the
// accumulators: the weight planes go global -> LDS directly (global_load_lds_dwordx4, the XOR swizzle applied to the SOURCE
// octet: no staging registers, no ds_write), one K-tile ahead into the other image (2 x 32 KB per workgroup), and the weight
// fragments are read two column fragments at a time into two alternating register sets.
// Same operands in the same order per accumulator as every other exact-split tile: identical bits (the launcher takes this
// kernel per LAUNCH, by the row count - batching and sharding still cannot change a bit). N a multiple of 256; row statistics
// (linear2's LayerNorm partials) are written per 128 columns, as the 128-wide tiles write them.
//
// Generalised in the same round (WNF, GEN): WNF = 12 is the 128 x 192 tile for N = 192 / 384 layers; GEN = true gives the
// activation fragments CONV addressing - a row's K axis is S1 runs of seg0 contiguous floats with whole-tap padding
// (igemm_common.h StageWalk): with Cin a multiple of 8 a lane's 8 consecutive k never leave one tap, so the fragment is still
// two 16-byte loads, from (row base) + (run offset) where the row's tap bit is set and from the zero page where it is not. The
// strided convs (k8 s4), the 3x3 / k3 rewrites, the transposed convs (K = 2 Cin) and the 1x1 rewrites of the C = 96 / 192 /
// 384 levels were two (N = 192) to six (N = 768) 96- or 128-wide tiles that each fetched AND split the same activation rows;
// here a row is fetched and split once per 192 / 256 columns and never passes through LDS. K-tile order, chunk -> (run, tap)
// walk and term order are those of igemm_split_kernel: the same bits (per-launch choice, tools/gpu_lin_ab.py).
// The narrow forms (WNF = 6 / 4 / 2: 96 / 64 / 32 columns) have the staged tiles' workgroup count; what they drop is the LDS round
// trip of the activations, and their small register files fit three to four workgroups per CU (launch_wide_conv /
// launch_split_narrow; measured per form: profiles/r06_experiments/).
template <int WNF, int EPI, bool GEN>
__global__ __launch_bounds__(256, WNF == 2 ? 4 : WNF <= 6 ? 3 : 2) void igemm_split_linw_kernel(const GemmArgs p)
{
    constexpr int KT = 32, WMF = 2, BM = 128, BN = 16 * WNF, NG = WNF / 2;
    constexpr int PW = WNF / 2; // weight pieces (1 KB = 16 columns of one plane) per wave and K-tile
    static_assert(WNF == 16 || WNF == 12 || WNF == 6 || WNF == 4 || WNF == 2,
                  "128 x 256 / 128 x 192 / 128 x 96 (N = 96 layers: three workgroups per CU) / 128 x 64 (the last transposed conv of the "
                  "frequency branch, N = 64: three) / 128 x 32 (the C = 192 DConv K1, N = 24: four)");
    __shared__ u32x4 Bp[2][2][BN][4]; // [image][plane][column][octet slot]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    unsigned tileM, tileN;
    if (!tile_of_block(p, tileM, tileN))
        return;
    const i64 m0 = (i64)tileM * BM;
    const int n0 = (int)tileN * BN;
    auto rowinfo_of = [&](int r) -> int4 { return row_info(p, m0 + r); };

    // activation rows of this lane: block i = rows wave * 32 + 16 i + l15, k-octet kq (rows beyond M read the last valid row;
    // their accumulators are never stored). Linear addressing: one 32-bit byte offset per block that advances by a K-tile.
    // GEN: the row's base for (tap 0, k 0) - it may lie before the tensor where padding starts a row - and one validity bit per
    // tap (StageWalk's rule: padding starts and ends at whole taps); the lane walks its octet's (run s1, offset in the run, tap)
    // without divisions, the same for both blocks.
    unsigned aOff[WMF];
    const float *aRow[WMF];
    unsigned aMask[WMF];
    int gS1 = 0, gOffb = kq * 8, gTapC = 0, gTapOff = kq * 8;
    const int rowLenI = p.L0 * p.Cin;
    {
        const i64 rowLen = (i64)p.L0 * p.Cin;
#pragma unroll
        for (int i = 0; i < WMF; ++i)
        {
            const i64 m = min(m0 + wave * (WMF * 16) + i * 16 + l15, p.M - 1);
            const int4 ri = row_info(p, m);
            aOff[i] = 0, aRow[i] = p.X, aMask[i] = 0;
            if constexpr (GEN)
            {
                const int taps0 = p.seg0 / p.Cin;
                const int in1_0 = ri.y * p.stride1 - p.pad1;
                const int e0 = (ri.z * p.stride0 - p.pad0) * p.Cin;
                aRow[i] = p.X + (i64)ri.x * p.xBS + (i64)in1_0 * rowLen + e0;
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
                        aMask[i] |= m0bits << (s * taps0);
                }
            }
            else
                aOff[i] = (unsigned)(((i64)ri.x * p.xBS + (i64)ri.y * rowLen + (i64)ri.z * p.stride0 * p.Cin) * 4 + kq * 32);
        }
        if constexpr (GEN)
        {
            if (p.S1 > 1)
                while (gOffb >= p.seg0)
                {
                    gOffb -= p.seg0;
                    ++gS1;
                }
            while (gTapOff >= p.Cin)
            {
                gTapOff -= p.Cin;
                ++gTapC;
            }
        }
    }
    // weight pieces: one wave instruction moves 1 KB = 16 columns x 4 octet slots of one plane; lane l lands at (column l / 4,
    // slot l % 4) and therefore FETCHES octet (l % 4) ^ swz(column). Piece c = PW wave + i: plane c / WNF, columns 16 (c % WNF) ...
    // The per-lane part of the address is the same for all pieces (swz depends on (column / 4) % 4 = lane / 16 only); the piece's
    // part is uniform.
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    unsigned bVoff = ((unsigned)(lane >> 2) * (unsigned)p.Kp + 8u * (unsigned)((lane & 3) ^ swz(lane >> 2))) * 2u;
    const char *bBase[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i)
    {
        const int c = wv * PW + i, pl = c / WNF, cb = c % WNF;
        bBase[i] = reinterpret_cast<const char *>(pl ? p.Wb2 : p.Wb1) + (size_t)(n0 + 16 * cb) * (size_t)p.Kp * 2u;
    }
    auto dma_B = [&](int img) {
#pragma unroll
        for (int i = 0; i < PW; ++i)
        {
            const int c = wv * PW + i, pl = c / WNF, cb = c % WNF;
            load_to_lds_b128(reinterpret_cast<const float *>(bBase[i] + bVoff), reinterpret_cast<float4 *>(&Bp[img][pl][16 * cb][0]));
        }
        bVoff += KT * 2;
    };

    f32x4 aRaw[2][WMF][2];
    u32x4 aPl[2][WMF][3];
    auto load_A = [&](auto setTag) {
        constexpr int SET = decltype(setTag)::value;
        if constexpr (GEN)
        {
            const unsigned tapBit = gTapC < 32 ? 1u << gTapC : 0u; // taps beyond K have no bit in any row mask
            const int segOff = gS1 * p.dil1 * rowLenI + gOffb;     // fits int32 for every layer of the model
#pragma unroll
            for (int i = 0; i < WMF; ++i)
            {
                const float *src = (aMask[i] & tapBit) ? aRow[i] + segOff : p.zero; // (the zero page holds 64 floats)
                aRaw[SET][i][0] = *reinterpret_cast<const f32x4 *>(src);
                aRaw[SET][i][1] = *reinterpret_cast<const f32x4 *>(src + 4);
            }
            gOffb += KT;
            if (p.S1 > 1 && gOffb >= p.seg0) // (seg0 >= 32: the launcher checks)
            {
                gOffb -= p.seg0;
                ++gS1;
            }
            gTapOff += KT;
            while (gTapOff >= p.Cin)
            {
                gTapOff -= p.Cin;
                ++gTapC;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < WMF; ++i)
        {
            const char *src = reinterpret_cast<const char *>(p.X) + aOff[i];
            aRaw[SET][i][0] = *reinterpret_cast<const f32x4 *>(src);
            aRaw[SET][i][1] = *reinterpret_cast<const f32x4 *>(src + 16);
            aOff[i] += KT * 4;
        }
    };
    auto split_block = [&](auto setTag, int i) {
        constexpr int SET = decltype(setTag)::value;
        const f32x4 lo = aRaw[SET][i][0], hi = aRaw[SET][i][1];
        unsigned h1[4], h2[4], h3[4];
        split3_pk(lo[0], lo[1], h1[0], h2[0], h3[0]);
        split3_pk(lo[2], lo[3], h1[1], h2[1], h3[1]);
        split3_pk(hi[0], hi[1], h1[2], h2[2], h3[2]);
        split3_pk(hi[2], hi[3], h1[3], h2[3], h3[3]);
        u32x4 q1{h1[0], h1[1], h1[2], h1[3]}, q2{h2[0], h2[1], h2[2], h2[3]}, q3{h3[0], h3[1], h3[2], h3[3]};
        asm volatile("" : "+v"(q1), "+v"(q2), "+v"(q3)); // (computed HERE, between the MFMA groups)
        aPl[SET][i][0] = q1;
        aPl[SET][i][1] = q2;
        aPl[SET][i][2] = q3;
    };

    f32x4 acc[WMF][WNF];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
        for (int j = 0; j < WNF; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // pipeline: iteration kt multiplies tile kt (weight image PAR, activation planes aPl[PAR]); at its start it requests the
    // activations of tile kt + 2 into register set PAR (free: split one iteration ago) and the weight pieces of tile kt + 1
    // into the other image (free: every wave passed the barrier behind its last reads); between its MFMA groups it splits the
    // activations of tile kt + 1. The activation loads are issued BEFORE the pieces, so waiting for them (vmcnt is in order)
    // never waits for a piece; the pieces are waited for at the end, before the barrier that publishes them.
    const std::integral_constant<int, 0> set0{};
    const std::integral_constant<int, 1> set1{};
    const int nk = (p.Kp + 31) >> 5; // (GEN: Kp may be an odd multiple of 16 - the last half tile meets zero-page activations)
    load_A(set0);
    load_A(set1);
    dma_B(0);
    split_block(set0, 0);
    split_block(set0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int fslot = kq ^ swz(l15);
    auto iteration = [&](auto parTag) {
        constexpr int PAR = decltype(parTag)::value;
        const std::integral_constant<int, PAR ^ 1> other{};
        load_A(parTag); // tile kt + 2 (beyond K: the row's next bytes, never used)
        __builtin_amdgcn_sched_barrier(0);
        dma_B(PAR ^ 1); // tile kt + 1
        u32x4 b1[2][2], b2[2][2];
        auto read_group = [&](int g) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
            {
                const int r = (2 * g + u) * 16 + l15;
                b1[g & 1][u] = Bp[PAR][0][r][fslot];
                b2[g & 1][u] = Bp[PAR][1][r][fslot];
            }
        };
        auto term = [&](int g, u32x4(&b)[2], int plane) {
#pragma unroll
            for (int i = 0; i < WMF; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                    if constexpr (EPI == EPI_VT) // the transposed product (igemm_split_lin_kernel): activations as the matrix pipe's A operand
                        acc[i][2 * g + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aPl[PAR][i][plane]), __builtin_bit_cast(bf16x8, b[u]), acc[i][2 * g + u], 0, 0, 0);
                    else
                        acc[i][2 * g + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[u]), __builtin_bit_cast(bf16x8, aPl[PAR][i][plane]), acc[i][2 * g + u], 0, 0, 0);
                }
        };
        read_group(0);
#pragma unroll
        for (int g = 0; g < NG; ++g)
        {
            if (g + 1 < NG)
                read_group(g + 1);
            // smallest terms first: a3 w1, a2 w2, a1 w2, a2 w1, a1 w1
            term(g, b1[g & 1], 2);
            term(g, b2[g & 1], 1);
            if (g == (WNF == 16 ? 2 : WNF == 12 ? 1 : 0))
                split_block(other, 0);
            if (g == (WNF == 16 ? 5 : WNF == 12 ? 4 : WNF == 6 ? 2 : WNF == 4 ? 1 : 0))
                split_block(other, 1);
            term(g, b2[g & 1], 0);
            term(g, b1[g & 1], 1);
            term(g, b1[g & 1], 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2)
    {
        iteration(set0);
        if (kt + 1 < nk)
            iteration(set1);
    }

    // ---- epilogue, four column fragments at a time (the generic epilogue would hold bias / scale / residual of all 16)
    if constexpr (EPI == EPI_KPL || EPI == EPI_VT)
    {
        float2(*none)[1] = nullptr;
        igemm_epilogue<1, WMF, WNF, EPI, 256>(p, acc, rowinfo_of, none, m0, n0, tileN, wave, 0, BM);
    }
    else if constexpr (WNF == 2)
    {
        // one column block with row statistics (the DConv K1: its GroupNorm partials per row, plan.h rowstat): the shared epilogue
        // whole; the scratch aliases the weight images (every wave is past the K loop's last barrier, no piece is in flight)
        float2(*rsum)[1] = reinterpret_cast<float2(*)[1]>(&Bp[0][0][0][0]);
        igemm_epilogue<1, WMF, WNF, EPI, 256>(p, acc, rowinfo_of, rsum, m0, n0, tileN, wave, 0, BM); // (EPI_TRCONV: no statistics)
    }
    else if constexpr (GEN || WNF != 16)
    {
        // the shared epilogue (igemm_common.h), four column fragments at a time: GLU pairs are adjacent fragments, the
        // transposed-conv scatter and the linear epilogue are per fragment. No row statistics on this path (the launcher checks).
        float2(*none)[1] = nullptr;
        constexpr int JC = WNF % 4 == 0 ? 4 : 6;
#pragma unroll
        for (int jc = 0; jc < WNF; jc += JC)
        {
            f32x4 part[WMF][JC];
#pragma unroll
            for (int i = 0; i < WMF; ++i)
#pragma unroll
                for (int j = 0; j < JC; ++j)
                    part[i][j] = acc[i][jc + j];
            igemm_epilogue<1, WMF, JC, EPI, 256>(p, part, rowinfo_of, none, m0, n0 + 16 * jc, tileN, wave, 0, BM);
        }
    }
    else
    {
        static_assert(EPI == EPI_LINEAR || EPI == EPI_SCALE_RES, "plain linear layers");
        const bool hasRes = EPI == EPI_SCALE_RES || p.res != nullptr;
        // row statistics (linear2 feeds a LayerNorm): per 128 COLUMNS, as the 128-wide tiles write them - a run of four fragments
        // summed per lane in fragment order, reduced over the row's four lanes, two runs added per entry (igemm_epilogue, SSEG = 2):
        // this tile fills the entries 2 tileN and 2 tileN + 1 of its rows, same bits. The scratch aliases the weight images
        // (every wave is past the last barrier of the K loop, no piece is in flight).
        const bool wantStats = p.rowstat != nullptr;
        float2(*rsum)[4] = reinterpret_cast<float2(*)[4]>(&Bp[0][0][0][0]);
#pragma unroll
        for (int jc = 0; jc < WNF; jc += 4)
        {
            float4 biasv[4], scalev[4], resv[WMF][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                const int n = n0 + (jc + j) * 16 + 4 * kq;
                biasv[j] = *reinterpret_cast<const float4 *>(p.bias + n);
                if (EPI == EPI_SCALE_RES)
                    scalev[j] = *reinterpret_cast<const float4 *>(p.scale + n);
#pragma unroll
                for (int i = 0; i < WMF; ++i)
                {
                    const i64 m = m0 + wave * (WMF * 16) + i * 16 + l15;
                    resv[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (hasRes && m < p.M)
                        resv[i][j] = *reinterpret_cast<const float4 *>(p.res + m * p.ldy + n);
                }
            }
#pragma unroll
            for (int i = 0; i < WMF; ++i)
            {
                const int rl = wave * (WMF * 16) + i * 16 + l15;
                const i64 m = m0 + rl;
                const bool rowOk = m < p.M;
                float sm = 0.f, ss = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    const int n = n0 + (jc + j) * 16 + 4 * kq;
                    const f32x4 a = acc[i][jc + j];
                    if (rowOk)
                    {
                        float4 v = make_float4(a[0] + biasv[j].x, a[1] + biasv[j].y, a[2] + biasv[j].z, a[3] + biasv[j].w);
                        if (EPI == EPI_LINEAR)
                        {
                            if (p.act)
                                v = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
                            v.x += resv[i][j].x, v.y += resv[i][j].y, v.z += resv[i][j].z, v.w += resv[i][j].w;
                        }
                        else
                            v = make_float4(resv[i][j].x + v.x * scalev[j].x, resv[i][j].y + v.y * scalev[j].y, resv[i][j].z + v.z * scalev[j].z,
                                            resv[i][j].w + v.w * scalev[j].w);
                        *reinterpret_cast<float4 *>(p.Y + m * p.ldy + n) = v;
                        sm += (v.x + v.y) + (v.z + v.w);
                        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                    }
                }
                if (wantStats)
                {
                    sm += __shfl_xor(sm, 16);
                    ss += __shfl_xor(ss, 16);
                    sm += __shfl_xor(sm, 32);
                    ss += __shfl_xor(ss, 32);
                    if (kq == 0)
                    {
                        rsum[rl][jc >> 2].x = sm;
                        rsum[rl][jc >> 2].y = ss;
                    }
                }
            }
        }
        if (wantStats)
        {
            __syncthreads();
            for (int e = tid; e < 2 * BM; e += 256)
            {
                const int r = e >> 1, hb = e & 1;
                const i64 m = m0 + r;
                if (m < p.M)
                {
                    float sm = 0.f, ss = 0.f;
#pragma unroll
                    for (int w = 0; w < 2; ++w)
                    {
                        sm += rsum[r][2 * hb + w].x;
                        ss += rsum[r][2 * hb + w].y;
                    }
                    float *dst = p.rowstat + (m * p.NB + 2 * tileN + hb) * 2;
                    dst[0] = sm;
                    dst[1] = ss;
                }
            }
        }
    }
}

// the kernels' activation split on an array (dmx_debug_split_activations: unit test of the split itself)
__global__ void split3_debug_kernel(const float *x, i64 n, unsigned short *planes)
{
    const i64 i = ((i64)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n)
        return;
    const float x0 = x[i], x1 = i + 1 < n ? x[i + 1] : 0.f;
    unsigned h1, h2, h3;
    split3_pk(x0, x1, h1, h2, h3);
    planes[i] = (unsigned short)h1, planes[n + i] = (unsigned short)h2, planes[2 * n + i] = (unsigned short)h3;
    if (i + 1 < n)
        planes[i + 1] = (unsigned short)(h1 >> 16), planes[n + i + 1] = (unsigned short)(h2 >> 16), planes[2 * n + i + 1] = (unsigned short)(h3 >> 16);
}
void launch_split3_debug(const float *d_x, i64 n, unsigned short *d_planes, hipStream_t s)
{
    const i64 pairs = (n + 1) / 2;
    hipLaunchKernelGGL(split3_debug_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, d_x, n, d_planes);
}

// per-row scales of a linear layer's A operand (GEMM_FP16X3): a 16-lane group per row, rows of K contiguous floats
__global__ __launch_bounds__(256) void rowscale_kernel(const GemmArgs p, float *out)
{
    const int tid = threadIdx.x, l15 = tid & 15;
    const i64 m = (i64)blockIdx.x * 16 + (tid >> 4);
    if (m >= p.M)
        return;
    const int4 ri = row_info(p, m);
    const float *row = p.X + (i64)ri.x * p.xBS + ((i64)ri.y * p.L0 + (i64)ri.z * p.stride0) * p.Cin;
    // (fmaxf ignores NaN operands: a NaN element still poisons its products through the split; an inf row maximum gives 2^-114.
    // A maximum does not depend on the order it is taken in: four loads in flight per lane, same bits as one)
    float m4[4] = {0.f, 0.f, 0.f, 0.f};
    auto take = [&](int j, int k) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(row + k);
        m4[j] = fmaxf(fmaxf(m4[j], fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    };
    int k = l15 * 4;
    for (; k + 192 < p.K; k += 256)
    {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            take(j, k + 64 * j);
    }
    for (; k < p.K; k += 64)
        take(0, k);
    float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
#pragma unroll
    for (int off = 8; off > 0; off >>= 1)
        mx = fmaxf(mx, __shfl_xor(mx, off));
    if (l15 == 0)
        *reinterpret_cast<float2 *>(out + 2 * m) = rowscale_of(mx);
}
void launch_rowscale(const GemmArgs &a0, float *out, hipStream_t s)
{
    GemmArgs a = a0;
    a.dP0 = make_fastdiv((unsigned)a.P0), a.dP1 = make_fastdiv((unsigned)a.P1);
    hipLaunchKernelGGL(rowscale_kernel, dim3((unsigned)((a.M + 15) / 16)), dim3(256), 0, s, a, out);
}
// the kernels' fp16 split on an array under one scale 2^sexp (dmx_debug_split_activations_fp16: unit test of the split itself)
__global__ void split3h_debug_kernel(const float *x, i64 n, float scale, unsigned short *planes)
{
    const i64 i = ((i64)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n)
        return;
    const float x0 = x[i] * scale, x1 = (i + 1 < n ? x[i + 1] : 0.f) * scale;
    unsigned h1, h2, h3;
    split3h_pk(x0, x1, h1, h2, h3);
    planes[i] = (unsigned short)h1, planes[n + i] = (unsigned short)h2, planes[2 * n + i] = (unsigned short)h3;
    if (i + 1 < n)
        planes[i + 1] = (unsigned short)(h1 >> 16), planes[n + i + 1] = (unsigned short)(h2 >> 16), planes[2 * n + i + 1] = (unsigned short)(h3 >> 16);
}
void launch_split3h_debug(const float *d_x, i64 n, int sexp, unsigned short *d_planes, hipStream_t s)
{
    const i64 pairs = (n + 1) / 2;
    hipLaunchKernelGGL(split3h_debug_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, d_x, n, ldexpf(1.0f, sexp), d_planes);
}

// arith 0: bf16 terms (every tile of the table below); arith 1: fp16 terms - only where igemm_split_lin_kernel applies
// (linear-layer addressing, 128-wide tiles of at least 64 rows); returns -1 where it does not (the caller falls back to
// arith 0). dry: decide only (needs the op's full geometry for arith 1).
// The 128 x 256 linear-layer tile (igemm_split_linw_kernel) is taken per LAUNCH - it produces the bits of the 128-wide tiles -
// where it pays: N a multiple of 256 and enough row tiles that the half as many, twice as large workgroups
// still fill the 64 slots of an XCD at least `kWideMinRounds` times (the tile map deals row tiles to XCDs). DMX_SPLIT_LIN: 0 = the
// staged 2 x 2-wave kernel, 2 = never the wide tile, 3 = the wide tile wherever it exists (bitwise A/B: tools/gpu_lin_ab.py).
static constexpr double kWideMinRounds = 2.5; // (2.62 rounds: linear1 1 - 3 % faster on the wide tile; 1.75: level)
static bool wide_tile_pays(const GemmArgs &a, int mode)
{
    if (a.N % 256 != 0 || a.Np != a.N || a.Kp % 32 != 0 || mode == 2 || mode == 0 || (a.rowstat && a.NB != a.N / 128))
        return false;
    if (mode == 3)
        return true;
    const i64 tm = (a.M + 127) / 128;
    return (double)(((tm + 7) / 8) * (a.N / 256)) / 64.0 >= kWideMinRounds;
}

// The conv-addressed wide tiles (igemm_split_linw_kernel<WNF, EPI, true>): column fragments per wave (16 = 128 x 256, 12 =
// 128 x 192) this launch takes, or 0. Ops of the full-height 128 x 128 / 128 x 96 tiles without prologue and row statistics whose
// A rows are runs of whole 8-float pieces (Cin a multiple of 8) and whose width is whole wide tiles; per LAUNCH, by the same
// occupancy rule as wide_tile_pays (the results are the bits of the narrow tiles). mode = DMX_SPLIT_LIN (0 / 2: never, 3: always).
static int wide_conv_width(const GemmArgs &a, int pro, int epi, int mode)
{
    if (mode == 0 || mode == 2 || pro != PRO_NONE || !(epi == EPI_LINEAR || epi == EPI_GLU || epi == EPI_TRCONV) || a.rowstat)
        return 0;
    const int wnf = a.N % 256 == 0 ? 16 : a.N % 192 == 0 ? 12 : a.N == 96 ? 6 : 0;
    if (!wnf || a.Np != a.N || a.Kp % 16 != 0 || a.Cin % 8 != 0 || a.seg0 % a.Cin != 0 || a.seg0 < 32 || a.S1 * (a.seg0 / a.Cin) > 32 ||
        (i64)a.S1 * a.dil1 * a.L0 * a.Cin + a.seg0 >= (1ll << 31) || (epi == EPI_TRCONV && a.Cout % 4 != 0))
        return 0;
    if (mode == 3)
        return wnf;
    if (mode == 4 && wnf == 6) // (A/B switch: the 96-wide layers stay on the staged tile)
        return 0;
    // Measured at 1 - 42 segments per call against the narrow tile the plan chose (profiles/r06_experiments/wide_tile_rounds.txt):
    // the 96-wide form has the narrow tile's workgroup count and wins everywhere (13 - 24 %); an N = 192 layer replaces TWO 96-wide
    // tiles and is never slower from 0.66 rounds of an XCD's 64 slots on; the others pay from about one round (N = 768 at 0.98
    // rounds: 5 - 14 % faster; N = 384 / 512 at 0.66 rounds: 10 - 25 % slower)
    if (wnf == 6)
        return wnf;
    const i64 tm = (a.M + 127) / 128;
    return (double)(((tm + 7) / 8) * (a.N / (16 * wnf))) / 64.0 >= (a.N == 192 ? 0.6 : 0.9) ? wnf : 0;
}
template <int EPI>
static void launch_wide_conv(GemmArgs a, int wnf, hipStream_t s)
{
    a.tilesM = (unsigned)((a.M + 127) / 128);
    a.tilesN = (unsigned)(a.N / (16 * wnf));
    const dim3 grid(8u * ((a.tilesM + 7u) / 8u) * a.tilesN);
    if (wnf == 16)
        hipLaunchKernelGGL((igemm_split_linw_kernel<16, EPI, true>), grid, dim3(256), 0, s, a);
    else if (wnf == 12)
        hipLaunchKernelGGL((igemm_split_linw_kernel<12, EPI, true>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((igemm_split_linw_kernel<6, EPI, true>), grid, dim3(256), 0, s, a);
}

// The narrow layers in exact-split arithmetic on the direct-fragment kernel, taken at EVERY batch size (full-height tile cfg and
// its half-height sibling alike: one arithmetic per op); -1 where the conditions fail - the op keeps its fp32 kernel. dry: decided
// from the op's whole geometry (api.cpp split_ok fills it).
//   cfg 5 / 12, N <= 32 (128 x 32, four workgroups per CU): the DConv K1 of the C = 192 levels (Conv1d(192 -> 24, k3): K = 576; row
//     statistics) - its fp32 tile (igemm_128x32) fetched every input row three times through LDS staging at 2.7x the algorithmic
//     HBM traffic; here the 88 split operations per 20 MFMAs that kept the staged split tile level with fp32 (round 4) are spread
//     over four workgroups per CU with nothing else to do (1.10 -> 1.03 ms per 42-segment step: the op is bound by its fragment loads);
//   cfg 6 / 13, N = 64 (128 x 64, three workgroups per CU): the frequency branch's last transposed conv of a 4-source model
//     (48 -> 4 x 16, K = 96), which sat on the fp32 matrix pipe at 60 % of its peak in the direct kernel (plan.cpp finish).
static int launch_split_narrow(int cfg, const GemmArgs &a0, hipStream_t s, int arith, bool dry)
{
    const GemmArgs &a = a0;
    const int np = (cfg == 5 || cfg == 12) ? 32 : 64;
    if (arith != 0 || a.pro != PRO_NONE || !(a.epi == EPI_LINEAR || (a.epi == EPI_TRCONV && np == 64)) || (a.epi == EPI_LINEAR && a.res) || a.N > np || a.Np != np ||
        a.Kp % 16 != 0 || a.Cin % 8 != 0 || a.Cin <= 0 || a.seg0 % a.Cin != 0 || a.seg0 < 32 || a.S1 * (a.seg0 / a.Cin) > 32 ||
        (i64)a.S1 * a.dil1 * a.L0 * a.Cin + a.seg0 >= (1ll << 31) || a.NB != 1 || (a.epi == EPI_TRCONV && (a.rowstat || a.Cout % 4 != 0)) ||
        (np == 64 && a.rowstat))
        return -1;
    if (dry)
        return 0;
    GemmArgs k = a;
    k.tilesM = (unsigned)((k.M + 127) / 128);
    k.tilesN = 1;
    k.xcdMap = 1;
    k.dP0 = make_fastdiv((unsigned)k.P0), k.dP1 = make_fastdiv((unsigned)k.P1);
    const dim3 grid(8u * ((k.tilesM + 7u) / 8u));
    if (np == 32)
        hipLaunchKernelGGL((igemm_split_linw_kernel<2, EPI_LINEAR, true>), grid, dim3(256), 0, s, k);
    else if (a.epi == EPI_LINEAR)
        hipLaunchKernelGGL((igemm_split_linw_kernel<4, EPI_LINEAR, true>), grid, dim3(256), 0, s, k);
    else
        hipLaunchKernelGGL((igemm_split_linw_kernel<4, EPI_TRCONV, true>), grid, dim3(256), 0, s, k);
    return 0;
}

template <int WM_, int WN_, int MF, int NF, int PRO, int EPI>
static int launch_split_one(const GemmArgs &a0, hipStream_t s, int arith, bool dry)
{
    // (the K / V plane projections exist only where linear addressing applies: decided below, in dry mode too, from the op's
    // whole geometry - api.cpp split_ok fills it)
    if (dry && arith == 0 && EPI != EPI_KPL && EPI != EPI_VT)
        return 0;
    constexpr int BM = WM_ * MF * 16, BN = WN_ * NF * 16;
    GemmArgs a = a0;
    a.tilesM = (unsigned)((a.M + BM - 1) / BM);
    a.tilesN = (unsigned)((a.N + BN - 1) / BN);
    a.xcdMap = 1;
    a.dP0 = make_fastdiv((unsigned)a.P0), a.dP1 = make_fastdiv((unsigned)a.P1);
    const unsigned blocks = 8u * ((a.tilesM + 7u) / 8u) * a.tilesN;
    // (LIN also requires every staging offset to fit 32 bits: the kernel addresses a row as base + 32-bit byte offset)
    const bool lin = gemm_is_linear(a, PRO, EPI, 32) && ((i64)a.B * a.xBS + 64) * 4 < (1ll << 32) &&
                     ((i64)(a.Wb2 - a.Wb1) + (i64)a.Np * a.Kp + 64) * 2 < (1ll << 32);
    if constexpr (EPI == EPI_KPL || EPI == EPI_VT)
    {
        // the K / V plane projections exist on the linear-layer kernel only (plan.cpp plane_linear keeps them on 128- / 64-row
        // tiles; the V^T form needs its transposed MFMAs)
        if constexpr (PRO == PRO_NONE && WM_ == 2 && WN_ == 2 && NF == 4 && MF >= 2)
            if (lin)
            {
                if (dry)
                    return 0;
                static const int mode = [] { const char *e = getenv("DMX_SPLIT_LIN"); return e ? atoi(e) : 1; }();
                if (arith == 1)
                    hipLaunchKernelGGL((igemm_split_lin_kernel<MF / 2, 8, EPI, 1>), dim3(blocks), dim3(256), 0, s, a);
                else if (MF == 4 && wide_tile_pays(a, mode == 0 ? 1 : mode))
                {
                    a.tilesN = (unsigned)(a.N / 256);
                    hipLaunchKernelGGL((igemm_split_linw_kernel<16, EPI, false>), dim3(8u * ((a.tilesM + 7u) / 8u) * a.tilesN), dim3(256), 0, s, a);
                }
                else
                    hipLaunchKernelGGL((igemm_split_lin_kernel<MF / 2, 8, EPI>), dim3(blocks), dim3(256), 0, s, a);
                return 0;
            }
        (void)BM;
        return -1; // dry: get_plan rebuilds the plan in the fp32-K/V form; a real launch: launch_op reports the error
    }
    else if constexpr (PRO == PRO_NONE && (EPI == EPI_LINEAR || EPI == EPI_SCALE_RES || EPI == EPI_GLU))
    {
        // plain linear layers of a width the 128 x 256 linear tile takes go there (below); every other full-height op of these
        // epilogues - strided convs, 3x3 / k3 / 1x1 rewrites - to the conv-addressed wide tiles where they exist and pay
        if constexpr (EPI != EPI_SCALE_RES && ((WM_ == 2 && WN_ == 2 && MF == 4 && NF == 4) || (WM_ == 4 && WN_ == 1 && MF == 2 && NF == 6)))
            if (arith == 0 && !(lin && EPI == EPI_LINEAR && a.N % 256 == 0))
            {
                static const int mode = [] { const char *e = getenv("DMX_SPLIT_LIN"); return e ? atoi(e) : 1; }();
                if (const int wnf = wide_conv_width(a, PRO, EPI, mode))
                {
                    launch_wide_conv<EPI>(a, wnf, s);
                    return 0;
                }
            }
        if (lin)
        {
            // 128-wide tiles of at least 64 rows: activation fragments straight into registers (igemm_split_lin_kernel);
            // same tile size and map, same bits. DMX_SPLIT_LIN=0 keeps the staged form (A/B comparison). Measured at 42
            // segments (profiles/DESIGN_history_r1-r4.md 7.6): the 34 linear-layer launches 24.62 -> 24.44 ms - the loop is bound by the energy
            // of the bytes it moves from L2, which this form does not change; a 256 x 128 tile with one workgroup per
            // CU (one wave per SIMD, 445 registers) was 15 % slower and is not kept.
            if constexpr (WM_ == 2 && WN_ == 2 && NF == 4 && MF >= 2)
            {
                static const int mode = [] { const char *e = getenv("DMX_SPLIT_LIN"); return e ? atoi(e) : 1; }();
                if (arith == 1) // fp16 terms exist on this kernel only
                {
                    if (!dry)
                        hipLaunchKernelGGL((igemm_split_lin_kernel<MF / 2, 8, EPI, 1>), dim3(blocks), dim3(256), 0, s, a);
                    return 0;
                }
                if constexpr (MF == 4 && EPI != EPI_GLU)
                    if (wide_tile_pays(a, mode))
                    {
                        a.tilesN = (unsigned)(a.N / 256);
                        hipLaunchKernelGGL((igemm_split_linw_kernel<16, EPI, false>), dim3(8u * ((a.tilesM + 7u) / 8u) * a.tilesN), dim3(256), 0, s, a);
                        return 0;
                    }
                if (mode != 0)
                {
                    hipLaunchKernelGGL((igemm_split_lin_kernel<MF / 2, 8, EPI>), dim3(blocks), dim3(256), 0, s, a);
                    return 0;
                }
            }
            if (arith == 1)
                return -1;
            hipLaunchKernelGGL((igemm_split_kernel<WM_, WN_, MF, NF, PRO, EPI, true>), dim3(blocks), dim3(256), 0, s, a);
            return 0;
        }
    }
    if (arith == 1)
        return -1;
    if constexpr (PRO == PRO_NONE && EPI == EPI_TRCONV && ((WM_ == 2 && WN_ == 2 && MF == 4 && NF == 4) || (WM_ == 4 && WN_ == 1 && MF == 2 && NF == 6)))
    {
        static const int mode = [] { const char *e = getenv("DMX_SPLIT_LIN"); return e ? atoi(e) : 1; }();
        if (const int wnf = wide_conv_width(a, PRO, EPI, mode))
        {
            launch_wide_conv<EPI>(a, wnf, s);
            return 0;
        }
    }
    if constexpr (EPI != EPI_KPL && EPI != EPI_VT)
        hipLaunchKernelGGL((igemm_split_kernel<WM_, WN_, MF, NF, PRO, EPI, false>), dim3(blocks), dim3(256), 0, s, a);
    return 0;
}

// does launch_igemm_split run this op (tile cfg, geometry as api.cpp fill_gemm_geometry fills it) on a wide tile of
// igemm_split_linw_kernel? 0, or the tile's width 256 / 192 (the per-op profile labels such launches igemm_split_128x256 /
// igemm_split_128x192: dmx_debug_profile, bench.py)
int igemm_split_is_wide(int cfg, const GemmArgs &a)
{
    static const int mode = [] { const char *e = getenv("DMX_SPLIT_LIN"); return e ? atoi(e) : 1; }();
    if (cfg != 0 && cfg != 2)
        return 0;
    const bool lin = gemm_is_linear(a, a.pro, a.epi, 32) && ((i64)a.B * a.xBS + 64) * 4 < (1ll << 32);
    if (cfg == 0 && lin && (a.epi == EPI_LINEAR || a.epi == EPI_SCALE_RES || a.epi == EPI_KPL || a.epi == EPI_VT) &&
        wide_tile_pays(a, (a.epi == EPI_KPL || a.epi == EPI_VT) && mode == 0 ? 1 : mode))
        return 256;
    if (lin && a.epi == EPI_LINEAR && a.N % 256 == 0) // (launch_split_one: these stay with the linear-layer kernels)
        return 0;
    return 16 * wide_conv_width(a, a.pro, a.epi, mode);
}

// The MFMA-bound tile families only (plan.h kTileCfgs): 0 / 7 / 15 (2x2 waves, 4 column fragments), 9 / 16 (2 column
// fragments), 2 / 10 (4x1 waves, 6 column fragments), 3 / 11 (4x1 waves, 3 column fragments, plain convs only).
// Returns -1 for anything else: the op keeps its fp32 kernel.
int launch_igemm_split(int cfg, const GemmArgs &a, hipStream_t s, bool dry, int arith)
{
    if (a.M >= (1ll << 31) - 256 || !a.Wb1 || !a.Wb2)
        return -1;
#define DMX_CASE(cfgid, WM_, WN_, MF, NF, PRO, EPI) \
    case (cfgid * 100 + PRO * 10 + EPI):           \
        return launch_split_one<WM_, WN_, MF, NF, PRO, EPI>(a, s, arith, dry);
    if (cfg == 5 || cfg == 12 || cfg == 6 || cfg == 13) // 128x32 / 64x32, 128x64 / 64x64 (4 x 1 waves)
        return launch_split_narrow(cfg, a, s, arith, dry);
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
        // K / V projections that write the attention kernel's operand planes (plan.cpp plane_linear)
        DMX_CASE(0, 2, 2, 4, 4, PRO_NONE, EPI_KPL)
        DMX_CASE(0, 2, 2, 4, 4, PRO_NONE, EPI_VT)
        DMX_CASE(7, 2, 2, 2, 4, PRO_NONE, EPI_KPL)
        DMX_CASE(7, 2, 2, 2, 4, PRO_NONE, EPI_VT)
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
        // 48-wide tiles (4 x 1 waves, 3 column fragments): the deepest DConv K1 convs (K = 3 C = 1152, N = C / 8 = 48: 72 flops
        // per byte, above what the fp32 MFMA feeds at HBM speed): 96-102 -> 118-143 TFLOP/s. The 32-wide tiles of the
        // level below (N = 24) were measured on this path too: no change (88 split operations per 20 MFMAs), they keep fp32.
        DMX_CASE(3, 4, 1, 2, 3, PRO_NONE, EPI_LINEAR)
        DMX_CASE(11, 4, 1, 1, 3, PRO_NONE, EPI_LINEAR)
    default:
        return -1;
    }
#undef DMX_CASE
}

} // namespace dmx

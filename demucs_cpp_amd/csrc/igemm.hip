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
// operands, so the 16x16x4 k-slots never need a shuffle. LDS image: row-major, a row = its 4*KS k-quads
// (128 B at KS = 2) with the quad index XOR-swizzled, [row][kq ^ ((row / RPB) % LPR)] float4 (RPB = rows per
// 256 B): every 16-lane ds_read_b128 group (16 rows, one k-quad) hits 16 distinct 16-byte slots of the 64
// banks, and a staging wave writes whole rows (8 lanes = one 128-byte row, 64 lanes = 1 KB contiguous).
// That contiguity is what the DIRECT staging needs: kernels without a prologue transform fetch their tiles
// with global_load_lds_dwordx4 (lane l of a wave writes LDS[M0 base + 16 l]; semantics checked by
// tools/micro/lds_dma.hip) - no staging registers, no ds_write; the swizzle moves to the GLOBAL side (a lane
// fetches k-quad slot ^ f(row) of its row, still one full 128-byte line per 8 lanes).
//
// Tile = (WAVES_M*WMF*16) x (WAVES_N*WNF*16) x (16*KS), 256 threads, LDS double buffered, staging loads
// branch-free (out-of-range chunks read a zero page; validity kept as a bit mask for the prologues).
// K loop (IL, the default): tile t is multiplied while tile t+1 goes registers -> LDS and tile t+2 is
// requested from memory, both in small pieces between the MFMA groups; one barrier per K-tile in the
// middle of the iteration (see the loop). The plain three-phase loop (loads | MFMA block | ds_write +
// addresses) remains for the kernels the interleaved loop does not cover. Prologue, epilogue, the linear-layer
// addressing mode and the loop kind are template parameters (one specialised kernel per combination the
// plan emits).
//
// Alignment contract (checked by the host, dmx_ctx_create): every float4 staging chunk is
// either entirely inside or entirely outside the valid input, and 16-byte aligned:
// Cin*L0, Cin*stride0, Cin*pad0, seg0, K, xBatchStride are multiples of 4 elements.
#include "igemm_common.h"
#include <cstdlib>
#include <type_traits>

#ifndef DMX_SMALL_KS
#define DMX_SMALL_KS 2
#endif
#ifndef DMX_CFG2_KS
#define DMX_CFG2_KS 2
#endif
#ifndef DMX_BIG_KS
#define DMX_BIG_KS 2 // experiment: 1 = 16-deep K-tiles for the 128x128 / 64x128 tiles (half the LDS: 3 workgroups per CU)
#endif
#ifndef DMX_IGEMM_DIRECT
#define DMX_IGEMM_DIRECT 0 // 1: kernels without a prologue transform stage global -> LDS directly (global_load_lds_dwordx4).
                           // Measured at batch 24 on the same box: 120.8 (direct) vs 125.2 TFLOP/s (registers) for the
                           // 128x128 tile, 107.9 vs 112.0 for 128x96 - the LDS write happens either way, and the direct
                           // form needs a full vmcnt(0) drain in front of the barrier. Kept for A/B builds.
#endif
#ifndef DMX_KS1_WAVES
#define DMX_KS1_WAVES 1 // experiment: min waves per SIMD the KS == 1 kernels are compiled for
#endif

namespace dmx
{

// LIN: "linear layer" addressing - one contiguous run of K floats per row (S1 == 1, no padding, K a
// multiple of the K-tile): the staging addresses of a row just advance by one K-tile per iteration, no
// per-tile bounds checks, tap bookkeeping or pointer selects (transformer linears, 1x1 rewrites).
// IL: interleaved main loop (KS == 2). Instead of three phases per K-tile [issue loads | MFMA block |
// ds_write + addresses], the ds_writes of tile t+1 and the global loads of tile t+2 are issued in eight
// small pieces BETWEEN the eight 16-MFMA groups of tile t (pinned with sched_barrier), so a wave's
// instruction stream is a uniform MFMA-dominated mix with no long matrix-idle stretch.
template <int WAVES_M, int WAVES_N, int WMF, int WNF, int KS, int PRO, int EPI, bool LIN, bool IL>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, KS == 2 || WMF * WNF >= 24 ? 2 : DMX_KS1_WAVES) void igemm_kernel(const GemmArgs p)
{
    static_assert(!IL || KS == 2, "interleaved loop is written for 2 k-chunks per tile");
    constexpr int NT = WAVES_M * WAVES_N * 64; // 256 threads; 512 for the double-height tile (4 x 2 waves, ONE workgroup per CU)
    constexpr int BM = WAVES_M * WMF * 16;
    constexpr int BN = WAVES_N * WNF * 16;
    constexpr int LPR = 4 * KS;              // lanes per staged row: one float4 each = the row's 16*KS floats (full 128-B lines at KS=2)
    constexpr int RP = NT / LPR;             // rows staged per pass
    constexpr int RPB = 16 / LPR;            // rows per 256 B of the LDS image (swizzle period)
    constexpr int AR = BM / RP;              // A rows staged per thread (row = tid/LPR + i*RP)
    constexpr int BR = (BN + RP - 1) / RP;   // B rows staged per thread
    static_assert(NT == 256 || NT == 512, "4 or 8 waves");
    static_assert(BM % RP == 0, "BM multiple of the staging pass");
    static_assert((RP / RPB) % LPR == 0 && (16 / RPB) % LPR == 0, "swizzle term constant per lane");
    constexpr bool DIRECT = DMX_IGEMM_DIRECT && IL && PRO == PRO_NONE; // global -> LDS without registers
    constexpr int BRP = BR * RP;             // B rows held (>= BN: the staging passes are whole)

    // two buffers as DISTINCT objects, selected at compile time inside the interleaved loop: the compiler
    // can then tell a direct load into one buffer from the fragment reads of the other (no conservative
    // vmcnt wait in front of every ds_read)
    __shared__ float4 As0[BM][LPR], As1[BM][LPR];
    __shared__ float4 Bs0[BRP][LPR], Bs1[BRP][LPR];
    auto bufA = [&](int b) -> float4(*)[LPR] { return b ? As1 : As0; };
    auto bufB = [&](int b) -> float4(*)[LPR] { return b ? Bs1 : Bs0; };
    __shared__ int4 rowinfo[BM];         // b, p1, p0, group (-1: row >= M)
    __shared__ float2 rsum[BM][WAVES_N]; // cross-wave row statistics

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    unsigned tileM, tileN;
    if (!tile_of_block(p, tileM, tileN)) // XCD-aware workgroup -> tile map (igemm_common.h)
        return;                          // whole workgroup, before any barrier
    const i64 m0 = (i64)tileM * BM;
    const int n0 = (int)tileN * BN;

    for (int r = tid; r < BM; r += NT)
        rowinfo[r] = row_info(p, m0 + r);
    __syncthreads();

    // ---- per-thread staging state: AR rows of A and BR rows of B, all at k-quad `slaneK` (igemm_common.h StageWalk)
    const int slane = tid % LPR, srow = tid / LPR;
    const int slaneK = slane ^ ((srow / RPB) % LPR); // k-quad this lane fetches (LDS slot `slane` of its rows)
    StageWalk<AR, BR, 16 * KS, PRO, LIN> w(p, slaneK);
    w.init([&](int r) { return rowinfo[r]; }, [&](int i) { return srow + i * RP; }, [&](int i) { return srow + i * RP; }, n0, BN);

    // staging registers are NATIVE vectors: a float4 (struct) copied whole is lowered to a memcpy through a
    // private-memory alloca that SROA does not split, i.e. the tile would travel global -> scratch -> LDS
    f32x4 aReg[AR], bReg[BR], gW, gB;
    const int nk16 = p.Kp >> 4;
    const int nk = (nk16 + KS - 1) / KS;

    // Addresses of the NEXT tile are computed one iteration ahead (after the MFMA block), so the loop body starts with
    // nothing but the global loads:  loads(t+1) ; MFMA(t) ; transform+ds_write(t+1) ; addresses(t+2) ; barrier
    unsigned maskHeld = 0;
    auto compute_addrs = [&]() { w.compute_addrs(); };
    auto addr_piece = [&](int c) { w.addr_piece(c); };
    auto issue_loads = [&]() {
#pragma unroll
        for (int i = 0; i < AR; ++i)
            aReg[i] = *reinterpret_cast<const f32x4 *>(w.addrA[i]);
#pragma unroll
        for (int i = 0; i < BR; ++i)
            bReg[i] = *reinterpret_cast<const f32x4 *>(w.addrB[i]);
        if (PRO == PRO_GN_GELU)
        {
            gW = *reinterpret_cast<const f32x4 *>(w.addrG);
            gB = *reinterpret_cast<const f32x4 *>(w.addrG + (p.proB - p.proW));
        }
        maskHeld = w.maskNext;
    };
    // prologue transform (+ zero fill where a transform would make padding non-zero) + LDS write.
    // LDS image [row][slot] float4, slot = k-quad ^ f(row): this lane holds k-quad slaneK = slane ^ f(row), i.e. slot `slane`.
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AR; ++i)
        {
            const f32x4 v = w.transform(aReg[i], i, (maskHeld >> i) & 1u, gW, gB);
            *reinterpret_cast<f32x4 *>(&bufA(buf)[srow + i * RP][slane]) = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
            *reinterpret_cast<f32x4 *>(&bufB(buf)[srow + i * RP][slane]) = bReg[i];
    };

    // piece-wise staging for the interleaved loop: piece 0/1 = first / second half of the A rows,
    // piece 2/3 = first / second half of the B rows
    constexpr int AH = (AR + 1) / 2, BH = (BR + 1) / 2;
    auto load_piece = [&](int piece) {
        if (piece < 2)
        {
#pragma unroll
            for (int i = 0; i < AR; ++i)
                if ((i < AH) == (piece == 0))
                    aReg[i] = *reinterpret_cast<const f32x4 *>(w.addrA[i]);
            if (PRO == PRO_GN_GELU && piece == 0)
            {
                gW = *reinterpret_cast<const f32x4 *>(w.addrG);
                gB = *reinterpret_cast<const f32x4 *>(w.addrG + (p.proB - p.proW));
            }
        }
        else
        {
#pragma unroll
            for (int i = 0; i < BR; ++i)
                if ((i < BH) == (piece == 2))
                    bReg[i] = *reinterpret_cast<const f32x4 *>(w.addrB[i]);
        }
    };
    auto store_piece = [&](int buf, int piece) {
        if (piece < 2)
        {
#pragma unroll
            for (int i = 0; i < AR; ++i)
                if ((i < AH) == (piece == 0))
                {
                    const f32x4 v = w.transform(aReg[i], i, (maskHeld >> i) & 1u, gW, gB);
                    *reinterpret_cast<f32x4 *>(&bufA(buf)[srow + i * RP][slane]) = v;
                }
        }
        else
        {
#pragma unroll
            for (int i = 0; i < BR; ++i)
                if ((i < BH) == (piece == 2))
                    *reinterpret_cast<f32x4 *>(&bufB(buf)[srow + i * RP][slane]) = bReg[i];
        }
    };
    // DIRECT staging: one global_load_lds_dwordx4 per row block; the wave's 64 lanes fill the 64 / LPR rows
    // (wave * 64 / LPR + i * RP ...) of the image, 1 KB contiguous from the wave-uniform base in M0
    auto dload_piece = [&](int buf, int piece) {
        if (piece < 2)
        {
#pragma unroll
            for (int i = 0; i < AR; ++i)
                if ((i < AH) == (piece == 0))
                    load_to_lds_b128(w.addrA[i], &bufA(buf)[wave * (64 / LPR) + i * RP][0]);
        }
        else
        {
#pragma unroll
            for (int i = 0; i < BR; ++i)
                if ((i < BH) == (piece == 2))
                    load_to_lds_b128(w.addrB[i], &bufB(buf)[wave * (64 / LPR) + i * RP][0]);
        }
    };

    f32x4 acc[WMF][WNF];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
        for (int j = 0; j < WNF; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    compute_addrs();
    if constexpr (DIRECT)
    {
        // tiles 0 and 1 go straight to their buffers; the loop requests tile kt+2 into the buffer of tile kt
        // in the second half of iteration kt (every wave has read tile kt completely before the barrier in
        // the middle of the iteration)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            dload_piece(0, c);
        compute_addrs();
#pragma unroll
        for (int c = 0; c < 4; ++c)
            dload_piece(1, c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    else
    {
        issue_loads();
        compute_addrs();
        store_tiles(0);
        if (IL)
            issue_loads(); // tile 1 stays in registers across the first iteration; the loop computes the
                           // addresses of tile kt+2 in its first half and issues its loads in the second
    }
    __syncthreads();
    int cur = 0;
    const int l15 = lane & 15, kq = lane >> 4;
    if constexpr (IL)
    {
        const int fsw = (l15 / RPB) % LPR; // swizzle term of this lane's fragment rows
        auto read_frags = [&](int buf, int ch, f32x4 *a, f32x4 *b) {
#pragma unroll
            for (int i = 0; i < WMF; ++i)
                a[i] = *reinterpret_cast<const f32x4 *>(&bufA(buf)[wm * (WMF * 16) + i * 16 + l15][(ch * 4 + kq) ^ fsw]);
#pragma unroll
            for (int j = 0; j < WNF; ++j)
                b[j] = *reinterpret_cast<const f32x4 *>(&bufB(buf)[wn * (WNF * 16) + j * 16 + l15][(ch * 4 + kq) ^ fsw]);
        };
        // (Measured, kept out: s_setprio 1 / 3 around every MFMA group - the idea being that the co-resident wave gets
        // the issue slots for its memory instructions meanwhile - costs 1 %: 123.7 vs 125.0 TFLOP/s.)
        auto mfma16 = [&](const f32x4 *a, const f32x4 *b, int c) {
#pragma unroll
            for (int i = 0; i < WMF; ++i)
#pragma unroll
                for (int j = 0; j < WNF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(b[j], c), f4c(a[i], c), acc[i][j], 0, 0, 0);
        };
        // ONE barrier per K-tile, in the MIDDLE of the iteration: the ds_writes (or, DIRECT, the landing of the
        // direct loads) of tile kt+1 all sit before it, every read of tile kt's LDS image too (its k-chunk 0
        // fragments were fetched at the end of the previous iteration, k-chunk 1 at group 1), so after the
        // barrier tile kt+1 is complete and tile kt's buffer is free. The second half can therefore already
        // fetch the first fragments of tile kt+1: no ds_read latency is exposed behind the barrier, and a wave
        // waiting at the barrier sits between two MFMA groups while the co-resident workgroup keeps the matrix
        // pipe busy. The buffer index is a compile-time constant (two copies of the body).
        f32x4 a0[WMF], b0[WNF], a1[WMF], b1[WNF];
        read_frags(0, 0, a0, b0);
        auto iteration = [&](auto curTag) {
            constexpr int CUR = decltype(curTag)::value;
            // k-chunk 0 of tile kt  |  ds_write of tile kt+1 (loaded one iteration ago; not DIRECT) and the
            // addresses of tile kt+2, each in four pieces
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                mfma16(a0, b0, c);
                if (!DIRECT)
                    store_piece(CUR ^ 1, c);
                addr_piece(c); // addresses of tile kt+2 (fetched in the second half)
                if (c == 1)
                    read_frags(CUR, 1, a1, b1); // fragments of k-chunk 1 arrive behind the rest of chunk 0
                if (c == 3)
                    maskHeld = w.maskNext; // tile kt+1 is written out: from here on the validity of tile kt+2
                __builtin_amdgcn_sched_barrier(0);
            }
            // DIRECT: this wave's part of tile kt+1 must have LANDED in LDS before the barrier publishes it. The
            // compiler tracks direct loads only against this wave's own LDS reads (and, in the generated code,
            // not across the loop back edge), so the wait is explicit.
            if constexpr (DIRECT)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // k-chunk 1  |  global loads of tile kt+2 (into the registers just written out; DIRECT: into the
            // buffer of tile kt), the first fragments of tile kt+1
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                mfma16(a1, b1, c);
                if constexpr (DIRECT)
                    dload_piece(CUR, c);
                else
                    load_piece(c);
                if (c == 2)
                    read_frags(CUR ^ 1, 0, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int kt = 0; kt < nk; kt += 2)
        {
            iteration(std::integral_constant<int, 0>{});
            if (kt + 1 < nk)
                iteration(std::integral_constant<int, 1>{});
        }
    }
    else
    for (int kt = 0; kt < nk; ++kt)
    {
        issue_loads(); // tile kt+1 (zero page beyond the end: no branch)
#pragma unroll
        for (int ch = 0; ch < KS; ++ch)
        {
            f32x4 a[WMF], b[WNF];
#pragma unroll
            for (int i = 0; i < WMF; ++i)
                a[i] = *reinterpret_cast<const f32x4 *>(&bufA(cur)[wm * (WMF * 16) + i * 16 + l15][(ch * 4 + kq) ^ ((l15 / RPB) % LPR)]);
#pragma unroll
            for (int j = 0; j < WNF; ++j)
                b[j] = *reinterpret_cast<const f32x4 *>(&bufB(cur)[wn * (WNF * 16) + j * 16 + l15][(ch * 4 + kq) ^ ((l15 / RPB) % LPR)]);
            // k sub-step outermost: consecutive MFMAs hit DIFFERENT accumulators (the 16x16x4 f32
            // MFMA has a 40-cycle dependent latency vs a 32-cycle issue interval)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < WMF; ++i)
#pragma unroll
                    for (int j = 0; j < WNF; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(b[j], c), f4c(a[i], c), acc[i][j], 0, 0, 0); // operands swapped: C^T
        }
        store_tiles(cur ^ 1);
        compute_addrs();
        __syncthreads();
        cur ^= 1;
    }

    // ------------------------------------------------------------------ epilogue (igemm_common.h)
    igemm_epilogue<WAVES_N, WMF, WNF, EPI, NT>(p, acc, [&](int r) { return rowinfo[r]; }, rsum, m0, n0, tileN, wm, wn, BM);
}

template <int WM_, int WN_, int MF, int NF, int KS, int PRO, int EPI>
static void launch_one(const GemmArgs &a0, hipStream_t s)
{
    constexpr int BM = WM_ * MF * 16, BN = WN_ * NF * 16;
    constexpr int xcdMap = 1; // XCD-aware tile map (0 = plain row-major tiles: measured slower, round 2)
    GemmArgs a = a0;
    a.tilesM = (unsigned)((a.M + BM - 1) / BM);
    a.tilesN = (unsigned)((a.N + BN - 1) / BN);
    a.xcdMap = xcdMap;
    a.dP0 = make_fastdiv((unsigned)a.P0), a.dP1 = make_fastdiv((unsigned)a.P1);
    const unsigned blocks = xcdMap ? 8u * ((a.tilesM + 7u) / 8u) * a.tilesN : a.tilesM * a.tilesN;
    constexpr int linOn = 1, ilOn = 1; // linear-layer addressing and the interleaved K loop wherever they apply
    constexpr bool CAN_IL = KS == 2;
    if constexpr (PRO == PRO_NONE && (EPI == EPI_LINEAR || EPI == EPI_SCALE_RES || EPI == EPI_GLU) && KS == 2)
    {
        if (linOn && KS == 2 && gemm_is_linear(a, PRO, EPI, 16 * KS))
        {
            if (CAN_IL && ilOn)
                hipLaunchKernelGGL((igemm_kernel<WM_, WN_, MF, NF, KS, PRO, EPI, true, CAN_IL>), dim3(blocks), dim3(WM_ * WN_ * 64), 0, s, a);
            else
                hipLaunchKernelGGL((igemm_kernel<WM_, WN_, MF, NF, KS, PRO, EPI, true, false>), dim3(blocks), dim3(WM_ * WN_ * 64), 0, s, a);
            return;
        }
    }
    if (CAN_IL && ilOn)
        hipLaunchKernelGGL((igemm_kernel<WM_, WN_, MF, NF, KS, PRO, EPI, false, CAN_IL>), dim3(blocks), dim3(WM_ * WN_ * 64), 0, s, a);
    else
        hipLaunchKernelGGL((igemm_kernel<WM_, WN_, MF, NF, KS, PRO, EPI, false, false>), dim3(blocks), dim3(WM_ * WN_ * 64), 0, s, a);
}

// Instantiated (tile, prologue, epilogue) combinations = exactly what plan.cpp emits for the
// 4- and 6-source models at any segment length (enumerated with tests/cpu_interp.cpp
// interp_combos). key = cfg*100 + pro*10 + epi.
int launch_igemm(int cfg, const GemmArgs &a, hipStream_t s, bool dry)
{
    if (a.M >= (1ll << 31) - 256)
        return -1; // 32-bit row arithmetic in the kernel prologue
    if (cfg == 19) // 256x128, four waves of 128x64, linear layers only (igemm_lin256.hip)
        return launch_igemm_lin256(a, s, dry);
#define DMX_CASE(cfgid, WM_, WN_, MF, NF, KS, PRO, EPI) \
    case (cfgid * 100 + PRO * 10 + EPI):                \
        if (!dry)                                       \
            launch_one<WM_, WN_, MF, NF, KS, PRO, EPI>(a, s); \
        return 0;
    switch (cfg * 100 + a.pro * 10 + a.epi)
    {
        // cfg 0: 128x128, cfg 7: 64x128 (same column decomposition)
        DMX_CASE(0, 2, 2, 4, 4, DMX_BIG_KS, PRO_NONE, EPI_LINEAR)
        DMX_CASE(0, 2, 2, 4, 4, DMX_BIG_KS, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(0, 2, 2, 4, 4, DMX_BIG_KS, PRO_NONE, EPI_GLU)
        DMX_CASE(0, 2, 2, 4, 4, DMX_BIG_KS, PRO_NONE, EPI_TRCONV)
        DMX_CASE(0, 2, 2, 4, 4, DMX_BIG_KS, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES)
        DMX_CASE(0, 2, 2, 4, 4, DMX_BIG_KS, PRO_GN_GELU, EPI_STATS_ONLY)
        DMX_CASE(0, 2, 2, 4, 4, DMX_BIG_KS, PRO_GN_GELU, EPI_STATS_FACT) // Demucs v3 level 3: hidden 96 -> 98 factor columns
        // cfg 17: 256x128 = the double-height sibling of 0 (4 x 2 waves of 64x64, one workgroup per CU: 6 instead of 8
        // staged float4 per lane and K-tile; same column decomposition, bit-identical). Experiment, DMX_TALL=1.
        DMX_CASE(17, 4, 2, 4, 4, 2, PRO_NONE, EPI_LINEAR)
        DMX_CASE(17, 4, 2, 4, 4, 2, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(17, 4, 2, 4, 4, 2, PRO_NONE, EPI_GLU)
        DMX_CASE(17, 4, 2, 4, 4, 2, PRO_NONE, EPI_TRCONV)
        // cfg 18: 256x128 with FOUR waves of 128x64 and 16-deep K-tiles (two workgroups per CU stay independent; 12 instead
        // of 16 fragment reads and 6 instead of 8 staged float4 per 128 MFMAs). Experiment, DMX_TALL=2; plain loop only.
        DMX_CASE(18, 2, 2, 8, 4, 1, PRO_NONE, EPI_LINEAR)
        DMX_CASE(18, 2, 2, 8, 4, 1, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(18, 2, 2, 8, 4, 1, PRO_NONE, EPI_GLU)
        DMX_CASE(18, 2, 2, 8, 4, 1, PRO_NONE, EPI_TRCONV)
        // cfg 20: 256x96 with four waves of 64x96 and 16-deep K-tiles, plain loop: the short-K ops of the 128x96 family
        DMX_CASE(20, 4, 1, 4, 6, 1, PRO_NONE, EPI_LINEAR)
        DMX_CASE(20, 4, 1, 4, 6, 1, PRO_NONE, EPI_GLU)
        DMX_CASE(20, 4, 1, 4, 6, 1, PRO_NONE, EPI_TRCONV)
        DMX_CASE(7, 2, 2, 2, 4, DMX_BIG_KS, PRO_NONE, EPI_LINEAR)
        DMX_CASE(7, 2, 2, 2, 4, DMX_BIG_KS, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(7, 2, 2, 2, 4, DMX_BIG_KS, PRO_NONE, EPI_GLU)
        DMX_CASE(7, 2, 2, 2, 4, DMX_BIG_KS, PRO_NONE, EPI_TRCONV)
        DMX_CASE(7, 2, 2, 2, 4, DMX_BIG_KS, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES)
        DMX_CASE(7, 2, 2, 2, 4, DMX_BIG_KS, PRO_GN_GELU, EPI_STATS_ONLY)
        DMX_CASE(7, 2, 2, 2, 4, DMX_BIG_KS, PRO_GN_GELU, EPI_STATS_FACT)
        // cfg 2: 128x96
        DMX_CASE(2, 4, 1, 2, 6, DMX_CFG2_KS, PRO_NONE, EPI_LINEAR)
        DMX_CASE(2, 4, 1, 2, 6, DMX_CFG2_KS, PRO_NONE, EPI_GLU)
        DMX_CASE(2, 4, 1, 2, 6, DMX_CFG2_KS, PRO_NONE, EPI_TRCONV)
        DMX_CASE(2, 4, 1, 2, 6, DMX_CFG2_KS, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES)
        DMX_CASE(2, 4, 1, 2, 6, DMX_CFG2_KS, PRO_GN_GELU, EPI_STATS_ONLY)
        // cfg 3: 128x48
        DMX_CASE(3, 4, 1, 2, 3, DMX_SMALL_KS, PRO_NONE, EPI_LINEAR)
        DMX_CASE(3, 4, 1, 2, 3, DMX_SMALL_KS, PRO_NONE, EPI_TRCONV)
        DMX_CASE(3, 4, 1, 2, 3, DMX_SMALL_KS, PRO_AFFINE, EPI_LINEAR)
        // cfg 4: 256x16, cfg 5: 128x32, cfg 6: 128x64
        DMX_CASE(4, 4, 1, 4, 1, DMX_SMALL_KS, PRO_NONE, EPI_LINEAR)
        DMX_CASE(4, 4, 1, 4, 1, DMX_SMALL_KS, PRO_GN_GELU, EPI_STATS_FACT)
        DMX_CASE(5, 4, 1, 2, 2, DMX_SMALL_KS, PRO_GN_GELU, EPI_STATS_FACT)
        DMX_CASE(6, 4, 1, 2, 4, DMX_SMALL_KS, PRO_GN_GELU, EPI_STATS_FACT)
        DMX_CASE(5, 4, 1, 2, 2, DMX_SMALL_KS, PRO_NONE, EPI_LINEAR)
        DMX_CASE(5, 4, 1, 2, 2, DMX_SMALL_KS, PRO_NONE, EPI_TRCONV)
        DMX_CASE(6, 4, 1, 2, 4, DMX_SMALL_KS, PRO_NONE, EPI_TRCONV)
        DMX_CASE(6, 4, 1, 2, 4, DMX_SMALL_KS, PRO_NONE, EPI_LINEAR)
        // half-height siblings (plan.cpp refine_cfg): cfg 9: 64x64 of 7; 10: 64x96 of 2; 11: 64x48 of 3; 12: 64x32 of 5;
        // 13: 64x64 (4 x 1 waves) of 6; 14: 128x16 of 4
        DMX_CASE(9, 2, 2, 2, 2, 2, PRO_NONE, EPI_LINEAR)
        DMX_CASE(9, 2, 2, 2, 2, 2, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(9, 2, 2, 2, 2, 2, PRO_NONE, EPI_GLU)
        DMX_CASE(9, 2, 2, 2, 2, 2, PRO_NONE, EPI_TRCONV)
        DMX_CASE(9, 2, 2, 2, 2, 2, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES)
        DMX_CASE(10, 4, 1, 1, 6, DMX_CFG2_KS, PRO_NONE, EPI_LINEAR)
        DMX_CASE(10, 4, 1, 1, 6, DMX_CFG2_KS, PRO_NONE, EPI_GLU)
        DMX_CASE(10, 4, 1, 1, 6, DMX_CFG2_KS, PRO_NONE, EPI_TRCONV)
        DMX_CASE(10, 4, 1, 1, 6, DMX_CFG2_KS, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES)
        DMX_CASE(10, 4, 1, 1, 6, DMX_CFG2_KS, PRO_GN_GELU, EPI_STATS_ONLY)
        DMX_CASE(11, 4, 1, 1, 3, DMX_SMALL_KS, PRO_NONE, EPI_LINEAR)
        DMX_CASE(11, 4, 1, 1, 3, DMX_SMALL_KS, PRO_NONE, EPI_TRCONV)
        DMX_CASE(11, 4, 1, 1, 3, DMX_SMALL_KS, PRO_AFFINE, EPI_LINEAR)
        DMX_CASE(12, 4, 1, 1, 2, DMX_SMALL_KS, PRO_GN_GELU, EPI_STATS_FACT)
        DMX_CASE(12, 4, 1, 1, 2, DMX_SMALL_KS, PRO_NONE, EPI_LINEAR)
        DMX_CASE(12, 4, 1, 1, 2, DMX_SMALL_KS, PRO_NONE, EPI_TRCONV)
        DMX_CASE(13, 4, 1, 1, 4, DMX_SMALL_KS, PRO_GN_GELU, EPI_STATS_FACT)
        DMX_CASE(13, 4, 1, 1, 4, DMX_SMALL_KS, PRO_NONE, EPI_TRCONV)
        DMX_CASE(13, 4, 1, 1, 4, DMX_SMALL_KS, PRO_NONE, EPI_LINEAR)
        DMX_CASE(14, 4, 1, 2, 1, DMX_SMALL_KS, PRO_NONE, EPI_LINEAR)
        DMX_CASE(14, 4, 1, 2, 1, DMX_SMALL_KS, PRO_GN_GELU, EPI_STATS_FACT)
        // quarter-height siblings: cfg 15: 32x128 of 7 (same column decomposition); 16: 32x64 of 9
        DMX_CASE(15, 2, 2, 1, 4, 2, PRO_NONE, EPI_LINEAR)
        DMX_CASE(15, 2, 2, 1, 4, 2, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(15, 2, 2, 1, 4, 2, PRO_NONE, EPI_GLU)
        DMX_CASE(15, 2, 2, 1, 4, 2, PRO_NONE, EPI_TRCONV)
        DMX_CASE(15, 2, 2, 1, 4, 2, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES)
        DMX_CASE(15, 2, 2, 1, 4, 2, PRO_GN_GELU, EPI_STATS_ONLY)
        DMX_CASE(15, 2, 2, 1, 4, 2, PRO_GN_GELU, EPI_STATS_FACT)
        DMX_CASE(16, 2, 2, 1, 2, 2, PRO_NONE, EPI_LINEAR)
        DMX_CASE(16, 2, 2, 1, 2, 2, PRO_NONE, EPI_SCALE_RES)
        DMX_CASE(16, 2, 2, 1, 2, 2, PRO_NONE, EPI_GLU)
        DMX_CASE(16, 2, 2, 1, 2, 2, PRO_NONE, EPI_TRCONV)
        DMX_CASE(16, 2, 2, 1, 2, 2, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES)
    default:
        return -1;
    }
#undef DMX_CASE
}

} // namespace dmx

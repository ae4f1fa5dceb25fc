// igemm_lin256.hip — fp32 MFMA GEMM for the LINEAR layers of HTDemucs on a 256x128 tile with FOUR waves (gfx950).
//
// The transformer linears (/root/reference/src/layers.cpp:426-440,488-511) and the 1x1 channel up / down samplers
// (src/crosstransformer.cpp:213-227,327-339) are 40 % of the time of the dominant igemm class, have short K (512) and
// the simplest addressing (one contiguous run of K floats per row). For them the 128x128 tile of igemm.hip (four
// waves of 64x64) spends 16 ds_read_b128 and 8 staged float4 per lane on every 128 MFMAs. Here a workgroup is still
// four waves - two independent workgroups per CU keep overlapping one's epilogue with the other's loop, which is why
// the EIGHT-wave 256x128 tile lost (profiles/DESIGN_history_r1-r4.md section 7.1) - but a wave owns 128x64 outputs (8 x 4 fragments, 128
// accumulator registers) and a K-tile is 16 deep, so that two double-buffered LDS images (48 KB) fit twice per CU:
// 12 fragment reads and 6 staged float4 per 128 MFMAs, one barrier per K-tile as before.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 with the operands swapped (weights as A, activations as B: the accumulator holds
// C^T) and k ascending, i.e. every output element is the same fmaf chain as in igemm.hip / dgemm.hip: results are
// bit-identical to the 128x128 tile (tested), the choice between the two is a per-launch speed decision (plan.cpp).
//
// K loop (one K-tile = 16 k, 8 groups of 16 MFMAs, group i = row fragment i against the 4 column fragments):
//   groups 0-3 | ds_write of tile t+1 (registers loaded one iteration ago), ds_read of A fragments 4-7 of tile t
//   barrier    | tile t+1 complete; nobody reads tile t's image any more
//   groups 4-7 | global loads of tile t+2, ds_read of the B fragments (second register set) and A fragments 0-3 of t+1
// A fragments are single-buffered: fragment i is dead after group i, so fragments 0-3 of the next tile arrive during
// groups 6-7 and fragments 4-7 during groups 0-1 of the next iteration.
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

namespace dmx
{

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float l2_gelu(float v) { return dmx_gelu(v); }

// STAT: the op also writes per-row (sum, sum of squares) partials of its 128-column block (plan.h rowstat), summed in the
// order of igemm.hip: per lane over its 4 fragments, across the 4 lanes of a row (xor 16, 32), then wave 0 + wave 1.
template <int EPI, bool STAT>
__global__ __launch_bounds__(256, 2) void igemm_lin256_kernel(const GemmArgs p)
{
    constexpr int BM = 256, BN = 128, WMF = 8, WNF = 4;
    __shared__ f32x4 As0[BM][4], As1[BM][4];
    __shared__ f32x4 Bs0[BN][4], Bs1[BN][4];
    __shared__ i64 rowBase[BM]; // element offset of a row's K-run in X, -1: row >= M
    __shared__ float2 rsum[STAT ? BM : 1][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, kq = lane >> 4;
    // workgroup -> tile: all column tiles of a row tile on ONE XCD, adjacent in dispatch order (igemm.hip)
    unsigned tileM, tileN;
    {
        const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const unsigned mi = j / p.tilesN;
        tileN = j - mi * p.tilesN;
        tileM = mi * 8u + xcd;
        if (tileM >= p.tilesM)
            return;
    }
    const i64 m0 = (i64)tileM * BM;
    const int n0 = (int)tileN * BN;

    const i64 rowLen = (i64)p.L0 * p.Cin;
    for (int r = tid; r < BM; r += 256)
    {
        const i64 m = m0 + r;
        i64 off = -1;
        if (m < p.M)
        {
            const unsigned mu = (unsigned)m;
            const unsigned t = p.dP0.magic ? (__umulhi(mu, p.dP0.magic) >> p.dP0.shift) : (mu >> p.dP0.shift);
            const int p0 = (int)(mu - t * (unsigned)p.P0);
            const unsigned b = p.dP1.magic ? (__umulhi(t, p.dP1.magic) >> p.dP1.shift) : (t >> p.dP1.shift);
            const int p1 = (int)(t - b * (unsigned)p.P1);
            off = (i64)b * p.xBS + (i64)p1 * rowLen + (i64)p0 * p.stride0 * p.Cin;
        }
        rowBase[r] = off;
    }
    __syncthreads();

    // ---- staging: thread -> (row srow + 64 i, LDS slot slane); it fetches k-quad slane ^ f(row), f = (-(row >> 2)) & 3.
    // The swizzle is made for the REAL lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31}, ...): a fragment read takes rows l15 at quad kq, lane = 16 kq + l15, so a group mixes rows
    // q = l15 >> 2 in {0, 3} at one kq with rows q in {1, 2} at kq ^ 1; f(q) = {0, 3, 2, 1} puts its 16 accesses on 16
    // distinct 16-byte slots of the 256-byte bank row. (f = q, the rule for 16 CONSECUTIVE lanes, was two-way conflicted:
    // SQ_LDS_BANK_CONFLICT = 33 % of the LDS cycles of the first version of this kernel.)
    const int slane = tid & 3, srow = tid >> 2;
    const int slaneK = slane ^ ((0 - (srow >> 2)) & 3);
    const float *aPtr[4], *bPtr[2];
    int aStep[4], bStep[2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const i64 off = rowBase[srow + 64 * i];
        aPtr[i] = off >= 0 ? p.X + off + 4 * slaneK : p.zero;
        aStep[i] = off >= 0 ? 16 : 0;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
    {
        const int n = n0 + srow + 64 * i;
        const bool ok = n < p.Np;
        bPtr[i] = ok ? p.Wt + (i64)n * p.Kp + 4 * slaneK : p.zero;
        bStep[i] = ok ? 16 : 0;
    }
    const int nk = p.Kp >> 4;
    f32x4 sA[4], sB[2];
    auto load_A = [&](int lo, int hi) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i >= lo && i < hi)
            {
                sA[i] = *reinterpret_cast<const f32x4 *>(aPtr[i]);
                aPtr[i] += aStep[i];
            }
    };
    auto load_B = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
        {
            sB[i] = *reinterpret_cast<const f32x4 *>(bPtr[i]);
            bPtr[i] += bStep[i];
        }
    };
    auto store_A = [&](int buf, int lo, int hi) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i >= lo && i < hi)
                (buf ? As1 : As0)[srow + 64 * i][slane] = sA[i];
    };
    auto store_B = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            (buf ? Bs1 : Bs0)[srow + 64 * i][slane] = sB[i];
    };
    // beyond the last tile the walk would leave the rows: park every pointer on the zero page
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            aPtr[i] = p.zero, aStep[i] = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            bPtr[i] = p.zero, bStep[i] = 0;
    };

    f32x4 acc[WMF][WNF];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
        for (int j = 0; j < WNF; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fsw = (0 - (l15 >> 2)) & 3; // swizzle term of this lane's fragment rows (fragment bases are multiples of 16)
    f32x4 aF[WMF], bF[2][WNF];
    auto read_A = [&](int buf, int lo, int hi) {
#pragma unroll
        for (int i = 0; i < WMF; ++i)
            if (i >= lo && i < hi)
                aF[i] = (buf ? As1 : As0)[wm * 128 + i * 16 + l15][kq ^ fsw];
    };
    auto read_B = [&](int buf, int set) {
#pragma unroll
        for (int j = 0; j < WNF; ++j)
            bF[set][j] = (buf ? Bs1 : Bs0)[wn * 64 + j * 16 + l15][kq ^ fsw];
    };
    auto mfma_group = [&](int i, int set) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < WNF; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bF[set][j][c], aF[i][c], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: tile 0 -> LDS buffer 0, tile 1 -> registers, fragments 0-3 and B of tile 0
    load_A(0, 4);
    load_B();
    store_A(0, 0, 4);
    store_B(0);
    if (nk > 1)
    {
        load_A(0, 4);
        load_B();
    }
    if (nk <= 2)
        park();
    __syncthreads();
    read_A(0, 0, 4);
    read_B(0, 0);

    // one K-tile; CUR = buffer of tile t = fragment register set of B
    auto iteration = [&](auto curTag, int t) {
        constexpr int CUR = decltype(curTag)::value;
        // groups 0-3 | write tile t+1, fetch A fragments 4-7 of tile t
        read_A(CUR, 4, 8);
        mfma_group(0, CUR);
        store_A(CUR ^ 1, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(1, CUR);
        store_A(CUR ^ 1, 2, 4);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(2, CUR);
        store_B(CUR ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(3, CUR);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // groups 4-7 | request tile t+2, fetch the fragments of tile t+1
        mfma_group(4, CUR);
        load_A(0, 2);
        read_B(CUR ^ 1, CUR ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(5, CUR);
        load_A(2, 4);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(6, CUR);
        load_B();
        read_A(CUR ^ 1, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(7, CUR);
        read_A(CUR ^ 1, 2, 4);
        if (t + 3 >= nk) // tile t+3 does not exist: the next iteration's loads stay on the zero page
            park();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = 0; t < nk; t += 2)
    {
        iteration(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nk)
            iteration(std::integral_constant<int, 1>{}, t + 1);
    }

    // ---- epilogue (C^T layout): lane (l15, kq) owns row wm*128 + 16 i + l15 and channels n0 + wn*64 + 16 j + 4 kq .. +3
    const int colBase = n0 + wn * 64 + 4 * kq;
    f32x4 biasv[WNF], scalev[WNF];
#pragma unroll
    for (int j = 0; j < WNF; ++j)
    {
        const int n = colBase + 16 * j;
        biasv[j] = *reinterpret_cast<const f32x4 *>(n < p.N ? p.bias + n : p.zero);
        scalev[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI == EPI_SCALE_RES && n < p.N)
            scalev[j] = *reinterpret_cast<const f32x4 *>(p.scale + n);
    }
#pragma unroll
    for (int i = 0; i < WMF; ++i)
    {
        const int rl = wm * 128 + i * 16 + l15;
        const i64 m = m0 + rl;
        const bool rowOk = m < p.M;
        float rs = 0.f, rss = 0.f;
        f32x4 resv[WNF];
#pragma unroll
        for (int j = 0; j < WNF; ++j)
        {
            const int n = colBase + 16 * j;
            resv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (rowOk && n < p.N && (EPI == EPI_SCALE_RES || p.res))
                resv[j] = *reinterpret_cast<const f32x4 *>(p.res + m * p.ldy + n); // may alias Y element-wise: read before written
        }
#pragma unroll
        for (int j = 0; j < WNF; ++j)
        {
            const int n = colBase + 16 * j;
            if (rowOk && n < p.N)
            {
                f32x4 v = acc[i][j] + biasv[j];
                if (EPI == EPI_LINEAR)
                {
                    if (p.act)
                        v = f32x4{l2_gelu(v[0]), l2_gelu(v[1]), l2_gelu(v[2]), l2_gelu(v[3])};
                    v += resv[j];
                }
                else
                    v = resv[j] + v * scalev[j];
                *reinterpret_cast<f32x4 *>(p.Y + m * p.ldy + n) = v;
                if (STAT)
                {
                    rs += (v[0] + v[1]) + (v[2] + v[3]);
                    rss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                }
            }
        }
        if (STAT)
        {
            rs += __shfl_xor(rs, 16);
            rss += __shfl_xor(rss, 16);
            rs += __shfl_xor(rs, 32);
            rss += __shfl_xor(rss, 32);
            if (kq == 0)
            {
                rsum[rl][wn].x = rs;
                rsum[rl][wn].y = rss;
            }
        }
    }
    if (STAT)
    {
        __syncthreads();
        for (int r = tid; r < BM; r += 256)
        {
            const i64 m = m0 + r;
            if (m < p.M)
            {
                float *dst = p.rowstat + (m * p.NB + tileN) * 2;
                dst[0] = rsum[r][0].x + rsum[r][1].x;
                dst[1] = rsum[r][0].y + rsum[r][1].y;
            }
        }
    }
}

// shapes this kernel covers: "linear layer" addressing (igemm.hip is_linear), no prologue
bool lin256_ok(const GemmArgs &a)
{
    return a.pro == PRO_NONE && (a.epi == EPI_LINEAR || a.epi == EPI_SCALE_RES) && a.S1 == 1 && a.pad0 == 0 && a.seg0 == a.K &&
           a.K == a.Kp && a.K % 16 == 0 && a.Np % 4 == 0 && a.N % 4 == 0 && (i64)(a.P0 - 1) * a.stride0 * a.Cin + a.seg0 <= (i64)a.L0 * a.Cin &&
           a.P1 == a.L1 && a.stride1 == 1 && a.pad1 == 0 && a.M < (1ll << 31) - 256 && (a.epi != EPI_SCALE_RES || (a.res && a.scale));
}

int launch_igemm_lin256(const GemmArgs &a0, hipStream_t s, bool dry)
{
    if (dry) // availability check at context creation (partial arguments): the plan decided the shape with full information
        return a0.pro == PRO_NONE && (a0.epi == EPI_LINEAR || a0.epi == EPI_SCALE_RES) ? 0 : -1;
    if (!lin256_ok(a0))
        return -1;
    GemmArgs a = a0;
    a.tilesM = (unsigned)((a.M + 255) / 256);
    a.tilesN = (unsigned)((a.N + 127) / 128);
    a.xcdMap = 1;
    a.dP0 = make_fastdiv((unsigned)a.P0), a.dP1 = make_fastdiv((unsigned)a.P1);
    const unsigned blocks = 8u * ((a.tilesM + 7u) / 8u) * a.tilesN;
    if (a.epi == EPI_LINEAR && !a.rowstat)
        hipLaunchKernelGGL((igemm_lin256_kernel<EPI_LINEAR, false>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.epi == EPI_LINEAR)
        hipLaunchKernelGGL((igemm_lin256_kernel<EPI_LINEAR, true>), dim3(blocks), dim3(256), 0, s, a);
    else if (!a.rowstat)
        hipLaunchKernelGGL((igemm_lin256_kernel<EPI_SCALE_RES, false>), dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((igemm_lin256_kernel<EPI_SCALE_RES, true>), dim3(blocks), dim3(256), 0, s, a);
    return 0;
}

} // namespace dmx

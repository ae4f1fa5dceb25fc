// dgemm.hip — "direct" fp32 MFMA implicit-GEMM for the HBM-bound layers of HTDemucs (gfx950).
//
// Level-0/1 layers (C = 48 / 96: first encoders, last decoders and their DConv branches;
// /root/reference/src/encdec.cpp:8-164,166-361, src/layers.cpp:152-375) have K or N so small that
// the whole weight matrix fits in one wave's registers and an A element is used by exactly one
// wave. For those the LDS-staged tile kernel (igemm.hip) only adds barriers and a round trip
// through LDS; they are bound by HBM, 57 % of the reference's unfused bytes.
// Here:
//   * one wave owns 16 output rows x ALL N columns at a time (NF column fragments of 16);
//   * the weights are loaded ONCE per wave into registers, in MFMA B-operand order;
//   * A is read straight from the channels-last activation into MFMA A-operand registers:
//     lane (row = lane&15, h = lane>>4) loads 16 B at k = 16j + 4h of its row, so the four
//     k-slots of v_mfma_f32_16x16x4_f32 take real k = 16j + 4h + c without any shuffle;
//     runs that are not a multiple of 16 long (DConv hidden width 8 / 12) use k = RPL*h + c;
//   * no LDS, no barrier; waves walk the row fragments with a grid stride (persistent);
//   * the same prologues / epilogues as igemm.hip (plan.h), same row-statistics contract.
// Semantics are specified by plan.h and tests/cpu_interp.cpp exactly like igemm.hip.
#include "kernels.h"
#include <cstdlib>

namespace dmx
{

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dgelu(float v) { return dmx_gelu(v); }
// 1 / (1 + e^-v) with v_rcp_f32 (1 ulp): the IEEE division expands to ~10 VALU instructions, and a GLU
// epilogue evaluates one sigmoid per output
__device__ __forceinline__ float dsigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv f)
{
    return f.magic ? (__umulhi(n, f.magic) >> f.shift) : (n >> f.shift);
}
// `ok ? *ptr : zero` as written selects between a global pointer and a private temporary and loads
// through a FLAT pointer (plus a scratch slot); select the address against the zero page instead
__device__ __forceinline__ float4 ld4z(const float *ptr, bool ok, const float *zero)
{
    const f32x4 v = *reinterpret_cast<const f32x4 *>(ok ? ptr : zero);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float f4get(const float4 &v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

template <int NF, int S1, int SEG0, int PRO, int EPI, int PIPE>
__global__ __launch_bounds__(256) void dgemm_kernel(const GemmArgs pa, const FastDiv dP0, const FastDiv dP1)
{
    // column chunks (launch_dgemm: the deep-level DConv k3): workgroups of grid row y own the packed columns
    // [y N, (y + 1) N) of a wider op - N is the chunk width here - i.e. output channels [y N / 2, (y + 1) N / 2)
    GemmArgs p = pa;
    if (gridDim.y > 1)
    {
        const int n0 = (int)blockIdx.y * pa.N;
        p.Wt += (i64)n0 * pa.Kp, p.bias += n0, p.epiW += n0, p.epiB += n0, p.scale += n0 / 2;
        p.Y += n0 / 2, p.res += n0 / 2;
    }
    constexpr int NV4 = SEG0 / 16;          // float4 steps per contiguous run
    constexpr int RPL = (SEG0 % 16) / 4;    // remainder floats per lane (0, 2 or 3)
    constexpr int NV4A = NV4 > 0 ? NV4 : 1; // array extents must not be 0
    constexpr int RPLA = RPL > 0 ? RPL : 1;
    const int lane = threadIdx.x & 63, l15 = lane & 15, h = lane >> 4;
    const int gwave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int rowLen = p.L0 * p.Cin;

    // ---- weights -> registers (B operand: lane holds column n = 16 fj + l15)
    float4 Bv[NF][S1][NV4A];
    float Br[NF][S1][RPLA];
#pragma unroll
    for (int fj = 0; fj < NF; ++fj)
    {
        const int n = fj * 16 + l15;
        const float *w = p.Wt + (i64)(n < p.Np ? n : 0) * p.Kp;
        const bool nOk = n < p.Np;
#pragma unroll
        for (int s = 0; s < S1; ++s)
        {
#pragma unroll
            for (int j = 0; j < NV4; ++j)
                Bv[fj][s][j] = ld4z(w + s * SEG0 + 16 * j + 4 * h, nOk, p.zero);
#pragma unroll
            for (int c = 0; c < RPL; ++c)
                Br[fj][s][c] = nOk ? w[s * SEG0 + 16 * NV4 + RPL * h + c] : 0.f;
        }
    }
    // prologue constants of this lane's k positions (PRO_GN_GELU has S1 == 1)
    float4 gW4[NV4A], gB4[NV4A];
    float gWr[RPLA], gBr[RPLA];
    if (PRO == PRO_GN_GELU)
    {
#pragma unroll
        for (int j = 0; j < NV4; ++j)
        {
            gW4[j] = *reinterpret_cast<const float4 *>(p.proW + 16 * j + 4 * h);
            gB4[j] = *reinterpret_cast<const float4 *>(p.proB + 16 * j + 4 * h);
        }
#pragma unroll
        for (int c = 0; c < RPL; ++c)
        {
            gWr[c] = p.proW[16 * NV4 + RPL * h + c];
            gBr[c] = p.proB[16 * NV4 + RPL * h + c];
        }
    }
    // per-column epilogue constants. The MFMAs are issued with the operands SWAPPED (weights as A,
    // activations as B), i.e. they produce C^T: lane (l15, h) then holds, for ITS OWN row m = 16 frag +
    // l15, the 4 consecutive channels n = 16 fj + 4h + r -> float4 global accesses, no shuffles.
    // The GroupNorm+GLU epilogue needs 3.5 float4 constants per column fragment: those live in LDS
    // (broadcast ds_read_b128, 16 lanes per address) so that the kernel keeps >= 4 waves per SIMD.
    constexpr bool CST_LDS = EPI == EPI_GN_GLU_SCALE_RES;
    __shared__ float4 cst[CST_LDS ? 4 : 1][NF][4]; // [bias | gn weight | gn bias | layer scale][fj][h]
    float4 biasv[CST_LDS ? 1 : NF];
    int trR[NF], trC[NF];
    if (CST_LDS)
    {
        for (int i = threadIdx.x; i < NF * 4; i += 256)
        {
            const int fj = i >> 2, hh = i & 3, n = fj * 16 + 4 * hh;
            const bool ok = n < p.N;
            cst[0][fj][hh] = ld4z(p.bias + n, ok, p.zero);
            cst[CST_LDS ? 1 : 0][fj][hh] = ld4z(p.epiW + n, ok, p.zero);
            cst[CST_LDS ? 2 : 0][fj][hh] = ld4z(p.epiB + n, ok, p.zero);
            cst[CST_LDS ? 3 : 0][fj][hh] = ld4z(p.scale + (fj >> 1) * 16 + 4 * hh, ok && (fj & 1) == 0, p.zero);
        }
        __syncthreads();
    }
#pragma unroll
    for (int fj = 0; fj < NF; ++fj)
    {
        const int n = fj * 16 + 4 * h; // N, Np, Cout are multiples of 4
        if (!CST_LDS)
            biasv[fj] = ld4z(p.bias + n, n < p.N, p.zero);
        trR[fj] = trC[fj] = 0;
        if (EPI == EPI_TRCONV)
        {
            trR[fj] = n / p.Cout;
            trC[fj] = n - trR[fj] * p.Cout;
        }
    }

    // ---- software pipeline over this wave's row fragments: everything fragment f+1 needs from memory
    // (A runs, normalisation statistics of its row group, residual / table operands) is requested
    // BEFORE the MFMAs and the epilogue of fragment f, so a wave always has one fragment in flight.
    constexpr int NRES = (EPI == EPI_GLU || EPI == EPI_GN_GLU_SCALE_RES) ? (NF + 1) / 2 : 1;
    struct Stage
    {
        unsigned m, b;
        int p0, p1;
        bool rowOk;
        float aMean, aScale, eMean, eSc;
        f32x4 a4[S1][NV4A]; // native vectors: whole-struct float4 copies (cur = nxt) would go through scratch
        float ar[S1][RPLA];
        unsigned ok4, okr; // validity bits [s * NV4 + j] / [s]
        f32x4 resv[NRES];
    };
    const int nfrag = (int)((p.M + 15) >> 4);
    const int C2 = p.N >> 1;
    auto fetch = [&](int frag, Stage &st) {
        const unsigned m = (unsigned)frag * 16u + (unsigned)l15;
        const bool rowOk = frag < nfrag && (i64)m < p.M;
        const unsigned mm = rowOk ? m : 0u;
        const unsigned t1 = fdiv(mm, dP0);
        const int p0 = (int)(mm - t1 * (unsigned)p.P0);
        const unsigned b = fdiv(t1, dP1);
        const int p1 = (int)(t1 - b * (unsigned)p.P1);
        const int grp = (int)b * p.G0 + (p.G0 > 1 ? p0 : 0);
        st.m = m, st.b = b, st.p0 = p0, st.p1 = p1, st.rowOk = rowOk;
        st.aMean = 0.f, st.aScale = 1.f, st.eMean = 0.f, st.eSc = 1.f;
        if (PRO == PRO_AFFINE)
        {
            st.aMean = p.proStats[b * 4];
            st.aScale = p.proStats[b * 4 + 1];
        }
        if (PRO == PRO_GN_GELU)
        {
            st.aMean = p.proStats[grp * 4];
            st.aScale = p.proStats[grp * 4 + 1];
        }
        if (EPI == EPI_GN_GLU_SCALE_RES)
        {
            st.eMean = p.epiStats[grp * 4];
            st.eSc = p.epiStats[grp * 4 + 1];
        }
        const int e0 = (p0 * p.stride0 - p.pad0) * p.Cin;
        const float *xb = p.X + (i64)b * p.xBS;
        st.ok4 = 0, st.okr = 0;
#pragma unroll
        for (int s = 0; s < S1; ++s)
        {
            const int in1 = p1 * p.stride1 + s * p.dil1 - p.pad1;
            const bool ok1 = rowOk && in1 >= 0 && in1 < p.L1;
            const float *rowp = xb + (i64)(ok1 ? in1 : 0) * rowLen;
#pragma unroll
            for (int j = 0; j < NV4; ++j)
            {
                const int e = e0 + 16 * j + 4 * h;
                const bool ok = ok1 && e >= 0 && e < rowLen;
                st.ok4 |= (ok ? 1u : 0u) << (s * NV4 + j);
                st.a4[s][j] = *reinterpret_cast<const f32x4 *>(ok ? rowp + e : p.zero);
            }
            if (RPL > 0)
            {
                const int e = e0 + 16 * NV4 + RPL * h;
                const bool ok = ok1 && e >= 0 && e < rowLen;
                st.okr |= (ok ? 1u : 0u) << s;
                const float *src = ok ? rowp + e : p.zero;
#pragma unroll
                for (int c = 0; c < RPL; ++c)
                    st.ar[s][c] = src[c];
            }
        }
#pragma unroll
        for (int r = 0; r < NRES; ++r)
            st.resv[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI == EPI_GLU || EPI == EPI_GN_GLU_SCALE_RES)
        {
#pragma unroll
            for (int r = 0; r < NRES; ++r)
            {
                const int c = r * 16 + 4 * h;
                if (rowOk && c < C2)
                {
                    if (EPI == EPI_GN_GLU_SCALE_RES) // res may alias Y: rows are private to one lane, read before written
                        st.resv[r] = *reinterpret_cast<const f32x4 *>(p.res + (i64)m * p.ldy + c);
                    else if (p.table)
                        st.resv[r] = *reinterpret_cast<const f32x4 *>(p.table + (i64)p0 * C2 + c);
                }
            }
        }
    };
    auto compute = [&](const Stage &st) {
        const float aMean = st.aMean, aScale = st.aScale;
        // ---- prologue + MFMAs
        f32x4 acc[NF];
#pragma unroll
        for (int fj = 0; fj < NF; ++fj)
            acc[fj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S1; ++s)
        {
#pragma unroll
            for (int j = 0; j < NV4; ++j)
            {
                f32x4 v = st.a4[s][j];
                if (PRO == PRO_AFFINE)
                {
                    v.x = (v.x - aMean) * aScale, v.y = (v.y - aMean) * aScale;
                    v.z = (v.z - aMean) * aScale, v.w = (v.w - aMean) * aScale;
                }
                if (PRO == PRO_GN_GELU)
                {
                    v.x = dgelu((v.x - aMean) * aScale * gW4[j].x + gB4[j].x);
                    v.y = dgelu((v.y - aMean) * aScale * gW4[j].y + gB4[j].y);
                    v.z = dgelu((v.z - aMean) * aScale * gW4[j].z + gB4[j].z);
                    v.w = dgelu((v.w - aMean) * aScale * gW4[j].w + gB4[j].w);
                }
                if (PRO != PRO_NONE && !((st.ok4 >> (s * NV4 + j)) & 1u))
                    v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int fj = 0; fj < NF; ++fj)
                        acc[fj] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(Bv[fj][s][j], c), v[c], acc[fj], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < RPL; ++c)
            {
                float v = st.ar[s][c];
                if (PRO == PRO_AFFINE)
                    v = (v - aMean) * aScale;
                if (PRO == PRO_GN_GELU)
                    v = dgelu((v - aMean) * aScale * gWr[c] + gBr[c]);
                if (PRO != PRO_NONE && !((st.okr >> s) & 1u))
                    v = 0.f;
#pragma unroll
                for (int fj = 0; fj < NF; ++fj)
                    acc[fj] = __builtin_amdgcn_mfma_f32_16x16x4f32(Br[fj][s][c], v, acc[fj], 0, 0, 0);
            }
        }
        // ---- epilogue (C^T layout): this lane's row m, channels n = 16 fj + 4h + {0,1,2,3}
        const i64 em = st.m;
        const bool eOk = st.rowOk;
        const int p0 = st.p0, p1 = st.p1;
        const unsigned b = st.b;
        float s = 0.f, ss = 0.f;
        if (EPI == EPI_LINEAR || EPI == EPI_STATS_ONLY || EPI == EPI_STATS_FACT)
        {
            // no residual operand here (launch_dgemm refuses shapes that carry one): a load issued in the
            // epilogue would have to be waited for with vmcnt(0), i.e. behind the whole prefetched stage
#pragma unroll
            for (int fj = 0; fj < NF; ++fj)
            {
                const int n = fj * 16 + 4 * h;
                if (eOk && n < p.N)
                {
                    const float4 bi = biasv[CST_LDS ? 0 : fj];
                    float4 v = make_float4(acc[fj][0] + bi.x, acc[fj][1] + bi.y, acc[fj][2] + bi.z, acc[fj][3] + bi.w);
                    if (EPI == EPI_LINEAR)
                    {
                        if (p.act)
                            v = make_float4(dgelu(v.x), dgelu(v.y), dgelu(v.z), dgelu(v.w));
                        *reinterpret_cast<float4 *>(p.Y + em * p.ldy + n) = v;
                    }
                    if (EPI == EPI_STATS_FACT)
                    {
                        const int hid = p.Cout; // factorised statistics, see plan.h
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
            if (p.rowstat)
            {
                s += __shfl_xor(s, 16);
                ss += __shfl_xor(ss, 16);
                s += __shfl_xor(s, 32);
                ss += __shfl_xor(ss, 32);
                if (h == 0 && eOk)
                    *reinterpret_cast<float2 *>(p.rowstat + em * 2) = make_float2(s, ss); // NB == 1
            }
        }
        else if (EPI == EPI_GLU || EPI == EPI_GN_GLU_SCALE_RES)
        {
            if constexpr (NF % 2 == 0)
            {
                const float mean = st.eMean, sc = st.eSc;
#pragma unroll
                for (int fj = 0; fj < NF; fj += 2)
                {
                    const int c = (fj >> 1) * 16 + 4 * h;
                    if (eOk && c < C2)
                    {
                        const float4 ba = CST_LDS ? cst[0][fj][h] : biasv[CST_LDS ? 0 : fj];
                        const float4 bg = CST_LDS ? cst[0][fj + 1][h] : biasv[CST_LDS ? 0 : fj + 1];
                        float av[4] = {acc[fj][0] + ba.x, acc[fj][1] + ba.y, acc[fj][2] + ba.z, acc[fj][3] + ba.w};
                        float gv[4] = {acc[fj + 1][0] + bg.x, acc[fj + 1][1] + bg.y, acc[fj + 1][2] + bg.z, acc[fj + 1][3] + bg.w};
                        const f32x4 r4 = st.resv[fj >> 1];
                        const float rv[4] = {r4[0], r4[1], r4[2], r4[3]};
                        float ov[4];
                        if (EPI == EPI_GN_GLU_SCALE_RES)
                        {
                            const float4 gw4 = cst[CST_LDS ? 1 : 0][fj][h], gb4 = cst[CST_LDS ? 2 : 0][fj][h];
                            const float4 hw4 = cst[CST_LDS ? 1 : 0][fj + 1][h], hb4 = cst[CST_LDS ? 2 : 0][fj + 1][h];
                            const float4 sv4 = cst[CST_LDS ? 3 : 0][fj][h];
                            const float gw[4] = {gw4.x, gw4.y, gw4.z, gw4.w};
                            const float gb[4] = {gb4.x, gb4.y, gb4.z, gb4.w};
                            const float hw[4] = {hw4.x, hw4.y, hw4.z, hw4.w};
                            const float hb[4] = {hb4.x, hb4.y, hb4.z, hb4.w};
                            const float sv[4] = {sv4.x, sv4.y, sv4.z, sv4.w};
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                            {
                                const float a = (av[r] - mean) * sc * gw[r] + gb[r];
                                const float g = (gv[r] - mean) * sc * hw[r] + hb[r];
                                ov[r] = rv[r] + sv[r] * (a * dsigmoid(g));
                            }
                        }
                        else
                        {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                ov[r] = av[r] * dsigmoid(gv[r]) + p.tableScale * rv[r];
                        }
                        *reinterpret_cast<float4 *>(p.Y + em * p.ldy + c) = make_float4(ov[0], ov[1], ov[2], ov[3]);
                    }
                }
            }
        }
        else // EPI_TRCONV
        {
            i64 offs[NF];
#pragma unroll
            for (int fj = 0; fj < NF; ++fj)
            {
                const int n = fj * 16 + 4 * h;
                const int jj = p.trS * p0 + trR[fj] - p.trOff;
                offs[fj] = (eOk && n < p.N && jj >= 0 && jj < p.Lout) ? (i64)b * p.yBS + ((i64)p1 * p.Lout + jj) * p.ldy + trC[fj] : -1;
            }
#pragma unroll
            for (int fj = 0; fj < NF; ++fj)
                if (offs[fj] >= 0)
                {
                    const float4 bi = biasv[CST_LDS ? 0 : fj];
                    float4 v = make_float4(acc[fj][0] + bi.x, acc[fj][1] + bi.y, acc[fj][2] + bi.z, acc[fj][3] + bi.w);
                    if (p.act)
                        v = make_float4(dgelu(v.x), dgelu(v.y), dgelu(v.z), dgelu(v.w));
                    *reinterpret_cast<float4 *>(p.Y + offs[fj]) = v;
                }
        }
    };

    // Fragment pipeline. PIPE 0: one hand-over `cur = nxt` (the register allocator coalesces both stages
    // and sinks the prefetch behind the last MFMA that reads it: smallest footprint, most waves per SIMD);
    // PIPE 1: two explicit stages, loop unrolled by two, scheduler free; PIPE 2: same with the prefetch
    // pinned ahead of the other stage's MFMAs. Chosen per kernel shape from measurements (launch table).
    if constexpr (PIPE == 0)
    {
        Stage cur, nxt;
        fetch(gwave, cur);
        for (int frag = gwave; frag < nfrag; frag += nwaves)
        {
            fetch(frag + nwaves, nxt); // beyond the end: every load goes to the zero page
            compute(cur);
            cur = nxt;
        }
    }
    else if constexpr (PIPE == 3)
    {
        // two fragments in flight per wave: these kernels are bound by the bytes a CU keeps in flight (HBM latency
        // under load ~3 us x 6 TB/s = 74 KB per CU), not by issue - a third stage costs registers (fewer waves) but
        // each wave then covers twice the latency
        Stage s0, s1, s2;
        fetch(gwave, s0);
        fetch(gwave + nwaves, s1);
        for (int frag = gwave; frag < nfrag; frag += 3 * nwaves)
        {
            fetch(frag + 2 * nwaves, s2);
            compute(s0);
            fetch(frag + 3 * nwaves, s0);
            compute(s1);
            fetch(frag + 4 * nwaves, s1);
            compute(s2);
        }
    }
    else
    {
        Stage sa, sb;
        fetch(gwave, sa);
        for (int frag = gwave; frag < nfrag; frag += 2 * nwaves)
        {
            fetch(frag + nwaves, sb); // beyond the end: every load goes to the zero page, nothing is stored
            if (PIPE == 2)
                __builtin_amdgcn_sched_barrier(0);
            compute(sa);
            if (PIPE == 2)
                __builtin_amdgcn_sched_barrier(0);
            fetch(frag + 2 * nwaves, sa);
            if (PIPE == 2)
                __builtin_amdgcn_sched_barrier(0);
            compute(sb);
            if (PIPE == 2)
                __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// --------------------------------------------------------------------------- DConv K1: ONE read per input row
// K1 = Conv1d(C -> C/8, k3, dilation d) along TIME. The three taps of an output row are the input rows at t-d, t, t+d,
// so the generic kernel above fetches every input row three times through L1 / the texture addresser, which is what
// bounds it (profiles/DESIGN_history_r1-r4.md: TA busy 76-84 %, 2.5-3.6 TB/s of algorithmic bytes). Here every LANE ROW of a wave
// walks the time axis and keeps the fragments of rows t-d .. t+d in a register ring (2d+2 slots: the window plus one
// row in flight); each step loads only row t+d+1:
//   frequency branch [B][T][F][C]: the 16 lane rows are 16 consecutive bins, all at the same t (a chunk of the walk
//                                  re-reads 2d halo rows);
//   time branch      [B][L][C]:    the 16 lane rows are 16 consecutive sub-chunks of `len` time steps each, lane row r
//                                  walks [t0 + r len, t0 + (r+1) len) (its halo rows are its neighbours' first / last rows).
// MFMA order (tap, k, sub-step) and the epilogue are those of dgemm_kernel: every output bit is unchanged, whatever
// the chunk length (chosen per launch for occupancy).
template <int SEG0, int DIL, bool TIME>
__global__ __launch_bounds__(256) void dgemm_k1_ring_kernel(const GemmArgs p, const int len, const int nChunks, const int nFb)
{
    constexpr int NV4 = SEG0 / 16;
    constexpr int RING = 2 * DIL + 2;
    const int lane = threadIdx.x & 63, l15 = lane & 15, h = lane >> 4;
    const int gwave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int T = p.P1, F = p.P0; // TIME: F == 1
    const i64 rowLen = (i64)F * SEG0;

    f32x4 Bv[3][NV4]; // weights, MFMA A operand (operands swapped: C^T), lane = column n = l15
    {
        const bool nOk = l15 < p.Np;
        const float *w = p.Wt + (i64)(nOk ? l15 : 0) * p.Kp;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < NV4; ++j)
                Bv[s][j] = *reinterpret_cast<const f32x4 *>(nOk ? w + s * SEG0 + 16 * j + 4 * h : p.zero);
    }
    const int n = 4 * h;
    const float4 biasv = ld4z(p.bias + n, n < p.N, p.zero);

    const int ntask = p.B * nChunks * nFb;
    for (int task = gwave; task < ntask; task += nwaves)
    {
        const int fb = task % nFb, ck = (task / nFb) % nChunks, b = task / (nFb * nChunks);
        const int f = TIME ? 0 : fb * 16 + l15;
        const bool fOk = f < F;
        // this lane row walks [tl0, tl1); the loop below runs `len` steps for the whole wave
        const int tl0 = TIME ? (ck * 16 + l15) * len : ck * len;
        const int tl1 = min(T, tl0 + len);
        const float *xb = p.X + (i64)b * p.xBS + (i64)(fOk ? f : 0) * SEG0 + 4 * h;
        f32x4 ring[RING][NV4];
        auto load_row = [&](int slot, int t) {
            const bool ok = fOk && t >= 0 && t < T && t < tl1 + DIL; // rows past the walk's halo are never used
            const float *src = ok ? xb + (i64)t * rowLen : p.zero;
#pragma unroll
            for (int j = 0; j < NV4; ++j)
                ring[slot][j] = *reinterpret_cast<const f32x4 *>(ok ? src + 16 * j : src);
        };
#pragma unroll
        for (int i = 0; i <= 2 * DIL; ++i)
            load_row(i, tl0 - DIL + i);
        const int steps = TIME ? len : min(T, tl0 + len) - tl0; // wave-uniform
        for (int ub = 0; ub < steps; ub += RING)
        {
#pragma unroll
            for (int u = 0; u < RING; ++u)
            {
                if (ub + u < steps) // wave-uniform
                {
                    const int t = tl0 + ub + u;
                    load_row((u + 2 * DIL + 1) % RING, t + DIL + 1);
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 3; ++s)
#pragma unroll
                        for (int j = 0; j < NV4; ++j)
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Bv[s][j][c], ring[(u + s * DIL) % RING][j][c], acc, 0, 0, 0);
                    const bool rOk = fOk && t < tl1;
                    const i64 em = ((i64)b * T + t) * F + f;
                    float sm = 0.f, ss = 0.f;
                    if (rOk && n < p.N)
                    {
                        float4 v = make_float4(acc[0] + biasv.x, acc[1] + biasv.y, acc[2] + biasv.z, acc[3] + biasv.w);
                        if (p.act)
                            v = make_float4(dgelu(v.x), dgelu(v.y), dgelu(v.z), dgelu(v.w));
                        *reinterpret_cast<float4 *>(p.Y + em * p.ldy + n) = v;
                        sm += (v.x + v.y) + (v.z + v.w);
                        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                    }
                    if (p.rowstat)
                    {
                        sm += __shfl_xor(sm, 16);
                        ss += __shfl_xor(ss, 16);
                        sm += __shfl_xor(sm, 32);
                        ss += __shfl_xor(ss, 32);
                        if (h == 0 && rOk)
                            *reinterpret_cast<float2 *>(p.rowstat + em * 2) = make_float2(sm, ss); // NB == 1
                    }
                }
            }
        }
    }
}

template <int SEG0>
static void launch_k1_ring(const GemmArgs &a, hipStream_t s)
{
    const bool time = a.P0 == 1;
    const int nFb = time ? 1 : (a.P0 + 15) / 16;
    // length of a lane row's walk: long enough to amortise the 2d halo rows, short enough for >= ~4 waves per SIMD
    const i64 target = 4096;
    const i64 lanesRows = time ? (i64)a.B * a.P1 / 16 : (i64)a.B * nFb * a.P1; // walk steps summed over all waves
    int len = (int)((lanesRows + target - 1) / target);
    len = len < 8 ? 8 : (len > 56 ? 56 : len);
    const int nChunks = time ? (a.P1 + 16 * len - 1) / (16 * len) : (a.P1 + len - 1) / len;
    const i64 ntask = (i64)a.B * nChunks * nFb;
    int blocks = (int)((ntask + 3) / 4);
    if (blocks > 256 * 8)
        blocks = 256 * 8;
#define DMX_K1(D, TM) hipLaunchKernelGGL((dgemm_k1_ring_kernel<SEG0, D, TM>), dim3(blocks), dim3(256), 0, s, a, len, nChunks, nFb)
    if (time && a.dil1 == 1)
        DMX_K1(1, true);
    else if (time)
        DMX_K1(2, true);
    else if (a.dil1 == 1)
        DMX_K1(1, false);
    else
        DMX_K1(2, false);
#undef DMX_K1
}
// K1 shapes the ring kernels cover: taps along axis 1; >= 16 bins per time step (frequency branch) or one (time branch)
static bool k1_ring_ok(const GemmArgs &a)
{
    const char *e = getenv("DMX_K1_RING"); // read per call: A/B runs inside one process (tools/gpu_k1ring.sh); 0 = off, 2 = frequency branch only, 3 = time branch always
    const int on = e ? atoi(e) : 1;
    // time branch: measured at 42 / 1 segments per call, C = 96 (L = 21499): 153 -> 110 us / 15.6 -> 17.5 us; C = 48 (L = 85995):
    // 223 -> 257 us / 12.7 -> 11.5 us (a lane row's 32-byte output pieces are scattered `len` rows apart, and short walks
    // re-read up to half of their rows as halo): only the C = 96 shape, and only when the walk is >= 12 steps long
    const bool timeOk = a.P0 == 1 && on != 2 && (on == 3 || (a.seg0 == 96 && (i64)a.B * a.P1 >= 12ll * 16 * 4096));
    return on && a.pro == PRO_NONE && a.epi == EPI_LINEAR && a.S1 == 3 && a.N <= 16 && (a.P0 >= 16 || timeOk) && a.L0 == a.P0 &&
           a.L1 == a.P1 && a.stride0 == 1 && a.stride1 == 1 && a.pad0 == 0 && a.pad1 == a.dil1 && (a.dil1 == 1 || a.dil1 == 2) &&
           a.seg0 == a.Cin && a.NB == 1 && !a.res;
}

template <int NF, int S1, int SEG0, int PRO, int EPI, int PIPE_>
static void launch_d(const GemmArgs &a, hipStream_t s, int chunks = 1)
{
    const int nfrag = (int)((a.M + 15) >> 4);
    int blocks = (nfrag + 3) / 4;
    if (blocks > 256 * 8 / chunks)
        blocks = 256 * 8 / chunks; // persistent: 8 workgroups per CU at most, waves stride over fragments
    constexpr int PIPE = PIPE_;
    hipLaunchKernelGGL((dgemm_kernel<NF, S1, SEG0, PRO, EPI, PIPE>), dim3(blocks, chunks), dim3(256), 0, s, a, make_fastdiv((unsigned)a.P0),
                       make_fastdiv((unsigned)a.P1));
}

// Direct kernel table. key = NF*1000000 + S1*100000 + SEG0*100 + PRO*10 + EPI. Returns 0, or -1 if
// the combination is not instantiated (the plan then keeps the LDS-tiled igemm).
int launch_dgemm(const GemmArgs &a, hipStream_t s, bool dry)
{
    const int NF = (a.N + 15) / 16;
    if (a.M >= (1ll << 31) - 16 || (i64)a.L0 * a.Cin >= (1ll << 31))
        return -1;
    if ((a.epi == EPI_LINEAR || a.epi == EPI_TRCONV) && a.res)
        return -1; // residual adds of these epilogues stay with the LDS-tiled kernel
#define DMX_D(NF_, S1_, SEG_, PRO_, EPI_, PIPE_)                              \
    case (NF_ * 1000000 + S1_ * 100000 + SEG_ * 100 + PRO_ * 10 + EPI_):      \
        if (!dry)                                                             \
            launch_d<NF_, S1_, SEG_, PRO_, EPI_, PIPE_>(a, s);                \
        return 0;
    if (k1_ring_ok(a) && (a.seg0 == 48 || a.seg0 == 96))
    {
        if (!dry)
            a.seg0 == 48 ? launch_k1_ring<48>(a, s) : launch_k1_ring<96>(a, s);
        return 0;
    }
    // DConv k3 of the deep levels (C = 192 / 384: hidden 24 / 48 -> 2C = 384 / 768 packed columns; layers.cpp:204-259): the
    // weight matrix no longer fits one wave's registers, so the op runs as COLUMN CHUNKS of 192 (12 fragments = 6 GLU pairs
    // = 96 output channels) through the 12-fragment kernel: every chunk streams all rows once (hidden row in, its 96
    // channels of the residual in and out) with its 192 x K weights resident, instead of a 128x128 GEMM tile whose single
    // half-empty K-tile made the op all prologue and epilogue (round 3: 1.7-2.3 TB/s). The chunks touch disjoint columns of
    // the in-place tensor; arithmetic per output = the direct kernels' (k ascending, remainder k-slots last).
    if (a.pro == PRO_GN_GELU && a.epi == EPI_GN_GLU_SCALE_RES && a.S1 == 1 && (a.seg0 == 24 || a.seg0 == 48) && a.N > 192 && a.N % 192 == 0)
    {
        if (!dry)
        {
            // chunk width: 192 columns keep 12 x K weights in registers (268 / 351 VGPRs: one wave per SIMD) and read the hidden
            // row N / 192 times; 96 columns (6 fragments, ~130 VGPRs: three to four waves per SIMD) read it N / 96 times.
            // 96 is the measured winner (profiles/DESIGN_history_r1-r4.md section 7.6)
            const int cw = 96;
            GemmArgs c = a; // ONE launch: grid row y = chunk y (the kernel offsets its column-indexed operands by y * N)
            c.N = c.Np = cw;
            const int chunks = a.N / cw;
            if (cw == 192)
                a.seg0 == 24 ? launch_d<12, 1, 24, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES, 1>(c, s, chunks) : launch_d<12, 1, 48, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES, 0>(c, s, chunks);
            else
                a.seg0 == 24 ? launch_d<6, 1, 24, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES, 0>(c, s, chunks) : launch_d<6, 1, 48, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES, 0>(c, s, chunks);
        }
        return 0;
    }
    switch (NF * 1000000 + a.S1 * 100000 + a.seg0 * 100 + a.pro * 10 + a.epi)
    {
        // DConv k1: Conv1d(C -> C/8, k3): C = 48, 96 (time branch; the frequency branch takes the ring kernel above)
        DMX_D(1, 3, 48, PRO_NONE, EPI_LINEAR, 0)
        DMX_D(1, 3, 96, PRO_NONE, EPI_LINEAR, 1)
        // DConv k2 / k3: hidden 8 (C=48) / 12 (C=96) -> 2C, statistics / final
        DMX_D(6, 1, 8, PRO_GN_GELU, EPI_STATS_ONLY, 2)
        DMX_D(1, 1, 8, PRO_GN_GELU, EPI_STATS_FACT, 0)
        DMX_D(1, 1, 12, PRO_GN_GELU, EPI_STATS_FACT, 0)
        DMX_D(6, 1, 8, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES, 0)
        DMX_D(12, 1, 12, PRO_GN_GELU, EPI_STATS_ONLY, 0)
        DMX_D(12, 1, 12, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES, 1)
        // Demucs v3 DConv (hidden C/4): k3 for hidden 12 (C=48) / 24 (C=96), factorised statistics for hidden 24
        DMX_D(6, 1, 12, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES, 0)
        DMX_D(12, 1, 24, PRO_GN_GELU, EPI_GN_GLU_SCALE_RES, 1)
        DMX_D(2, 1, 24, PRO_GN_GELU, EPI_STATS_FACT, 0)
        // first encoder convs (z-norm prologue) k8 s4: Cin = 4 (freq), 2 (time)
        DMX_D(3, 1, 32, PRO_AFFINE, EPI_LINEAR, 0)
        DMX_D(3, 1, 16, PRO_AFFINE, EPI_LINEAR, 0)
        // level-0 rewrites 48 -> 96 + GLU
        DMX_D(6, 1, 48, PRO_NONE, EPI_GLU, 0)
        // last transposed convs 48 -> 4*Cout: Cout = 16 / 8 (4 sources), 24 / 12 (6 sources)
        DMX_D(4, 1, 96, PRO_NONE, EPI_TRCONV, 0)
        DMX_D(2, 1, 96, PRO_NONE, EPI_TRCONV, 0)
        DMX_D(6, 1, 96, PRO_NONE, EPI_TRCONV, 0)
        DMX_D(3, 1, 96, PRO_NONE, EPI_TRCONV, 0)
    default:
        return -1;
    }
#undef DMX_D
}

} // namespace dmx

// v3.hip — kernels of the Demucs v3 (hdemucs_mmi) levels 4 / 5 for gfx950 (CDNA4): GroupNorm with channel
// groups, the bidirectional LSTM recurrence, and the LocalState attention core. Semantics: plan.h (OP_GROUP_STATS,
// OP_GN_ACT, OP_LSTM, OP_LOCAL_ATTN) and, executable, tests/cpu_interp.cpp. Reference:
//   /root/reference/src/layers.hpp:125-225 (generalized_group_norm), src/lstm.cpp:68-147 (lstm_forward),
//   src/layers.cpp:533-721 (local_attention).
#include "kernels.h"
#include <algorithm>
#include <string>

namespace dmx
{

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ float v3_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }
// the LSTM's gates sit on the critical path of 2016 sequential steps: v_exp_f32 / v_rcp_f32 forms (1 ulp each; the
// absolute error of the tanh form is < 2e-7, the parity tolerance 1e-4)
__device__ __forceinline__ float lstm_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float lstm_tanh(float v) { return 2.0f * lstm_sigmoid(2.0f * v) - 1.0f; }

// --------------------------------------------------------------------------- GroupNorm statistics
// Two launches: kStatChunks workgroups per (group, batch element) each reduce a fixed row range to (sum, sum of
// squares) in DOUBLE (one pass: with fp64 accumulators the subtraction n*mean^2 costs nothing for |mean| / sigma
// up to 1e6), then one wave per (group, batch element) adds the chunk partials in index order. The record depends
// on nothing but the group's data and the fixed chunking (rows per chunk = ceil(rows / kStatChunks)).
static const int kStatChunks = 32;
__global__ __launch_bounds__(256) void group_stats_partial_kernel(const GroupStatsArgs p)
{
    const int ch = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const int gs = p.C / p.G, gs4 = gs >> 2;
    const int rpc = (p.rows + kStatChunks - 1) / kStatChunks;
    const int r0 = ch * rpc, r1 = min(p.rows, r0 + rpc);
    const float *x = p.x + (i64)b * p.rows * p.C + (i64)g * gs;
    double s = 0, q = 0;
    const i64 n4 = (i64)max(r1 - r0, 0) * gs4;
    for (i64 i = threadIdx.x; i < n4; i += 256)
    {
        const i64 r = r0 + i / gs4;
        const int c4 = (int)(i % gs4);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + r * p.C + 4 * c4);
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            s += (double)v[k];
            q += (double)v[k] * (double)v[k];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        s += __shfl_xor(s, off);
        q += __shfl_xor(q, off);
    }
    __shared__ double red[4][2];
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6][0] = s, red[threadIdx.x >> 6][1] = q;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double *o = p.partials + (((i64)b * p.G + g) * kStatChunks + ch) * 2;
        o[0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        o[1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}
__global__ __launch_bounds__(64) void group_stats_final_kernel(const GroupStatsArgs p)
{
    const int bg = blockIdx.x; // b * G + g
    const double *pp = p.partials + (i64)bg * kStatChunks * 2;
    if (threadIdx.x != 0)
        return;
    double s = 0, q = 0;
    for (int ch = 0; ch < kStatChunks; ++ch)
        s += pp[2 * ch], q += pp[2 * ch + 1];
    const double cnt = (double)p.rows * (p.C / p.G);
    const double mean = s / cnt;
    double var = (q - cnt * mean * mean) / (cnt - 1.0); // unbiased (Q3)
    if (var < 0)
        var = 0;
    float *o = p.out + (i64)bg * 4;
    o[0] = (float)mean;
    o[1] = (float)(1.0 / sqrt(var + (double)p.eps));
    o[2] = (float)sqrt(var);
    o[3] = 0.f;
}

void launch_group_stats(const GroupStatsArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(group_stats_partial_kernel, dim3(kStatChunks, a.G, a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(group_stats_final_kernel, dim3(a.G * a.B), dim3(64), 0, s, a);
}

// --------------------------------------------------------------------------- GroupNorm apply (+GELU | +GLU), crop, scale, residual
template <int MODE>
__global__ __launch_bounds__(256) void gn_act_kernel(const GnActArgs p)
{
    const int Co = MODE == 2 ? p.C / 2 : p.C, Co4 = Co >> 2, gs = p.C / p.G;
    const i64 total = (i64)p.B * p.rowsOut * Co4;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < total; i += (i64)gridDim.x * 256)
    {
        const int c = 4 * (int)(i % Co4);
        const i64 br = i / Co4;
        const int r = (int)(br % p.rowsOut), b = (int)(br / p.rowsOut);
        const float *x = p.x + ((i64)b * p.rowsIn + r + p.rowOff) * p.C;
        auto gn4 = [&](int cc) -> f32x4 {
            const float *st = p.stats + ((i64)b * p.G + cc / gs) * 4; // 4 consecutive channels share a group (gs % 4 == 0)
            const float mean = st[0], rs = st[1];
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + cc);
            const f32x4 w = *reinterpret_cast<const f32x4 *>(p.w + cc), bb = *reinterpret_cast<const f32x4 *>(p.b + cc);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = (v[k] - mean) * rs * w[k] + bb[k];
            return o;
        };
        f32x4 v = gn4(c);
        if (MODE == 1)
        {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = dmx_gelu(v[k]);
        }
        if (MODE == 2)
        {
            const f32x4 g = gn4(c + Co);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = v[k] * v3_sigmoid(g[k]);
        }
        if (p.scale)
        {
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(p.scale + c);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] *= sc[k];
        }
        const i64 o = ((i64)b * p.rowsOut + r) * Co + c;
        if (p.res)
        {
            const f32x4 rr = *reinterpret_cast<const f32x4 *>(p.res + o);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] += rr[k];
        }
        *reinterpret_cast<f32x4 *>(p.y + o) = v;
    }
}

void launch_gn_act(const GnActArgs &a, hipStream_t s)
{
    const int Co = a.mode == 2 ? a.C / 2 : a.C;
    const i64 total = (i64)a.B * a.rowsOut * (Co / 4);
    const unsigned blocks = (unsigned)std::min<i64>((total + 255) / 256, 256 * 16);
    if (a.mode == 0)
        hipLaunchKernelGGL(gn_act_kernel<0>, dim3(blocks), dim3(256), 0, s, a);
    else if (a.mode == 1)
        hipLaunchKernelGGL(gn_act_kernel<1>, dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(gn_act_kernel<2>, dim3(blocks), dim3(256), 0, s, a);
}

// --------------------------------------------------------------------------- bidirectional LSTM layer (recurrent part)
// The recurrence h_t = f(W_hh h_{t-1} + xproj_t) is sequential in t (336 / 168 steps) and W_hh of one direction
// (590 KB at H = 192, 2.36 MB at H = 384, fp32) fits neither one CU's registers nor its LDS, so one recurrence =
// one (direction, group of 16 batch columns) is run by P cooperating workgroups of NW = 4 (or 8) waves:
//   * a wave owns FR fragments of 16 gate rows = 4 hidden units x (i, f, g, o); its slice of W_hh stays in registers
//     for all steps as MFMA A operands (v_mfma_f32_16x16x4_f32: gate rows x batch columns, k = hidden units);
//     the accumulator starts from the input projection, so a lane ends a step holding the four gates of ONE
//     (unit, batch column): the cell update is lane-local and c never leaves its register;
//   * h_t is exchanged as 8-byte {tag = step + 1, value} granules written with ONE agent-scope (sc1) store each and
//     polled with agent-scope loads: the data is the flag, no fence, no counter, and the result does not depend on
//     where the workgroups run (cdna_hip_programming.md, guideline 16, form R2). Two granule buffers alternate by
//     step parity: a producer can run at most one step ahead of its slowest consumer, because step t+1 needs every
//     workgroup's h_t. Every workgroup stages the polled h into LDS ([column][unit], padded so that both the
//     scattered staging writes and the ds_read_b128 operand reads are conflict-free) and feeds its MFMAs from there;
//   * the workgroups of one recurrence are placed on one XCD (block b runs on XCD b % 8: speed only).
// The granule area is zeroed by the launcher before every launch (tags of an earlier launch must never match).
// Spins are bounded: on a time-out the kernel raises `status` and carries on with whatever it read.
template <int H, int FR, int NW>
__global__ __launch_bounds__(64 * NW) void lstm_kernel(const LstmArgs p)
{
    constexpr int UW = 4 * FR;   // hidden units per wave
    constexpr int UWG = NW * UW; // per workgroup
    constexpr int P = H / UWG;   // workgroups per recurrence
    constexpr int HP = H + 4;    // LDS row pitch (floats): 16 columns x ds_read_b128 hit 64 distinct banks
    constexpr int NJ = H / 16;   // float4 k-steps per fragment
    constexpr int NPOLL = H / (4 * NW); // granule loads per lane and step (a wave sweeps H/NW units x 16 columns)
    static_assert(H % UWG == 0, "units split evenly");
    __shared__ float hs[2][16][HP];

    const int nGroups = 2 * ((p.B + 15) / 16);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int g = xcd + 8 * (j / P), slot = j % P;
    if (g >= nGroups)
        return;
    const int dir = g & 1, bg = g >> 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int bcol = bg * 16 + l15;
    const bool valid = bcol < p.B;
    const int ub = slot * UWG + wave * UW; // first hidden unit of this wave

    // ---- recurrent weights -> registers, MFMA A operand: lane (row = l15, k-slot kq) holds W[row][16 jj + 4 kq + c]
    f32x4 wreg[FR][NJ];
#pragma unroll
    for (int f = 0; f < FR; ++f)
    {
        const float *wrow = p.whh + ((i64)dir * 4 * H + 4 * (ub + 4 * f) + l15) * H + 4 * kq;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
            wreg[f][jj] = *reinterpret_cast<const f32x4 *>(wrow + 16 * jj);
    }
    gu64 *gran = (gu64 *)p.gran + (i64)g * 2 * H * 16;
    float cst[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f)
        cst[f] = 0.f;

    for (int step = 0; step < p.T; ++step)
    {
        const int t = dir ? p.T - 1 - step : step;
        // input projection of this step: the four gates of (unit ub + 4 f + kq, column bcol)
        f32x4 acc[FR];
#pragma unroll
        for (int f = 0; f < FR; ++f)
        {
            acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (valid)
                acc[f] = *reinterpret_cast<const f32x4 *>(p.xproj + ((i64)bcol * p.T + t) * 8 * H + (i64)dir * 4 * H + 4 * (ub + 4 * f + kq));
        }
        if (step > 0)
        {
            // ---- poll h_{step-1}: this wave sweeps units [wave H/NW, +H/NW) of all 16 columns until every tag == step
            const gu64 *src = gran + (i64)((step - 1) & 1) * H * 16 + (i64)(wave * (H / NW)) * 16 + lane;
            float hv[NPOLL];
            unsigned spins = 0;
            for (;;)
            {
                bool ok = true;
#pragma unroll
                for (int n = 0; n < NPOLL; ++n)
                {
                    const unsigned long long x = __hip_atomic_load(src + n * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hv[n] = __uint_as_float((unsigned)x);
                    ok &= (unsigned)(x >> 32) == (unsigned)step;
                }
                if (__all(ok))
                    break;
                if (++spins > (1u << 24)) // ~ seconds: a partner never arrived
                {
                    if (lane == 0)
                        __hip_atomic_store((gu32 *)p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            float(*hb)[HP] = hs[step & 1];
#pragma unroll
            for (int n = 0; n < NPOLL; ++n)
                hb[l15][wave * (H / NW) + 4 * n + kq] = hv[n]; // granule (unit, column) -> hs[column][unit]
            __syncthreads();
            // ---- gates += W_hh h: B operand lane (column l15, k-slot kq) holds h[16 jj + 4 kq + c]. Two accumulator
            // chains per fragment (even / odd k sub-step): the 16x16x4 f32 MFMA issues every 32 cycles but a dependent one
            // waits 40; the second chain starts from zero and is added at the end (fixed order).
            f32x4 acc2[FR];
#pragma unroll
            for (int f = 0; f < FR; ++f)
                acc2[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
            {
                const f32x4 hq = *reinterpret_cast<const f32x4 *>(&hb[l15][16 * jj + 4 * kq]);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int f = 0; f < FR; ++f)
                    {
                        if ((c & 1) == 0)
                            acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[f][jj][c], hq[c], acc[f], 0, 0, 0);
                        else
                            acc2[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[f][jj][c], hq[c], acc2[f], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int f = 0; f < FR; ++f)
                acc[f] += acc2[f];
        }
        // ---- cell update (lstm.cpp:109-126), lane-local; publish h
        gu64 *dst = gran + (i64)(step & 1) * H * 16;
#pragma unroll
        for (int f = 0; f < FR; ++f)
        {
            const float ig = lstm_sigmoid(acc[f][0]), fg = lstm_sigmoid(acc[f][1]), gg = lstm_tanh(acc[f][2]), og = lstm_sigmoid(acc[f][3]);
            const float cn = fg * cst[f] + ig * gg;
            cst[f] = cn;
            const float h = og * lstm_tanh(cn);
            const int unit = ub + 4 * f + kq;
            if (step + 1 < p.T)
                __hip_atomic_store(dst + (i64)unit * 16 + l15, ((unsigned long long)(unsigned)(step + 1) << 32) | __float_as_uint(h),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (valid)
                p.out[((i64)bcol * p.T + t) * 2 * H + (i64)dir * H + unit] = h;
        }
    }
}

// EXPERIMENT (DMX_LSTM_XCHG=x4): 4-byte values in a sentinel-filled image X[group][t][unit][column], written once per
// (t, unit, column): no tags, no re-arming. The launcher fills the image with 0xFFFFFFFF (a NaN no arithmetic produces), a
// value is published by ONE agent-scope 4-byte store, the partners poll 8 bytes = two columns of a unit at a time (a
// dword is either the sentinel or final, so tearing is harmless). Half the bytes and half the loads of the granule form.
template <int H, int FR, int NW>
__global__ __launch_bounds__(64 * NW) void lstm_x4_kernel(const LstmArgs p)
{
    constexpr int UW = 4 * FR, UWG = NW * UW, P = H / UWG, HP = H + 4, NJ = H / 16;
    constexpr int UQ = H / NW, NPOLL = UQ / 8;
    static_assert(H % UWG == 0 && UQ % 8 == 0, "whole sweeps");
    __shared__ float hs[2][16][HP];
    constexpr unsigned SENT = 0xFFFFFFFFu;
    const int nGroups = 2 * ((p.B + 15) / 16);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int g = xcd + 8 * (j / P), slot = j % P;
    if (g >= nGroups)
        return;
    const int dir = g & 1, bg = g >> 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int bcol = bg * 16 + l15;
    const bool valid = bcol < p.B;
    const int ub = slot * UWG + wave * UW;
    f32x4 wreg[FR][NJ];
#pragma unroll
    for (int f = 0; f < FR; ++f)
    {
        const float *wrow = p.whh + ((i64)dir * 4 * H + 4 * (ub + 4 * f) + l15) * H + 4 * kq;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
            wreg[f][jj] = *reinterpret_cast<const f32x4 *>(wrow + 16 * jj);
    }
    gu32 *X = (gu32 *)p.gran + (i64)g * p.T * H * 16; // [t][unit][column]
    // poll positions of this lane: unit = wave UQ + 8 n + lane / 8, columns 2 cp, 2 cp + 1. Lanes whose columns lie beyond
    // the batch re-read pair 0 (column 0 always exists) - the loads stay unconditional - and keep zeros
    const int cp = lane & 7, pu0 = wave * UQ + (lane >> 3);
    const int nValid = min(16, p.B - bg * 16);
    const bool v0 = 2 * cp < nValid, v1 = 2 * cp + 1 < nValid;
    const bool chkHi = v0 ? v1 : nValid > 1; // does the high dword of the pair this lane reads get published?
    const int cpr = v0 ? cp : 0;
    float cst[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f)
        cst[f] = 0.f;
    for (int step = 0; step < p.T; ++step)
    {
        const int t = dir ? p.T - 1 - step : step;
        f32x4 acc[FR];
#pragma unroll
        for (int f = 0; f < FR; ++f)
        {
            acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (valid)
                acc[f] = *reinterpret_cast<const f32x4 *>(p.xproj + ((i64)bcol * p.T + t) * 8 * H + (i64)dir * 4 * H + 4 * (ub + 4 * f + kq));
        }
        if (step > 0)
        {
            const gu64 *src = (const gu64 *)(X + ((i64)(step - 1) * H + pu0) * 16 + 2 * cpr);
            unsigned long long hv[NPOLL];
            unsigned spins = 0;
            for (;;)
            {
                bool ok = true;
#pragma unroll
                for (int n = 0; n < NPOLL; ++n)
                {
                    hv[n] = __hip_atomic_load(src + n * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // 8 units further
                    ok &= (unsigned)hv[n] != SENT && (!chkHi || (unsigned)(hv[n] >> 32) != SENT);
                }
                if (__all(ok))
                    break;
                if (++spins > (1u << 24))
                {
                    if (lane == 0)
                        __hip_atomic_store((gu32 *)p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            float(*hb)[HP] = hs[step & 1];
#pragma unroll
            for (int n = 0; n < NPOLL; ++n)
            {
                hb[2 * cp][pu0 + 8 * n] = v0 ? __uint_as_float((unsigned)hv[n]) : 0.f;
                hb[2 * cp + 1][pu0 + 8 * n] = v1 ? __uint_as_float((unsigned)(hv[n] >> 32)) : 0.f;
            }
            __syncthreads();
            f32x4 acc2[FR];
#pragma unroll
            for (int f = 0; f < FR; ++f)
                acc2[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
            {
                const f32x4 hq = *reinterpret_cast<const f32x4 *>(&hb[l15][16 * jj + 4 * kq]);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int f = 0; f < FR; ++f)
                    {
                        if ((c & 1) == 0)
                            acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[f][jj][c], hq[c], acc[f], 0, 0, 0);
                        else
                            acc2[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[f][jj][c], hq[c], acc2[f], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int f = 0; f < FR; ++f)
                acc[f] += acc2[f];
        }
#pragma unroll
        for (int f = 0; f < FR; ++f)
        {
            const float ig = lstm_sigmoid(acc[f][0]), fg = lstm_sigmoid(acc[f][1]), gg = lstm_tanh(acc[f][2]), og = lstm_sigmoid(acc[f][3]);
            const float cn = fg * cst[f] + ig * gg;
            cst[f] = cn;
            const float h = og * lstm_tanh(cn);
            const int unit = ub + 4 * f + kq;
            if (valid)
            {
                if (step + 1 < p.T)
                    __hip_atomic_store(X + ((i64)step * H + unit) * 16 + l15, __float_as_uint(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                p.out[((i64)bcol * p.T + t) * 2 * H + (i64)dir * H + unit] = h;
            }
        }
    }
}

// The recurrence kernels spin on partner workgroups, so every workgroup of a launch must be able to be resident at once:
// the grid is checked against (occupancy of this instantiation) x (compute units) of the current device, minus a margin
// of one workgroup per CU (the occupancy API can be one block per CU high, MI355X_MICROARCH.md "Correctness boundaries").
template <typename K>
static bool lstm_grid_fits(K kernel, unsigned blocks, int threads)
{
    int dev = 0, perCu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kernel, threads, 0) != hipSuccess)
        return false;
    const long cap = (long)std::max(perCu - 1, 1) * prop.multiProcessorCount;
    return (long)blocks <= cap;
}
template <int H, int FR, int NW>
static int launch_lstm_t(const LstmArgs &a, hipStream_t s, bool x4)
{
    constexpr int P = H / (4 * FR * NW);
    const int nGroups = 2 * ((a.B + 15) / 16);
    const unsigned blocks = 8u * P * (unsigned)((nGroups + 7) / 8);
    static int fitsUpTo[2] = {0, 0}; // largest grid already checked, per kernel form (the check costs a runtime query)
    if ((int)blocks > fitsUpTo[x4 ? 1 : 0])
    {
        const bool ok = x4 ? lstm_grid_fits(lstm_x4_kernel<H, FR, NW>, blocks, 64 * NW) : lstm_grid_fits(lstm_kernel<H, FR, NW>, blocks, 64 * NW);
        if (!ok)
            return -2;
        fitsUpTo[x4 ? 1 : 0] = (int)blocks;
    }
    if (x4)
        hipLaunchKernelGGL((lstm_x4_kernel<H, FR, NW>), dim3(blocks), dim3(64 * NW), 0, s, a);
    else
        hipLaunchKernelGGL((lstm_kernel<H, FR, NW>), dim3(blocks), dim3(64 * NW), 0, s, a);
    return 0;
}

int launch_lstm(const LstmArgs &a, hipStream_t s)
{
    static const bool x4 = getenv("DMX_LSTM_XCHG") && std::string(getenv("DMX_LSTM_XCHG")) == "x4"; // experiment
    if (x4)
    {
        const size_t nGroups = 2 * (size_t)((a.B + 15) / 16);
        if (hipMemsetAsync(a.gran, 0xFF, nGroups * a.T * a.H * 16 * sizeof(float), s) != hipSuccess)
            return -1;
    }
    // tags of an earlier launch must never satisfy a poll of this one
    else if (hipMemsetAsync(a.gran, 0, (size_t)lstm_sync_floats(a.B, a.H) * sizeof(float), s) != hipSuccess)
        return -1;
    // workgroup shape: 4 waves = ONE wave per SIMD, i.e. the 32-cycle fp32 MFMAs of
    // a step are not shared with a second wave (the recurrence is latency-bound: a step cannot start before the
    // previous one's h has crossed the chip); the price is twice the workgroups polling the same granules
    constexpr int nw = 4;
    if (a.H == 192)
        return nw == 8 ? launch_lstm_t<192, 1, 8>(a, s, x4) : launch_lstm_t<192, 1, 4>(a, s, x4);
    if (a.H == 384)
        return nw == 8 ? launch_lstm_t<384, 1, 8>(a, s, x4) : launch_lstm_t<384, 1, 4>(a, s, x4);
    return -1;
}

// --------------------------------------------------------------------------- LocalState attention core
// One workgroup = 16 queries of one (batch element, head): 4 waves x 4 queries. Keys and content of the head are
// staged in LDS once per workgroup when they fit (T x head-dim x 2 x 4 B <= 144 KB: both production shapes,
// 336 x 48 and 168 x 96), else read from global memory (L2). A wave scores its query against all keys (lane = key),
// adds the decay penalty -(n+1) |t-s| / 2 * sigmoid(d_n) / 2, masks the diagonal with -100, normalises over the keys
// with the reference's max / exp / sum / divide sequence (layers.cpp:652-679), then accumulates the weighted content
// (lane = channel). Accumulation orders are the reference's (c ascending, t ascending).
template <int HD, bool STAGE>
__global__ __launch_bounds__(256) void local_attn_kernel(const LocalAttnArgs p)
{
    extern __shared__ float smem[];
    const int T = p.T, H = p.H;
    const int h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int KP = HD + 4; // LDS row pitch (floats)
    float *wbuf = smem + (size_t)wave * T; // [4][T] softmax weights of the wave's current query
    float *ks = smem + (size_t)4 * T;      // [T][KP] keys, [T][KP] content (STAGE)
    float *vs = ks + (size_t)T * KP;
    const float *base = p.qkvd + (i64)b * T * p.ld;
    if (STAGE)
    {
        for (int i = tid; i < T * (HD / 4); i += 256)
        {
            const int t = i / (HD / 4), c4 = i - t * (HD / 4);
            *reinterpret_cast<f32x4 *>(ks + (size_t)t * KP + 4 * c4) = *reinterpret_cast<const f32x4 *>(base + (i64)t * p.ld + H + h * HD + 4 * c4);
            *reinterpret_cast<f32x4 *>(vs + (size_t)t * KP + 4 * c4) = *reinterpret_cast<const f32x4 *>(base + (i64)t * p.ld + 2 * H + h * HD + 4 * c4);
        }
        __syncthreads();
    }
    const int kvs = STAGE ? KP : p.ld;
    // key row t / content element (t, c): LDS image or global memory (two typed paths: no flat addressing)
    auto krow4 = [&](int t, int c) -> f32x4 {
        if constexpr (STAGE)
            return *reinterpret_cast<const f32x4 *>(ks + (size_t)t * KP + c);
        else
            return *reinterpret_cast<const f32x4 *>(base + (i64)t * p.ld + H + h * HD + c);
    };
    auto vat = [&](int t, int c) -> float {
        if constexpr (STAGE)
            return vs[(size_t)t * KP + c];
        else
            return base[(i64)t * p.ld + 2 * H + h * HD + c];
    };
    (void)kvs;
    const float rsq = sqrtf((float)HD);
    for (int qi = 0; qi < 4; ++qi)
    {
        const int s = blockIdx.x * 16 + wave * 4 + qi;
        if (s >= T)
            break; // wave-uniform
        const float *qrow = base + (i64)s * p.ld;
        float q[HD];
#pragma unroll
        for (int c = 0; c < HD; c += 4)
        {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(qrow + h * HD + c);
            q[c] = v[0], q[c + 1] = v[1], q[c + 2] = v[2], q[c + 3] = v[3];
        }
        float dq[4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
            dq[n] = 0.5f * v3_sigmoid(qrow[3 * H + h * 4 + n]);
        float mx = -INFINITY;
        for (int t = lane; t < T; t += 64)
        {
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < HD; c += 4)
            {
                const f32x4 kv = krow4(t, c);
                dot += q[c] * kv[0];
                dot += q[c + 1] * kv[1];
                dot += q[c + 2] * kv[2];
                dot += q[c + 3] * kv[3];
            }
            const float delta = fabsf((float)(t - s));
            float decay = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n)
                decay += (-(float)(n + 1) * delta / 2.0f) * dq[n];
            const float v = t != s ? dot / rsq + decay : -100.0f;
            wbuf[t] = v;
            mx = fmaxf(mx, v);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            mx = fmaxf(mx, __shfl_xor(mx, off));
        float sum = 0.f;
        for (int t = lane; t < T; t += 64)
        {
            const float e = expf(wbuf[t] - mx);
            wbuf[t] = e;
            sum += e;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            sum += __shfl_xor(sum, off);
        for (int t = lane; t < T; t += 64)
            wbuf[t] = wbuf[t] / sum;
        // wbuf is private to the wave; its lanes exchange through LDS: wait for the writes above
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < HD; c += 64)
        {
            float acc = 0.f;
            for (int t = 0; t < T; ++t)
                acc += wbuf[t] * vat(t, c);
            p.out[((i64)b * T + s) * H + h * HD + c] = acc;
        }
        __builtin_amdgcn_wave_barrier(); // next query overwrites wbuf
    }
}

int launch_local_attn(const LocalAttnArgs &a, hipStream_t s)
{
    const int hd = a.H / 4;
    const size_t stageBytes = ((size_t)4 * a.T + (size_t)2 * a.T * (hd + 4)) * sizeof(float);
    const int stage = stageBytes <= 144 * 1024 ? 1 : 0;
    const size_t smem = stage ? stageBytes : (size_t)4 * a.T * sizeof(float);
    if (smem > 160 * 1024)
        return -1;
    const dim3 grid((a.T + 15) / 16, 4, a.B);
    // (the attribute is per device: set on every launch, it is a cheap host-side call)
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a);
    };
    if (hd == 48)
        stage ? go(local_attn_kernel<48, true>) : go(local_attn_kernel<48, false>);
    else if (hd == 96)
        stage ? go(local_attn_kernel<96, true>) : go(local_attn_kernel<96, false>);
    else
        return -1;
    return 0;
}

} // namespace dmx

// misc.hip — normalisation / statistics / track-level kernels for gfx950 (HBM-bound,
// wave64 shuffle reductions, fixed summation order => bit-reproducible results).
//
// Reference semantics (/root/reference):
//   layer_norm      src/layers.cpp:121-150   (UNBIASED variance, Q3: layers.hpp:76-95)
//   group_norm      src/layers.cpp:9-49      (1 group; statistics come from the producing
//                                             GEMM's per-row partials, reduced here in fp64)
//   z-norm          src/model_inference.cpp:115-124,138-144 : (x-mean)/(std+1e-5)
//   track apply     src/model_apply.cpp:60-288
#include "kernels.h"

namespace dmx
{

// --------------------------------------------------------------------------- stats reduce
// rowstat [B][R][NB][2] -> out [B*G][4] = {mean, scale, std, 0}; G = max(G0,1).
// grid (ceil(G/GX), B); block 256 = GX groups x SL slices.
template <int GX>
__global__ __launch_bounds__(256) void stats_reduce_kernel(const ReduceArgs p)
{
    constexpr int SL = 256 / GX;
    __shared__ double sh[SL][GX][2];
    const int G = p.G0 > 1 ? p.G0 : 1;
    const int gx = threadIdx.x % GX, sl = threadIdx.x / GX;
    const int gi = blockIdx.x * GX + gx, b = blockIdx.y;
    double s = 0.0, q = 0.0;
    if (gi < G)
    {
        const int rowsPerGroup = (p.R - gi + G - 1) / G;
        for (int i = sl; i < rowsPerGroup; i += SL)
        {
            const int r = gi + i * G;
            const float2 *src = reinterpret_cast<const float2 *>(p.rowstat) + ((i64)b * p.R + r) * p.NB;
            for (int nb = 0; nb < p.NB; ++nb)
            {
                const float2 v = src[nb];
                s += (double)v.x;
                q += (double)v.y;
            }
        }
    }
    sh[sl][gx][0] = s;
    sh[sl][gx][1] = q;
    __syncthreads();
    if (sl == 0 && gi < G)
    {
        double S = 0.0, Q = 0.0;
        for (int i = 0; i < SL; ++i)
        {
            S += sh[i][gx][0];
            Q += sh[i][gx][1];
        }
        const double mean = S / p.count;
        double var = (Q - p.count * mean * mean) / (p.count - 1.0);
        if (var < 0.0)
            var = 0.0;
        float4 o;
        o.x = (float)mean;
        o.z = (float)sqrt(var);
        o.y = p.mode == MODE_RSTD ? (float)(1.0 / sqrt(var + (double)p.eps)) : (float)(1.0 / (sqrt(var) + (double)p.eps));
        o.w = 0.f;
        reinterpret_cast<float4 *>(p.out)[(i64)b * G + gi] = o;
    }
}

// G0 <= 1 (one group per batch element, up to 86k rows): two stages.
// stage 1: grid (nchunk, B); each workgroup sums a contiguous chunk of the [R*NB] float2 entries
__global__ __launch_bounds__(256) void stats_partial_kernel(const ReduceArgs p)
{
    __shared__ double red[4][2];
    const int b = blockIdx.y;
    const i64 entries = (i64)p.R * p.NB;
    const i64 per = (entries + p.nchunk - 1) / p.nchunk;
    const i64 lo = (i64)blockIdx.x * per, hi = lo + per < entries ? lo + per : entries;
    const float2 *src = reinterpret_cast<const float2 *>(p.rowstat) + (i64)b * entries;
    double s = 0.0, q = 0.0;
    for (i64 i = lo + threadIdx.x; i < hi; i += 256)
    {
        const float2 v = src[i];
        s += (double)v.x;
        q += (double)v.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        s += __shfl_xor(s, off);
        q += __shfl_xor(q, off);
    }
    if ((threadIdx.x & 63) == 0)
    {
        red[threadIdx.x >> 6][0] = s;
        red[threadIdx.x >> 6][1] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double *dst = p.scratch + ((i64)b * p.nchunk + blockIdx.x) * 2;
        dst[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        dst[1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    }
}
// stage 2: one wave per batch element sums the nchunk partials in index order
__global__ __launch_bounds__(64) void stats_final_kernel(const ReduceArgs p)
{
    const int b = blockIdx.x;
    if (threadIdx.x != 0)
        return;
    double S = 0.0, Q = 0.0;
    for (int i = 0; i < p.nchunk; ++i)
    {
        S += p.scratch[((i64)b * p.nchunk + i) * 2];
        Q += p.scratch[((i64)b * p.nchunk + i) * 2 + 1];
    }
    const double mean = S / p.count;
    double var = (Q - p.count * mean * mean) / (p.count - 1.0);
    if (var < 0.0)
        var = 0.0;
    float4 o;
    o.x = (float)mean;
    o.z = (float)sqrt(var);
    o.y = p.mode == MODE_RSTD ? (float)(1.0 / sqrt(var + (double)p.eps)) : (float)(1.0 / (sqrt(var) + (double)p.eps));
    o.w = 0.f;
    reinterpret_cast<float4 *>(p.out)[b] = o;
}

void launch_stats_reduce(const ReduceArgs &a, hipStream_t s)
{
    const int G = a.G0 > 1 ? a.G0 : 1;
    if (G >= 32)
        hipLaunchKernelGGL(stats_reduce_kernel<32>, dim3((G + 31) / 32, a.B), dim3(256), 0, s, a);
    else if (G > 1)
        hipLaunchKernelGGL(stats_reduce_kernel<8>, dim3((G + 7) / 8, a.B), dim3(256), 0, s, a);
    else
    {
        hipLaunchKernelGGL(stats_partial_kernel, dim3(a.nchunk, a.B), dim3(256), 0, s, a);
        hipLaunchKernelGGL(stats_final_kernel, dim3(a.B), dim3(64), 0, s, a);
    }
}

// --------------------------------------------------------------------------- LayerNorm
// one wave64 per row; two-pass (mean, then sum of squared deviations) in registers.
__global__ __launch_bounds__(256) void layernorm_kernel(const LnArgs p)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows)
        return;
    const float *x = p.x + (i64)row * p.D;
    float v[8]; // D <= 512
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
        const int c = lane + i * 64;
        v[i] = c < p.D ? x[c] : 0.f;
        s += v[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_xor(s, off);
    const float mean = s / (float)p.D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
        const int c = lane + i * 64;
        const float d = c < p.D ? v[i] - mean : 0.f;
        q += d * d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        q += __shfl_xor(q, off);
    const float rstd = 1.0f / sqrtf(q / (float)(p.D - 1) + p.eps);
    float *y = p.y + (i64)row * p.D;
    const float *pe = p.pe ? p.pe + (i64)(row % p.rowsPerBatch) * p.D : nullptr;
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
        const int c = lane + i * 64;
        if (c < p.D)
        {
            float o = (v[i] - mean) * rstd * p.w[c] + p.b[c];
            if (pe)
                o += pe[c];
            y[c] = o;
        }
    }
}

void launch_layernorm(const LnArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(layernorm_kernel, dim3((a.rows + 3) / 4), dim3(256), 0, s, a);
}

// --------------------------------------------------------------------------- GroupNorm apply / add
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnArgs p)
{
    const int b = blockIdx.y;
    const i64 per = (i64)p.rows * p.C; // multiple of 4
    float mean = 0.f, sc = 1.f;
    if (p.stats)
    {
        mean = p.stats[b * 4];
        sc = p.stats[b * 4 + 1];
    }
    const float4 *x = reinterpret_cast<const float4 *>(p.x + (i64)b * per);
    const float4 *res = p.res ? reinterpret_cast<const float4 *>(p.res + (i64)b * per) : nullptr;
    float4 *y = reinterpret_cast<float4 *>(p.y + (i64)b * per);
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < per / 4; i += (i64)gridDim.x * 256)
    {
        float4 v = x[i];
        if (p.stats)
        {
            const int c = (int)((i * 4) % p.C);
            const float4 w = *reinterpret_cast<const float4 *>(p.w + c);
            const float4 bb = *reinterpret_cast<const float4 *>(p.b + c);
            v.x = (v.x - mean) * sc * w.x + bb.x;
            v.y = (v.y - mean) * sc * w.y + bb.y;
            v.z = (v.z - mean) * sc * w.z + bb.z;
            v.w = (v.w - mean) * sc * w.w + bb.w;
        }
        if (res)
        {
            const float4 r = res[i];
            v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
        }
        y[i] = v;
    }
}

void launch_gn_apply(const GnArgs &a, hipStream_t s)
{
    i64 n4 = (i64)a.rows * a.C / 4;
    int gx = (int)((n4 + 255) / 256);
    if (gx > 2048)
        gx = 2048;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gx, a.B), dim3(256), 0, s, a);
}

// --------------------------------------------------------------------------- track level
// ref = mean over channels; mean / unbiased std of ref (model_apply.cpp:72-82)
__global__ __launch_bounds__(256) void track_stats_kernel(const float *audio, i64 n, double *partials)
{
    __shared__ double red[4][2];
    const float2 *a = reinterpret_cast<const float2 *>(audio);
    double s = 0.0, q = 0.0;
    const i64 per = (n + gridDim.x - 1) / gridDim.x;
    const i64 lo = (i64)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (i64 i = lo + threadIdx.x; i < hi; i += 256)
    {
        const float2 v = a[i];
        const float r = (v.x + v.y) / 2.0f;
        s += (double)r;
        q += (double)r * (double)r;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        s += __shfl_xor(s, off);
        q += __shfl_xor(q, off);
    }
    if ((threadIdx.x & 63) == 0)
    {
        red[threadIdx.x >> 6][0] = s;
        red[threadIdx.x >> 6][1] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        partials[blockIdx.x * 2] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        partials[blockIdx.x * 2 + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    }
}
__global__ void track_stats_final_kernel(const double *partials, int nblk, i64 n, float *stats)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
    {
        double s = 0.0, q = 0.0;
        for (int i = 0; i < nblk; ++i)
        {
            s += partials[2 * i];
            q += partials[2 * i + 1];
        }
        const double mean = s / (double)n;
        double var = (q - (double)n * mean * mean) / (double)(n - 1);
        if (var < 0.0)
            var = 0.0;
        stats[0] = (float)mean;
        stats[1] = (float)sqrt(var);
    }
}
void launch_track_stats(const float *audio, i64 n, double *partials, int nblk, hipStream_t s)
{
    hipLaunchKernelGGL(track_stats_kernel, dim3(nblk), dim3(256), 0, s, audio, n, partials);
}
void launch_track_stats_final(const double *partials, int nblk, i64 n, float *stats, hipStream_t s)
{
    hipLaunchKernelGGL(track_stats_final_kernel, dim3(1), dim3(64), 0, s, partials, nblk, n, stats);
}

// shift + zero pad + normalise + chunk + centre (model_apply.cpp:21-43,93-138,189-194,250-262).
// The segment indices travel by value in the kernel arguments: no device index buffer, no host
// synchronisation between consecutive gathers.
__global__ __launch_bounds__(256) void track_gather_kernel(const float *audio, i64 n, const float *stats, int shiftOffset,
                                                           i64 seg, i64 stride, i64 len, TrackSegIdx segIdx, float *mixes)
{
    const int which = blockIdx.y;
    const i64 off = (i64)segIdx.v[which] * stride;
    const i64 chunk = seg < len - off ? seg : len - off;
    const i64 left = (seg - chunk) / 2; // floor(total_padding / 2)
    const float mean = stats[0], stdv = stats[1];
    const float2 *a = reinterpret_cast<const float2 *>(audio);
    float2 *dst = reinterpret_cast<float2 *>(mixes) + (i64)which * seg;
    const i64 maxShift = 22050;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < seg; i += (i64)gridDim.x * 256)
    {
        float2 v = make_float2(0.f, 0.f);
        const i64 k = i - left;
        if (k >= 0 && k < chunk)
        {
            const i64 src = off + k + shiftOffset - maxShift; // index into the un-padded track
            if (src >= 0 && src < n)
            {
                const float2 x = a[src];
                v = make_float2((x.x - mean) / stdv, (x.y - mean) / stdv);
            }
        }
        dst[i] = v;
    }
}
void launch_track_gather(const float *audio, i64 n, const float *stats, int shiftOffset, i64 seg, i64 stride, i64 len,
                         const int *segIdx, int nIdx, float *mixes, hipStream_t s)
{
    int gx = (int)((seg + 255) / 256);
    if (gx > 512)
        gx = 512;
    for (int i0 = 0; i0 < nIdx; i0 += TrackSegIdx::kMax)
    {
        TrackSegIdx t{};
        const int nb = nIdx - i0 < TrackSegIdx::kMax ? nIdx - i0 : TrackSegIdx::kMax;
        for (int i = 0; i < nb; ++i)
            t.v[i] = segIdx[i0 + i];
        hipLaunchKernelGGL(track_gather_kernel, dim3(gx, nb), dim3(256), 0, s, audio, n, stats, shiftOffset, seg, stride, len, t,
                           mixes + (i64)i0 * seg * 2);
    }
}

// weighted overlap-add in segment order, /sum_weight, trim, de-normalise
// (model_apply.cpp:171-179,207-246,129-135,88). Each output sample is covered by at most
// ceil(seg/stride) = 2 segments; they are accumulated in increasing segment index like the
// reference loop, so the result does not depend on how segments were sharded over GPUs.
// planes [planeBase, planeBase + gridDim.y) (plane = s*2 + ch) and samples [i0, i1) of the output are
// produced: the bag takes stem m from model m, and a track can be finished (and copied out) in pieces
// as soon as the segments covering a piece are done.
__global__ __launch_bounds__(256) void track_ola_kernel(const float *segOut, int nSeg, int S, i64 seg, i64 stride, i64 len,
                                                        i64 n, int shiftOffset, const float *stats, float *out, int layout,
                                                        int planeBase, i64 i0, i64 i1, int gBase)
{
    const int plane = blockIdx.y + planeBase;
    const float mean = stats[0], stdv = stats[1];
    const i64 maxShift = 22050;
    const float half = (float)(seg / 2);
    for (i64 i = i0 + (i64)blockIdx.x * 256 + threadIdx.x; i < i1; i += (i64)gridDim.x * 256)
    {
        const i64 j = i + maxShift - shiftOffset; // position in the shifted track
        float acc = 0.f, sw = 0.f;
        i64 first = (j - seg + stride) / stride; // smallest g with g*stride + seg > j
        if (j - seg + 1 <= 0)
            first = 0;
        for (i64 g = first; g < nSeg && g * stride <= j; ++g)
        {
            const i64 off = g * stride;
            const i64 chunk = seg < len - off ? seg : len - off;
            const i64 k = j - off;
            if (k >= chunk)
                continue;
            const i64 left = (seg - chunk) / 2;
            // triangle weight, indexed from 0 even for short chunks (Q8)
            const i64 kk = k < seg / 2 ? k + 1 : seg - k;
            const float w = (float)kk / half;
            acc += w * segOut[((i64)(g - gBase) * S * 2 + plane) * seg + left + k]; // segOut[0] holds segment gBase
            sw += w;
        }
        const float v = (acc / sw) * stdv + mean;
        if (layout == 0)
            out[(i64)plane * n + i] = v;
        else
        {
            const int sIdx = plane >> 1, c = plane & 1;
            out[sIdx + (i64)S * (c + 2 * i)] = v;
        }
    }
}
void launch_track_ola(const float *segOut, int nSeg, int S, i64 seg, i64 stride, i64 len, i64 n, int shiftOffset,
                      const float *stats, float *out, int layout, int planeBase, int nPlanes, i64 i0, i64 i1, hipStream_t s, int gBase)
{
    if (i1 <= i0 || nPlanes <= 0)
        return;
    int gx = (int)((i1 - i0 + 255) / 256);
    if (gx > 4096)
        gx = 4096;
    hipLaunchKernelGGL(track_ola_kernel, dim3(gx, nPlanes), dim3(256), 0, s, segOut, nSeg, S, seg, stride, len, n, shiftOffset,
                       stats, out, layout, planeBase, i0, i1, gBase);
}

// rows x width floats between two pitched images (packing / unpacking the segment tails the OWNER finish mode exchanges)
__global__ __launch_bounds__(256) void copy_rows_kernel(float *dst, i64 dpitch, const float *src, i64 spitch, i64 width)
{
    const i64 r = blockIdx.y;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < width; i += (i64)gridDim.x * 256)
        dst[r * dpitch + i] = src[r * spitch + i];
}
void launch_copy_rows(float *dst, i64 dpitch, const float *src, i64 spitch, i64 width, int rows, hipStream_t s)
{
    if (width <= 0 || rows <= 0)
        return;
    int gx = (int)((width + 255) / 256);
    if (gx > 1024)
        gx = 1024;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(gx, rows), dim3(256), 0, s, dst, dpitch, src, spitch, width);
}

__global__ void planar_to_interleaved_kernel(const float *src, float *dst, i64 n)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
        reinterpret_cast<float2 *>(dst)[i] = make_float2(src[i], src[n + i]);
}
void launch_planar_to_interleaved(const float *src, float *dst, i64 n, hipStream_t s)
{
    int gx = (int)((n + 255) / 256);
    if (gx > 4096)
        gx = 4096;
    hipLaunchKernelGGL(planar_to_interleaved_kernel, dim3(gx), dim3(256), 0, s, src, dst, n);
}

__global__ void f32_to_f16_kernel(const float *src, unsigned short *dst, i64 n)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
        const _Float16 h = (_Float16)src[i];
        dst[i] = __builtin_bit_cast(unsigned short, h);
    }
}
void launch_f32_to_f16(const float *src, unsigned short *dst, i64 n, hipStream_t s)
{
    if (n > 0)
        hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
}

} // namespace dmx

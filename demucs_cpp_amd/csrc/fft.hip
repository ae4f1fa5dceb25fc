// fft.hip — STFT / ISTFT kernels for gfx950.
//
// Restates /root/reference/src/dsp.cpp:51-185 (stft/istft, nfft 4096, hop 1024,
// periodic Hann, 1/sqrt(N) scaling, unscaled inverse) fused with the surrounding glue
// of src/model_inference.cpp: symmetric padding :22-46 (Q2), frame slice [2, 2+le)
// :75-78, CaC packing :88-99 and its statistics :115-118 on the way in; de-normalise
// :380-393, CaC undo + zero Nyquist :409-444 on the way out.
//
// Both stereo channels of a frame share ONE 4096-point complex FFT (z = x_L + i x_R;
// the two half spectra are separated with the conjugate-symmetry identity), done as a
// 3-stage radix-16 Stockham autosort FFT in LDS by a 256-thread workgroup (one 16-point DFT per thread and stage). The forward transform
// runs one workgroup per frame; the inverse runs fused with the overlap-add, crop and time-branch
// sum (istft_ola_kernel: one workgroup per (batch, source, chunk of frames), the inverse frames stay
// in registers). istft_kernel + ola_kernel are the two-kernel form of the same arithmetic: the
// executable specification (CPU interpreter) and the DMX_FUSE_ISTFT=0 path.
#include "kernels.h"

namespace dmx
{

#define FFT_N 4096
#define FFT_LOG 12

// twiddle w_N^idx of the requested direction: tw[k] = exp(-2 pi i k / 4096), k < 2048; exponents in
// [2048, 4096) use w^(k) = -w^(k - 2048)
template <int SIGN>
__device__ __forceinline__ float2 fft_tw(const float2 *__restrict__ tw, int idx)
{
    float2 w = tw[idx & 2047];
    if (idx & 2048)
        w = make_float2(-w.x, -w.y);
    if (SIGN > 0)
        w.y = -w.y;
    return w;
}
__device__ __forceinline__ float2 cmul(const float2 a, const float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }

// LDS position of element i: a XOR swizzle of the low 4 bits with the next 4. Every access pattern of the three
// radix-16 stages below (reads x[j + 256 k] / 16-element blocks, writes y[16 p + k] / y[q + 256 p + 16 k] /
// y[j + 256 k]) then puts the 16 lanes of a ds_read/write_b64 group on 16 distinct bank pairs.
#define FSW(i) ((i) ^ (((i) >> 4) & 15))

// in place 4-point DFT X_d = sum_e w4^(e d) x_e, w4 = -i (forward) / +i (inverse)
template <int SIGN>
__device__ __forceinline__ void dft4(float2 &x0, float2 &x1, float2 &x2, float2 &x3)
{
    const float2 s02 = make_float2(x0.x + x2.x, x0.y + x2.y), d02 = make_float2(x0.x - x2.x, x0.y - x2.y);
    const float2 s13 = make_float2(x1.x + x3.x, x1.y + x3.y), d13 = make_float2(x1.x - x3.x, x1.y - x3.y);
    const float2 j13 = SIGN < 0 ? make_float2(d13.y, -d13.x) : make_float2(-d13.y, d13.x); // w4 * d13
    x0 = make_float2(s02.x + s13.x, s02.y + s13.y);
    x1 = make_float2(d02.x + j13.x, d02.y + j13.y);
    x2 = make_float2(s02.x - s13.x, s02.y - s13.y);
    x3 = make_float2(d02.x - j13.x, d02.y - j13.y);
}
template <int SIGN>
__device__ __forceinline__ float2 mulw16(const float2 v, const float c, const float sn) // v * (c + i SIGN sn)
{
    const float si = SIGN < 0 ? -sn : sn;
    return make_float2(v.x * c - v.y * si, v.x * si + v.y * c);
}

// In-LDS Stockham radix-16 FFT of FFT_N = 16^3 complex points: 3 autosort stages, every thread does ONE 16-point DFT per
// stage in registers (4 x 4 decomposition, constants w16^(c d)). Half the LDS traffic of the 6-stage radix-4 form it
// replaced. IN PLACE since round 5: a stage reads its 16 inputs into registers, all threads meet at a barrier, then the
// outputs go back into the same 32 KB image (the autosort permutation moves every element, hence the barrier between the
// reads and the writes; a second one publishes the stage). One image instead of two: 32 KB per transform, so five
// (STFT) / three (fused ISTFT, with its twiddle table) workgroups share a CU instead of two - the kernels are bound by
// the latency of their barrier-separated stages, which more resident workgroups hide (stft 0.46 -> see DESIGN.md, same bits).
// Input and result: a[FSW(i)]. sign = -1 forward (w = exp(-2 pi i p/n)), +1 inverse. tw[k] = exp(-2 pi i k / 4096), k < 2048.
// Stage with stride s (= 16^st), sub-transform length n = N/s, m = n/16, p < m, q < s:
//   y[q + s (16 p + k)] = w_n^(p k) * sum_j w16^(j k) x[q + s (p + j m)]
template <int SIGN>
__device__ __forceinline__ void fft4096(float2 *a, const float2 *__restrict__ tw, int tid)
{
    float2 *x = a, *y = a;
#pragma unroll 1
    for (int st = 0; st < FFT_LOG; st += 4)
    {
        const int s = 1 << st, m = 256 >> st;
        const int p = tid >> st, q = tid & (s - 1);
        float2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k)
        {
            const int i = q + s * (p + k * m);
            v[k] = x[FSW(i)];
        }
        // 16-point DFT: b_{c,d} = DFT4 over e of v[c + 4e] (in place -> v[c + 4d]); times w16^(c d); Y_{d + 4f} =
        // DFT4 over c (in place -> v[f + 4d])
#pragma unroll
        for (int c = 0; c < 4; ++c)
            dft4<SIGN>(v[c], v[c + 4], v[c + 8], v[c + 12]);
        v[1 + 4] = mulw16<SIGN>(v[1 + 4], 0.92387953251128674f, 0.38268343236508977f);   // w16^1
        v[2 + 4] = mulw16<SIGN>(v[2 + 4], 0.70710678118654752f, 0.70710678118654752f);   // w16^2
        v[3 + 4] = mulw16<SIGN>(v[3 + 4], 0.38268343236508977f, 0.92387953251128674f);   // w16^3
        v[1 + 8] = mulw16<SIGN>(v[1 + 8], 0.70710678118654752f, 0.70710678118654752f);   // w16^2
        v[2 + 8] = mulw16<SIGN>(v[2 + 8], 0.0f, 1.0f);                                   // w16^4
        v[3 + 8] = mulw16<SIGN>(v[3 + 8], -0.70710678118654752f, 0.70710678118654752f);  // w16^6
        v[1 + 12] = mulw16<SIGN>(v[1 + 12], 0.38268343236508977f, 0.92387953251128674f); // w16^3
        v[2 + 12] = mulw16<SIGN>(v[2 + 12], -0.70710678118654752f, 0.70710678118654752f); // w16^6
        v[3 + 12] = mulw16<SIGN>(v[3 + 12], -0.92387953251128674f, -0.38268343236508977f); // w16^9
#pragma unroll
        for (int d = 0; d < 4; ++d)
            dft4<SIGN>(v[4 * d], v[4 * d + 1], v[4 * d + 2], v[4 * d + 3]);
        const int o = q + s * 16 * p;
        __syncthreads(); // every thread holds its inputs: the image may be overwritten
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int f = 0; f < 4; ++f)
            {
                const int k = d + 4 * f;
                float2 r = v[f + 4 * d];
                if (st < 8 && k > 0) // the last stage has p = 0
                    r = cmul(r, fft_tw<SIGN>(tw, (p * k) << st));
                const int i = o + s * k;
                y[FSW(i)] = r;
            }
        __syncthreads();
    }
}

__device__ __forceinline__ double block_sum(double v, double *red, int tid)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off);
    __syncthreads();
    if ((tid & 63) == 0)
        red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void stft_kernel(const StftArgs p)
{
    __shared__ float2 bufA[FFT_N];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int t = blockIdx.x, b = blockIdx.y;
#ifdef DMX_FFT_LDS_PAD // diagnostic builds: a larger LDS footprint changes which workgroups can share the CU
    __shared__ char ldsPad[DMX_FFT_LDS_PAD];
    if (p.T < 0)
        ldsPad[tid] = 1;
#endif
    const float2 *tw = reinterpret_cast<const float2 *>(p.twiddle);
    const float2 *mix = reinterpret_cast<const float2 *>(p.mix) + (i64)b * p.seg;

    // windowed frame, both channels packed as one complex signal (re = L, im = R)
    for (int i = tid; i < FFT_N; i += 256)
    {
        i64 j = (i64)t * 1024 + i - p.pad;
        if (j < 0)
            j = -1 - j; // symmetric (edge-duplicating) reflection, Q2
        if (j >= p.seg)
            j = 2 * (i64)p.seg - 1 - j;
        const float2 v = mix[j];
        const float w = p.window[i];
        bufA[FSW(i)] = make_float2(v.x * w, v.y * w);
    }
    // raw-mix statistics of this hop (time-branch z-norm, model_inference.cpp:138-141)
    double sT = 0.0, qT = 0.0;
    for (int i = tid; i < 1024; i += 256)
    {
        i64 j = (i64)t * 1024 + i;
        if (j < p.seg)
        {
            const float2 v = mix[j];
            sT += (double)v.x + (double)v.y;
            qT += (double)v.x * v.x + (double)v.y * v.y;
        }
    }
    __syncthreads();
    fft4096<-1>(bufA, tw, tid);

    // split the two real spectra, scale by 1/sqrt(N), write CaC (re0, im0, re1, im1)
    float4 *out = reinterpret_cast<float4 *>(p.x) + ((i64)b * p.T + t) * 2048;
    double sF = 0.0, qF = 0.0;
    const float sc = 1.0f / 64.0f;
    for (int k = tid; k < 2048; k += 256)
    {
        const float2 zk = bufA[FSW(k)];
        const float2 zn = bufA[FSW((FFT_N - k) & (FFT_N - 1))];
        // X0 = (Z[k] + conj(Z[N-k]))/2 ; X1 = -i (Z[k] - conj(Z[N-k]))/2
        const float re0 = 0.5f * (zk.x + zn.x) * sc, im0 = 0.5f * (zk.y - zn.y) * sc;
        const float re1 = 0.5f * (zk.y + zn.y) * sc, im1 = -0.5f * (zk.x - zn.x) * sc;
        out[k] = make_float4(re0, im0, re1, im1);
        sF += (double)re0 + (double)im0 + (double)re1 + (double)im1;
        qF += (double)re0 * re0 + (double)im0 * im0 + (double)re1 * re1 + (double)im1 * im1;
    }
    sF = block_sum(sF, red, tid);
    qF = block_sum(qF, red, tid);
    sT = block_sum(sT, red, tid);
    qT = block_sum(qT, red, tid);
    if (tid == 0)
    {
        float *rs = p.rowstat + ((i64)b * p.T + t) * 2;
        rs[0] = (float)sF;
        rs[1] = (float)qF;
        float *rt = p.rowstatT + ((i64)b * p.T + t) * 2;
        rt[0] = (float)sT;
        rt[1] = (float)qT;
    }
}

void launch_stft(const StftArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(stft_kernel, dim3(a.T, a.B), dim3(256), 0, s, a);
}

// One workgroup per (frame t, source s, batch b): inverse transform of both channels.
__global__ __launch_bounds__(256) void istft_kernel(const IstftArgs p)
{
    __shared__ float2 bufA[FFT_N];
    const int tid = threadIdx.x;
    const int t = blockIdx.x, src = blockIdx.y, b = blockIdx.z;
    const float2 *tw = reinterpret_cast<const float2 *>(p.twiddle);
    const int CS = 4 * p.S;
    const float mean = p.stats[b * 4], stdv = p.stats[b * 4 + 2];
    const float *xin = p.x + (((i64)b * p.T + t) * 2048) * CS + src * 4;

    // Z[k] = X0[k] + i X1[k] with Hermitian extension; X*sqrt(N) (dsp.cpp:160-165);
    // DC imaginary parts ignored, Nyquist bin = 0 (model_inference.cpp:439-442)
    for (int k = tid; k < 2048; k += 256)
    {
        const float4 v = *reinterpret_cast<const float4 *>(xin + (i64)k * CS);
        float re0 = (stdv * v.x + mean) * 64.0f, im0 = (stdv * v.y + mean) * 64.0f;
        float re1 = (stdv * v.z + mean) * 64.0f, im1 = (stdv * v.w + mean) * 64.0f;
        if (k == 0)
        {
            im0 = 0.f;
            im1 = 0.f;
        }
        // X0 + i X1 = (re0 - im1) + i (im0 + re1)
        bufA[FSW(k)] = make_float2(re0 - im1, im0 + re1);
        if (k > 0) // conj(X0) + i conj(X1) = (re0 + im1) + i (re1 - im0)
            bufA[FSW(FFT_N - k)] = make_float2(re0 + im1, re1 - im0);
    }
    if (tid == 0)
        bufA[FSW(2048)] = make_float2(0.f, 0.f);
    __syncthreads();
    fft4096<+1>(bufA, tw, tid);
    float *f0 = p.frames + ((((i64)b * p.S + src) * 2 + 0) * p.T + t) * 4096;
    float *f1 = p.frames + ((((i64)b * p.S + src) * 2 + 1) * p.T + t) * 4096;
    for (int i = tid; i < FFT_N; i += 256)
    {
        const float2 z = bufA[FSW(i)];
        const float w = p.window[i];
        f0[i] = z.x * w;
        f1[i] = z.y * w;
    }
}

void launch_istft(const IstftArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(istft_kernel, dim3(a.T, a.S, a.B), dim3(256), 0, s, a);
}

// Overlap-add of the windowed inverse frames with window-sum-square normalisation
// (dsp.cpp:174-183), crop (dsp.cpp:113-116, model_inference.cpp:454-455) and sum with the
// de-normalised time branch (model_inference.cpp:396-405,460).
__global__ __launch_bounds__(256) void ola_kernel(const OlaArgs p)
{
    const int plane = blockIdx.y; // (b*S + src)*2 + ch
    const int ch = plane & 1, src = (plane >> 1) % p.S, b = (plane >> 1) / p.S;
    const float meanT = p.statsT[b * 4], stdT = p.statsT[b * 4 + 2];
    const float *fr = p.frames + (i64)plane * p.T * 4096;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < p.seg; i += gridDim.x * 256)
    {
        const int n = 2048 + p.pad + i;
        const int f1 = n >> 10;
        float acc = 0.f;
#pragma unroll
        for (int f = f1 - 3; f <= f1; ++f)
        {
            const int t = f - 2;
            if (t >= 0 && t < p.T)
            {
                const float yv = fr[(i64)t * 4096 + (n - f * 1024)];
                acc += yv * 1.0f / 4096.0f / (p.wss[n] + 1e-8f);
            }
        }
        const float tb = stdT * p.xt[((i64)b * p.seg + i) * (2 * p.S) + src * 2 + ch] + meanT;
        p.out[(i64)plane * p.seg + i] = acc + tb;
    }
}

void launch_ola(const OlaArgs &a, hipStream_t s)
{
    int gx = (a.seg + 255) / 256;
    if (gx > 1024)
        gx = 1024;
    hipLaunchKernelGGL(ola_kernel, dim3(gx, a.B * a.S * 2), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------------------
// ISTFT + overlap-add + crop + time-branch sum in ONE kernel: the inverse frames never reach HBM (the
// two-kernel form above writes and re-reads [B][S][2][T][4096] floats = 1.06 GB per 24-segment step).
//
// A workgroup owns (batch b, source src, chunk of `fpc` consecutive frames) and walks its frames in order.
// After the inverse FFT thread `tid` holds the samples i = tid + 256 j (j < 16) of the frame; sample i of
// frame f lands on position n = 1024 f + i of the padded signal, i.e. in hop H = f + (j >> 2) at offset
// tid + 256 (j & 3): always one of THIS thread's 16 accumulators, ring slot H & 3. A hop receives its last
// contribution from frame H itself, so after frame f has been added slot f & 3 is complete: it is written
// out (window-sum-square normalisation per term, crop, + de-normalised time branch: the summation order of
// istft_kernel + ola_kernel; the two divisions per term of that form - 32 IEEE divisions per thread and frame - are one
// multiplication by the precomputed (1 / 4096) / (wss + 1e-8), <= 1 ulp per term apart) and cleared. The ring rotates with f, so
// the step is instantiated for the 4 values of f & 3 (register indices must be compile-time constants).
// Chunks start 3 frames early (recomputed halo, no output) so that every hop they emit has all 4 addends.
//
// The four (six) sources of one (b, chunk) read the same 64-byte lines of x (16 B each): they are dealt to the
// SAME XCD, adjacent in dispatch order, so the lines come from HBM once and from that XCD's L2 afterwards.
template <int PH>
__device__ __forceinline__ void istft_ola_step(const IstftOlaArgs &p, float2 (&acc)[4][4], const float2 *bufA, int f, int tid, bool add,
                                               bool emit, int b, int src, float meanT, float stdT, const float (&wreg)[16], const float (&rdI)[4])
{
    if (add)
    {
        // (1 / 4096) / (wss + 1e-8): one table entry per position of the padded signal (plan.cpp). Where all four overlapping
        // frames exist - frames 3 .. T of the T + 4 - the window sum-square, summed in frame order, is the same float for
        // every position with the same offset in its hop: 4 registers per thread instead of 16 loads per frame (and the
        // window's 16 values are the thread's own in every frame). Only the first and the last frame read the table.
        const bool edge = f < 3 || f > p.T;
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            const int i = tid + 256 * j;
            const float2 z = bufA[FSW(i)]; // `bufA` here is the buffer that holds the transformed frame
            const float w = wreg[j];
            const float rd = edge ? p.rden[f * 1024 + i] : rdI[j & 3];
            const float y0 = z.x * w, y1 = z.y * w; // the value istft_kernel stores in `frames`
            float2 &a = acc[(PH + (j >> 2)) & 3][j & 3];
            a.x += y0 * rd; // the reference divides, y / 4096 / (wss + 1e-8) per term (dsp.cpp:151-185): <= 1 ulp apart
            a.y += y1 * rd;
        }
    }
    if (emit)
    {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
        {
            const int n = f * 1024 + tid + 256 * jj;
            const int i = n - (2048 + p.pad);
            if (i >= 0 && i < p.seg)
            {
                const float2 tb = *reinterpret_cast<const float2 *>(p.xt + ((i64)b * p.seg + i) * (2 * p.S) + src * 2);
                p.out[(((i64)b * p.S + src) * 2 + 0) * p.seg + i] = acc[PH][jj].x + (stdT * tb.x + meanT);
                p.out[(((i64)b * p.S + src) * 2 + 1) * p.seg + i] = acc[PH][jj].y + (stdT * tb.y + meanT);
            }
        }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
        acc[PH][jj] = make_float2(0.f, 0.f); // slot f & 3 now collects hop f + 4
}

// workgroups per CU the fused kernel is compiled for: 2 (210 registers) or 3 (168 registers; its 48 KB of LDS allow three)
#ifndef DMX_ISTFT_WGS
#define DMX_ISTFT_WGS 2
#endif
__global__ __launch_bounds__(256, DMX_ISTFT_WGS) void istft_ola_kernel(const IstftOlaArgs p)
{
    __shared__ float2 bufA[FFT_N];
    __shared__ float2 twS[2048]; // twiddles in LDS: a chunk of ~30 frames amortises the 16 KB; the one-frame-per-workgroup
                                 // kernels above take them from L1/L2 (3 loads per butterfly and stage in the dependency chain)
    const int tid = threadIdx.x;
#ifdef DMX_FFT_LDS_PAD
    __shared__ char ldsPad[DMX_FFT_LDS_PAD];
    if (p.T < 0)
        ldsPad[tid] = 1;
#endif
    // workgroup -> (group = (b, chunk), source): all sources of a group on one XCD (block id % 8)
    const unsigned xcd = blockIdx.x & 7u, jq = blockIdx.x >> 3;
    const int src = (int)(jq % (unsigned)p.S);
    const unsigned group = (jq / (unsigned)p.S) * 8u + xcd;
    if (group >= (unsigned)(p.B * p.nch))
        return;
    const int b = (int)(group / (unsigned)p.nch), ch = (int)(group - (unsigned)b * (unsigned)p.nch);
    for (int k = tid; k < 2048; k += 256)
        twS[k] = reinterpret_cast<const float2 *>(p.twiddle)[k];
    const int CS = 4 * p.S;
    const float mean = p.stats[b * 4], stdv = p.stats[b * 4 + 2];
    const float meanT = p.statsT[b * 4], stdT = p.statsT[b * 4 + 2];
    // model frames t in [t0, t1) are this chunk's; padded-signal frame index f = t + 2 (frames 0, 1, T+2, T+3
    // of the reference are zero, model_inference.cpp:439-444). The last chunk also emits the 3 hops behind
    // the last frame (their addends all exist by then).
    const int t0 = ch * p.fpc, t1 = min(p.T, t0 + p.fpc);
    const int fEnd = (ch == p.nch - 1) ? p.T + 2 + 3 : t1 + 2;
    float2 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            acc[a][c] = make_float2(0.f, 0.f);
    // the spectrum of the NEXT frame travels global -> registers while the current frame is transformed
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 xr[8];
    auto fetch = [&](int f) {
        const int t = min(f - 2, p.T - 1); // beyond the last frame: a harmless re-read
        const float *xin = p.x + (((i64)b * p.T + t) * 2048) * CS + src * 4;
#pragma unroll
        for (int m = 0; m < 8; ++m)
            xr[m] = *reinterpret_cast<const f32x4 *>(xin + (i64)(tid + 256 * m) * CS);
    };
    float wreg[16], rdI[4];
#pragma unroll
    for (int j = 0; j < 16; ++j)
        wreg[j] = p.window[tid + 256 * j];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        rdI[c] = p.rden[3072 + tid + 256 * c]; // frame 3, first hop: an interior position
    const int fBeg = max(2, t0 + 2 - 3);
    fetch(fBeg);
    __syncthreads(); // twS
    for (int f = fBeg; f < fEnd; ++f)
    {
        const int t = f - 2;
        const bool add = t < p.T;
        if (add)
        {
            // Z[k] = X0[k] + i X1[k] with Hermitian extension; X*sqrt(N) (dsp.cpp:160-165);
            // DC imaginary parts ignored, Nyquist bin = 0 (model_inference.cpp:439-442)
#pragma unroll
            for (int m = 0; m < 8; ++m)
            {
                const int k = tid + 256 * m;
                const f32x4 v = xr[m];
                float re0 = (stdv * v[0] + mean) * 64.0f, im0 = (stdv * v[1] + mean) * 64.0f;
                float re1 = (stdv * v[2] + mean) * 64.0f, im1 = (stdv * v[3] + mean) * 64.0f;
                if (k == 0)
                {
                    im0 = 0.f;
                    im1 = 0.f;
                }
                bufA[FSW(k)] = make_float2(re0 - im1, im0 + re1);
                if (k > 0)
                    bufA[FSW(FFT_N - k)] = make_float2(re0 + im1, re1 - im0);
            }
            if (tid == 0)
                bufA[FSW(2048)] = make_float2(0.f, 0.f);
            fetch(f + 1);
            __syncthreads();
            fft4096<+1>(bufA, twS, tid); // ends with a barrier: bufA holds the frame
        }
        const bool emit = t >= t0; // halo frames only build up the ring
        switch (f & 3)
        {
        case 0:
            istft_ola_step<0>(p, acc, bufA, f, tid, add, emit, b, src, meanT, stdT, wreg, rdI);
            break;
        case 1:
            istft_ola_step<1>(p, acc, bufA, f, tid, add, emit, b, src, meanT, stdT, wreg, rdI);
            break;
        case 2:
            istft_ola_step<2>(p, acc, bufA, f, tid, add, emit, b, src, meanT, stdT, wreg, rdI);
            break;
        default:
            istft_ola_step<3>(p, acc, bufA, f, tid, add, emit, b, src, meanT, stdT, wreg, rdI);
            break;
        }
        __syncthreads(); // the image is rewritten by the next frame's spectrum
    }
}

void launch_istft_ola(const IstftOlaArgs &a0, hipStream_t s)
{
    IstftOlaArgs a = a0;
    // Chunks per (batch item, source): every chunk recomputes a 3-frame halo and loads the twiddles (~2 frames' worth), so
    // no shorter than 8 frames - and the workgroups run in ROUNDS of DMX_ISTFT_WGS per CU, all the same length:
    // the launch lasts rounds x (frames per chunk + 5). The first form asked for "about four workgroups per CU" and got
    // 1184 workgroups = 2.3 rounds at 42 segments: the third round ran a third full. Take the chunk count with the
    // shortest launch (42 x 4 sources: 3 chunks = 504 workgroups, one round, 1.39 -> 1.05 ms; chunk boundaries do not
    // change a bit: every hop is summed in frame order).
    static int slots = 0;
    if (!slots)
    {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        slots = DMX_ISTFT_WGS * (cus > 0 ? cus : 256);
    }
    const int maxch = a.T / 8 > 0 ? a.T / 8 : 1;
    long best = -1;
    for (int n = 1; n <= maxch; ++n)
    {
        const int fpc = (a.T + n - 1) / n, nch = (a.T + fpc - 1) / fpc;
        const long wgs = 8l * ((long)(a.B * nch + 7) / 8) * a.S;
        const long cost = ((wgs + slots - 1) / slots) * (long)(fpc + 5);
        if (best < 0 || cost < best)
            best = cost, a.nch = nch, a.fpc = fpc;
    }
    const unsigned groups = (unsigned)(a.B * a.nch);
    hipLaunchKernelGGL(istft_ola_kernel, dim3(8u * ((groups + 7u) / 8u) * (unsigned)a.S), dim3(256), 0, s, a);
}

} // namespace dmx

// resample.hip — sample-rate conversion on the GPU, the step in front of (and behind) the hot path for input that is
// not 44.1 kHz (the role libnyquist plays for the reference's CLIs,
// /root/reference/cli-apps/demucs.cpp:21-76; the reference itself REJECTS other rates, :30-36, so there is no
// reference arithmetic to reproduce - the specification below is this file's own, restated in numpy in
// oracle/resample_oracle.py and pinned there against scipy.signal.resample_poly).
//
// Rational-ratio polyphase FIR, the textbook form (zero-stuff by L, low-pass, keep every M-th sample) evaluated per
// output sample:
//     L / M = rate_out / rate_in in lowest terms,  R = max(L, M),  c = 16 R  (16 zero crossings per side)
//     h[i]  = sinc((i - c) / R) * kaiser_8.6(i),  i = 0 .. 2c,  scaled so that sum(h) = L
//     y[k]  = sum_j x[j] h[c + k M - j L],        k = 0 .. ceil(n L / M) - 1,  x = 0 outside [0, n)
// With u = c + k M, p = u mod L, j_hi = u div L this is  y[k] = sum_{i=0}^{T-1} x[j_hi - i] hp[p][i],
// hp[p][i] = h[p + i L] (zero beyond 2c), T = ceil((2c + 1) / L): one fp32 fmaf chain in ascending i, so the result
// is bit-reproducible and independent of the launch geometry. Algorithmic traffic: every input sample read once,
// every output written once (the table of L*T coefficients, 20 KB for 48 kHz <-> 44.1 kHz, is re-staged per
// workgroup from L2); the inner loop is LDS-bound (2 T reads per output and plane pair).
#include "api_internal.h"

#include <cmath>
#include <map>
#include <mutex>
#include <numeric>

using namespace dmx;

namespace
{
const int kZeroCrossings = 16;
const double kKaiserBeta = 8.6;

double bessel_i0(double x)
{
    double s = 1.0, t = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 200; ++k)
    {
        t *= q / ((double)k * (double)k);
        s += t;
        if (t < 1e-17 * s)
            break;
    }
    return s;
}

struct Filter
{
    int L = 1, M = 1, T = 0;
    i64 c = 0;               // centre tap
    std::vector<float> taps; // h[0 .. 2c]
    std::vector<float> poly; // hp[p][i], L rows of Tp = T | 1 floats (odd row length: LDS banks), zero padded
};

int design(int rate_in, int rate_out, Filter &f)
{
    if (rate_in < 1 || rate_out < 1 || rate_in > 768000 || rate_out > 768000)
        return dmx_fail(DMX_ERR_ARG, "resample: sample rates must be in [1, 768000] (are %d -> %d)", rate_in, rate_out);
    const int g = std::gcd(rate_in, rate_out);
    f.L = rate_out / g;
    f.M = rate_in / g;
    const int R = std::max(f.L, f.M);
    f.c = (i64)kZeroCrossings * R;
    const i64 len = 2 * f.c + 1;
    f.T = (int)((len + f.L - 1) / f.L);
    if ((i64)f.L * f.T > (i64)1 << 24)
        return dmx_fail(DMX_ERR_ARG, "resample: the ratio %d/%d needs a polyphase table of %lld coefficients (limit 16 M); "
                                     "convert through a common rate first",
                        f.L, f.M, (long long)f.L * f.T);
    std::vector<double> h((size_t)len);
    const double i0b = bessel_i0(kKaiserBeta);
    double sum = 0.0;
    for (i64 i = 0; i < len; ++i)
    {
        const double t = (double)(i - f.c) / (double)R;
        const double s = t == 0.0 ? 1.0 : std::sin(M_PI * t) / (M_PI * t);
        const double r = (double)(i - f.c) / (double)f.c;
        const double w = bessel_i0(kKaiserBeta * std::sqrt(std::max(0.0, 1.0 - r * r))) / i0b;
        h[(size_t)i] = s * w;
        sum += h[(size_t)i];
    }
    f.taps.resize((size_t)len);
    for (i64 i = 0; i < len; ++i)
        f.taps[(size_t)i] = (float)(h[(size_t)i] * (double)f.L / sum);
    const int Tp = f.T | 1;
    f.poly.assign((size_t)f.L * Tp, 0.f);
    for (int p = 0; p < f.L; ++p)
        for (int i = 0; i < f.T; ++i)
        {
            const i64 t = (i64)p + (i64)i * f.L;
            if (t < len)
                f.poly[(size_t)p * Tp + i] = f.taps[(size_t)t];
        }
    return DMX_OK;
}

// A workgroup produces KB consecutive output samples of NP planes. The input window those need
// (KB * M / L + T samples per plane) and, when it fits, the polyphase table are staged in LDS with coalesced
// loads; every output is then T LDS reads of the signal (neighbouring lanes read neighbouring or identical
// samples: conflict-free / broadcast), T of the table (rows padded to an odd length so that lanes in different
// phases fall into different banks) and T fmaf per plane, in ascending tap order - the arithmetic and its order
// are those of the one-line definition in the header. Two planes per thread share the coefficient reads
// (interleaved stereo: NP = 2 with plane stride 1; a (S, 2, n) stem tensor: plane pairs).
template <int NP, bool TAB_LDS>
__global__ __launch_bounds__(256) void resample_kernel(const float *x, i64 n, i64 inPlane, i64 inStep, float *y, i64 nOut, i64 outPlane,
                                                       i64 outStep, const float *hp, int L, int M, int T, i64 c, int KB, int win,
                                                       int planes)
{
    extern __shared__ float smem[];
    float *xs = smem;            // [win][NP]: the planes of a thread side by side (NP = 2: one ds_read_b64 per tap)
    float *hs = smem + NP * win; // [L][Tp], Tp = T | 1
    const int Tp = T | 1;
    const int tid = threadIdx.x;
    const int plane0 = blockIdx.y * NP;
    const i64 k0 = (i64)blockIdx.x * KB;
    const i64 k1 = k0 + KB < nOut ? k0 + KB : nOut;
    const i64 jlo = (c + k0 * M) / L - (T - 1);
    const int count = (int)((c + (k1 - 1) * M) / L - jlo + 1);
    for (int idx = tid; idx < count; idx += 256)
    {
        const i64 j = jlo + idx;
        const bool ok = j >= 0 && j < n;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            xs[idx * NP + pl] = (ok && plane0 + pl < planes) ? x[(i64)(plane0 + pl) * inPlane + j * inStep] : 0.f;
    }
    if (TAB_LDS)
        for (int idx = tid; idx < L * Tp; idx += 256)
            hs[idx] = hp[idx];
    __syncthreads();
    for (i64 k = k0 + tid; k < k1; k += 256)
    {
        const i64 u = c + k * M;
        const i64 jhi = u / L;
        const int p = (int)(u - jhi * L);
        const int base = (int)(jhi - jlo);
        const float *h = TAB_LDS ? hs + p * Tp : hp + (i64)p * Tp;
        float acc[NP];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            acc[pl] = 0.f;
#pragma unroll 4
        for (int i = 0; i < T; ++i)
        {
            const float coef = h[i];
            if constexpr (NP == 2)
            {
                const float2 v = *reinterpret_cast<const float2 *>(&xs[(base - i) * 2]);
                acc[0] = fmaf(v.x, coef, acc[0]);
                acc[1] = fmaf(v.y, coef, acc[1]);
            }
            else
                acc[0] = fmaf(xs[base - i], coef, acc[0]);
        }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            if (plane0 + pl < planes)
                y[(i64)(plane0 + pl) * outPlane + k * outStep] = acc[pl];
    }
}

struct DeviceFilter
{
    Filter f;
    float *dPoly = nullptr;
};
std::mutex g_mu;
std::map<std::tuple<int, int, int>, DeviceFilter> g_filters; // (device, L, M) -> table; lives as long as the process

int device_filter(int device, int rate_in, int rate_out, const DeviceFilter **out)
{
    const int g = std::gcd(rate_in, rate_out);
    const auto key = std::make_tuple(device, rate_out / g, rate_in / g);
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_filters.find(key);
    if (it == g_filters.end())
    {
        DeviceFilter df;
        DMXCHK(design(rate_in, rate_out, df.f));
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipMalloc((void **)&df.dPoly, sizeof(float) * df.f.poly.size()));
        HIPCHK(hipMemcpy(df.dPoly, df.f.poly.data(), sizeof(float) * df.f.poly.size(), hipMemcpyHostToDevice));
        it = g_filters.emplace(key, std::move(df)).first;
    }
    *out = &it->second;
    return DMX_OK;
}
} // namespace

extern "C" int64_t dmx_resample_length(int64_t n_in, int rate_in, int rate_out)
{
    if (n_in < 0 || rate_in < 1 || rate_out < 1)
        return -1;
    const int g = std::gcd(rate_in, rate_out);
    const i64 L = rate_out / g, M = rate_in / g;
    return (n_in * L + M - 1) / M;
}

extern "C" int dmx_resample_filter(int rate_in, int rate_out, int *up, int *down, int *n_taps, float *taps, int cap)
{
    Filter f;
    DMXCHK(design(rate_in, rate_out, f));
    if (up)
        *up = f.L;
    if (down)
        *down = f.M;
    if (n_taps)
        *n_taps = (int)f.taps.size();
    if (taps)
    {
        if (cap < (int)f.taps.size())
            return dmx_fail(DMX_ERR_ARG, "dmx_resample_filter: %d taps do not fit %d", (int)f.taps.size(), cap);
        std::copy(f.taps.begin(), f.taps.end(), taps);
    }
    return DMX_OK;
}

extern "C" int dmx_resample_device(int device, const float *d_in, int64_t n_in, int planes, int64_t in_plane_stride, int64_t in_sample_stride,
                                   int rate_in, int rate_out, float *d_out, int64_t out_plane_stride, int64_t out_sample_stride, void *stream)
{
    if (!d_in || !d_out || n_in < 0 || planes < 1 || planes > 65535 || in_sample_stride < 1 || out_sample_stride < 1)
        return dmx_fail(DMX_ERR_ARG, "dmx_resample_device: invalid argument");
    if (dmx_device_count() <= device || device < 0)
        return dmx_fail(DMX_ERR_NO_DEVICE, "dmx_resample_device: no HIP device %d (there is no CPU fallback)", device);
    const DeviceFilter *df = nullptr;
    DMXCHK(device_filter(device, rate_in, rate_out, &df));
    const i64 nOut = dmx_resample_length(n_in, rate_in, rate_out);
    if (nOut == 0)
        return DMX_OK;
    HIPCHK(hipSetDevice(device));
    const Filter &f = df->f;
    // outputs per workgroup: 1024 (4096 would amortise the table staging further but leaves one workgroup per CU:
    // measured slower), fewer when the input window of a decimation would not fit 32 KB of LDS
    int KB = 1024, NP = planes >= 2 ? 2 : 1;
    auto window = [&](int kb) { return (int)(((i64)kb * f.M) / f.L + f.T + 2); };
    while (KB > 64 && sizeof(float) * (size_t)NP * window(KB) > 32 * 1024)
        KB /= 2;
    if (sizeof(float) * (size_t)NP * window(KB) > 64 * 1024)
        NP = 1;
    const int win = window(KB);
    const int Tp = f.T | 1;
    const size_t ldsSig = sizeof(float) * (size_t)NP * win, ldsTab = sizeof(float) * (size_t)f.L * Tp;
    const bool tabLds = ldsSig + ldsTab <= 64 * 1024; // (48 <-> 44.1 kHz: 20 KB; 96 -> 44.1 kHz: 42 KB; else from L1/L2)
    const size_t lds = ldsSig + (tabLds ? ldsTab : 0);
    if (lds > 64 * 1024)
        return dmx_fail(DMX_ERR_ARG, "dmx_resample_device: the ratio %d/%d needs a %zu-byte input window per workgroup", f.L, f.M, ldsSig);
    const dim3 grid((unsigned)((nOut + KB - 1) / KB), (unsigned)((planes + NP - 1) / NP));
    hipStream_t st = (hipStream_t)stream;
#define DMX_RS_LAUNCH(NP_, TAB_)                                                                                                       \
    hipLaunchKernelGGL((resample_kernel<NP_, TAB_>), grid, dim3(256), lds, st, d_in, n_in, in_plane_stride, in_sample_stride, d_out, nOut, \
                       out_plane_stride, out_sample_stride, df->dPoly, f.L, f.M, f.T, f.c, KB, win, planes)
    if (NP == 2 && tabLds)
        DMX_RS_LAUNCH(2, true);
    else if (NP == 2)
        DMX_RS_LAUNCH(2, false);
    else if (tabLds)
        DMX_RS_LAUNCH(1, true);
    else
        DMX_RS_LAUNCH(1, false);
#undef DMX_RS_LAUNCH
    HIPCHK(hipGetLastError());
    return DMX_OK;
}

extern "C" int dmx_resample(int device, const float *in, int64_t n_in, int planes, int interleaved, int rate_in, int rate_out, float *out)
{
    if (!in || !out || n_in < 0 || planes < 1)
        return dmx_fail(DMX_ERR_ARG, "dmx_resample: invalid argument");
    if (dmx_device_count() <= device || device < 0)
        return dmx_fail(DMX_ERR_NO_DEVICE, "dmx_resample: no HIP device %d (there is no CPU fallback)", device);
    const i64 nOut = dmx_resample_length(n_in, rate_in, rate_out);
    if (nOut < 0)
        return dmx_fail(DMX_ERR_ARG, "dmx_resample: invalid rates %d -> %d", rate_in, rate_out);
    if (n_in == 0 || nOut == 0)
        return DMX_OK;
    HIPCHK(hipSetDevice(device));
    float *dIn = nullptr, *dOut = nullptr;
    HIPCHK(hipMalloc((void **)&dIn, sizeof(float) * (size_t)(n_in * planes)));
    if (hipMalloc((void **)&dOut, sizeof(float) * (size_t)(nOut * planes)) != hipSuccess)
    {
        (void)hipFree(dIn);
        return dmx_fail(DMX_ERR_HIP, "dmx_resample: out of device memory");
    }
    int rc = DMX_OK;
    hipError_t e = hipMemcpy(dIn, in, sizeof(float) * (size_t)(n_in * planes), hipMemcpyHostToDevice);
    if (e == hipSuccess)
    {
        rc = interleaved ? dmx_resample_device(device, dIn, n_in, planes, 1, planes, rate_in, rate_out, dOut, 1, planes, nullptr)
                         : dmx_resample_device(device, dIn, n_in, planes, n_in, 1, rate_in, rate_out, dOut, nOut, 1, nullptr);
        if (rc == DMX_OK)
            e = hipMemcpy(out, dOut, sizeof(float) * (size_t)(nOut * planes), hipMemcpyDeviceToHost);
    }
    (void)hipFree(dIn);
    (void)hipFree(dOut);
    if (rc != DMX_OK)
        return rc;
    if (e != hipSuccess)
        return dmx_fail(DMX_ERR_HIP, "dmx_resample: copy failed: %s", hipGetErrorString(e));
    return DMX_OK;
}

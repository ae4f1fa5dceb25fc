// split_fp16.hip — the exactness domain of a three-term fp16 split on the device (the "next lever" of profiles/DESIGN_history_r1-r4.md 7.8; not
// product code). x = h1 + h2 + h3, h1 = fp16(x), h2 = fp16(x - h1), h3 = fp16(x - h1 - h2), conversions round-to-nearest-even.
// Prediction: exact for 0.5 <= |x| <= 65504 (11 + 11 + 2 significand bits, the last one at 2^-24 = fp16's smallest
// subnormal); below 0.5 the error is at most 2^-25; above 65504 h1 is inf. Prints, per binade of |x|, how many of the
// values are reproduced exactly and the largest absolute error, for the device's conversions (and whether they keep
// fp16 subnormals).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void split_kernel(const float *x, float *sum, unsigned *terms, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float v = x[i];
    const _Float16 h1 = (_Float16)v;
    const float r1 = v - (float)h1;
    const _Float16 h2 = (_Float16)r1;
    const float r2 = r1 - (float)h2;
    const _Float16 h3 = (_Float16)r2;
    sum[i] = ((float)h1 + (float)h2) + (float)h3;
    terms[i] = (unsigned)__builtin_bit_cast(unsigned short, h3);
}

int main()
{
    const int per = 20000, lo = -40, hi = 17; // binades 2^lo .. 2^hi
    std::vector<float> h;
    unsigned s = 12345u;
    for (int e = lo; e <= hi; ++e)
        for (int k = 0; k < per; ++k)
        {
            s = s * 1664525u + 1013904223u;
            const float m = 1.0f + (float)(s >> 9) * (1.0f / 8388608.0f); // [1, 2), all 23 fraction bits random
            h.push_back(((s >> 3) & 1u ? -1.0f : 1.0f) * ldexpf(m, e));
        }
    const int n = (int)h.size();
    float *dx, *ds;
    unsigned *dt;
    hipMalloc(&dx, n * 4), hipMalloc(&ds, n * 4), hipMalloc(&dt, n * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dx, ds, dt, n);
    std::vector<float> r(n);
    std::vector<unsigned> t(n);
    hipMemcpy(r.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(t.data(), dt, n * 4, hipMemcpyDeviceToHost);
    printf("binade   exact   max |sum - x|      (x 2^25)   subnormal third terms\n");
    for (int e = lo; e <= hi; ++e)
    {
        int exact = 0, sub = 0, nonfin = 0;
        double worst = 0;
        for (int k = 0; k < per; ++k)
        {
            const int i = (e - lo) * per + k;
            if (!std::isfinite(r[i]))
            {
                ++nonfin;
                continue;
            }
            exact += r[i] == h[i];
            worst = fmax(worst, fabs((double)r[i] - (double)h[i]));
            sub += (t[i] & 0x7c00u) == 0 && (t[i] & 0x3ffu) != 0;
        }
        if (e < -30 && e % 4)
            continue;
        printf("2^%-4d  %6d   %.3e   %8.3f   %6d%s\n", e, exact, worst, worst * 33554432.0, sub, nonfin ? "   (non-finite sums)" : "");
    }
    return 0;
}

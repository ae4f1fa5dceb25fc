// Micro test: semantics of global_load_lds_dwordx4 on gfx950 (direct global -> LDS load, 16 B per lane).
// Expected: lane l of a wave writes LDS[base(M0) + imm offset + 16 l .. +16) with the 16 bytes at ITS global address.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float *g, float *out)
{
    __shared__ float4 buf[2][256];
    const int tid = threadIdx.x, wave = tid >> 6;
    __builtin_amdgcn_global_load_lds(g + 4 * (tid ^ 5), (__attribute__((address_space(3))) void *)&buf[0][64 * wave], 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g + 4 * (tid ^ 9) + 1024, (__attribute__((address_space(3))) void *)&buf[1][64 * wave], 16, 0, 0);
    __syncthreads();
    for (int b = 0; b < 2; ++b)
    {
        float4 a = buf[b][tid];
        out[(b * 256 + tid) * 4 + 0] = a.x;
        out[(b * 256 + tid) * 4 + 1] = a.y;
        out[(b * 256 + tid) * 4 + 2] = a.z;
        out[(b * 256 + tid) * 4 + 3] = a.w;
    }
}
int main()
{
    std::vector<float> h(2048), o(2048);
    for (int i = 0; i < 2048; ++i)
        h[i] = (float)i;
    float *dg, *dout;
    hipMalloc(&dg, 8192);
    hipMalloc(&dout, 8192);
    hipMemcpy(dg, h.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dg, dout);
    hipMemcpy(o.data(), dout, 8192, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 2; ++b)
        for (int t = 0; t < 256; ++t)
            for (int c = 0; c < 4; ++c)
            {
                const float want = (float)(4 * (t ^ (b ? 9 : 5)) + c + 1024 * b);
                if (o[(b * 256 + t) * 4 + c] != want && bad++ < 5)
                    printf("mismatch b=%d t=%d c=%d got %g want %g\n", b, t, c, o[(b * 256 + t) * 4 + c], want);
            }
    printf("global_load_lds_dwordx4 semantics: %s (%d mismatches)\n", bad ? "UNEXPECTED" : "as expected", bad);
    return bad != 0;
}

#!/usr/bin/env python
"""Builds an INSTRUMENTED copy of csrc/dconv_row.hip (never the product): wave 0 of every workgroup stamps s_memtime after each
barrier of the row kernel and adds the phase durations to a device array; the launcher of the copy synchronises, prints the
per-phase averages (cycles per workgroup) to stderr and clears the array. Usage:
    python tools/micro/dconv_row_timing.py            # writes build/timing/dconv_row.hip
    make variant1src NAME=rowtiming FILE=dconv_row SRC=build/timing/dconv_row.hip
    DMX_LIB=demucs_cpp_amd/lib/libdemucs_hip_rowtiming.so PB=42 python tools/prof_ops.py rowtiming
The stamps go where the product source carries the barriers; the product itself holds no timing code."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "demucs_cpp_amd", "csrc", "dconv_row.hip")).read()

prelude = '''
__device__ unsigned long long g_rowdbg[64];
#define DMX_STAMP(k)                                                                   \\
    do                                                                                 \\
    {                                                                                  \\
        if (tid0 == 0)                                                                 \\
        {                                                                              \\
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                \\
            atomicAdd(&g_rowdbg[(k)], t_ - tprev_);                                    \\
            tprev_ = t_;                                                               \\
        }                                                                              \\
    } while (0)
'''
anchor = "namespace\n{\n// geometry shared by the kernel and its launcher"
assert src.count(anchor) == 1
src = src.replace(anchor, prelude + anchor, 1)
# start of the kernel body: the previous-stamp variable
src = src.replace("    const int tid0 = threadIdx.x, nthr = blockDim.x, nw = nthr >> 6;",
                  "    const int tid0 = threadIdx.x, nthr = blockDim.x, nw = nthr >> 6;\n    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();", 1)
# stamps per row: x loads issued (0), then per layer (base 1 + 8 * layer): operands landed + barrier (0), K1 + outer taps to LDS (1),
# taps meet + reduce 1 (2), GELU + hn written (3), k2f + reduce 2 (4), K3 (5); stores issued (17)
def put(text, stamp, after=True):
    global src
    assert src.count(text) == 1, text
    src = src.replace(text, (text + stamp) if after else (stamp + text), 1)
put("#pragma unroll 1\n        for (int layer = 0; layer < 2; ++layer)", "        DMX_STAMP(0);\n", after=False)
put('            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n            __syncthreads();\n', "            DMX_STAMP(1 + 8 * layer + 0);\n")
put("            __syncthreads();\n            if (!RES)\n", "            DMX_STAMP(1 + 8 * layer + 1);\n", after=False)
src = src.replace("            DMX_STAMP(1 + 8 * layer + 1);\n            __syncthreads();\n            if (!RES)\n", "            __syncthreads();\n            DMX_STAMP(1 + 8 * layer + 1);\n            if (!RES)\n", 1)
put("            block_sum2(s1, q1, red + (0 * 32), w, lane, nw);\n", "            DMX_STAMP(1 + 8 * layer + 2);\n")
put("            __syncthreads();\n\n            // ---- GroupNorm(1, 2C) statistics", "            DMX_STAMP(1 + 8 * layer + 3);\n", after=False)
src = src.replace("            DMX_STAMP(1 + 8 * layer + 3);\n            __syncthreads();\n", "            __syncthreads();\n            DMX_STAMP(1 + 8 * layer + 3);\n", 1)
put("            block_sum2(s2, q2, red + (1 * 32), w, lane, nw);\n", "            DMX_STAMP(1 + 8 * layer + 4);\n")
put("            // (no barrier here: the next phase that writes LDS", "            DMX_STAMP(1 + 8 * layer + 5);\n", after=False)
put("                    *reinterpret_cast<f32x4 *>(dst + 16 * j) = xr[i][j];\n            }\n", "        DMX_STAMP(17);\n")
# the launcher: dump after every launch
dump = '''        hipEvent_t e0_, e1_;
        (void)hipEventCreate(&e0_), (void)hipEventCreate(&e1_);
        (void)hipEventRecord(e0_, s);
        hipLaunchKernelGGL(kern, dim3(8 * slots), dim3(64 * nw), smem, s, k);
        (void)hipEventRecord(e1_, s);
        {
            (void)hipDeviceSynchronize();
            float ms_ = 0.f;
            (void)hipEventElapsedTime(&ms_, e0_, e1_);
            unsigned long long h[64];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rowdbg), sizeof(h));
            const double nr = (double)rows;
            static const char *nm[6] = {"wait", "K1", "meet+red1", "gelu", "k2f+red2", "K3"};
            double tot = 0;
            for (int q = 0; q < 18; ++q)
                tot += (double)h[q];
            fprintf(stderr, "[rowtiming] C=%d rows=%d wgs=%d (%d/CU) %.3f ms | ticks per ROW: load-issue %.0f |", a.C, rows, 8 * slots, perCu, ms_, h[0] / nr);
            for (int l = 0; l < 2; ++l)
                for (int q = 0; q < 6; ++q)
                    fprintf(stderr, " L%d.%s %.0f", l, nm[q], h[1 + 8 * l + q] / nr);
            fprintf(stderr, " | store %.0f | total %.0f | ticks/us if %d rows in flight: %.0f\\n", h[17] / nr, tot / nr, 8 * slots, tot / (ms_ * 1e3 * 8 * slots));
            unsigned long long z[64] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rowdbg), z, sizeof(z));
        }
'''
old = "        hipLaunchKernelGGL(kern, dim3(8 * slots), dim3(64 * nw), smem, s, k);\n"
assert src.count(old) == 1
src = src.replace(old, dump, 1)
src = src.replace('#include "igemm_common.h"', '#include "igemm_common.h"\n#include <cstdio>', 1)
out = os.path.join(ROOT, "build", "timing")
os.makedirs(out, exist_ok=True)
open(os.path.join(out, "dconv_row.hip"), "w").write(src)
print("wrote", os.path.join(out, "dconv_row.hip"))

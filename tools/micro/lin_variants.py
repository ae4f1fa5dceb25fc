#!/usr/bin/env python
"""Ablation / tuning copies of csrc/igemm_split.hip and csrc/plan.cpp for the linear-layer kernel (never the product).
    linhalf : igemm_split_lin_kernel loads and splits the activations of every OTHER K-tile only (odd tiles reuse the planes of the
              tile before; results wrong): the loop with half the activation loads and half the split work per MFMA - what a
              128 x 256 tile would change, with everything else (weight staging, fragment reads, tile count) as it is
    lin64   : plan.cpp keeps every linear layer on 64-row tiles (cfg 7) at any batch size
    lin64x3 : lin64 + the 64-row kernel compiled for THREE workgroups per CU (168 registers)
Writes build/variants/<file>_<name>.{hip,cpp}; build with
    make variant1src NAME=linhalf FILE=igemm_split SRC=build/variants/igemm_split_linhalf.hip
    make variant2src NAME=lin64x3 FILE=igemm_split SRC=build/variants/igemm_split_x3.hip FILE2=plan SRC2=build/variants/plan_lin64.cpp"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(ROOT, "build", "variants")
os.makedirs(out, exist_ok=True)
src = open(os.path.join(ROOT, "demucs_cpp_amd", "csrc", "igemm_split.hip")).read()
plan = open(os.path.join(ROOT, "demucs_cpp_amd", "csrc", "plan.cpp")).read()


def sub(s, old, new):
    assert s.count(old) == 1, old
    return s.replace(old, new)


loadA = """        for (int i = 0; i < WMF; ++i)
        {
            const char *src = reinterpret_cast<const char *>(p.X) + aOff[i];
            aRaw[SET][i][0] = *reinterpret_cast<const f32x4 *>(src);
            aRaw[SET][i][1] = *reinterpret_cast<const f32x4 *>(src + 16);
            aOff[i] += KT * 4;
        }"""
s = sub(src, loadA, loadA.replace("        for (int i = 0; i < WMF; ++i)", "        for (int i = 0; i < (SET == 0 ? WMF : 0); ++i)"))
splitHead = """        constexpr int SET = decltype(setTag)::value, DST = decltype(dstTag)::value;
        const f32x4 lo = aRaw[SET][i][0], hi = aRaw[SET][i][1];"""
s = sub(s, splitHead, """        constexpr int SET = decltype(setTag)::value, DST = decltype(dstTag)::value;
        if constexpr (DST == 1)
        {
            aPl[1][i][0] = aPl[0][i][0], aPl[1][i][1] = aPl[0][i][1], aPl[1][i][2] = aPl[0][i][2];
            return;
        }
        const f32x4 lo = aRaw[SET][i][0], hi = aRaw[SET][i][1];""")
open(os.path.join(out, "igemm_split_linhalf.hip"), "w").write(s)

lb = """template <int WMF, int WNF, int EPI, int ARITH = 0>
__global__ __launch_bounds__(256, 2) void igemm_split_lin_kernel(const GemmArgs p)"""
open(os.path.join(out, "igemm_split_x3.hip"), "w").write(sub(src, lb, lb.replace("__launch_bounds__(256, 2)", "__launch_bounds__(256, WMF == 1 ? 3 : 2)")))

cfgline = "        g.cfg = refine_cfg(choose_cfg((i64)g.P1 * g.P0, g.N, paired), (i64)g.B * g.P1 * g.P0, g.N, g.rowstat >= 0);"
open(os.path.join(out, "plan_lin64.cpp"), "w").write(sub(plan, cfgline, cfgline + """
        if (opts.gemm != GEMM_F32 && g.cfg == 0 && g.pro == PRO_NONE && g.S1 == 1 && g.seg0 == g.K && g.K % 32 == 0 && g.N % 128 == 0 &&
            (g.epi == EPI_LINEAR || g.epi == EPI_SCALE_RES || g.epi == EPI_KPL || g.epi == EPI_VT))
            g.cfg = 7; // (experiment: the linear layers on 64-row tiles at any batch)"""))
print("wrote igemm_split_linhalf.hip igemm_split_x3.hip plan_lin64.cpp")

#!/usr/bin/env python
"""Ablation / tuning copies of csrc/dconv_row.hip (never the product).
    copy     : rows are loaded and stored, no layer is computed (the access pattern and occupancy of the product kernel)
    nostore  : both layers, nothing written back
    w5/w7/w8 : the C = 48 kernels compiled for 5 / 7 / 8 waves per SIMD (96 / 72 / 64 registers) instead of 6 (80)
    nomem    : both layers on register contents that never came from memory, nothing written back (the compute phases alone)
    stagN    : the workgroups that share a CU start N * 64 * 127 cycles apart (are the co-resident workgroups in lockstep?)
Writes build/variants/dconv_row_<name>.hip; build each with
    make variant1src NAME=row<name> FILE=dconv_row SRC=build/variants/dconv_row_<name>.hip"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "demucs_cpp_amd", "csrc", "dconv_row.hip")).read()
loop = "        for (int layer = 0; layer < 2; ++layer)"
store = "            if (tOk[i])\n            {\n                const unsigned off = ((unsigned)(((tids"
assert src.count(loop) == 1 and src.count(store) == 1
out = os.path.join(ROOT, "build", "variants")
os.makedirs(out, exist_ok=True)
def emit(name, s):
    open(os.path.join(out, f"dconv_row_{name}.hip"), "w").write(s)
    print("wrote", name)
emit("copy", src.replace(loop, "        for (int layer = 0; layer < 0; ++layer)"))
emit("nostore", src.replace(store, store.replace("if (tOk[i])", "if (tOk[i] && p.T < 0)")))
for w in (5, 7, 8):
    s = src.replace("dconv_row_kernel<48, 6, FPW, 6>", f"dconv_row_kernel<48, 6, FPW, {w}>").replace("dconv_row_kernel<48, 12, FPW, 6>", f"dconv_row_kernel<48, 12, FPW, {w}>")
    assert s != src
    emit(f"w{w}", s)

load = "                xr[i][j] = *reinterpret_cast<const f32x4 *>(xrow + off + 64 * j);"
assert src.count(load) == 1
nomem = src.replace(load, "                xr[i][j] = f32x4{1e-3f * (float)(tid0 + j), 0.5f, -0.25f, 1e-2f * (float)i};")
emit("nomem", nomem.replace(store, store.replace("if (tOk[i])", "if (tOk[i] && p.T < 0)")))
walk = "    const int rows = p.B * F, slots = (int)(gridDim.x >> 3);"
assert src.count(walk) == 1
for n in (2, 4, 8):
    emit(f"stag{n}", src.replace(walk, walk + f"""
    {{
        // co-resident workgroups (dispatch order within an XCD: one per CU, then the second per CU, ...) start apart
        const int slot = (int)(blockIdx.x >> 3) / 32;
        for (int k = 0; k < slot * {n} * (C == 48 ? 1 : 2); ++k)
            __builtin_amdgcn_s_sleep(127);
    }}"""))

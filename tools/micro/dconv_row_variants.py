#!/usr/bin/env python
"""Ablation / tuning copies of csrc/dconv_row.hip (never the product).
    copy     : rows are loaded and stored, no layer is computed (the access pattern and occupancy of the product kernel)
    nostore  : both layers, nothing written back
    w5/w7/w8 : the C = 48 kernels compiled for 5 / 7 / 8 waves per SIMD (96 / 72 / 64 registers) instead of 6 (80)
Writes build/variants/dconv_row_<name>.hip; build each with
    make variant1src NAME=row<name> FILE=dconv_row SRC=build/variants/dconv_row_<name>.hip"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "demucs_cpp_amd", "csrc", "dconv_row.hip")).read()
loop = "        for (int layer = 0; layer < 2; ++layer)"
store = "        // ---- x back, once\n#pragma unroll\n        for (int i = 0; i < FPW; ++i)\n            if (tOk[i])"
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

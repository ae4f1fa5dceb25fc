#!/usr/bin/env python
"""CPU study (no GPU; test infrastructure like the oracle it drives, not product code): what would operand splits with fp16
terms do to the results?

Binds tests/study/quant_sgemm.c into the ORACLE through its BLAS hook, so every convolution / linear layer of the
oracle sees its activation operand quantised the way a split GEMM would see it (QK^T and PV stay fp32), and compares
  * the default synthetic models with the fp64 torch goldens (tests/golden/golden_seg_{4s,6s}.npz), and
  * the stress models (dc, illcond, initscale, input gains) with the oracle's own plain fp32 run.
    python tests/study/split_study.py [--quick]
"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/study/ -> repo root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc  # noqa: E402
from demucs_cpp_amd.weights import write_synthetic_model  # noqa: E402

MODES = ["f32", "bf16x3", "fp16x3row", "fp16x3mat", "fp16x3"]


def build():
    so = os.path.join(tempfile.gettempdir(), "libquant_sgemm.so")
    subprocess.check_call(["gcc", "-O2", "-march=native", "-mf16c", "-fopenmp", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests", "study", "quant_sgemm.c"), "-lm"])
    return so


def relerr(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / np.abs(b).max())


def main():
    quick = "--quick" in sys.argv
    so = build()
    q = ctypes.CDLL(so)
    q.quant_stats.argtypes = [ctypes.c_void_p]
    assert orc.lib().orc_use_blas(so.encode(), b"quant_cblas_sgemm64") == 0
    d = tempfile.mkdtemp()
    stats = np.zeros(3, np.int64)

    def run(model, mix, mode):
        os.environ["QMODE"] = mode
        out = model.segment(mix)
        q.quant_stats(stats.ctypes.data)
        return out, stats.copy()

    print("== default synthetic weights vs the fp64 torch golden (relative max-abs error of the output)")
    for ns, seed in ((4, 0), (6, 3)):
        g = np.load(os.path.join(ROOT, "tests", "golden", f"golden_seg_{ns}s.npz"))
        path = os.path.join(d, f"m{ns}.bin")
        write_synthetic_model(path, ns, seed)
        m = orc.OracleModel(path)
        row = []
        for mode in MODES:
            out, st = run(m, g["mix"], mode)
            row.append((mode, relerr(out, g["out"]), int(st[1]), int(st[2])))
        print(f"  htdemucs-{ns}s: " + "   ".join(f"{mode} {e:.3e}" for mode, e, _, _ in row) + f"   (inexact fp16 weights seen: {row[2][2]}, overflows: {row[2][3]})")
        m.close()
        if quick:
            break

    print("== stress models / input gains vs the oracle's own fp32 run (relative max-abs difference of the output)")
    rng = np.random.default_rng(11)
    n = 12000
    mix = (0.1 * rng.standard_normal((2, n))).astype(np.float32)
    cases = [("default", 1.0), ("dc", 1.0), ("illcond", 1.0), ("initscale", 1.0), ("default", 0.25), ("default", 4.0), ("default", 64.0), ("default", 1e-3)]
    if quick:
        cases = cases[:2]
    for variant, gain in cases:
        path = os.path.join(d, f"s_{variant}.bin")
        if not os.path.exists(path):
            write_synthetic_model(path, 4, 0, variant)
        m = orc.OracleModel(path)
        ref, _ = run(m, mix * np.float32(gain), "f32")
        row = []
        for mode in MODES[1:]:
            out, st = run(m, mix * np.float32(gain), mode)
            fin = bool(np.isfinite(out).all())
            row.append(f"{mode} {relerr(out, ref) if fin else float('nan'):.3e}" + ("" if fin else " (non-finite)") + (f" [{int(st[2])} overflows]" if st[2] else ""))
        print(f"  {variant:9s} input gain {gain:<6g} |out|max {np.abs(ref).max():.3e}: " + "   ".join(row))
        m.close()
    orc.lib().orc_use_blas(None, None)


if __name__ == "__main__":
    main()

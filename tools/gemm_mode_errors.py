#!/usr/bin/env python
"""Error of every GEMM mode against the fp64 golden model (tests/golden/golden_seg_{4s,6s}.npz, golden_seg_v3.npz):
max |out - golden| / max |golden| per model and mode, and the modes against each other. One line per model. (GPU box.)"""
import glob
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demucs_cpp_amd import binding as dmx  # noqa: E402
from demucs_cpp_amd.weights import write_synthetic_model  # noqa: E402


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


d = tempfile.mkdtemp()
models = {"4s": (4, 0, "htdemucs"), "6s": (6, 3, "htdemucs"), "v3": (4, 5, "v3")}  # seeds of tests/golden/make_golden*.py
for key, gname in (("4s", "golden_seg_4s.npz"), ("6s", "golden_seg_6s.npz"), ("v3", "golden_seg_v3.npz")):
    path = os.path.join(ROOT, "tests", "golden", gname)
    if not os.path.exists(path):
        cands = glob.glob(os.path.join(ROOT, "tests", "golden", f"*{key}*.npz"))
        if not cands:
            continue
        path = cands[0]
    g = np.load(path)
    if "mix" not in g or "out" not in g:
        continue
    ns, seed, arch = models[key]
    p = os.path.join(d, f"m_{key}.bin")
    write_synthetic_model(p, ns, seed, "default", arch) if arch == "v3" else write_synthetic_model(p, ns, seed)
    m = dmx.Model(p)
    outs = {}
    for mode in (dmx.GEMM_F32, dmx.GEMM_BF16X3, dmx.GEMM_FP16X3):
        c = dmx.Context(m, int(g["seg"]), 1, gemm=mode)
        outs[dmx.GEMM_NAMES[mode]] = c.segment(g["mix"])
        c.close()
    m.close()
    print(key, os.path.basename(path), "seg", int(g["seg"]),
          " ".join(f"{k} vs fp64 {relerr(v, g['out']):.3e}" for k, v in outs.items()),
          f"| fp16x3 vs bf16x3 {relerr(outs['fp16x3'], outs['bf16x3']):.3e}  bf16x3 vs f32 {relerr(outs['bf16x3'], outs['f32']):.3e}", flush=True)

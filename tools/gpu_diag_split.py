#!/usr/bin/env python
"""Diagnostic: where do bf16x3 contexts stop being bit-reproducible? (run on the GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model

which = sys.argv[1] if len(sys.argv) > 1 else "4"
path = f"/tmp/diag_{which}.bin"
if which == "3":
    write_synthetic_model(path, 4, 5, "default", "v3")
else:
    write_synthetic_model(path, int(which), 0 if which == "4" else 3)
dmx.set_default_gemm(dmx.GEMM_BF16X3)
SEG = 343980
rng = np.random.default_rng(7)
mix = (0.1 * rng.standard_normal((2, SEG))).astype(np.float32)
m = dmx.Model(path)
TAPS = ["x_cac", "x_0", "xt_0", "x_1", "xt_1", "x_2", "xt_2", "x_3", "xt_3", "ct_in_x", "ct_in_xt", "ct_x", "ct_xt", "dec_0", "tdec_0", "dec_3", "tdec_3"]

def run(ctx, taps=False):
    o = ctx.segment(mix)
    t = {k: ctx.tap(k) for k in TAPS} if taps else {}
    return o, t

def cmp(tag, a, b):
    d = np.abs(a - b)
    print(f"{tag}: equal={np.array_equal(a, b)} maxabs={d.max():.3e} n_diff={(d > 0).sum()} of {d.size}", flush=True)

c1 = dmx.Context(m, 0, 2)
o1, t1 = run(c1, True)
for r in range(3):
    o, t = run(c1, True)
    cmp(f"same ctx repeat {r}", o, o1)
    for k in TAPS:
        if t[k] is not None and not np.array_equal(t[k], t1[k]):
            print("   first differing tap:", k, np.abs(t[k] - t1[k]).max()); break
c2 = dmx.Context(m, 0, 2)
o2, t2 = run(c2, True)
cmp("second ctx same model", o2, o1)
for k in TAPS:
    if t2[k] is not None and not np.array_equal(t2[k], t1[k]):
        print("   first differing tap:", k, np.abs(t2[k] - t1[k]).max()); break
c3 = dmx.Context(m, 0, 3)
o3, t3 = run(c3, True)
cmp("ctx max_batch 3", o3, o1)
stride = 257985
n = 2 * stride + 1000
audio = (0.1 * np.random.default_rng(41).standard_normal((2, n)) + 0.02).astype(np.float32)
ref = c1.track(audio, 4033)
cmp("track repeat", c1.track(audio, 4033), ref)
cmp("track ctx2", c2.track(audio, 4033), ref)
S = m.n_sources
for devs in ([0], [0, 0], [0, 0, 0]):
    eng = dmx.Engine([path], devs, max_batch=2)
    got = eng.track(audio, [4033])
    cmp(f"engine {devs}", got, ref)
    if not np.array_equal(got, ref):
        d = np.abs(got - ref).max(axis=(0, 1))
        nz = np.nonzero(d)[0]
        print("   differing sample range:", nz.min(), nz.max(), "of", n, " segments boundaries at multiples of", stride, "- shift 18017")
    cmp(f"engine {devs} repeat", eng.track(audio, [4033]), got)
    eng.close()

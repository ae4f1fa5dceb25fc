#!/usr/bin/env python
"""Diagnostic 4: two bf16x3 contexts run concurrently; the victim's STFT output (tap x_cac, independent of every later op)
is compared with the one of a quiet run. Works with ablated libraries (DMX_LIB) whose later results are garbage."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model

path = "/tmp/diag_4.bin"
write_synthetic_model(path, 4, 0)
dmx.set_default_gemm(dmx.GEMM_BF16X3 if os.environ.get("MODE", "bf16x3") == "bf16x3" else dmx.GEMM_F32)
SEG = 343980
mix = (0.1 * np.random.default_rng(7).standard_normal((2, SEG))).astype(np.float32)
m = dmx.Model(path)
MB = int(os.environ.get("MB", "2"))
c = dmx.Context(m, 0, MB)
c.segment(mix)
ref = c.tap("x_cac")
c2 = dmx.Context(m, 0, MB)
bad = {"a": 0, "b": 0}
R = int(os.environ.get("RUNS", "12"))
def work(ctx, key):
    for r in range(R):
        ctx.segment(mix)
        t = ctx.tap("x_cac")
        if not np.array_equal(t, ref):
            bad[key] += 1
th = [threading.Thread(target=work, args=(c, "a")), threading.Thread(target=work, args=(c2, "b"))]
[t.start() for t in th]; [t.join() for t in th]
print(f"[{os.environ.get('DMX_LIB', 'product')[-24:]} {os.environ.get('MODE', 'bf16x3')} {os.environ.get('EXTRA', '')}] runs with a wrong STFT output: a {bad['a']}/{R}, b {bad['b']}/{R}", flush=True)

#!/usr/bin/env python
"""Diagnostic 3: two bf16x3 contexts concurrently; for a wrong run, the first differing tap and where it differs."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model

path = "/tmp/diag_4.bin"
write_synthetic_model(path, 4, 0)
dmx.set_default_gemm(dmx.GEMM_BF16X3)
SEG = 343980
mix = (0.1 * np.random.default_rng(7).standard_normal((2, SEG))).astype(np.float32)
m = dmx.Model(path)
TAPS = ["x_cac", "x_0", "xt_0", "x_1", "xt_1", "x_2", "xt_2", "x_3", "xt_3", "x_3_up", "ct_in_x", "ct_in_xt", "ct_x", "ct_xt", "dec_0", "tdec_0", "dec_1", "tdec_1", "dec_2", "tdec_2", "dec_3", "tdec_3"]
MB = int(os.environ.get("MB", "2"))
c = dmx.Context(m, 0, MB)
ref = c.segment(mix)
rt = {k: c.tap(k) for k in TAPS}
c2 = dmx.Context(m, 0, MB)
lock = threading.Lock()
def work(ctx, key):
    for r in range(int(os.environ.get("RUNS", "6"))):
        o = ctx.segment(mix)
        if not np.array_equal(o, ref):
            t = {k: ctx.tap(k) for k in TAPS}
            with lock:
                print(f"ctx {key} run {r}: out maxabs {np.abs(o - ref).max():.3e}", flush=True)
                status = []
                for k in TAPS:
                    if t[k] is None or rt[k] is None:
                        continue
                    d = np.abs(t[k] - rt[k])
                    status.append(f"{k}:{'=' if d.max() == 0 else '%.1e/%d' % (d.max(), int((d > 0).sum()))}")
                print("   " + "  ".join(status), flush=True)
th = [threading.Thread(target=work, args=(c, "a")), threading.Thread(target=work, args=(c2, "b"))]
[t.start() for t in th]; [t.join() for t in th]
print("done")

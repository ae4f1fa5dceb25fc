#!/usr/bin/env python
"""Diagnostic 2: batch 2 vs batch 1 per tap in bf16x3 contexts at full size, and two contexts running concurrently."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model

which = sys.argv[1] if len(sys.argv) > 1 else "4"
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
path = f"/tmp/diag_{which}.bin"
if which == "3":
    write_synthetic_model(path, 4, 5, "default", "v3")
else:
    write_synthetic_model(path, int(which), 0 if which == "4" else 3)
dmx.set_default_gemm(dmx.GEMM_BF16X3 if mode == "bf16x3" else dmx.GEMM_F32)
SEG = 343980
rng = np.random.default_rng(7)
mixes = (0.1 * rng.standard_normal((2, 2, SEG))).astype(np.float32)
m = dmx.Model(path)
S = m.n_sources
TAPS4 = ["x_cac", "x_0", "xt_0", "x_1", "xt_1", "x_2", "xt_2", "x_3", "xt_3", "x_3_up", "ct_in_x", "ct_in_xt", "ct_x", "ct_xt", "dec_0", "tdec_0", "dec_1", "tdec_1", "dec_2", "tdec_2", "dec_3", "tdec_3"]
c = dmx.Context(m, 0, 2)
single, staps = [], []
for b in range(2):
    single.append(c.segment(mixes[b]))
    staps.append({k: c.tap(k) for k in TAPS4})
d_mix = torch.from_numpy(np.ascontiguousarray(mixes.transpose(0, 2, 1))).cuda()
d_out = torch.zeros((2, S, 2, SEG), device="cuda")
torch.cuda.synchronize()
for rep in range(2):
    c.segment_device(d_mix.data_ptr(), d_out.data_ptr(), 2); c.synchronize()
    got = d_out.cpu().numpy()
    print(f"[{mode} model {which}] batch 2 vs singles, rep {rep}:", [bool(np.array_equal(got[b], single[b])) for b in range(2)],
          [float(np.abs(got[b] - single[b]).max()) for b in range(2)], flush=True)
    for k in TAPS4:
        t = c.tap(k)
        if t is None:
            continue
        for b in range(2):
            s = staps[b][k]
            if s is not None and not np.array_equal(t[b], s[0]):
                d = np.abs(t[b] - s[0])
                idx = np.argwhere(d > 0)
                print(f"   tap {k} b={b}: maxabs {d.max():.3e} n_diff {len(idx)} of {d.size}; first {idx[0]}, last {idx[-1]}; shape {t.shape}")
                break
        else:
            continue
        break
# two contexts concurrently from two host threads
c2 = dmx.Context(m, 0, 2)
res = {}
def work(ctx, key):
    outs = []
    for r in range(4):
        outs.append(ctx.segment(mixes[0]))
    res[key] = outs
th = [threading.Thread(target=work, args=(c, "a")), threading.Thread(target=work, args=(c2, "b"))]
[t.start() for t in th]; [t.join() for t in th]
for key in ("a", "b"):
    print(f"[{mode} model {which}] concurrent ctx {key}:", [bool(np.array_equal(o, single[0])) for o in res[key]], [float(np.abs(o - single[0]).max()) for o in res[key]], flush=True)

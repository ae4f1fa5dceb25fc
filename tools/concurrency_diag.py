#!/usr/bin/env python
"""Does a plan run change a bit when NT contexts of one GPU run at the same time? NT threads, each with its own context, run REPS
plan runs of PB segments (device buffers) against a quiet reference; mismatching runs are listed with the layer taps that differ
and the batch elements hit. MODE=f32|bf16x3, NS=4|6. (Round 6: this found the missing prologue barrier of attention.hip's head-dim-48
form - profiles/r06_experiments/attention48_prologue_race.txt.)   MODE=f32 NS=6 python tools/concurrency_diag.py"""
import os, sys, threading, numpy as np, torch
sys.path.insert(0, os.getcwd())
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
mode = {"f32": dmx.GEMM_F32, "bf16x3": dmx.GEMM_BF16X3}[os.environ.get("MODE", "f32")]
dmx.set_default_gemm(mode)
ns = int(os.environ.get("NS", "6")); NT = int(os.environ.get("NT", "8")); REPS = int(os.environ.get("REPS", "8")); PB = int(os.environ.get("PB", "6"))
path = f"/tmp/diag_{ns}s.bin"
write_synthetic_model(path, ns, 3 if ns == 6 else 0)
TAPS = ["x_cac", "x_0", "xt_0", "x_3", "xt_3", "ct_in_x", "ct_in_xt", "ct_x", "ct_xt", "dec_0", "tdec_0", "dec_3", "tdec_3"]
m = dmx.Model(path)
S = m.n_sources
seg = 343980
g = torch.Generator(device="cpu").manual_seed(5)
mix_h = 0.1 * torch.randn((PB, seg, 2), generator=g) + 0.02
def run(ctx, d_mix, d_out):
    ctx.segment_device(d_mix.data_ptr(), d_out.data_ptr(), PB)
    ctx.synchronize()
ctx0 = dmx.Context(m, 0, PB)
d_mix0 = mix_h.cuda(); d_out0 = torch.zeros(PB, S, 2, seg, device="cuda")
run(ctx0, d_mix0, d_out0)
ref_out = d_out0.cpu().numpy().copy()
ref = {t: ctx0.tap(t) for t in TAPS}
run(ctx0, d_mix0, d_out0)
print("quiet run-to-run equal:", np.array_equal(d_out0.cpu().numpy(), ref_out))
ctxs = [dmx.Context(m, 0, PB) for _ in range(NT)]
bufs = [(mix_h.cuda(), torch.zeros(PB, S, 2, seg, device="cuda")) for _ in range(NT)]
torch.cuda.synchronize()
res = [[] for _ in range(NT)]
def work(i):
    for rep in range(REPS):
        run(ctxs[i], bufs[i][0], bufs[i][1])
        out = bufs[i][1].cpu().numpy()
        if not np.array_equal(out, ref_out):
            bad = [t for t in TAPS if not np.array_equal(ctxs[i].tap(t), ref[t])]
            per = [(b, int((out[b] != ref_out[b]).sum()), float(np.abs(out[b] - ref_out[b]).max())) for b in range(PB) if not np.array_equal(out[b], ref_out[b])]
            res[i].append((rep, bad, per))
ths = [threading.Thread(target=work, args=(i,)) for i in range(NT)]
[t.start() for t in ths]; [t.join() for t in ths]
print("mismatching runs of", NT * REPS, ":", sum(len(r) for r in res))
for i in range(NT):
    for r in res[i][:3]:
        print(i, r)

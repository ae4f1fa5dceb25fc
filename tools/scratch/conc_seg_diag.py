import os, sys, threading, numpy as np
sys.path.insert(0, os.getcwd())
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
mode = {"f32": dmx.GEMM_F32, "bf16x3": dmx.GEMM_BF16X3}[os.environ.get("MODE", "f32")]
dmx.set_default_gemm(mode)
ns = int(os.environ.get("NS", "6")); NT = int(os.environ.get("NT", "8")); REPS = int(os.environ.get("REPS", "24"))
path = f"/tmp/diag_{ns}s.bin"
write_synthetic_model(path, ns, 3 if ns == 6 else 0)
TAPS = ["x_cac", "x_0", "xt_0", "x_1", "xt_1", "x_2", "x_3", "ct_x", "dec_0", "dec_1", "dec_2", "dec_3", "tdec_3"]
mix = (0.1 * np.random.default_rng(6).standard_normal((2, 343980)) + 0.02).astype(np.float32)
m = dmx.Model(path)
ctx0 = dmx.Context(m, 0, 1)
ref_out = ctx0.segment(mix)
ref = {t: ctx0.tap(t) for t in TAPS}
ctx0.close()
ctxs = [dmx.Context(m, 0, 1) for _ in range(NT)]
res = [[] for _ in range(NT)]
def work(i):
    for rep in range(REPS):
        out = ctxs[i].segment(mix)
        ok = bool(np.array_equal(out, ref_out))
        if not ok:
            bad = [t for t in TAPS if not np.array_equal(ctxs[i].tap(t), ref[t])]
            d = out != ref_out
            res[i].append((rep, bad, int(d.sum()), float(np.abs(out - ref_out).max())))
ths = [threading.Thread(target=work, args=(i,)) for i in range(NT)]
[t.start() for t in ths]; [t.join() for t in ths]
print("mismatching runs of", NT * REPS, ":", sum(len(r) for r in res))
for i in range(NT):
    if res[i]:
        print(i, res[i][:4])

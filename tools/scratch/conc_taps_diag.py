import os, sys, threading, numpy as np
sys.path.insert(0, os.getcwd())
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
mode = {"f32": dmx.GEMM_F32, "bf16x3": dmx.GEMM_BF16X3}[os.environ.get("MODE", "f32")]
dmx.set_default_gemm(mode)
ns = int(os.environ.get("NS", "6"))
NT = int(os.environ.get("NT", "8"))
path = f"/tmp/diag_{ns}s.bin"
write_synthetic_model(path, ns, 3 if ns == 6 else 0)
TAPS = ["x_cac", "x_0", "xt_0", "x_1", "xt_1", "x_2", "x_3", "ct_x", "dec_0", "dec_1", "dec_2", "dec_3", "tdec_3"]
n = 257985 * 5 + 343980  # six segments
audio = (0.1 * np.random.default_rng(6).standard_normal((2, n)) + 0.02).astype(np.float32)
m = dmx.Model(path)
ctx0 = dmx.Context(m, 0, 6)
ref_out = ctx0.track(audio, 4033)
ref = {t: ctx0.tap(t) for t in TAPS}
ctx0.close()
ctxs = [dmx.Context(m, 0, 6) for _ in range(NT)]
res = [[] for _ in range(NT)]
def work(i):
    for rep in range(int(os.environ.get('REPS','6'))):
        out = ctxs[i].track(audio, 4033)
        bad = [t for t in TAPS if not np.array_equal(ctxs[i].tap(t), ref[t])]
        ok = bool(np.array_equal(out, ref_out))
        info = ""
        if not ok:
            d = out != ref_out
            idx = np.argwhere(d)
            pos = idx[:, 2]
            info = "ndiff %d stems %s range %d-%d" % (d.sum(), sorted(set(idx[:, 0].tolist())), pos.min(), pos.max())
            # structure of the error in the first differing stem: fit out = a ref + b over the differing range
            s0 = int(idx[0, 0]); c0 = int(idx[0, 1]); lo, hi = int(pos.min()), int(pos.max()) + 1
            x = ref_out[s0, c0, lo:hi].astype(np.float64); y = out[s0, c0, lo:hi].astype(np.float64)
            a, b = np.polyfit(x, y, 1)
            r = y - (a * x + b)
            info += " fit a-1 %.3e b %.3e resid rms %.3e err rms %.3e" % (a - 1, b, np.sqrt((r * r).mean()), np.sqrt(((y - x) ** 2).mean()))
        res[i].append((ok, bad, info))
ths = [threading.Thread(target=work, args=(i,)) for i in range(NT)]
[t.start() for t in ths]; [t.join() for t in ths]
for i in range(NT):
    print(i, res[i])
# where inside the first bad tap?

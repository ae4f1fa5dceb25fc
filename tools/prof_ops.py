#!/usr/bin/env python
"""Per-op profile of the segment plan (dmx_debug_profile: HIP events around `reps` launches of every op)
at batch PB, written to gpurun_out/ops_<label>.tsv; prints the per-kernel-class summary. The env switches of
the kernels (DMX_*) are read once per process, so A/B runs are separate invocations with different env."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demucs_cpp_amd import binding as dmx  # noqa: E402
from demucs_cpp_amd.weights import write_synthetic_model  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "base"
B = int(os.environ.get("PB", "24"))
ns = int(os.environ.get("NS", "4"))
path = f"/tmp/prof_ops_{ns}s.bin"
if not os.path.exists(path):
    if ns == 3:  # hdemucs_mmi (Demucs v3)
        write_synthetic_model(path, 4, 5, "default", "v3")
    else:
        write_synthetic_model(path, ns, 0 if ns == 4 else 3)
m = dmx.Model(path)
ctx = dmx.Context(m, 0, B)
prof = ctx.profile(B, int(os.environ.get("REPS", "3")))
os.makedirs("gpurun_out", exist_ok=True)
with open(f"gpurun_out/ops_{label}.tsv", "w") as f:
    for r in prof:
        f.write("\t".join(str(x) for x in r) + "\t" + ctx.profile_geometry.get(r[0], "") + "\n")
agg = {}
for nm, k, ms, fl, by in prof:
    d = agg.setdefault(k, [0, 0, 0, 0])
    d[0] += ms
    d[1] += fl
    d[2] += by
    d[3] += 1
tot = sum(v[0] for v in agg.values())
print(f"[{label}] B={B}: total {tot:.3f} ms per batch = {tot / B:.4f} ms/segment")
for k, (ms, fl, by, n) in sorted(agg.items(), key=lambda x: -x[1][0]):
    print(f"  {k:16s} n={n:3d} {ms:8.3f} ms {100 * ms / tot:5.1f}%  {fl / ms / 1e9 if ms else 0:7.1f} TF/s {by / ms / 1e6 if ms else 0:8.1f} GB/s")

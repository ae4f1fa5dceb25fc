#!/usr/bin/env python
"""ROOT and OWNER finish of an 8-logical-device engine on one GPU against one context's track result, three rounds; prints where
(stems, sample range, segments) and by how much a mismatching run differs. MODE=f32|bf16x3, NS=4|6. (Round 6: the first reproducer
of the attention.hip head-dim-48 race; tools/concurrency_diag.py localises by layer taps.)"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
mode = {"f32": dmx.GEMM_F32, "bf16x3": dmx.GEMM_BF16X3}[os.environ.get("MODE", "f32")]
dmx.set_default_gemm(mode)
ns = int(os.environ.get("NS", "6"))
path = f"/tmp/diag_{ns}s.bin"
write_synthetic_model(path, ns, 3 if ns == 6 else 0)
n = 240 * 44100
audio = (0.1 * np.random.default_rng(6).standard_normal((2, n)) + 0.02).astype(np.float32)
m = dmx.Model(path); ctx = dmx.Context(m, 0, 6)
ref = ctx.track(audio, 4033)
ref2 = ctx.track(audio, 4033)
print("ref run-to-run equal:", np.array_equal(ref, ref2))
ctx.close(); m.close()
got = np.zeros_like(ref)
eng = dmx.Engine([path], [0] * 8, max_batch=6)
for rep in range(3):
    for fin, nm in ((dmx.FINISH_ROOT, "ROOT"), (dmx.FINISH_OWNER, "OWNER")):
        eng.set_finish(fin)
        out = eng.track(audio, [4033], out=got)
        d = out != ref
        if d.any():
            idx = np.argwhere(d)
            seg_stride = 257985
            pos = idx[:, 2]
            print(rep, nm, "MISMATCH", d.sum(), "stems", sorted(set(idx[:, 0])), "ch", sorted(set(idx[:, 1])), "sample range", pos.min(), pos.max(),
                  "segments", sorted(set((pos // seg_stride).tolist()))[:20], "maxabs", float(np.abs(out - ref).max()))
        else:
            print(rep, nm, "equal")
eng.close()

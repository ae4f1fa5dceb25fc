#!/usr/bin/env python
"""A/B of the linear-layer split kernels (DMX_SPLIT_LIN = 0 staged / 1 default: fragments in registers, 128 x 256 tile where
it pays / 2 never the 128 x 256 tile / 3 the 128 x 256 tile wherever it exists): every mode must produce the same bits. `run <out.npz>` writes the outputs of this process's mode; `cmp a.npz b.npz` compares."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ok = True
    for k in a.files:
        eq = np.array_equal(a[k], b[k])
        ok &= eq
        print(f"{sys.argv[2]} vs {sys.argv[3]} {k}: bitwise equal = {eq}" + ("" if eq else f" (max abs diff {np.abs(a[k] - b[k]).max():.3e}, {np.count_nonzero(a[k] != b[k])} of {a[k].size})"))
    sys.exit(0 if ok else 1)

from demucs_cpp_amd import binding as dmx  # noqa: E402
from demucs_cpp_amd.weights import write_synthetic_model  # noqa: E402

out = {}
for which in ("4", "6"):
    path = f"/tmp/linab_{which}.bin"
    if not os.path.exists(path):
        write_synthetic_model(path, int(which), 0 if which == "4" else 3)
    m = dmx.Model(path)
    stride = 257985
    n = 5 * stride + 1234
    audio = (0.1 * np.random.default_rng(41).standard_normal((2, n)) + 0.02).astype(np.float32)
    for mb in (1, 6):
        ctx = dmx.Context(m, 0, mb, gemm=dmx.GEMM_BF16X3)
        out[f"{which}s_track_b{mb}"] = ctx.track(audio, 4033)
        ctx.close()
    print(which, "b1 == b6:", np.array_equal(out[f"{which}s_track_b1"], out[f"{which}s_track_b6"]), flush=True)
np.savez(sys.argv[2], **out)

#!/usr/bin/env python
"""sha256 of the stems of a fixed synthetic input (env: SEG, TB = batch, MODEL = 4s|6s): one line per process, for A/B runs of
environment switches that must not change a bit (DMX_LIN_DEPTH, DMX_KV_PLANES, DMX_STREAMS ...)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
ns = 6 if os.environ.get("MODEL", "4s") == "6s" else 4
path = f"/tmp/out_hash_{ns}.bin"
write_synthetic_model(path, ns, 3 if ns == 6 else 0)
seg, B = int(os.environ.get("SEG", "16384")), int(os.environ.get("TB", "3"))
m = dmx.Model(path)
c = dmx.Context(m, seg, B)
mixes = (0.1 * np.random.default_rng(5).standard_normal((B, seg, 2))).astype(np.float32)
d_mix = torch.from_numpy(mixes).cuda()
d_out = torch.zeros((B, ns, 2, seg), device="cuda")
torch.cuda.synchronize()
c.segment_device(d_mix.data_ptr(), d_out.data_ptr(), B)
c.synchronize()
o = d_out.cpu().numpy()
print("stems sha256", hashlib.sha256(o.tobytes()).hexdigest()[:16], "finite", bool(np.isfinite(o).all()), "absmax %.6f" % float(np.abs(o).max()), flush=True)

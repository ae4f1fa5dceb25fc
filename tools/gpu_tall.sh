#!/bin/bash
# EXPERIMENT: the double-height igemm tile (cfg 17, 256x128, 8 waves, one workgroup per CU; DMX_TALL=1) against the
# 128x128 tile: bit-equality at batch 12, then per-op profile of both at batch 42
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -60
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
write_synthetic_model('/tmp/pm.bin', 4, 0)
m = dmx.Model('/tmp/pm.bin')
PB = 12
mix = (0.1 * np.random.default_rng(3).standard_normal((PB, 2, 343980))).astype(np.float32)
outs = {}
for tall in ("0", "1"):
    os.environ["DMX_TALL"] = tall
    ctx = dmx.Context(m, 0, PB)
    outs[tall] = np.stack([ctx.segment(mix[b]) for b in range(1)] ) if False else None
    d_mix = torch.from_numpy(np.ascontiguousarray(mix.transpose(0, 2, 1))).cuda()
    d_out = torch.zeros(PB, 4, 2, 343980, device='cuda')
    ctx.segment_device(d_mix.data_ptr(), d_out.data_ptr(), PB); ctx.synchronize()
    outs[tall] = d_out.cpu().numpy()
    ctx.close()
print("tall == default bitwise at batch 12:", np.array_equal(outs["0"], outs["1"]), float(np.abs(outs["0"]).max()))
res = {}
for tall in ("0", "1"):
    os.environ["DMX_TALL"] = tall
    ctx = dmx.Context(m, 0, 42)
    prof = ctx.profile(42, 5)
    res[tall] = {r[0]: r for r in prof}
    tot = sum(r[2] for r in prof)
    print(f"DMX_TALL={tall}: sum of ops {tot:.3f} ms = {tot/42:.4f} ms/segment")
    ctx.close()
a, b = res["0"], res["1"]
ta = tb = 0
for n in a:
    if b[n][1] != a[n][1] or b[n][1] == "igemm_256x128":
        fa = a[n][3] / a[n][2] / 1e9; fb = b[n][3] / b[n][2] / 1e9
        ta += a[n][2]; tb += b[n][2]
        print(f"{n:40s} {a[n][1]:14s} {a[n][2]*1e3:8.1f} us {fa:6.1f} TF/s -> {b[n][1]:14s} {b[n][2]*1e3:8.1f} us {fb:6.1f} TF/s")
print(f"changed ops: {ta:.3f} ms -> {tb:.3f} ms")
PY
echo "== weight sweep tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "weight_scale" 2>&1 | grep -E "gain|passed|failed|Error" | tail -12

#!/bin/bash
# A/B of the 256x128 four-wave kernel for linear layers (igemm_lin256.hip; DMX_LIN256=0 keeps them on the 128x128 tile):
# bit-equality of a full-size batch (4s, 6s), then the per-op profile at 42 and 12 segments
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -60
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
for ns in (4, 6):
    write_synthetic_model(f'/tmp/pm{ns}.bin', ns, 0)
    m = dmx.Model(f'/tmp/pm{ns}.bin')
    PB = 26
    mix = (0.1 * np.random.default_rng(3).standard_normal((PB, 343980, 2))).astype(np.float32)
    outs = {}
    for on in ("0", "1"):
        os.environ["DMX_LIN256"] = on
        ctx = dmx.Context(m, 0, PB)
        d_mix = torch.from_numpy(mix).cuda(); d_out = torch.zeros(PB, ns, 2, 343980, device='cuda')
        ctx.segment_device(d_mix.data_ptr(), d_out.data_ptr(), PB); ctx.synchronize()
        outs[on] = d_out.cpu().numpy()
        ctx.close()
    print(f"{ns}s batch {PB}: lin256 == 128x128 bitwise:", np.array_equal(outs["0"], outs["1"]), float(np.abs(outs["0"]).max()), bool(np.isfinite(outs["1"]).all()))
    if not np.array_equal(outs["0"], outs["1"]):
        d = np.abs(outs["0"] - outs["1"]); print("   max abs diff", float(d.max()), "rel", float(d.max() / np.abs(outs["0"]).max()))
    m.close()
m = dmx.Model('/tmp/pm4.bin')
res = {}
for on in ("0", "1"):
    os.environ["DMX_LIN256"] = on
    for PB in (42, 12):
        ctx = dmx.Context(m, 0, PB)
        prof = ctx.profile(PB, 5)
        res[on, PB] = {r[0]: r for r in prof}
        print(f"DMX_LIN256={on} batch {PB}: sum of ops {sum(r[2] for r in prof):.3f} ms")
        ctx.close()
for PB in (42, 12):
    a, b = res["0", PB], res["1", PB]
    ta = tb = 0; seen = set()
    for n in a:
        if b[n][1] == "igemm_lin256x128":
            ta += a[n][2]; tb += b[n][2]
            key = (n.split(".")[-1] + ("_t" if "layers_t" in n else "")) if "crosstransformer" in n else n
            if key in seen: continue
            seen.add(key)
            f = lambda r: r[3] / r[2] / 1e9
            print(f"b{PB} {n:40s} {a[n][1]:14s} {a[n][2]*1e3:8.1f} us {f(a[n]):6.1f} TF/s -> {b[n][2]*1e3:8.1f} us {f(b[n]):6.1f} TF/s")
    print(f"b{PB} ops on the new kernel: {ta:.3f} ms -> {tb:.3f} ms")
PY

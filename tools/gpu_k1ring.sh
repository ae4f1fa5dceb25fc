#!/bin/bash
# EXPERIMENT: DConv K1 on the frequency branch with one read per input row (register ring along time, DMX_K1_RING=1)
# against the generic direct kernel (=0): bit-equality at batch 3 and 42, then the K1 ops of the per-op profile
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -60
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
for arch, nm in (("v4", "/tmp/pm4.bin"), ("v3", "/tmp/pm3.bin")):
    write_synthetic_model(nm, 4, 0, 'default', arch)
for nm in ("/tmp/pm4.bin", "/tmp/pm3.bin"):
    m = dmx.Model(nm)
    for PB, seg in ((3, 343980), (2, 30002)):
        mix = (0.1 * np.random.default_rng(3).standard_normal((PB, seg, 2))).astype(np.float32)
        outs = {}
        for ring in ("0", "1"):
            os.environ["DMX_K1_RING"] = ring
            ctx = dmx.Context(m, seg, PB)
            d_mix = torch.from_numpy(mix).cuda()
            d_out = torch.zeros(PB, 4, 2, seg, device='cuda')
            ctx.segment_device(d_mix.data_ptr(), d_out.data_ptr(), PB); ctx.synchronize()
            outs[ring] = d_out.cpu().numpy()
            ctx.close()
        print(nm, "batch", PB, "seg", seg, "ring == generic bitwise:", np.array_equal(outs["0"], outs["1"]), float(np.abs(outs["0"]).max()))
    m.close()
m = dmx.Model("/tmp/pm4.bin")
res = {}
for ring in ("0", "1"):
    os.environ["DMX_K1_RING"] = ring
    for PB in (1, 42):
        ctx = dmx.Context(m, 0, PB)
        prof = ctx.profile(PB, 5)
        res[ring, PB] = {r[0]: r for r in prof}
        print(f"DMX_K1_RING={ring} batch {PB}: sum of ops {sum(r[2] for r in prof):.3f} ms")
        ctx.close()
for PB in (1, 42):
    a, b = res["0", PB], res["1", PB]
    ta = tb = 0
    for n in a:
        if n.endswith(".k1") and a[n][1] == "dgemm_direct":
            ta += a[n][2]; tb += b[n][2]
            print(f"b{PB} {n:28s} {a[n][2]*1e3:8.1f} us {a[n][4]/a[n][2]/1e9:6.2f} TB/s -> {b[n][2]*1e3:8.1f} us {b[n][4]/b[n][2]/1e9:6.2f} TB/s")
    print(f"b{PB} K1 ops: {ta:.3f} ms -> {tb:.3f} ms")
PY

import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
for ns, arch in ((4, "v4"), (6, "v4"), (4, "v3")):
    write_synthetic_model('/tmp/pm.bin', ns, 0, 'default', arch)
    m = dmx.Model('/tmp/pm.bin')
    PB = 26
    mix = (0.1 * np.random.default_rng(3).standard_normal((PB, 343980, 2))).astype(np.float32)
    outs = {}
    for on in ("0", "1"):
        os.environ["DMX_SHORTK"] = on
        ctx = dmx.Context(m, 0, PB)
        d_mix = torch.from_numpy(mix).cuda(); d_out = torch.zeros(PB, ns, 2, 343980, device='cuda')
        ctx.segment_device(d_mix.data_ptr(), d_out.data_ptr(), PB); ctx.synchronize()
        outs[on] = d_out.cpu().numpy()
        if ns == 4: print(arch, "DMX_SHORTK", on, "sum of ops", round(sum(r[2] for r in ctx.profile(PB, 3)), 3), [r[0] for r in ctx.profile(PB,1) if r[1]=="igemm_256x96"])
        ctx.close()
    print(f"{arch} {ns}s batch {PB}: short-K tile == 128x96 bitwise:", np.array_equal(outs["0"], outs["1"]), bool(np.isfinite(outs["1"]).all()))
    m.close()

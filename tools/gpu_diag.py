#!/usr/bin/env python
"""First-contact GPU diagnostic: runs the HIP path against the oracle at reduced and full
size, prints per-tap errors, timings and the per-op profile. Not a test (see tests/)."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc  # noqa: E402
import parity_utils as pu  # noqa: E402
from demucs_cpp_amd import binding as dmx  # noqa: E402
from demucs_cpp_amd.weights import write_synthetic_model  # noqa: E402


def section(t):
    print("\n==== " + t, flush=True)


def main():
    full = "--no-full" not in sys.argv
    print("devices:", dmx.device_count())
    paths = {4: "/tmp/diag_m4.bin", 6: "/tmp/diag_m6.bin"}
    write_synthetic_model(paths[4], 4, 0)
    write_synthetic_model(paths[6], 6, 3)
    for ns, seg in ((4, 10000), (6, 6000)):
        section(f"{ns}s reduced segment seg={seg}")
        try:
            g = np.load(os.path.join(ROOT, "tests", "golden", f"golden_seg_{ns}s.npz"))
            m = dmx.Model(paths[ns])
            om = orc.OracleModel(paths[ns])
            ctx = dmx.Context(m, seg, 2)
            errs, out, ref = pu.compare_segment(ctx, om, g["mix"])
            for k, v in errs.items():
                print(f"  {k:8s} relerr {v:.3e}")
            print("  vs fp64 golden:", pu.relerr(out, g["out"]))
            # batch of 2 through the device API equals two single runs
            import torch
            rng = np.random.default_rng(5)
            mix2 = (0.1 * rng.standard_normal((2, 2, seg))).astype(np.float32)
            dm = torch.from_numpy(np.ascontiguousarray(mix2.transpose(0, 2, 1))).cuda()
            do = torch.zeros((2, ns, 2, seg), device="cuda")
            ctx.segment_device(dm.data_ptr(), do.data_ptr(), 2)
            ctx.synchronize()
            o2 = do.cpu().numpy()
            s0, s1 = ctx.segment(mix2[0]), ctx.segment(mix2[1])
            print("  batch2 vs single maxabs:", np.abs(o2[0] - s0).max(), np.abs(o2[1] - s1).max())
            # track level
            n = 3 * seg + 1234
            audio = (0.1 * rng.standard_normal((2, n))).astype(np.float32)
            t0 = time.time()
            rt = om.track(audio, 4033, seg)
            t1 = time.time()
            gt = ctx.track(audio, 4033)
            print(f"  track n={n}: relerr {pu.relerr(gt, rt):.3e} (oracle {t1 - t0:.1f}s)")
            ctx.close(); m.close(); om.close()
        except Exception:
            traceback.print_exc()
    if full:
        section("4s full segment 343980")
        try:
            m = dmx.Model(paths[4])
            om = orc.OracleModel(paths[4])
            ctx = dmx.Context(m, 0, 1)
            print("  arena MB:", ctx.arena_bytes / 1e6)
            rng = np.random.default_rng(0)
            mix = (0.1 * rng.standard_normal((2, 343980))).astype(np.float32)
            t0 = time.time()
            errs, out, ref = pu.compare_segment(ctx, om, mix)
            print(f"  oracle+gpu wall {time.time() - t0:.1f}s, oracle threads {orc.lib().orc_num_threads()}")
            for k, v in errs.items():
                print(f"  {k:8s} relerr {v:.3e}")
            for _ in range(2):
                ctx.segment(mix)
            t0 = time.time()
            for _ in range(5):
                ctx.segment(mix)
            dt = (time.time() - t0) / 5
            print(f"  GPU segment incl. H2D/D2H: {dt * 1e3:.2f} ms -> {7.8 / dt:.1f}x RT")
            prof = ctx.profile(1, 3)
            tot = sum(r[2] for r in prof)
            print(f"  per-op total {tot:.2f} ms over {len(prof)} ops; top 30:")
            for nm, k, t, fl, by in sorted(prof, key=lambda x: -x[2])[:30]:
                print(f"    {t:8.3f} ms {fl / t / 1e9 if t > 0 else 0:8.1f} TF/s {by / t / 1e6 if t > 0 else 0:8.1f} GB/s {k:14s} {nm}")
            with open(os.path.join(ROOT, "gpurun_out", "profile_ops.txt"), "w") as f:
                for r in prof:
                    f.write("\t".join(str(x) for x in r) + "\n")
        except Exception:
            traceback.print_exc()


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()

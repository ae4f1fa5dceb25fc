#!/usr/bin/env python
"""Secondary measurements quoted in DESIGN.md (not the bench.py headline):
  * BASELINE.json configs[1] read literally: ONE 7.8 s segment per call (latency, batch 1), host
    buffers in and out (PCIe inclusive) and device-resident;
  * configs[2]: a ~4-minute track (10 584 000 samples, 42 segments, shift offset 4033) through
    dmx_track_infer, host buffers in and out: H2D of the track, normalisation, segment gather,
    all segments in batches, overlap-add, D2H of (S, 2, N). Weight load and WAV I/O excluded.
Synthetic weights (seed 0) and 0.1*N(0,1) audio, fp32."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from demucs_cpp_amd import binding as dmx  # noqa: E402
from demucs_cpp_amd.weights import write_synthetic_model  # noqa: E402

SEG = 343980


def main():
    ns = int(os.environ.get("NS", "4"))
    mb = int(os.environ.get("MAXBATCH", "24"))
    path = f"/tmp/track_bench_{ns}s.bin"
    write_synthetic_model(path, ns, 0 if ns == 4 else 3)
    m = dmx.Model(path)
    rng = np.random.default_rng(1)
    res = {"model": f"htdemucs-{ns}s synthetic", "max_batch": mb}

    # ---- one segment per call
    c1 = dmx.Context(m, SEG, 1)
    mix = (0.1 * rng.standard_normal((2, SEG))).astype(np.float32)
    for _ in range(3):
        c1.segment(mix)
    t0 = time.perf_counter()
    R = 10
    for _ in range(R):
        c1.segment(mix)
    dt = (time.perf_counter() - t0) / R
    res["single_segment_host_ms"] = round(dt * 1e3, 3)
    res["single_segment_host_xRT"] = round(7.8 / dt, 1)
    dm = torch.from_numpy(np.ascontiguousarray(mix.T)).cuda()
    do = torch.zeros((ns, 2, SEG), device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        c1.segment_device(dm.data_ptr(), do.data_ptr(), 1)
    c1.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        c1.segment_device(dm.data_ptr(), do.data_ptr(), 1)
        c1.synchronize()
    dt = (time.perf_counter() - t0) / R
    res["single_segment_device_ms"] = round(dt * 1e3, 3)
    res["single_segment_device_xRT"] = round(7.8 / dt, 1)
    c1.close()

    # ---- 4-minute track
    n = 240 * 44100
    audio = (0.1 * rng.standard_normal((2, n))).astype(np.float32)
    ctx = dmx.Context(m, SEG, mb)
    _, nseg, _ = ctx.track_geometry(n, 4033)
    ctx.track(audio[:, : 3 * SEG], 4033)  # warm-up (plans, allocator)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = ctx.track(audio, 4033)
        ts.append(time.perf_counter() - t0)
    dt = min(ts)
    res["track_samples"] = n
    res["track_segments"] = nseg
    res["track_wall_s"] = [round(t, 4) for t in ts]
    res["track_xRT_pcie_inclusive"] = round(240.0 / dt, 1)
    res["track_outputs_finite"] = bool(np.isfinite(out).all())
    res["host_bytes_in_out_MB"] = round((audio.nbytes + out.nbytes) / 1e6, 1)
    print(json.dumps(res))
    ctx.close()
    m.close()


if __name__ == "__main__":
    main()

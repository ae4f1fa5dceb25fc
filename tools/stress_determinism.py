#!/usr/bin/env python
"""Run-to-run determinism stress at the bench workload: the same batch of full-size segments is pushed
through the hot path N times; every output must equal the first one bit for bit (a latent LDS race in the
interleaved K loop or a missing cross-stream join would show up as a mismatch). Also 4 segments (two-stream
mode) and the 6-source model."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demucs_cpp_amd import binding as dmx  # noqa: E402
from demucs_cpp_amd.weights import write_synthetic_model  # noqa: E402

SEG = 343980
N = int(os.environ.get("N", "25"))
bad = 0
for ns, B in ((4, 24), (4, 4), (4, 1), (6, 12)):  # 4 and 1: two-stream mode, graph replay, half-height tiles
    path = f"/tmp/stress_{ns}.bin"
    write_synthetic_model(path, ns, ns)
    m = dmx.Model(path)
    ctx = dmx.Context(m, SEG, B)
    g = torch.Generator().manual_seed(ns * 100 + B)
    mix = (0.1 * torch.randn((B, SEG, 2), generator=g)).cuda()
    out = torch.zeros((B, ns, 2, SEG), device="cuda")
    torch.cuda.synchronize()
    ctx.segment_device(mix.data_ptr(), out.data_ptr(), B)
    ctx.synchronize()
    ref = out.clone()
    mism = 0
    for i in range(N):
        out.zero_()
        torch.cuda.synchronize()
        ctx.segment_device(mix.data_ptr(), out.data_ptr(), B)
        ctx.synchronize()
        if not torch.equal(out, ref):
            mism += 1
    print(f"model {ns}s batch {B}: {N} repeats, {mism} mismatching, finite={bool(torch.isfinite(ref).all())}")
    bad += mism
    ctx.close(); m.close()
sys.exit(1 if bad else 0)

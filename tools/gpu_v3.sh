#!/bin/bash
# Demucs v3 GPU iteration: v3 parity tests, then per-op profile and a batch timing of the v3 plan
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_v3.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/pytest_v3.log
( timeout 600 python - <<'PY' 2>&1 | tail -80
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
write_synthetic_model('/tmp/pm3.bin', 4, 5, 'default', 'v3')
m = dmx.Model('/tmp/pm3.bin')
for PB in (1, 24):
    ctx = dmx.Context(m, 0, PB)
    prof = ctx.profile(PB, 3)
    with open(f'gpurun_out/profile_ops_v3_b{PB}.tsv', 'w') as f:
        for r in prof: f.write('\t'.join(str(x) for x in r) + '\n')
    agg = {}
    for nm, k, ms, fl, by in prof:
        d = agg.setdefault(k, [0, 0, 0, 0]); d[0] += ms; d[1] += fl; d[2] += by; d[3] += 1
    tot = sum(v[0] for v in agg.values())
    print(f'== v3 batch {PB}: sum of ops {tot:.3f} ms = {tot/PB:.3f} ms/segment')
    for k, (ms, fl, by, n) in sorted(agg.items(), key=lambda x: -x[1][0]):
        print(f'{k:16s} n={n:3d} {ms:8.3f} ms {100*ms/tot:5.1f}%  {fl/ms/1e9 if ms else 0:7.1f} TF/s {by/ms/1e6 if ms else 0:8.1f} GB/s')
    # wall time of whole batches, inputs resident
    seg = ctx.seg
    d_mix = (0.1 * torch.randn(PB, seg, 2, device='cuda'))
    d_out = torch.zeros(PB, 4, 2, seg, device='cuda')
    for _ in range(3):
        ctx.segment_device(d_mix.data_ptr(), d_out.data_ptr(), PB)
    ctx.synchronize()
    t0 = time.perf_counter(); n = 5
    for _ in range(n):
        ctx.segment_device(d_mix.data_ptr(), d_out.data_ptr(), PB)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'== v3 batch {PB}: {dt*1e3:.3f} ms per call = {dt*1e3/PB:.3f} ms/segment = {PB*seg/44100/dt:.1f} x RT (segment seconds)')
    ctx.close()
PY
) > gpurun_out/profile_v3.log
echo ---- pytest v3; cat gpurun_out/pytest_v3.log
echo ---- profile v3; cat gpurun_out/profile_v3.log

#!/bin/bash
# per-op HIP-event profile of a model (MODEL=4s|6s|v3) at the batch sizes in $PBS (default "1 42"), GEMM mode from DMX_GEMM
# (default: the library default, bf16x3): gpurun_out/profile_ops_<model>_b<B>.tsv
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
MODEL=${MODEL:-4s} PBS="${PBS:-1 42}" timeout 900 python - <<'PY' 2>&1 | tail -120
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
model = os.environ["MODEL"]
write_synthetic_model('/tmp/pm.bin', 6 if model == "6s" else 4, 3 if model == "6s" else 0, 'default', 'v3' if model == 'v3' else 'v4')
m = dmx.Model('/tmp/pm.bin')
for PB in [int(x) for x in os.environ["PBS"].split()]:
    ctx = dmx.Context(m, 0, PB)
    prof = ctx.profile(PB, 5)
    with open(f'gpurun_out/profile_ops_{model}_b{PB}.tsv', 'w') as f:
        for r in prof: f.write('\t'.join(str(x) for x in r) + '\n')
    agg = {}
    for nm, k, ms, fl, by in prof:
        d = agg.setdefault(k, [0, 0, 0, 0]); d[0] += ms; d[1] += fl; d[2] += by; d[3] += 1
    tot = sum(v[0] for v in agg.values())
    print(f'== {model} batch {PB}: sum of ops {tot:.3f} ms = {tot/PB:.3f} ms/segment')
    for k, (ms, fl, by, n) in sorted(agg.items(), key=lambda x: -x[1][0]):
        print(f'{k:16s} n={n:3d} {ms:8.3f} ms {100*ms/tot:5.1f}%  {fl/ms/1e9 if ms else 0:7.1f} TF/s {by/ms/1e6 if ms else 0:8.1f} GB/s')
    seg = ctx.seg
    S = m.n_sources
    d_mix = (0.1 * torch.randn(PB, seg, 2, device='cuda'))
    d_out = torch.zeros(PB, S, 2, seg, device='cuda')
    for _ in range(3):
        ctx.segment_device(d_mix.data_ptr(), d_out.data_ptr(), PB); ctx.synchronize()
    t0 = time.perf_counter(); n = 10
    for _ in range(n):
        ctx.segment_device(d_mix.data_ptr(), d_out.data_ptr(), PB); ctx.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'== {model} batch {PB}: {dt*1e3:.3f} ms per call = {dt*1e3/PB:.3f} ms/segment')
    ctx.close()
PY

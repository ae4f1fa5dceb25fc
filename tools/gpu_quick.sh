#!/bin/bash
# Quick GPU iteration: parity tests + bench + per-op profile dump at batch $PB (gpurun_out/profile_ops.tsv)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/bench.log
for once in 0; do
( timeout 600 python - <<'PY' 2>&1 | tail -60
import sys, os
sys.path.insert(0, os.getcwd())
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
write_synthetic_model('/tmp/pm4.bin', 4, 0)
m = dmx.Model('/tmp/pm4.bin'); ctx = dmx.Context(m, 0, int(os.environ.get("PB","4")))
prof = ctx.profile(int(os.environ.get("PB","4")), 3)
with open('gpurun_out/profile_ops.tsv', 'w') as f:
    for r in prof: f.write('\t'.join(str(x) for x in r) + '\n')
agg = {}
for nm, k, ms, fl, by in prof:
    d = agg.setdefault(k, [0, 0, 0, 0]); d[0] += ms; d[1] += fl; d[2] += by; d[3] += 1
tot = sum(v[0] for v in agg.values())
print(f'total {tot:.3f} ms per batch = {tot/int(os.environ.get("PB","4")):.3f} ms/segment')
for k, (ms, fl, by, n) in sorted(agg.items(), key=lambda x: -x[1][0]):
    print(f'{k:16s} n={n:3d} {ms:8.3f} ms {100*ms/tot:5.1f}%  {fl/ms/1e9 if ms else 0:7.1f} TF/s {by/ms/1e6 if ms else 0:8.1f} GB/s')
PY
) ; done > gpurun_out/profile_summary.log
echo ---- pytest; cat gpurun_out/pytest_gpu.log
echo ---- bench; cat gpurun_out/bench.log
echo ---- profile; cat gpurun_out/profile_summary.log

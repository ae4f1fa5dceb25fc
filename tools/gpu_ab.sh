#!/bin/bash
# A/B of environment-selected kernel variants: tools/gpu_ab.sh "DMX_IGEMM_WS=1" "DMX_IGEMM_WS=0" ...
# per variant: one small parity test (hang guard), then the per-kernel profile at batch $PB (default 12)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for v in "$@"; do
echo "=== $v"
( export $v; timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${ABTEST:-reduced}" 2>&1 | tail -3
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os
sys.path.insert(0, os.getcwd())
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
PB = int(os.environ.get("PB", "12"))
write_synthetic_model('/tmp/pm4.bin', 4, 0)
m = dmx.Model('/tmp/pm4.bin'); ctx = dmx.Context(m, 0, PB)
prof = ctx.profile(PB, 3)
agg = {}
for nm, k, ms, fl, by in prof:
    d = agg.setdefault(k, [0, 0, 0, 0]); d[0] += ms; d[1] += fl; d[2] += by; d[3] += 1
tot = sum(v[0] for v in agg.values())
print(f'total {tot:.3f} ms per batch = {tot/PB:.3f} ms/segment')
for k, (ms, fl, by, n) in sorted(agg.items(), key=lambda x: -x[1][0]):
    print(f'{k:16s} n={n:3d} {ms:8.3f} ms {100*ms/tot:5.1f}%  {fl/ms/1e9 if ms else 0:7.1f} TF/s {by/ms/1e6 if ms else 0:8.1f} GB/s')
if os.environ.get("ABOPS"):
    for nm, k, ms, fl, by in prof:
        if any(s in nm for s in os.environ["ABOPS"].split(",")):
            print(f'   {nm:44s} {k:14s} {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF/s {by/ms/1e6:8.1f} GB/s')
PY
)
done

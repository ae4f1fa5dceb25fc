#!/bin/bash
# per-op times of the direct kernels under each fragment-pipeline variant (libs built with -DDMX_DG_PIPE=m)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "" _dg1 _dg2; do
DMX_LIB=$R/demucs_cpp_amd/lib/libdemucs_hip$v.so timeout 600 python - "$v" <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os
sys.path.insert(0, os.getcwd())
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
write_synthetic_model('/tmp/pm4.bin', 4, 0)
m = dmx.Model('/tmp/pm4.bin'); ctx = dmx.Context(m, 0, 12)
prof = ctx.profile(12, 3)
tot = sum(r[2] for r in prof if r[1] == 'dgemm_direct')
print('VARIANT', sys.argv[1] or 'pipe0', 'dgemm total %.3f ms' % tot)
for nm, k, ms, fl, by in prof:
    if k == 'dgemm_direct' and not ('dconv1' in nm):
        print('  %-34s %8.1f' % (nm, ms * 1e3))
PY
done

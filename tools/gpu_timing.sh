#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "$@"; do
echo "=== variant $v"
DMX_LIB=$R/demucs_cpp_amd/lib/libdemucs_hip_$v.so timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os
sys.path.insert(0, os.getcwd())
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
write_synthetic_model('/tmp/pm4.bin', 4, 0)
m = dmx.Model('/tmp/pm4.bin'); ctx = dmx.Context(m, 0, 4)
prof = {r[0]: r for r in ctx.profile(4, 3)}
for op in ['crosstransformer.layers.0.linear1', 'crosstransformer.layers.0.linear2', 'crosstransformer.layers.0.qkv', 'decoder.0.rewrite', 'decoder.2.conv_tr', 'encoder.1.conv']:
    t = dmx.igemm_timing(ctx, 4, op)
    nm, k, ms, fl, by = prof[op]
    ph = '' if t is None else ' cyc/tile: loads %.0f mfma %.0f store %.0f addr %.0f barrier %.0f (tiles %.0f) sum %.0f' % (t[0], t[1], t[2], t[3], t[4], t[5], sum(t[:5]))
    print(f'{op:38s} {k:14s} {ms*1e3:7.1f} us {fl/ms/1e9:6.1f} TF/s |{ph}')
PY
done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "reduced_segment or full_size_segment or batch_equals or awkward or deterministic_inputs or track_vs_oracle" 2>&1 | tail -8 ) > gpurun_out/pytest_quick.log
cat gpurun_out/pytest_quick.log
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
for ns, seg in ((4, 343980), (6, 40000), (4, 4098)):
    p = f'/tmp/f{ns}.bin'; write_synthetic_model(p, ns, 1)
    m = dmx.Model(p)
    mix = (0.1*np.random.default_rng(3).standard_normal((2, seg))).astype(np.float32)
    os.environ['DMX_FUSE_ISTFT'] = '1'; c1 = dmx.Context(m, seg, 1); a = c1.segment(mix); c1.close()
    os.environ['DMX_FUSE_ISTFT'] = '0'; c0 = dmx.Context(m, seg, 1); b = c0.segment(mix); c0.close()
    print(ns, seg, 'fused == two-kernel bitwise:', np.array_equal(a, b), 'max diff', float(np.abs(a-b).max()))
    m.close()
PY
( timeout 300 python tools/prof_ops.py fuse1 2>&1 | grep -v amdgpu.ids | grep -E "^\[|istft|ola|stft" ) 
( DMX_FUSE_ISTFT=0 timeout 300 python tools/prof_ops.py fuse0 2>&1 | grep -v amdgpu.ids | grep -E "^\[|istft|ola|stft" ) 

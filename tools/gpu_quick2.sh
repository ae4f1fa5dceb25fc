#!/bin/bash
# quick iteration: selected parity tests + per-op profile (label $1)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "${TESTS:-reduced_segment or full_size_segment or batch_equals or awkward}" 2>&1 | tail -15 ) > gpurun_out/pytest_quick.log
( timeout 300 python tools/prof_ops.py ${1:-cur} 2>&1 | grep -v amdgpu.ids ) > gpurun_out/prof_${1:-cur}.log
cat gpurun_out/pytest_quick.log; cat gpurun_out/prof_${1:-cur}.log

#!/bin/bash
# EXPERIMENT: the exact-split bf16 igemm (DMX_GEMM=bf16x3) - parity on the v4 tests, then per-op profile next to fp32
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== parity with DMX_GEMM=bf16x3"
DMX_GEMM=bf16x3 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reduced_segment or full_size_segment or batch_equals or stress" 2>&1 | tail -8
for g in f32 bf16x3; do
  echo "== profile DMX_GEMM=$g"
  DMX_GEMM=$g MODEL=4s PBS="${PBS:-42}" bash tools/gpu_prof.sh 2>&1 | grep -E "^==|igemm|attention|dgemm"
  cp gpurun_out/profile_ops_4s_b42.tsv gpurun_out/profile_ops_4s_b42_$g.tsv
done

#!/bin/bash
# round 4, run F: deep-level DConv k3 in column chunks, A/B of the chunk width (one launch, grid row = chunk)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for c in 0 96 192; do
  ( DMX_K3_CHUNKS=$([ $c = 0 ] && echo 0 || echo 1) DMX_K3_CHUNK=$c PB=42 REPS=3 timeout 200 python tools/prof_ops.py r4f_k3_$c 2>&1 | grep -E "total|dgemm|igemm_128x128" ) > gpurun_out/r4f_k3_$c.log
  echo "== chunk $c"; cat gpurun_out/r4f_k3_$c.log
  grep -E "(encoder.2|encoder.3|tencoder.2|tencoder.3).dconv0.k3" gpurun_out/ops_r4f_k3_$c.tsv | awk -F'\t' '{printf "   %-26s %-14s %8.4f ms\n",$1,$2,$3}'
done
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemm_modes or reduced_segment or full_size_segment" 2>&1 | tail -3 ) > gpurun_out/r4f_pytest.log; cat gpurun_out/r4f_pytest.log

#!/bin/bash
# EXPERIMENT: attention workgroup shape (64 vs 128 queries) at small batches; double-height igemm tile incl. the linears
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for v in "" "DMX_ATT_BIG=1" "DMX_ATT_SMALL=1"; do
  echo "== $v"
  env $v MODEL=4s PBS="1 2 4" bash tools/gpu_prof.sh 2>&1 | grep -E "^==|attention"
done
echo "== DMX_TALL=1 batch 42"
DMX_TALL=1 MODEL=4s PBS="42" bash tools/gpu_prof.sh 2>&1 | grep -E "^==|igemm_|attention"
grep -P "igemm_256x128" gpurun_out/profile_ops_4s_b42.tsv | awk -F'\t' '{printf "%s %.1f us %.1f TF/s\n",$1,$3*1000,$4/$3/1e9}'

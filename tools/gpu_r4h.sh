#!/bin/bash
# round 4, run H: machine model of the split loop (tools/micro/mfma_mix)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( MIX_SET=4 timeout 400 tools/micro/mfma_mix 2>&1 ) > gpurun_out/r4h_mfma_mix4.log
cat gpurun_out/r4h_mfma_mix4.log

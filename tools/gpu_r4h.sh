#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( MIX_SET=${MIX_SET:-7} timeout 400 tools/micro/mfma_mix 2>&1 ) > gpurun_out/r4h_mfma_mix_set${MIX_SET:-7}.log
cat gpurun_out/r4h_mfma_mix_set${MIX_SET:-7}.log

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( timeout 40 tools/micro/gemm_fp16x3 112896 2048 512 ) > gpurun_out/r4o_gemm_fp16x3_linear1.log 2>&1; cat gpurun_out/r4o_gemm_fp16x3_linear1.log

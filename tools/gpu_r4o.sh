#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( timeout 30 tools/micro/gemm_fp16x3 32768 512 2048; timeout 30 tools/micro/gemm_fp16x3 16384 2048 4096 ) > gpurun_out/r4o_gemm_fp16x3_k2048.log 2>&1; cat gpurun_out/r4o_gemm_fp16x3_k2048.log

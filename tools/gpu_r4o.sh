#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( timeout 60 tools/micro/split_fp16 2>&1 ) > gpurun_out/r4o_split_fp16.log; cat gpurun_out/r4o_split_fp16.log

#!/bin/bash
# ms per call over the batch sizes that matter (1: configs[1]; 5-6: one track over 8 GPUs; 42: the bench)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
MODEL=${MODEL:-4s} PBS="${PBS:-1 2 4 6 12 24 42}" bash tools/gpu_prof.sh 2>&1 | grep -E "^=="

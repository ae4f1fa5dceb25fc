#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "activation_split" 2>&1 | grep -E "^E|assert|Error" | cut -c1-400 | head -30 ) > gpurun_out/r4d_pytest2.log
cat gpurun_out/r4d_pytest2.log

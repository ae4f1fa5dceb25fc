cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_v3.py -m gpu -x -q 2>&1 | tail -4
for w in 4 8; do echo "== DMX_LSTM_WAVES=$w"; DMX_LSTM_WAVES=$w MODEL=v3 PBS="1 42" bash tools/gpu_prof.sh 2>&1 | grep -E "^==|lstm|local_attn|group_stats|dgemm|igemm_128x96 "; done

cd $GRAFT_REPO_ROOT
rm -f gpurun_out/ab/summary.txt
LIBS="product attpipe0 attnopipe product attpipe0 attnopipe" OPS="\.attn$" bash tools/gpu_ab_ops.sh
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "kv_operand_planes or batch_equals_singles" 2>&1 | tail -4 )

cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "kv_operand_planes or reduced_segment_all_layers or batch_equals_singles or full_size_segment" 2>&1 | tail -4 ) > gpurun_out/l3_tests.log
rm -f gpurun_out/ab/summary.txt
LIBS="product attnopipe product attnopipe" OPS="\.attn$" bash tools/gpu_ab_ops.sh
cat gpurun_out/l3_tests.log

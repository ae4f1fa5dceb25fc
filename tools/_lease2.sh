cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "row_resident or reduced_segment_all_layers or batch_equals_singles" 2>&1 | tail -4 ) > gpurun_out/l2_tests.log
rm -f gpurun_out/ab/summary.txt
LIBS="product rowold rowcopy rownomem rowstag4" OPS="^(encoder|decoder)\.[0-3]\.dconv$" bash tools/gpu_ab_ops.sh
LIBS="product linhalf lin64 lin64x3" OPS="linear|out_proj|\.qk|\.q$|\.k$|\.v$|sampler" bash tools/gpu_ab_ops.sh
# PMC traffic of the new dconv_row
GEMM=bf16x3 PMC_SQ=0 bash tools/gpu_pmc.sh 42 > gpurun_out/l2_pmc.log 2>&1
grep -A6 "dconv_row" gpurun_out/pmc/traffic.json | head -40
cat gpurun_out/l2_tests.log

mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reduced_segment or full_size_segment or batch_equals_singles or stress_models or bench_batch_and_awkward" > gpurun_out/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t1.log
PB=42 timeout 300 python tools/prof_ops.py row > gpurun_out/prof_row.log 2>&1
DMX_DCONV_ROW=0 PB=42 timeout 300 python tools/prof_ops.py chain > gpurun_out/prof_chain.log 2>&1
PB=1 timeout 300 python tools/prof_ops.py row_b1 > gpurun_out/prof_row_b1.log 2>&1
DMX_DCONV_ROW=0 PB=1 timeout 300 python tools/prof_ops.py chain_b1 > gpurun_out/prof_chain_b1.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_row.json 2> gpurun_out/bench_row.err
tail -5 gpurun_out/t1.log; head -12 gpurun_out/prof_row.log; head -12 gpurun_out/prof_chain.log; cat gpurun_out/bench_row.json | head -c 1500

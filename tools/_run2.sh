mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reduced_segment or full_size_segment or batch_equals_singles or stress_models or bench_batch_and_awkward" > gpurun_out/t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t2.log
tail -4 gpurun_out/t2.log
PB=42 timeout 300 python tools/prof_ops.py row2 > gpurun_out/prof_row2.log 2>&1; head -8 gpurun_out/prof_row2.log; grep dconv_row gpurun_out/ops_row2.tsv
PB=1 timeout 300 python tools/prof_ops.py row2_b1 > gpurun_out/prof_row2_b1.log 2>&1; grep dconv_row gpurun_out/ops_row2_b1.tsv
DMX_LIB=demucs_cpp_amd/lib/libdemucs_hip_rowtiming.so PB=42 REPS=1 timeout 300 python tools/prof_ops.py rowtiming_b42 > gpurun_out/rowtiming_b42.log 2>&1
grep rowtiming gpurun_out/rowtiming_b42.log | head -8

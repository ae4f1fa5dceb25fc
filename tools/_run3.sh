mkdir -p gpurun_out
for v in copy nostore w5 w7 w8; do
  DMX_LIB=demucs_cpp_amd/lib/libdemucs_hip_row$v.so PB=42 REPS=5 timeout 300 python tools/prof_ops.py abl_$v > gpurun_out/abl_$v.log 2>&1
  echo "== $v"; grep dconv_row gpurun_out/ops_abl_$v.tsv | cut -f1,3
done
PB=42 REPS=5 timeout 300 python tools/prof_ops.py abl_base > gpurun_out/abl_base.log 2>&1; echo "== base"; grep dconv_row gpurun_out/ops_abl_base.tsv | cut -f1,3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reduced_segment or full_size_segment or stress_models" 2>&1 | tail -3

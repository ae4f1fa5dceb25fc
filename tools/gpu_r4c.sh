#!/bin/bash
# round 4, run C: the engine / replica tests again, SQ counters of the split mode, ablation builds of the split GEMM loop
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q -k "engine or bag or 6s_4min or shards or sharded or concurrent or activation_split or non_finite or gemm_modes" 2>&1 | tail -40 ) > gpurun_out/r4c_pytest.log
for v in nosplit nostore nobar noload pure; do
  ( DMX_LIB=$R/demucs_cpp_amd/lib/libdemucs_hip_abl_$v.so DMX_GEMM=bf16x3 PB=42 REPS=2 timeout 200 python tools/prof_ops.py r4c_abl_$v 2>&1 | grep -E "total|igemm_split" ) > gpurun_out/r4c_abl_$v.log
done
( DMX_GEMM=bf16x3 PB=42 REPS=2 timeout 200 python tools/prof_ops.py r4c_abl_base 2>&1 | grep -E "total|igemm_split" ) > gpurun_out/r4c_abl_base.log
( GEMM=bf16x3 PMC_SQ=1 timeout 900 bash tools/gpu_pmc.sh 42 2>&1 | tail -60 ) > gpurun_out/r4c_pmc.log
echo ---- pytest; tail -25 gpurun_out/r4c_pytest.log
echo ---- ablations; for v in base nosplit nostore nobar noload pure; do echo "== $v"; cat gpurun_out/r4c_abl_$v.log; done
echo ---- pmc; cat gpurun_out/pmc/pass_A_class.csv | head -12; cat gpurun_out/pmc/pass_B_class.csv | head -12

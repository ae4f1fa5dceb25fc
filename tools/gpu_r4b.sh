#!/bin/bash
# round 4, run B: the whole GPU suite in both GEMM modes (split attention in), bench 4s in both modes, per-op profile of the split mode
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( timeout 1700 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -70 ) > gpurun_out/r4b_pytest.log
( timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-track --no-single 2>&1 | tail -2 ) > gpurun_out/r4b_bench_4s.log
( DMX_GEMM=bf16x3 PB=42 timeout 300 python tools/prof_ops.py r4b_split 2>&1 | tail -24 ) > gpurun_out/r4b_prof_split.log
( DMX_GEMM=bf16x3 PB=1 timeout 300 python tools/prof_ops.py r4b_split_b1 2>&1 | tail -8 ) > gpurun_out/r4b_prof_split_b1.log
echo ---- pytest; tail -45 gpurun_out/r4b_pytest.log
echo ---- bench; cat gpurun_out/r4b_bench_4s.log
echo ---- prof; cat gpurun_out/r4b_prof_split.log gpurun_out/r4b_prof_split_b1.log

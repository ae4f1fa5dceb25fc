#!/bin/bash
# round 4, run A: the whole GPU suite in both GEMM modes, the new bench lines (4s with the other mode, 6s, ft), per-op profile of the split mode
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -60 ) > gpurun_out/r4a_pytest.log
( timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -4 ) > gpurun_out/r4a_bench_4s.log
( timeout 400 python bench.py --steps 5 --warmup 2 --model 6s --no-cpu-baseline 2>&1 | tail -4 ) > gpurun_out/r4a_bench_6s.log
( timeout 400 python bench.py --steps 3 --warmup 1 --model ft --no-cpu-baseline 2>&1 | tail -4 ) > gpurun_out/r4a_bench_ft.log
( DMX_GEMM=bf16x3 PB=42 timeout 300 python tools/prof_ops.py r4a_split 2>&1 | tail -30 ) > gpurun_out/r4a_prof_split.log
echo ---- pytest; tail -30 gpurun_out/r4a_pytest.log
echo ---- bench; cat gpurun_out/r4a_bench_4s.log gpurun_out/r4a_bench_6s.log gpurun_out/r4a_bench_ft.log
echo ---- prof; cat gpurun_out/r4a_prof_split.log

#!/bin/bash
# round 4, run N: the whole GPU suite + smoke + a bench line without the CPU leg at HEAD (after the ISTFT chunking change)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( timeout 560 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) > gpurun_out/r4n_gpu_tests.txt
( timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke ) >> gpurun_out/r4n_gpu_tests.txt
( timeout 100 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' ) > gpurun_out/r4n_bench_4s_nocpu.json
cat gpurun_out/r4n_gpu_tests.txt; cut -c1-400 gpurun_out/r4n_bench_4s_nocpu.json

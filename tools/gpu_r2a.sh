#!/bin/bash
# round-2 first GPU pass: full GPU suite (all failures shown), smoke, default bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -80 ) > gpurun_out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > gpurun_out/smoke.log
( timeout 900 python bench.py 2>&1 | tail -20 ) > gpurun_out/bench.log
echo ---- pytest; cat gpurun_out/pytest_gpu.log
echo ---- smoke; cat gpurun_out/smoke.log
echo ---- bench; cat gpurun_out/bench.log

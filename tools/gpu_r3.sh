#!/bin/bash
# round-3 GPU iteration: v3 tests, the bench line (configs[2] headline) for htdemucs-4s and hdemucs_mmi
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_v3.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_v3.log
( time timeout 900 python bench.py 2>gpurun_out/bench_r03.err | tail -1 > gpurun_out/bench_r03.json ) 2> gpurun_out/bench_r03.time
( timeout 600 python bench.py --model v3 --no-cpu-baseline 2>gpurun_out/bench_r03_v3.err | tail -1 > gpurun_out/bench_r03_v3.json )
echo ---- pytest v3; cat gpurun_out/pytest_v3.log
echo ---- bench; cat gpurun_out/bench_r03.json; tail -3 gpurun_out/bench_r03.err; cat gpurun_out/bench_r03.time
echo ---- bench v3; cat gpurun_out/bench_r03_v3.json; tail -3 gpurun_out/bench_r03_v3.err

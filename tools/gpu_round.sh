#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprofv3 kernel stats. Outputs -> gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > gpurun_out/smoke.log
( timeout 600 python bench.py --steps 5 --warmup 2 2>&1 | tail -20 ) > gpurun_out/bench.log
( timeout 600 python bench.py --steps 5 --warmup 2 --batch 1 --no-cpu-baseline --no-roofline --no-single --no-track 2>&1 | tail -5 ) > gpurun_out/bench_b1.log
( timeout 600 python bench.py --steps 5 --warmup 2 --batch 4 --no-cpu-baseline --no-roofline --no-single --no-track 2>&1 | tail -5 ) > gpurun_out/bench_b4.log
( timeout 600 python bench.py --steps 5 --warmup 2 --batch 12 --no-cpu-baseline --no-roofline --no-single --no-track 2>&1 | tail -5 ) > gpurun_out/bench_b12.log
cd /tmp && export TMPDIR=/tmp
( timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-single --no-track 2>&1 | tail -15 ) > $R/gpurun_out/rocprof.log
cd $R
db=$(find gpurun_out/prof -name "*.db" | head -1)
if [ -n "$db" ]; then
  python tools/pmc_summary.py $db > gpurun_out/kernel_stats.csv
  python tools/pmc_summary.py $db --class > gpurun_out/kernel_stats_by_class.csv
  head -20 gpurun_out/kernel_stats_by_class.csv
fi
echo ---- pytest; cat gpurun_out/pytest_gpu.log
echo ---- smoke; cat gpurun_out/smoke.log
echo ---- bench; cat gpurun_out/bench.log gpurun_out/bench_b1.log gpurun_out/bench_b4.log

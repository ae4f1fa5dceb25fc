#!/bin/bash
# Round-2 measurement artefacts at HEAD (copied to profiles/r02_* afterwards): full GPU suite, smoke, the bench
# line, rocprofv3 kernel trace of the bench command, PMC passes (each its own run), effective clock, track bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r02
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) > gpurun_out/r02/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) >> gpurun_out/r02/gpu_tests.txt
( timeout 900 python bench.py 2>&1 | grep '^{' ) > gpurun_out/r02/bench_b24.json
for b in 1 4 12; do ( timeout 600 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-track 2>&1 | grep '^{' ) >> gpurun_out/r02/bench_b1_b4_b12.jsonl; done
( timeout 600 python tools/track_bench.py 2>&1 | grep '^{' ) > gpurun_out/r02/track_bench.jsonl
( NS=6 timeout 600 python tools/track_bench.py 2>&1 | grep '^{' ) >> gpurun_out/r02/track_bench.jsonl
cd /tmp && export TMPDIR=/tmp
( timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-single --no-track 2>&1 | tail -3 ) > $R/gpurun_out/r02/rocprof.log
cd $R
db=$(find /tmp/prof -name "*.db" | head -1)
python tools/pmc_summary.py $db > gpurun_out/r02/kernel_stats_b24.csv
python tools/pmc_summary.py $db --class > gpurun_out/r02/kernel_stats_b24_by_class.csv
bash tools/gpu_pmc.sh 24 > gpurun_out/r02/pmc.log 2>&1
bash tools/gpu_clock.sh 24 > gpurun_out/r02/effective_clock.csv 2>&1
cp gpurun_out/pmc/pass_A_class.csv gpurun_out/r02/pmc_sq_b24_by_class.csv
cp gpurun_out/pmc/pass_B_class.csv gpurun_out/r02/pmc_insts_b24_by_class.csv
cp gpurun_out/pmc/pass_C_class.csv gpurun_out/r02/pmc_fetch_b24_by_class.csv
cp gpurun_out/pmc/pass_D_class.csv gpurun_out/r02/pmc_write_b24_by_class.csv
cp gpurun_out/pmc/traffic.json gpurun_out/r02/traffic.json
cat gpurun_out/r02/gpu_tests.txt; head -12 gpurun_out/r02/kernel_stats_b24_by_class.csv; cat gpurun_out/r02/bench_b24.json | cut -c1-600; cat gpurun_out/r02/track_bench.jsonl

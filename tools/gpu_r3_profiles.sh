#!/bin/bash
# Round-3 measurement artefacts at HEAD (copied to profiles/r03_* afterwards): full GPU suite, smoke, the bench lines
# (htdemucs-4s = the headline, hdemucs_mmi), rocprofv3 kernel trace of the bench command, PMC passes (each its own run),
# effective clock, per-op profiles. Bench workload: one 4-minute track = 42 segments per step.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r03
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) > gpurun_out/r03/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) >> gpurun_out/r03/gpu_tests.txt
( timeout 900 python bench.py 2>&1 | grep '^{' ) > gpurun_out/r03/bench_b42.json
( timeout 600 python bench.py --model v3 2>&1 | grep '^{' ) > gpurun_out/r03/bench_v3_b42.json
for b in 1 4 12 24; do ( timeout 600 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-track --no-split-probe 2>&1 | grep '^{' ) >> gpurun_out/r03/bench_b1_b4_b12_b24.jsonl; done
MODEL=4s PBS="1 42" bash tools/gpu_prof.sh > gpurun_out/r03/ops_4s.log 2>&1; cp gpurun_out/profile_ops_4s_b42.tsv gpurun_out/r03/ops_4s_b42.tsv; cp gpurun_out/profile_ops_4s_b1.tsv gpurun_out/r03/ops_4s_b1.tsv
MODEL=v3 PBS="1 42" bash tools/gpu_prof.sh > gpurun_out/r03/ops_v3.log 2>&1; cp gpurun_out/profile_ops_v3_b42.tsv gpurun_out/r03/ops_v3_b42.tsv; cp gpurun_out/profile_ops_v3_b1.tsv gpurun_out/r03/ops_v3_b1.tsv
DMX_GEMM=bf16x3 MODEL=4s PBS="42" bash tools/gpu_prof.sh > gpurun_out/r03/ops_4s_split.log 2>&1; cp gpurun_out/profile_ops_4s_b42.tsv gpurun_out/r03/ops_4s_b42_split_bf16x3.tsv
bash tools/gpu_split_err.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/r03/split_errors.txt
cd /tmp && export TMPDIR=/tmp
BENCH="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-single --no-track --no-split-probe"
( timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r3 -- python $R/bench.py $BENCH 2>&1 | tail -3 ) > $R/gpurun_out/r03/rocprof.log
( timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o r3v3 -- python $R/bench.py --model v3 $BENCH 2>&1 | tail -3 ) >> $R/gpurun_out/r03/rocprof.log
cd $R
db=$(find /tmp/prof -name "*.db" | head -1)
python tools/pmc_summary.py $db > gpurun_out/r03/kernel_stats_b42.csv
python tools/pmc_summary.py $db --class > gpurun_out/r03/kernel_stats_b42_by_class.csv
db3=$(find /tmp/prof3 -name "*.db" | head -1)
python tools/pmc_summary.py $db3 --class > gpurun_out/r03/kernel_stats_v3_b42_by_class.csv
bash tools/gpu_pmc.sh 42 > gpurun_out/r03/pmc.log 2>&1
bash tools/gpu_clock.sh 42 > gpurun_out/r03/effective_clock.csv 2>&1
cp gpurun_out/pmc/pass_A_class.csv gpurun_out/r03/pmc_sq_b42_by_class.csv
cp gpurun_out/pmc/pass_B_class.csv gpurun_out/r03/pmc_insts_b42_by_class.csv
cp gpurun_out/pmc/pass_C_class.csv gpurun_out/r03/pmc_fetch_b42_by_class.csv
cp gpurun_out/pmc/pass_D_class.csv gpurun_out/r03/pmc_write_b42_by_class.csv
cp gpurun_out/pmc/traffic.json gpurun_out/r03/traffic.json
cat gpurun_out/r03/gpu_tests.txt; head -12 gpurun_out/r03/kernel_stats_b42_by_class.csv; head -8 gpurun_out/r03/kernel_stats_v3_b42_by_class.csv; cut -c1-700 gpurun_out/r03/bench_b42.json; cat gpurun_out/r03/split_errors.txt

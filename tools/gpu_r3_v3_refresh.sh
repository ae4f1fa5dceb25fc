#!/bin/bash
# refresh of the round-3 hdemucs_mmi artefacts after the LocalState attention moved to the MFMA kernel: bench line, rocprofv3
# kernel trace of the bench command, per-op profiles, FETCH / WRITE passes; plus the full GPU suite at the same HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r03
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) > gpurun_out/r03/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) >> gpurun_out/r03/gpu_tests.txt
MODEL=v3 PMC_SQ=0 bash tools/gpu_pmc.sh 42 > gpurun_out/r03/pmc_v3.log 2>&1
cp gpurun_out/pmc/traffic.json profiles/r03_traffic_v3.json   # bench.py --model v3 quotes it below
cp gpurun_out/pmc/traffic.json gpurun_out/r03/traffic_v3.json
cp gpurun_out/pmc/pass_C_class.csv gpurun_out/r03/pmc_fetch_v3_b42_by_class.csv
cp gpurun_out/pmc/pass_D_class.csv gpurun_out/r03/pmc_write_v3_b42_by_class.csv
( timeout 600 python bench.py --model v3 2>&1 | grep '^{' ) > gpurun_out/r03/bench_v3_b42.json
MODEL=v3 PBS="1 42" bash tools/gpu_prof.sh > gpurun_out/r03/ops_v3.log 2>&1; cp gpurun_out/profile_ops_v3_b42.tsv gpurun_out/r03/ops_v3_b42.tsv; cp gpurun_out/profile_ops_v3_b1.tsv gpurun_out/r03/ops_v3_b1.tsv
cd /tmp && export TMPDIR=/tmp
BENCH="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-single --no-track --no-split-probe"
( timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o r3v3 -- python $R/bench.py --model v3 $BENCH 2>&1 | tail -3 ) > $R/gpurun_out/r03/rocprof_v3.log
cd $R
python tools/pmc_summary.py $(find /tmp/prof3 -name "*.db" | head -1) --class > gpurun_out/r03/kernel_stats_v3_b42_by_class.csv
cat gpurun_out/r03/gpu_tests.txt; head -12 gpurun_out/r03/kernel_stats_v3_b42_by_class.csv; cut -c1-600 gpurun_out/r03/bench_v3_b42.json

#!/bin/bash
# Round-4 measurement artefacts at HEAD (copied to profiles/r04_* afterwards): full GPU suite (both GEMM modes in process),
# smoke, the bench lines (htdemucs-4s = the headline, 6s = configs[3] workload, ft = configs[4] workload, hdemucs_mmi),
# rocprofv3 kernel traces of the bench command in both GEMM modes, PMC passes of the default mode (each its own run),
# effective clock, per-op profiles. Bench workload: one 4-minute track = 42 segments per step.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r04; mkdir -p $O
( timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) > $O/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke ) >> $O/gpu_tests.txt
( timeout 900 python bench.py 2>&1 | grep '^{' ) > $O/bench_4s_b42.json
( timeout 600 python bench.py --model 6s 2>&1 | grep '^{' ) > $O/bench_6s_b42.json
( timeout 600 python bench.py --model ft --steps 3 --warmup 1 2>&1 | grep '^{' ) > $O/bench_ft_b42.json
( timeout 600 python bench.py --model v3 2>&1 | grep '^{' ) > $O/bench_v3_b42.json
for b in 1 4 12 24; do ( timeout 600 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-track 2>&1 | grep '^{' ) >> $O/bench_4s_b1_b4_b12_b24.jsonl; done
MODEL=4s PBS="1 42" bash tools/gpu_prof.sh > $O/ops_4s.log 2>&1; cp gpurun_out/profile_ops_4s_b42.tsv $O/ops_4s_b42.tsv; cp gpurun_out/profile_ops_4s_b1.tsv $O/ops_4s_b1.tsv
DMX_GEMM=f32 MODEL=4s PBS="42" bash tools/gpu_prof.sh > $O/ops_4s_f32.log 2>&1; cp gpurun_out/profile_ops_4s_b42.tsv $O/ops_4s_b42_f32.tsv
MODEL=6s PBS="42" bash tools/gpu_prof.sh > $O/ops_6s.log 2>&1; cp gpurun_out/profile_ops_6s_b42.tsv $O/ops_6s_b42.tsv
MODEL=v3 PBS="1 42" bash tools/gpu_prof.sh > $O/ops_v3.log 2>&1; cp gpurun_out/profile_ops_v3_b42.tsv $O/ops_v3_b42.tsv; cp gpurun_out/profile_ops_v3_b1.tsv $O/ops_v3_b1.tsv
cd /tmp && export TMPDIR=/tmp
BENCH="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-single --no-track --no-other-gemm"
( timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r4 -- python $R/bench.py $BENCH 2>&1 | tail -3 ) > $R/$O/rocprof.log
( timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof32 -o r4f32 -- python $R/bench.py --gemm f32 $BENCH 2>&1 | tail -3 ) >> $R/$O/rocprof.log
( timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o r4v3 -- python $R/bench.py --model v3 $BENCH 2>&1 | tail -3 ) >> $R/$O/rocprof.log
cd $R
db=$(find /tmp/prof -name "*.db" | head -1)
python tools/pmc_summary.py $db > $O/kernel_stats_b42.csv
python tools/pmc_summary.py $db --class > $O/kernel_stats_b42_by_class.csv
db32=$(find /tmp/prof32 -name "*.db" | head -1)
python tools/pmc_summary.py $db32 --class > $O/kernel_stats_b42_f32_by_class.csv
db3=$(find /tmp/prof3 -name "*.db" | head -1)
python tools/pmc_summary.py $db3 --class > $O/kernel_stats_v3_b42_by_class.csv
GEMM=bf16x3 bash tools/gpu_pmc.sh 42 > $O/pmc.log 2>&1
GEMM=bf16x3 bash tools/gpu_clock.sh 42 > $O/effective_clock.csv 2>&1
cp gpurun_out/pmc/pass_A_class.csv $O/pmc_sq_b42_by_class.csv
cp gpurun_out/pmc/pass_B_class.csv $O/pmc_insts_b42_by_class.csv
cp gpurun_out/pmc/pass_C_class.csv $O/pmc_fetch_b42_by_class.csv
cp gpurun_out/pmc/pass_D_class.csv $O/pmc_write_b42_by_class.csv
cp gpurun_out/pmc/traffic.json $O/traffic.json
cat $O/gpu_tests.txt; head -12 $O/kernel_stats_b42_by_class.csv; head -6 $O/kernel_stats_b42_f32_by_class.csv; for m in 4s 6s ft v3; do cut -c1-330 $O/bench_${m}_b42.json; echo; done

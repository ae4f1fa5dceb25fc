#!/bin/bash
# A round's measurement artefacts, ALL collected by this one script at one commit (TAG=r06 -> gpurun_out/r06/*, copied to profiles/r06_* afterwards):
#   full GPU suite (all three GEMM modes in process, the erratum assertions on), smoke, the bench lines (4s = the headline, 6s,
#   ft, v3, batch sweep), per-op HIP-event tables, rocprofv3 kernel traces of the bench command in both GEMM modes, the four
#   PMC passes + effective clock of the default mode (each its own run), the erratum reproducers.
# Parts: PARTS="tests bench ops trace pmc erratum fp16" (default all but `headline` = the first bench line alone; fp16: bench line and per-op table of the opt-in DMX_GEMM=fp16x3 mode). Bench workload: one 4-minute track = 42 segments per step.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; TAG=${TAG:-r06}; O=gpurun_out/$TAG; mkdir -p $O
PARTS=${PARTS:-tests bench ops trace pmc erratum fp16}
has() { [[ " $PARTS " == *" $1 "* ]]; }
git rev-parse HEAD > $O/head.txt 2>/dev/null || true
if has tests; then
  ( DMX_TEST_ERRATUM=1 DMX_TEST_ALL_MODES=1 timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) > $O/gpu_tests.txt
  ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke ) >> $O/gpu_tests.txt
fi
if has bench || has headline; then
  ( timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{' ) > $O/bench_4s_b42.json
fi
if has bench; then
  ( timeout 600 python bench.py --model 6s 2>&1 | grep '^{' ) > $O/bench_6s_b42.json
  ( timeout 600 python bench.py --model ft --steps 3 --warmup 1 2>&1 | grep '^{' ) > $O/bench_ft_b42.json
  ( timeout 600 python bench.py --model v3 2>&1 | grep '^{' ) > $O/bench_v3_b42.json
  rm -f $O/bench_4s_b1_b4_b12_b24.jsonl
  for b in 1 4 12 24; do ( timeout 600 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-track 2>&1 | grep '^{' ) >> $O/bench_4s_b1_b4_b12_b24.jsonl; done
fi
if has ops; then
  MODEL=4s PBS="1 42" bash tools/gpu_prof.sh > $O/ops_4s.log 2>&1; cp gpurun_out/profile_ops_4s_b42.tsv $O/ops_4s_b42.tsv; cp gpurun_out/profile_ops_4s_b1.tsv $O/ops_4s_b1.tsv
  DMX_GEMM=f32 MODEL=4s PBS="42" bash tools/gpu_prof.sh > $O/ops_4s_f32.log 2>&1; cp gpurun_out/profile_ops_4s_b42.tsv $O/ops_4s_b42_f32.tsv
  MODEL=6s PBS="42" bash tools/gpu_prof.sh > $O/ops_6s.log 2>&1; cp gpurun_out/profile_ops_6s_b42.tsv $O/ops_6s_b42.tsv
  MODEL=v3 PBS="42" bash tools/gpu_prof.sh > $O/ops_v3.log 2>&1; cp gpurun_out/profile_ops_v3_b42.tsv $O/ops_v3_b42.tsv
fi
if has trace; then
  cd /tmp && export TMPDIR=/tmp
  BENCH="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-single --no-track --no-other-gemm"
  ( timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o trace -- python $R/bench.py $BENCH 2>&1 | tail -3 ) > $R/$O/rocprof.log
  ( timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof32 -o tracef32 -- python $R/bench.py --gemm f32 $BENCH 2>&1 | tail -3 ) >> $R/$O/rocprof.log
  cd $R
  db=$(find /tmp/prof -name "*.db" | head -1)
  python tools/pmc_summary.py $db > $O/kernel_stats_b42.csv
  python tools/pmc_summary.py $db --class > $O/kernel_stats_b42_by_class.csv
  db32=$(find /tmp/prof32 -name "*.db" | head -1)
  python tools/pmc_summary.py $db32 --class > $O/kernel_stats_b42_f32_by_class.csv
fi
if has pmc; then
  GEMM=bf16x3 bash tools/gpu_pmc.sh 42 > $O/pmc.log 2>&1
  GEMM=bf16x3 bash tools/gpu_clock.sh 42 > $O/effective_clock.csv 2>&1
  cp gpurun_out/pmc/pass_A_class.csv $O/pmc_sq_b42_by_class.csv
  cp gpurun_out/pmc/pass_B_class.csv $O/pmc_insts_b42_by_class.csv
  cp gpurun_out/pmc/pass_C_class.csv $O/pmc_fetch_b42_by_class.csv
  cp gpurun_out/pmc/pass_D_class.csv $O/pmc_write_b42_by_class.csv
  cp gpurun_out/pmc/traffic.json $O/traffic_bf16x3.json
fi
if has fp16; then
  ( timeout 900 python bench.py --gemm fp16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-track 2>&1 | grep '^{' ) > $O/bench_4s_b42_fp16x3.json
  DMX_GEMM=fp16x3 MODEL=4s PBS="42" bash tools/gpu_prof.sh > $O/ops_4s_fp16x3.log 2>&1; cp gpurun_out/profile_ops_4s_b42.tsv $O/ops_4s_b42_fp16x3.tsv
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/profh -o traceh -- python $R/bench.py --gemm fp16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-single --no-track --no-other-gemm 2>&1 | tail -2 ) >> $O/ops_4s_fp16x3.log
  python tools/pmc_summary.py $(find /tmp/profh -name "*.db" | head -1) --class > $O/kernel_stats_b42_fp16x3_by_class.csv
fi
if has erratum; then
  ( timeout 200 tests/_build/pk_f32_erratum 6 0 2>&1 | cut -c1-230 ) > $O/pk_f32_erratum_forms.log
  ( timeout 200 tests/_build/fft_mfma_repro_pk 16 12 0xff 2>&1 | cut -c1-230 ) > $O/fft_mfma_repro_packed_build.log
  ( timeout 200 tests/_build/fft_mfma_repro 16 12 0xff 2>&1 | cut -c1-230 ) > $O/fft_mfma_repro_product_build.log
  ( RUNS=32 TAG=product timeout 300 python tools/fft_erratum_diag.py 2>&1 | grep '^\[' ) > $O/two_contexts_stress.log
fi
cat $O/gpu_tests.txt 2>/dev/null; head -12 $O/kernel_stats_b42_by_class.csv 2>/dev/null; for m in 4s_b42 6s_b42 ft_b42 v3_b42 4s_b42_fp16x3; do cut -c1-330 $O/bench_${m}.json 2>/dev/null; echo; done

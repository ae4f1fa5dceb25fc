#!/bin/bash
# PMC passes over the bench workload (each its own run; never combined with other traces).
#   [MODEL=4s|6s|ft|v3] [GEMM=f32|bf16x3] [PMC_SQ=0] tools/gpu_pmc.sh [batch]     -> gpurun_out/pmc/pass_{A,B,C,D}.csv (+ per class) and traffic.json
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-24}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --model ${MODEL:-4s} --steps 1 --warmup 1 --batch $B --gemm ${GEMM:-f32} --no-cpu-baseline --no-roofline --no-single --no-track --no-other-gemm"
[ "${PMC_SQ:-1}" = "1" ] && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmcA -o a -- $CMD > /tmp/pmcA.log 2>&1
[ "${PMC_SQ:-1}" = "1" ] && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d /tmp/pmcB -o b -- $CMD > /tmp/pmcB.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcC -o c -- $CMD > /tmp/pmcC.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmcD -o d -- $CMD > /tmp/pmcD.log 2>&1
for x in A B C D; do
  f=$(find /tmp/pmc$x -name "*.db" 2>/dev/null | head -1)
  [ -z "$f" ] && continue
  echo "== pass $x ($f)"; tail -2 /tmp/pmc$x.log
  python $R/tools/pmc_summary.py $f > $R/gpurun_out/pmc/pass_$x.csv 2>&1
  python $R/tools/pmc_summary.py $f --class > $R/gpurun_out/pmc/pass_${x}_class.csv 2>&1
  head -8 $R/gpurun_out/pmc/pass_${x}_class.csv
done
python $R/tools/traffic_json.py $(find /tmp/pmcC -name "*.db" | head -1) $(find /tmp/pmcD -name "*.db" | head -1) $B > $R/gpurun_out/pmc/traffic.json
cat $R/gpurun_out/pmc/traffic.json | head -30

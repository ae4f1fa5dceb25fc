#!/bin/bash
# Exploratory PMC passes on the memory pipeline (texture addresser / L1 / L2) of the bench workload, per kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-24}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --batch $B --no-cpu-baseline --no-roofline --no-single --no-track --no-split-probe"
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA_[A-Z_]+|TCP_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TD_[A-Z_]+)\b" | sort -u | tr '\n' ' ' | cut -c1-3000 > $R/gpurun_out/pmc/mem_counters.txt
p=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  p=$((p+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcM$p -o m -- $CMD > /tmp/pmcM$p.log 2>&1
  f=$(find /tmp/pmcM$p -name "*.db" 2>/dev/null | head -1)
  echo "== pass $p: $set"; tail -2 /tmp/pmcM$p.log | cut -c1-200
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f > $R/gpurun_out/pmc/pass_M$p.csv 2>&1 && python $R/tools/pmc_summary.py $f --class > $R/gpurun_out/pmc/pass_M${p}_class.csv 2>&1 && head -6 $R/gpurun_out/pmc/pass_M${p}_class.csv
done

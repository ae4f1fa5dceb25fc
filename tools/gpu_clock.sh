#!/bin/bash
# Effective shader clock per kernel class under the bench workload: GRBM_GUI_ACTIVE (cycles the GPU
# was busy during the dispatch, summed by rocprofv3 over the 8 XCDs) / 8 / kernel duration
# (MI355X_MICROARCH.md §DVFS give-back). Own PMC pass. Short kernels read high (the counter window is wider
# than the kernel); the long MFMA kernels show whether the 2.4 GHz the peak is priced at is sustained.
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-24}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --batch $B --gemm ${GEMM:-bf16x3} --no-cpu-baseline --no-roofline --no-single --no-track --no-other-gemm"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d /tmp/pmcE -o e -- $CMD > /tmp/pmcE.log 2>&1
f=$(find /tmp/pmcE -name "*.db" | head -1)
tail -2 /tmp/pmcE.log
python $R/tools/pmc_summary.py $f --class > $R/gpurun_out/pmc/pass_E_class.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/pmc/pass_E_class.csv")))
print("class,calls,avg_us,effective_clock_GHz_per_XCD")
for r in rows[:12]:
    cyc = float(r.get("GRBM_GUI_ACTIVE", 0) or 0)
    us = float(r["total_us"])
    if us > 0 and cyc > 0:
        print(f'{r["kernel"]},{r["calls"]},{r["avg_us"]},{cyc / us / 1e3 / 8:.3f}')
PY

#!/bin/bash
# A/B of library builds (make variant / variant1) on the bench workload: bench value + per-class kernel time of one 42-segment plan run.
#   LIBS="product nopk noslp" bash tools/gpu_ab_libs.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/ab; mkdir -p $O
for n in ${LIBS:-product}; do
  lib=$R/demucs_cpp_amd/lib/libdemucs_hip$([ $n = product ] || echo _$n).so
  [ -f $lib ] || { echo "no $lib"; continue; }
  ( DMX_LIB=$lib timeout 600 python bench.py --steps ${STEPS:-8} --warmup 3 --gemm ${GEMM:-bf16x3} --no-cpu-baseline --no-single --no-track --no-other-gemm --no-roofline 2>&1 | grep '^{' ) > $O/bench_$n.json
  python - <<PY
import json
d = json.load(open("$O/bench_$n.json"))
print("$n: value %.1f, %.3f ms per step, %.3f ms per segment" % (d["value"], d["ms_per_step"], d["config"]["ms_per_segment"]))
PY
  DMX_LIB=$lib MODEL=${MODEL:-4s} PBS="${PBS:-42}" bash tools/gpu_prof.sh > $O/ops_$n.log 2>&1
  cp gpurun_out/profile_ops_${MODEL:-4s}_b${PBS:-42}.tsv $O/ops_$n.tsv
  python - <<PY
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0, 0.0])
for l in open("$O/ops_$n.tsv"):
    nm, k, ms, fl, by = l.rstrip("\n").split("\t")
    a = agg[k]; a[0] += 1; a[1] += float(ms); a[2] += float(fl)
tot = sum(a[1] for a in agg.values())
print("   per-op replay: %.2f ms per plan run;" % tot, "  ".join("%s %.2f" % (k, a[1]) for k, a in sorted(agg.items(), key=lambda x: -x[1][1])[:9]))
PY
done 2>&1 | tee $O/summary.txt

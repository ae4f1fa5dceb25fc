#!/bin/bash
# A/B of library builds by the per-op HIP-event replay alone (no bench line): per class, and the ops whose names match $OPS.
#   LIBS="product linhalf lin64 lin64x3" OPS="linear|proj|qk|\.q|\.k|\.v" bash tools/gpu_ab_ops.sh   -> gpurun_out/ab/ops_<lib>.tsv, summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/ab; mkdir -p $O
for n in ${LIBS:-product}; do
  lib=$R/demucs_cpp_amd/lib/libdemucs_hip$([ $n = product ] || echo _$n).so
  [ -f $lib ] || { echo "no $lib"; continue; }
  DMX_LIB=$lib MODEL=${MODEL:-4s} PBS="${PBS:-42}" bash tools/gpu_prof.sh > $O/ops_$n.log 2>&1
  cp gpurun_out/profile_ops_${MODEL:-4s}_b${PBS:-42}.tsv $O/ops_$n.tsv
  N=$n OPS="${OPS:-.}" python - <<PY
import os, re
from collections import defaultdict
n, pat = os.environ["N"], re.compile(os.environ["OPS"])
agg = defaultdict(float); sel = 0.0; nsel = 0
for l in open("$O/ops_%s.tsv" % n):
    f = l.rstrip("\n").split("\t")
    agg[f[1]] += float(f[2])
    if pat.search(f[0]) and f[1] != "stats_reduce":
        sel += float(f[2]); nsel += 1
tot = sum(agg.values())
print("%-10s plan run %.2f ms | ops matching: %d ops %.3f ms |" % (n, tot, nsel, sel), "  ".join("%s %.2f" % (k, v) for k, v in sorted(agg.items(), key=lambda x: -x[1])[:8]))
PY
done 2>&1 | tee -a $O/summary.txt

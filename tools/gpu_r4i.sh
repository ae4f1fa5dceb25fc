#!/bin/bash
# round 4, run I: narrow split tiles for the deep DConv K1 + the linear-layer kernel as default: whole GPU suite, per-op profiles, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( PB=42 timeout 300 python tools/prof_ops.py r4i_4s 2>&1 | tail -24 ) > gpurun_out/r4i_prof_4s.log
( PB=42 NS=3 timeout 300 python tools/prof_ops.py r4i_v3 2>&1 | tail -24 ) > gpurun_out/r4i_prof_v3.log
( timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/r4i_bench_4s.json
( timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --model v3 2>&1 | tail -1 ) > gpurun_out/r4i_bench_v3.json
( timeout 1700 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | tail -30 ) > gpurun_out/r4i_pytest.log
echo ---- prof; head -9 gpurun_out/r4i_prof_4s.log; head -9 gpurun_out/r4i_prof_v3.log
awk -F'\t' '$1 ~ /dconv.\.k1/ {printf "%-28s %-22s %7.3f ms %7.1f TF/s %7.1f GB/s\n",$1,$2,$3,$4/$3/1e9,$5/$3/1e6}' gpurun_out/ops_r4i_4s.tsv | sort | head -40
for m in 4s v3; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r4i_bench_$m.json")); c=d["config"]; print("bench $m", d["value"], d["ms_per_step"], c.get("f32_mfma_xRT"), d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["traffic"])
except Exception as e:
    print("bench $m failed", e, open("gpurun_out/r4i_bench_$m.json").read()[-400:])
PY
done
echo ---- pytest; tail -14 gpurun_out/r4i_pytest.log

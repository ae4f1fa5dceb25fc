#!/bin/bash
# round 4, run E: the whole GPU suite (both modes; default = bf16x3), smoke, bench lines 4s / 6s / ft / v3, per-op profiles
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( timeout 1700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 ) > gpurun_out/r4e_pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/r4e_smoke.log
( timeout 500 python bench.py --steps 5 --warmup 2 2>&1 | tail -1 ) > gpurun_out/r4e_bench_4s.json
( timeout 400 python bench.py --steps 5 --warmup 2 --model 6s --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/r4e_bench_6s.json
( timeout 400 python bench.py --steps 3 --warmup 1 --model ft --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/r4e_bench_ft.json
( timeout 400 python bench.py --steps 5 --warmup 2 --model v3 --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/r4e_bench_v3.json
( PB=42 timeout 300 python tools/prof_ops.py r4e_split 2>&1 | tail -24 ) > gpurun_out/r4e_prof_split.log
echo ---- pytest; tail -22 gpurun_out/r4e_pytest.log
echo ---- smoke; cat gpurun_out/r4e_smoke.log
echo ---- bench; for m in 4s 6s ft v3; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r4e_bench_$m.json")); c=d["config"]
    print("$m", d["value"], d["ms_per_step"], c["ms_per_segment"], {k:v for k,v in c.items() if "xRT" in k or "latency" in k}, d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"))
except Exception as e:
    print("$m failed", e, open("gpurun_out/r4e_bench_$m.json").read()[-600:])
PY
done
echo ---- prof; cat gpurun_out/r4e_prof_split.log

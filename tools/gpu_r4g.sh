#!/bin/bash
# round 4, run G: linear-layer split kernel with activation fragments in registers: bits A/B, per-op profile A/B, bench A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for mode in 0 1 2; do
  ( DMX_SPLIT_LIN=$mode timeout 300 python tools/gpu_lin_ab.py run /tmp/lin_$mode.npz 2>&1 | tail -4 ) > gpurun_out/r4g_ab_$mode.log
done
( python tools/gpu_lin_ab.py cmp /tmp/lin_0.npz /tmp/lin_1.npz; python tools/gpu_lin_ab.py cmp /tmp/lin_0.npz /tmp/lin_2.npz ) > gpurun_out/r4g_cmp.log 2>&1
for mode in 0 1 2; do
  ( DMX_SPLIT_LIN=$mode PB=42 timeout 300 python tools/prof_ops.py r4g_lin$mode 2>&1 | tail -24 ) > gpurun_out/r4g_prof_$mode.log
  ( DMX_SPLIT_LIN=$mode timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-gemm 2>&1 | tail -1 ) > gpurun_out/r4g_bench_$mode.json
done
echo ---- ab; cat gpurun_out/r4g_ab_*.log; cat gpurun_out/r4g_cmp.log
for mode in 0 1 2; do echo ---- prof $mode; head -8 gpurun_out/r4g_prof_$mode.log; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r4g_bench_$mode.json")); print("bench mode $mode", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"])
except Exception as e:
    print("bench $mode failed", e, open("gpurun_out/r4g_bench_$mode.json").read()[-400:])
PY
done
python - <<'PY'
# LIN ops side by side
rows={}
for mode in (0,1,2):
    for l in open(f"gpurun_out/ops_r4g_lin{mode}.tsv"):
        f=l.rstrip("\n").split("\t")
        rows.setdefault(f[0],{})[mode]=(f[1],float(f[2]),float(f[3]))
tot={0:0,1:0,2:0}
for nm,d in rows.items():
    if any(k in nm for k in ("linear","qkv","kv",".q","out_proj","attn_out","proj")) and "attn" != nm.split(".")[-1]:
        ms=[d[m][1] for m in (0,1,2)]
        for m in (0,1,2): tot[m]+=d[m][1]
        if d[0][1]>0.4: print(f"{nm:44s} {d[1][0]:22s} " + " ".join(f"{x:7.3f}" for x in ms) + f"  TF/s " + " ".join(f"{d[m][2]/d[m][1]/1e9:6.1f}" for m in (0,1,2)))
print("sum of linear-layer ops (ms):", tot)
PY

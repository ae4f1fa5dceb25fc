#!/bin/bash
# round 4, run J: ablations of the linear-layer split kernel (make variant NAME=ablN FLAGS=-DDMX_SPLIT_ABL=N; results wrong by construction)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for v in base 1 16 256 64 29 285; do
  lib=demucs_cpp_amd/lib/libdemucs_hip_abl$v.so; [ $v = base ] && lib=demucs_cpp_amd/lib/libdemucs_hip.so
  ( DMX_LIB=$R/$lib PB=42 REPS=2 timeout 200 python tools/prof_ops.py r4j_$v 2>&1 | tail -12 ) > gpurun_out/r4j_prof_$v.log
done
python - <<'PY'
rows={}
vs=["base","1","16","256","64","29","285"]
for v in vs:
    for l in open(f"gpurun_out/ops_r4j_{v}.tsv"):
        f=l.rstrip("\n").split("\t")
        rows.setdefault(f[0],{})[v]=(f[1],float(f[2]),float(f[3]))
print("%-40s "%"op"+" ".join("%8s"%v for v in vs))
tot={v:0 for v in vs}
for nm,d in rows.items():
    if d["base"][0]=="igemm_split_128x128" and any(k in nm for k in ("linear","qkv",".kv",".q","out_proj")):
        for v in vs: tot[v]+=d[v][1]
        if any(k in nm for k in ("layers.0.","layers_t.0.")):
            print("%-40s "%nm+" ".join("%8.3f"%d[v][1] for v in vs))
print("%-40s "%"all linear-layer launches (ms)"+" ".join("%8.3f"%tot[v] for v in vs))
nm="decoder.0.rewrite"; print("%-40s "%nm+" ".join("%8.3f"%rows[nm][v][1] for v in vs))
PY

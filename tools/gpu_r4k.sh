#!/bin/bash
# round 4, run K: first-round stagger of the two workgroups of a CU (make variant NAME=stagN FLAGS=-DDMX_STAGGER=N)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
VS="base abl64"
for v in $VS; do
  lib=demucs_cpp_amd/lib/libdemucs_hip_$v.so; [ $v = base ] && lib=demucs_cpp_amd/lib/libdemucs_hip.so
  ( DMX_LIB=$R/$lib PB=42 REPS=3 timeout 200 python tools/prof_ops.py r4k_$v 2>&1 | tail -12 ) > gpurun_out/r4k_prof_$v.log
  head -6 gpurun_out/r4k_prof_$v.log | grep -v amdgpu.ids
done
VS="$VS" python - <<'PY'
import os
rows={}
vs=os.environ["VS"].split()
for v in vs:
    for l in open(f"gpurun_out/ops_r4k_{v}.tsv"):
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
for nm in ("decoder.0.rewrite","decoder.1.rewrite","decoder.2.rewrite","decoder.3.rewrite","encoder.3.conv","decoder.1.conv_tr"):
    print("%-40s "%nm+" ".join("%8.3f"%rows[nm][v][1] for v in vs))
PY

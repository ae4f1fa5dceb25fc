#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for v in base sametap; do
  lib=$R/demucs_cpp_amd/lib/libdemucs_hip_$v.so; [ $v = base ] && lib=$R/demucs_cpp_amd/lib/libdemucs_hip.so
  ( DMX_LIB=$lib timeout 300 python tools/prof_ops.py abl_$v 2>&1 | grep -v amdgpu.ids | grep -E "^\[|dgemm" )
done
python - <<'PY'
for v in ["base","sametap"]:
    rows=[l.rstrip().split('\t') for l in open(f"gpurun_out/ops_abl_{v}.tsv")]
    k1=[r for r in rows if r[0].endswith('.k1') and r[1]=='dgemm_direct']
    print(v, ' '.join(f"{r[0].replace('encoder','e').replace('decoder','d').replace('dconv','')}:{float(r[2])*1e3:.0f}us" for r in k1))
PY

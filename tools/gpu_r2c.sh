#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; rm -f gpurun_out/abl.log
for v in base ks1w3; do
  lib=$R/demucs_cpp_amd/lib/libdemucs_hip_$v.so; [ $v = base ] && lib=$R/demucs_cpp_amd/lib/libdemucs_hip.so
  ( DMX_LIB=$lib timeout 300 python tools/prof_ops.py abl_$v 2>&1 | grep -v amdgpu.ids | grep -E "^\[|igemm" ) >> gpurun_out/abl.log
done
cat gpurun_out/abl.log
python - <<'PY'
ops=["crosstransformer.layers.0.linear1","crosstransformer.layers.0.linear2","crosstransformer.layers.0.qkv","decoder.0.rewrite","decoder.1.rewrite","decoder.1.conv_tr","encoder.3.conv"]
for v in ["base","ks1w3"]:
    rows={l.split('\t')[0]:l.rstrip().split('\t') for l in open(f"gpurun_out/ops_abl_{v}.tsv")}
    print(v.ljust(6)," ".join(f"{o.replace('crosstransformer.layers','ct')}:{float(rows[o][3])/float(rows[o][2])/1e9:5.1f}" for o in ops if o in rows))
PY

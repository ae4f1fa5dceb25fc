#!/bin/bash
# ablation of the attention tile loop (diagnostic variant libraries, results wrong by construction)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; rm -f gpurun_out/abl.log
for v in base anosm anostage anov anosmv apure; do
  lib=$R/demucs_cpp_amd/lib/libdemucs_hip_$v.so; [ $v = base ] && lib=$R/demucs_cpp_amd/lib/libdemucs_hip.so
  ( DMX_LIB=$lib timeout 300 python tools/prof_ops.py abl_$v 2>&1 | grep -v amdgpu.ids | grep -E "^\[|attention" ) >> gpurun_out/abl.log
done
cat gpurun_out/abl.log
python - <<'PY'
ops=["crosstransformer.layers.0.attn","crosstransformer.layers.1.attn","crosstransformer.layers_t.0.attn","crosstransformer.layers_t.1.attn"]
for v in ["base","anosm","anostage","anov","anosmv","apure"]:
    rows={l.split('\t')[0]:l.rstrip().split('\t') for l in open(f"gpurun_out/ops_abl_{v}.tsv")}
    print(v.ljust(8)," ".join(f"{o.split('.')[1]+'.'+o.split('.')[2]}:{float(rows[o][2]):.3f}ms/{float(rows[o][3])/float(rows[o][2])/1e9:5.1f}TF" for o in ops if o in rows))
PY

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; rm -f gpurun_out/abl.log
for v in base dgp0 dgp1 dgp3; do
  lib=$R/demucs_cpp_amd/lib/libdemucs_hip_$v.so; [ $v = base ] && lib=$R/demucs_cpp_amd/lib/libdemucs_hip.so
  ( DMX_LIB=$lib timeout 300 python tools/prof_ops.py abl_$v 2>&1 | grep -v amdgpu.ids | grep -E "^\[|dgemm" ) >> gpurun_out/abl.log
done
cat gpurun_out/abl.log
python - <<'PY'
import collections
def load(v):
    rows=[l.rstrip('\n').split('\t') for l in open(f'gpurun_out/ops_abl_{v}.tsv')]
    grp=collections.OrderedDict()
    for r in rows:
        if r[1]!='dgemm_direct': continue
        n=r[0]
        key=('K1' if n.endswith('.k1') else 'K2' if n.endswith('.k2') else 'K3' if n.endswith('.k3') else n.split('.')[-1])+' '+('t' if n.startswith('t') else 'f')+n.split('.')[1]
        g=grp.setdefault(key,0.0); grp[key]=g+float(r[2])
    return grp
vs=['base','dgp0','dgp1','dgp3']
d={v:load(v) for v in vs}
print('group'.ljust(16),' '.join(v.rjust(8) for v in vs))
for k in d['base']:
    print(k.ljust(16),' '.join(f"{d[v][k]:8.3f}" for v in vs))
PY

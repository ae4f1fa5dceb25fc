#!/bin/bash
# EXPERIMENT: is the tile-choice cost model (plan.cpp refine_cfg) right at 2 / 4 / 6 segments per call?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for v in "" "DMX_FORCE_HALF=1" "DMX_FORCE_HALF=2"; do
  echo "== $v"
  env $v MODEL=4s PBS="2 4 6" bash tools/gpu_prof.sh 2>&1 | grep -E "^=="
  for b in 2 4 6; do cp gpurun_out/profile_ops_4s_b$b.tsv "gpurun_out/half_${v##*=}_b$b.tsv"; done
done
python - <<'PY'
def load(f): return {l.split("\t")[0]: l.rstrip("\n").split("\t") for l in open(f)}
for b in (2, 4, 6):
    d = load(f"gpurun_out/half__b{b}.tsv"); h1 = load(f"gpurun_out/half_1_b{b}.tsv"); h2 = load(f"gpurun_out/half_2_b{b}.tsv")
    best = 0; cur = 0
    for n in d:
        if not d[n][1].startswith("igemm"): continue
        t = [float(x[n][2]) for x in (d, h1, h2)]
        cur += t[0]; best += min(t)
        if min(t) < 0.93 * t[0] and t[0] > 0.02:
            print(f"b{b} {n:40s} {d[n][1]:14s} {t[0]*1e3:7.1f} us | half {h1[n][1]} {t[1]*1e3:7.1f} | quarter {h2[n][1]} {t[2]*1e3:7.1f}")
    print(f"b{b}: igemm ops {cur:.3f} ms, per-op optimum over the siblings {best:.3f} ms")
PY

#!/bin/bash
# EXPERIMENT: 256x128 tile with FOUR waves of 128x64 and 16-deep K-tiles (cfg 18, DMX_TALL=2, plain loop) against the
# 128x128 tile in the SAME plain 16-deep loop (variant build -DDMX_BIG_KS=1) and against the product (interleaved 32-deep loop)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
run() { # label, env...
  lab=$1; shift
  env "$@" MODEL=4s PBS="42" bash tools/gpu_prof.sh 2>&1 | grep -E "^== .*sum of ops"
  cp gpurun_out/profile_ops_4s_b42.tsv gpurun_out/tall2_$lab.tsv
}
run product DMX_TALL=0
run w4 DMX_TALL=2
run ks1 DMX_LIB=$R/demucs_cpp_amd/lib/libdemucs_hip_ks1.so
python - <<'PY'
def load(f): return {l.split("\t")[0]: l.rstrip("\n").split("\t") for l in open(f)}
p, w, k = load("gpurun_out/tall2_product.tsv"), load("gpurun_out/tall2_w4.tsv"), load("gpurun_out/tall2_ks1.tsv")
tp = tw = tk = 0
for n in p:
    if w[n][1] == "igemm_256x128w4":
        f = lambda r: float(r[3]) / float(r[2]) / 1e9
        tp += float(p[n][2]); tw += float(w[n][2]); tk += float(k[n][2])
        print(f"{n:40s} product {p[n][1]:14s} {f(p[n]):6.1f} | same plain 16-deep loop: 128x128 {f(k[n]):6.1f}  256x128/4 waves {f(w[n]):6.1f} TF/s")
print(f"sum: product {tp:.3f} ms, plain 128x128 {tk:.3f} ms, plain 256x128 w4 {tw:.3f} ms")
PY

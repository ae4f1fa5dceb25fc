#!/bin/bash
# PROBE: 256x96 tile with four waves of 64x96 and 16-deep K-tiles (cfg 20, DMX_TALL=3, plain loop) against the 128x96 tile in
# the SAME plain 16-deep loop (variant build -DDMX_CFG2_KS=1) and against the product (interleaved 32-deep loop)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
run() { lab=$1; shift; env "$@" MODEL=4s PBS="42" bash tools/gpu_prof.sh 2>&1 | grep -E "^== .*sum of ops"; cp gpurun_out/profile_ops_4s_b42.tsv gpurun_out/tall3_$lab.tsv; }
run product DMX_TALL=0
run w4 DMX_TALL=3
run ks1 DMX_LIB=$R/demucs_cpp_amd/lib/libdemucs_hip_ks1c.so
python - <<'PY'
def load(f): return {l.split("\t")[0]: l.rstrip("\n").split("\t") for l in open(f)}
p, w, k = load("gpurun_out/tall3_product.tsv"), load("gpurun_out/tall3_w4.tsv"), load("gpurun_out/tall3_ks1.tsv")
tp = tw = tk = 0
for n in p:
    if w[n][1] == "igemm_256x96":
        f = lambda r: float(r[3]) / float(r[2]) / 1e9
        tp += float(p[n][2]); tw += float(w[n][2]); tk += float(k[n][2])
        print(f"{n:28s} product {p[n][1]:13s} {float(p[n][2])*1e3:7.0f} us {f(p[n]):6.1f} | plain 16-deep loop: 128x96 {f(k[n]):6.1f}  256x96/4 waves {f(w[n]):6.1f} TF/s")
print(f"sum: product {tp:.3f} ms, plain 128x96 {tk:.3f} ms, plain 256x96 w4 {tw:.3f} ms")
PY

#!/bin/bash
# DESIGN.md 'FFT frames next to bf16 MFMA waves': (1) the standalone reproducer (tools/micro/fft_mfma_repro.hip: the product's
# stft_kernel beside a bare MFMA loop) in four builds of the victim, (2) the library with its containment switched off, with
# the FFT kernels built without packed-fp32 (SLP) vectorisation / with an LDS footprint that keeps split workgroups off their CU.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/fft_erratum; mkdir -p $O
ROUNDS=${ROUNDS:-16}
for v in base noslp pad70k pad100k; do
  echo "== standalone victim build: $v"; timeout 120 tools/micro/fft_mfma_repro_$v $ROUNDS 12 ${MASK:-0xff}
done 2>&1 | tee $O/standalone.log
L=demucs_cpp_amd/lib
for cfg in "product:$L/libdemucs_hip.so:0" "fftnoslp:$L/libdemucs_hip_fftnoslp.so:0" "fftpad:$L/libdemucs_hip_fftpad.so:0" "product:$L/libdemucs_hip.so:1"; do
  IFS=: read tag lib lane <<< "$cfg"
  [ -f $lib ] || continue
  TAG=$tag DMX_LIB=$R/$lib DMX_PLAN_LANE=$lane RUNS=${RUNS:-24} timeout 300 python tools/fft_erratum_diag.py 2>&1 | grep '^\['
done | tee $O/library.log
TAG=f32 DMX_PLAN_LANE=0 MODE=f32 RUNS=12 timeout 300 python tools/fft_erratum_diag.py 2>&1 | grep '^\[' | tee -a $O/library.log

#!/bin/bash
# runs the reproducer on every code object build/pk/*.co (tools/micro/pk_bisect.py variants of the product's stft_kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/fft_erratum
for co in ${COS:-build/pk/*.co}; do
  REPRO_CO=$co timeout 60 tools/micro/fft_mfma_repro_base ${ROUNDS:-8} 12 ${MASK:-0x2} 2>&1 | grep -E '^idle|^aggressor' | cut -c1-110 | sed "s|^|$(basename $co .co): |"
done | tee gpurun_out/fft_erratum/pk_bisect_${TAGN:-classes}.log

#!/bin/bash
# runs the reproducer on every code object build/pk/*.co (tools/micro/pk_bisect.py variants of the product's stft_kernel). Make them
# from the PACKED build of the kernel:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o build/pk/fft_base.s demucs_cpp_amd/csrc/fft.hip
#   python tools/micro/pk_bisect.py list  build/pk/fft_base.s 11stft_kernel
#   python tools/micro/pk_bisect.py build build/pk/fft_base.s 11stft_kernel build/pk/keep_020.co keep:20
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/fft_erratum
for co in ${COS:-build/pk/*.co}; do
  REPRO_CO=$co timeout 60 tests/_build/fft_mfma_repro_pk ${ROUNDS:-8} 12 ${MASK:-0x2} 2>&1 | grep -E '^idle|^aggressor' | cut -c1-110 | sed "s|^|$(basename $co .co): |"
done | tee gpurun_out/fft_erratum/pk_bisect_${TAGN:-classes}.log

#!/usr/bin/env python
"""DESIGN.md 7.1 (the packed-fp32 erratum): two contexts of one model driven from two host threads on ONE GPU, nothing
ordering them. Every run's STFT output (tap x_cac - independent of every later op) and final stems are compared bit for bit
with a quiet single-context run. Parameters (environment): MODE=bf16x3|f32, RUNS (per thread), MB (segments per call),
DMX_LIB (a `make variant1` build, e.g. the FFT kernels WITH packed fp32 arithmetic:
`make variant1 NAME=fftpk FILE=fft FLAGS="-Xclang -target-feature -Xclang +packed-fp32-ops"` shows the failure), TAG (label).
(The round-5 logs under profiles/ were taken while the library still had its round-4 'plan lane', switched off for the run.)"""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model


def run(mode="bf16x3", runs=16, mb=2, seg=343980):
    path = "/tmp/fft_erratum_4s.bin"
    if not os.path.exists(path):
        write_synthetic_model(path, 4, 0)
    dmx.set_default_gemm(dmx.GEMM_BF16X3 if mode == "bf16x3" else dmx.GEMM_F32)
    mix = (0.1 * np.random.default_rng(7).standard_normal((2, seg))).astype(np.float32)
    m = dmx.Model(path)
    ctxs = [dmx.Context(m, 0, mb), dmx.Context(m, 0, mb)]
    ref_out = ctxs[0].segment(mix)
    ref_tap = ctxs[0].tap("x_cac")
    bad = [[0, 0], [0, 0]]  # per thread: wrong STFT outputs, wrong stems with a right STFT output (= the ISTFT)

    def work(k):
        for _ in range(runs):
            out = ctxs[k].segment(mix)
            t_ok = np.array_equal(ctxs[k].tap("x_cac"), ref_tap)
            bad[k][0] += 0 if t_ok else 1
            bad[k][1] += 1 if (t_ok and not np.array_equal(out, ref_out)) else 0
    th = [threading.Thread(target=work, args=(k,)) for k in (0, 1)]
    [t.start() for t in th]
    [t.join() for t in th]
    return bad


if __name__ == "__main__":
    mode = os.environ.get("MODE", "bf16x3")
    runs = int(os.environ.get("RUNS", "16"))
    bad = run(mode, runs, int(os.environ.get("MB", "2")))
    lib = os.path.basename(os.environ.get("DMX_LIB", "libdemucs_hip.so"))
    print(f"[{os.environ.get('TAG', '')} {lib} {mode}] "
          f"of {2 * runs} runs: wrong STFT output {bad[0][0] + bad[1][0]}, right STFT but wrong stems {bad[0][1] + bad[1][1]}", flush=True)

"""TEST INFRASTRUCTURE ONLY (see oracle/demucs_oracle.cpp header): numpy restatement of the coarse
chunk split / cross-fade of the reference's *_mt CLIs,
/root/reference/cli-apps/threaded_inference.hpp:29-193, with the chunk inference passed in.
Pinning: the reference has no test or golden vector for this driver ("parity unpinned" by the
reference itself); this file follows the source line by line and is the checker for
demucs_cpp_amd/host/threaded_inference_hip.hpp (tests/test_threaded_split.py)."""
import math

import numpy as np

OVERLAP_SAMPLES = int(math.floor(44100 * 0.75))  # :25-27


def threaded_split(audio: np.ndarray, num_threads: int, n_sources: int, infer):
    """audio (2, L) float32; infer(i, chunk (2, n)) -> (S, 2, n). Returns (S, 2, L) float32."""
    audio = np.asarray(audio, np.float32)
    L = audio.shape[1]
    OV = OVERLAP_SAMPLES
    seg_len = int(np.ceil(np.float32(L) / np.float32(num_threads)))  # :52-53 (float arithmetic)
    outs = []
    for i in range(num_threads):  # :57-96
        start = min(L, i * seg_len)
        end = min(L, start + seg_len)
        seg = np.zeros((2, end - start + 2 * OV), np.float32)
        if i == 0:
            seg[:, :OV] = audio[:, :1]
        else:
            lo = start - OV
            src = audio[:, max(lo, 0):start]
            seg[:, OV - src.shape[1]:OV] = src
        if i != num_threads - 1:
            src = audio[:, end:end + OV]
            seg[:, end - start + OV:end - start + OV + src.shape[1]] = src
        seg[:, OV:OV + end - start] = audio[:, start:end]
        outs.append(np.asarray(infer(i, seg), np.float32))
    k = np.arange(seg_len)
    ramp = np.minimum(k + 1, seg_len - k).astype(np.float32)  # :134-139
    ramp /= ramp.max()
    final = np.zeros((n_sources, 2, L), np.float32)
    sum_w = np.zeros(L, np.float32)
    for i, o in enumerate(outs):  # :143-171
        for j in range(min(seg_len + 2 * OV, o.shape[2])):
            g = i * seg_len + j - OV
            if g < 0 or g >= L:
                continue
            w = np.float32(1.0)
            if j < OV:
                w = ramp[j] if j < seg_len else np.float32(0)
            elif j >= seg_len:
                r = seg_len + 2 * OV - j - 1
                w = ramp[r] if 0 <= r < seg_len else np.float32(0)
            final[:, :, g] += o[:, :, j] * w
            for _ in range(2 * n_sources):  # the reference adds the weight once per (target, channel)
                sum_w[g] = np.float32(sum_w[g] + w)
    nz = sum_w > 0  # :174-189
    final[:, :, nz] /= (sum_w[nz] / np.float32(2.0 * n_sources))
    return final

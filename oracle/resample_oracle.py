"""TEST INFRASTRUCTURE ONLY (checker of demucs_cpp_amd/csrc/resample.hip; never imported by the product).

CPU restatement, in numpy / float64, of the resampler specification of include/demucs_hip.h (dmx_resample*):

    L / M = rate_out / rate_in in lowest terms,  R = max(L, M),  c = 16 R
    h[i]  = sinc((i - c) / R) * kaiser(2c + 1, beta = 8.6)[i],  scaled so that sum(h) = L
    y[k]  = sum_j x[j] h[c + k M - j L],   k = 0 .. ceil(n L / M) - 1,   x = 0 outside [0, n)

There is NO reference arithmetic for this step: /root/reference rejects input that is not 44.1 kHz
(cli-apps/demucs.cpp:30-36) and delegates decoding to libnyquist, which does not resample either. The oracle is
therefore pinned against an independent implementation of the same textbook operation:
scipy.signal.resample_poly(x, L, M, window=h / L) (upfirdn with the filter centred on the first input sample; scipy
multiplies a given filter by `up`) - see
tests/test_resample.py::test_oracle_equals_scipy_resample_poly."""
import math

import numpy as np

ZERO_CROSSINGS = 16
KAISER_BETA = 8.6


def ratio(rate_in: int, rate_out: int):
    g = math.gcd(rate_in, rate_out)
    return rate_out // g, rate_in // g  # L (up), M (down)


def design(rate_in: int, rate_out: int) -> np.ndarray:
    """h[0 .. 2c] in float64"""
    L, M = ratio(rate_in, rate_out)
    R = max(L, M)
    c = ZERO_CROSSINGS * R
    i = np.arange(2 * c + 1, dtype=np.float64)
    h = np.sinc((i - c) / R) * np.kaiser(2 * c + 1, KAISER_BETA)
    return h * (L / h.sum())


def out_length(n: int, rate_in: int, rate_out: int) -> int:
    L, M = ratio(rate_in, rate_out)
    return -(-n * L // M)


def resample(x: np.ndarray, rate_in: int, rate_out: int, taps: np.ndarray = None) -> np.ndarray:
    """x (..., n) -> (..., ceil(n L / M)), float64 arithmetic, the defining sum evaluated phase by phase."""
    L, M = ratio(rate_in, rate_out)
    h = design(rate_in, rate_out) if taps is None else np.asarray(taps, np.float64)
    c = (len(h) - 1) // 2
    x = np.asarray(x, np.float64)
    n = x.shape[-1]
    m = out_length(n, rate_in, rate_out)
    T = -(-len(h) // L)
    hp = np.zeros((L, T))
    for p in range(L):
        seg = h[p::L]
        hp[p, : len(seg)] = seg
    k = np.arange(m, dtype=np.int64)
    u = c + k * M
    jhi = u // L
    ph = (u - jhi * L).astype(np.int64)
    y = np.zeros(x.shape[:-1] + (m,))
    xp = np.concatenate([x, np.zeros(x.shape[:-1] + (1,))], axis=-1)  # index n = the zero outside the signal
    for i in range(T):
        j = jhi - i
        jj = np.where((j >= 0) & (j < n), j, n)
        y += xp[..., jj] * hp[ph, i]
    return y

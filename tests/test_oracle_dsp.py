"""Pins the oracle's DSP against the reference's own machine-checked test:
/root/reference/test/test_dsp.cpp:106-196 (STFT->ISTFT round trip, 2049 bins,
NEAR_TOLERANCE 1e-4) on random data and on the reference fixture gspi_mono.wav, plus
the fp64 torch STFT golden frames (tests/golden/golden_prims.npz)."""
import os

import numpy as np

import oracle_lib as orc
from wavio import read_wav

NEAR_TOLERANCE = 1e-4  # test/test_dsp.cpp:13


def test_wav_fixture_loads(golden_dir):
    # test/test_dsp.cpp:75-88: 262144 samples, mono duplicated -> L == R
    rate, a = read_wav(os.path.join(golden_dir, "gspi_mono.wav"))
    assert rate == 44100 and a.shape == (1, 262144)
    rate, s = read_wav(os.path.join(golden_dir, "gspi_stereo_short.wav"))  # has a LIST chunk before data
    assert rate == 44100 and s.shape[0] == 2 and s.shape[1] == 44100


def test_stft_roundtrip_random():
    # test/test_dsp.cpp:106-158: 4096 random samples in [0,1]
    rng = np.random.default_rng(0)
    x = rng.random((2, 4096)).astype(np.float32)
    spec = orc.stft(x)
    assert spec.shape[1] == 2049 and spec.shape[2] == 4096 // 1024 + 1
    y = orc.istft(spec, 4096)
    assert np.abs(x - y).max() < NEAR_TOLERANCE


def test_stft_roundtrip_glockenspiel(golden_dir):
    # test/test_dsp.cpp:162-196
    _, a = read_wav(os.path.join(golden_dir, "gspi_mono.wav"))
    x = np.concatenate([a, a], axis=0)
    spec = orc.stft(x)
    assert spec.shape[1] == 2049
    y = orc.istft(spec, x.shape[1])
    assert y.shape == x.shape
    assert np.abs(x - y).max() < NEAR_TOLERANCE


def test_stft_frames_match_fp64_torch(golden_dir):
    g = np.load(os.path.join(golden_dir, "golden_prims.npz"))
    x = g["stft_x"]
    spec = orc.stft(x)  # (2, 2049, 9): frames 2..6 are independent of the stft-internal padding
    ref = g["stft_z_re"] + 1j * g["stft_z_im"]
    got = spec[:, :, 2:2 + ref.shape[2]]
    assert np.abs(got - ref).max() < 2e-6 * max(1.0, np.abs(ref).max())


def test_stft_edge_frames_symmetric_padding():
    # Q2: frame 0 sees the symmetric (edge-duplicating) extension, src/dsp.cpp:19-38
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 8192)).astype(np.float32)
    spec = orc.stft(x)
    w = 0.5 * (1 - np.cos(2 * np.pi * np.arange(4096) / 4096))
    xp = np.pad(x.astype(np.float64), ((0, 0), (2048, 2048)), mode="symmetric")
    for fr in (0, spec.shape[2] - 1):
        ref = np.fft.rfft(xp[:, fr * 1024:fr * 1024 + 4096] * w, axis=-1) / 64.0
        assert np.abs(spec[:, :, fr] - ref).max() < 5e-6 * np.abs(ref).max()

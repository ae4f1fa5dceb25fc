"""Sample-rate conversion (include/demucs_hip.h dmx_resample*): the numpy oracle against
scipy (independent implementation of the same polyphase operation), the product's host-side filter design and length
arithmetic against the oracle (CPU), and - marked gpu - the HIP kernel against the oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import resample_oracle as ro  # noqa: E402

RATES = [(48000, 44100), (44100, 48000), (22050, 44100), (96000, 44100), (8000, 44100), (44100, 32000), (44100, 44100), (11025, 44100),
         (192000, 8000), (44100, 44056)]  # + a strong decimation (3073 taps per output) and a table that does not fit LDS


@pytest.fixture(scope="module")
def dmx():
    from demucs_cpp_amd import binding
    return binding


@pytest.mark.parametrize("rin,rout", RATES[:6])
def test_oracle_equals_scipy_resample_poly(rin, rout):
    """pins the oracle: the same filter through scipy.signal.resample_poly (upfirdn, filter centred on the first input
    sample, ceil(n L / M) outputs) gives the same numbers (float64, 1e-12)"""
    from scipy.signal import resample_poly
    rng = np.random.default_rng(1)
    L, M = ro.ratio(rin, rout)
    h = ro.design(rin, rout)
    for n in (1, 2, 37, 1000, 4801):
        x = rng.standard_normal((2, n))
        want = resample_poly(x, L, M, axis=-1, window=h / L)  # scipy multiplies the given filter by `up`
        got = ro.resample(x, rin, rout)
        assert got.shape == want.shape == (2, ro.out_length(n, rin, rout))
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())


def test_oracle_preserves_dc_and_a_passband_tone():
    x = np.ones((1, 20000))
    y = ro.resample(x, 48000, 44100)
    assert np.abs(y[0, 2000:-2000] - 1.0).max() < 1e-4  # unity DC gain away from the edges
    t = np.arange(48000) / 48000.0
    tone = np.sin(2 * np.pi * 1000.0 * t)[None]
    y = ro.resample(tone, 48000, 44100)[0]
    t2 = np.arange(len(y)) / 44100.0
    assert np.abs(y[3000:-3000] - np.sin(2 * np.pi * 1000.0 * t2)[3000:-3000]).max() < 1e-4
    t96 = np.arange(96000) / 96000.0
    hi = np.sin(2 * np.pi * 30000.0 * t96)[None]  # beyond the new Nyquist and the transition band: suppressed (aliasing)
    assert np.abs(ro.resample(hi, 96000, 44100)[0, 3000:-3000]).max() < 1e-4


@pytest.mark.parametrize("rin,rout", RATES)
def test_product_filter_and_length_match_the_oracle(rin, rout, dmx):
    """dmx_resample_filter / dmx_resample_length are host functions: no GPU needed"""
    up, down, taps = dmx.resample_filter(rin, rout)
    assert (up, down) == ro.ratio(rin, rout)
    h = ro.design(rin, rout)
    assert taps.shape == h.shape
    assert np.abs(taps - h).max() <= 2e-7 * np.abs(h).max()
    for n in (0, 1, 2, 999, 10584000):
        assert dmx.resample_length(n, rin, rout) == ro.out_length(n, rin, rout)
    assert dmx.resample_length(-1, rin, rout) == -1 and dmx.resample_length(10, 0, rout) == -1


def test_resample_fails_loudly_without_a_gpu_and_on_bad_arguments(dmx):
    with pytest.raises(dmx.DmxError):
        dmx.resample_filter(0, 44100)
    dmx.resample_filter(44101, 44100)  # an awkward ratio (44100 phases x 33 taps) is accepted ...
    with pytest.raises(dmx.DmxError):
        dmx.resample_filter(768001, 44100)  # ... a rate beyond 768 kHz is not
    with pytest.raises(dmx.DmxError):
        dmx.resample_filter(700001, 768000)  # ... nor a ratio whose polyphase table would hold > 16 M coefficients
    if dmx.device_count() == 0:
        with pytest.raises(dmx.DmxError) as e:
            dmx.resample(np.zeros((2, 100), np.float32), 48000, 44100)
        assert "no HIP device" in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("rin,rout", RATES)
def test_hip_resampler_vs_oracle(rin, rout, dmx):
    """fp32 fmaf chain on the GPU vs the float64 oracle (with the product's own fp32 taps): 2e-6 of max-abs; planar and
    interleaved layouts give the same bits; lengths 1, 2 and ragged"""
    rng = np.random.default_rng(7)
    _, _, taps = dmx.resample_filter(rin, rout)
    for n in (1, 2, 37, 4801, 100003):
        x = (0.3 * rng.standard_normal((2, n))).astype(np.float32)
        want = ro.resample(x, rin, rout, taps=taps)
        got = dmx.resample(x, rin, rout)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1e-3), (n, np.abs(got - want).max())
        got_i = dmx.resample(np.ascontiguousarray(x.T), rin, rout, interleaved=True)
        assert np.array_equal(got_i.T, got)
    six = (0.3 * rng.standard_normal((12, 5000))).astype(np.float32)  # the 6 x 2 planes of a stem tensor
    assert np.abs(dmx.resample(six, rin, rout) - ro.resample(six, rin, rout, taps=taps)).max() < 2e-6


@pytest.mark.gpu
def test_hip_resampler_round_trip_of_a_band_limited_signal(dmx):
    """size-independent property at track length: 44.1 kHz -> 48 kHz -> 44.1 kHz returns a band-limited signal
    (4 minutes, 10 584 000 samples) to within the filter's passband ripple"""
    n = 10584000
    t = np.arange(n, dtype=np.float64) / 44100.0
    x = (0.4 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 5000.0 * t + 1.0)).astype(np.float32)[None]
    up = dmx.resample(x, 44100, 48000)
    assert up.shape == (1, dmx.resample_length(n, 44100, 48000))
    back = dmx.resample(up, 48000, 44100)
    assert back.shape[1] == dmx.resample_length(up.shape[1], 48000, 44100) >= n
    assert np.abs(back[0, 5000:n - 5000] - x[0, 5000:n - 5000]).max() < 2e-4


@pytest.mark.gpu
def test_cli_accepts_48k_input_with_dmx_resample(dmx, tmp_path):
    """cli/demucs.cpp.main: a 48 kHz file is rejected with the reference's message (cli-apps/demucs.cpp:30-36) by
    default; with DMX_RESAMPLE=1 it is converted to 44.1 kHz on the GPU, separated, and the stems are written at
    48 kHz: bit-identical to resample -> Context.track -> resample through the library API."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from wavio import read_wav, write_wav_f32
    from demucs_cpp_amd.weights import write_synthetic_model
    exe = os.path.join(ROOT, "cli", "demucs.cpp.main")
    assert os.path.exists(exe), "CLI not built"
    model = str(tmp_path / "ggml-model-htdemucs-4s-f16.bin")
    write_synthetic_model(model, 4, 5)
    n48 = 48000 * 3 + 123
    t = np.arange(n48) / 48000.0
    audio48 = np.stack([0.3 * np.sin(2 * np.pi * 330.0 * t) + 0.05 * np.random.default_rng(3).standard_normal(n48),
                        0.2 * np.sin(2 * np.pi * 1234.0 * t + 0.5)]).astype(np.float32)
    wav = str(tmp_path / "in48k.wav")
    write_wav_f32(wav, audio48, rate=48000)
    out_dir = tmp_path / "stems"
    env = dict(os.environ, DMX_SHIFT_OFFSET="1337", DMX_BATCH="2")
    r = subprocess.run([exe, model, wav, str(out_dir)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "only supports the following sample rate" in r.stderr
    r = subprocess.run([exe, model, wav, str(out_dir)], env=dict(env, DMX_RESAMPLE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Converted 48000 Hz -> 44100 Hz on the GPU" in r.stdout
    a441 = dmx.resample(audio48, 48000, 44100)
    m = dmx.Model(model); ctx = dmx.Context(m, 0, 2)
    ref = ctx.track(a441, 1337)
    ctx.close(); m.close()
    for i, name in enumerate(["drums", "bass", "other", "vocals"]):
        rate, stem = read_wav(str(out_dir / f"target_{i}_{name}.wav"))
        want = dmx.resample(ref[i], 44100, 48000)
        # ceil(ceil(n 147/160) 160/147) may exceed n by a sample or two: the stems are cut back to the file's own length
        assert rate == 48000 and stem.shape == (2, n48) and want.shape[1] >= n48
        assert np.array_equal(stem, want[:, :n48])

"""The coarse chunk split / cross-fade of the *_mt CLIs: the product's C++
driver (demucs_cpp_amd/host/threaded_inference_hip.hpp) against the numpy restatement of
/root/reference/cli-apps/threaded_inference.hpp in oracle/threaded_split.py, with a stand-in chunk
inference (CPU only; the GPU-backed CLI is exercised in test_gpu_parity.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from threaded_split import OVERLAP_SAMPLES, threaded_split  # noqa: E402

EXE = os.path.join(ROOT, "tests", "_build", "threaded_harness")


@pytest.fixture(scope="module")
def harness():
    src = os.path.join(ROOT, "tests", "threaded_harness.cpp")
    hdr = os.path.join(ROOT, "demucs_cpp_amd", "host", "threaded_inference_hip.hpp")
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "demucs_cpp_amd", "host"), "-o", EXE, src])
    return EXE


def stand_in(S):
    def infer(i, chunk):
        out = np.zeros((S, 2, chunk.shape[1]), np.float32)
        for s in range(S):
            out[s] = np.float32(s + 1) * chunk + np.float32(0.01) * np.float32(s) * np.float32(i + 1)
        return out
    return infer


def test_overlap_constant():
    assert OVERLAP_SAMPLES == 33075


@pytest.mark.parametrize("L,T,S", [(400000, 4, 4), (400001, 3, 6), (300000, 1, 4), (2 * 33075 + 10, 2, 4), (123457, 5, 4)])
def test_product_driver_matches_restatement(L, T, S, harness, tmp_path):
    rng = np.random.default_rng(L + T)
    audio = rng.standard_normal((2, L)).astype(np.float32)
    fin, fout = str(tmp_path / "in.f32"), str(tmp_path / "out.f32")
    audio.tofile(fin)
    subprocess.check_call([harness, str(L), str(T), str(S), fin, fout])
    got = np.fromfile(fout, np.float32).reshape(S, 2, L)
    ref = threaded_split(audio, T, S, stand_in(S))
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6)


def test_single_chunk_of_identity_inference_returns_the_track(harness, tmp_path):
    """Property: with ONE chunk and an inference that returns its input for every stem, the
    recombination must hand back the track itself (weights normalise away)."""
    L, S = 200000, 4
    rng = np.random.default_rng(5)
    audio = rng.standard_normal((2, L)).astype(np.float32)
    ref = threaded_split(audio, 1, S, lambda i, c: np.broadcast_to(c, (S,) + c.shape).copy())
    # interior (away from the ramps) is exact; ramps only re-weight a single contribution
    for s in range(S):
        np.testing.assert_allclose(ref[s], audio, rtol=1e-5, atol=1e-6)

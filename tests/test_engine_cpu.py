"""CPU-side coverage of round-2 host logic: the (model, segment) dealing of csrc/engine.cpp, the
error behaviour of the engine entry points without a GPU, the stress weight variants through the
test-only plan interpreter against the oracle, and a compile check of the Eigen-typed shim overloads
(against tests/eigen_stub, a stand-in that is NOT Eigen - Eigen is absent from this image)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as orc
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("nseg,G", [([42], 8), ([42], 1), ([42], 3), ([42, 42, 42, 42], 8), ([42, 41, 42, 43], 5), ([3], 8), ([1, 1, 1, 1], 2)])
def test_partition_is_contiguous_balanced_and_complete(nseg, G):
    runs = dmx.engine_partition(nseg, G)
    T = sum(nseg)
    seen = {m: [] for m in range(len(nseg))}
    counts = []
    flat = []
    for l, rl in enumerate(runs):
        counts.append(sum(g1 - g0 for _, g0, g1 in rl))
        assert [m for m, _, _ in rl] == sorted(m for m, _, _ in rl)  # model-major within a device
        for m, g0, g1 in rl:
            seen[m].append((g0, g1))
            flat.append((m, g0, g1))
    assert max(counts) - min(counts) <= 1 and sum(counts) == T      # balanced, complete
    for m, rs in seen.items():                                      # every segment exactly once, in order
        rs.sort()
        assert rs[0][0] == 0 and rs[-1][1] == nseg[m]
        assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
    assert flat == sorted(flat)                                     # devices own increasing item ranges
    if nseg == [42, 42, 42, 42] and G == 8:                         # BASELINE configs[4]: 21 items = one batch per GPU
        assert counts == [21] * 8 and all(len(rl) == 1 for rl in runs)


def test_engine_entry_points_fail_loudly_without_a_gpu(tmp_models):
    if dmx.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(dmx.DmxError) as e:
        dmx.Engine([tmp_models[4]], [0])
    assert e.value.code == 3  # DMX_ERR_NO_DEVICE: no CPU fallback
    with pytest.raises(dmx.DmxError) as e:
        dmx.Engine(["/nonexistent/model.bin"], [0])
    assert e.value.code == 1  # file errors are reported like load_demucs_model does (model_load.cpp:64-69)


@pytest.fixture(scope="module")
def interp():
    so = os.path.join(ROOT, "tests", "_build", "libcpu_interp.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", ROOT, "interp"], stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(so)
    L.interp_create.restype = ctypes.c_void_p
    L.interp_create.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int]
    L.interp_free.argtypes = [ctypes.c_void_p]
    L.interp_run.argtypes = [ctypes.c_void_p] * 3
    return L


@pytest.mark.parametrize("variant", ["dc", "illcond", "initscale"])
def test_stress_weights_plan_vs_oracle(variant, interp, tmp_path):
    """The product's algorithmic choices that differ from the reference (one-pass sum / sum-of-squares
    statistics from fp32 row partials vs the two-pass calculate_variance of layers.hpp:76-95; the DConv
    statistics through a factor of W^T W) on models built to hurt them, DC-offset input included. The same
    models run at full size on the GPU in test_gpu_parity.py."""
    seg = 8000
    path = str(tmp_path / f"stress_{variant}-4s.bin")
    write_synthetic_model(path, 4, 7, variant)
    mix = (0.1 * np.random.default_rng(0).standard_normal((2, seg)) + 0.3).astype(np.float32)
    h = interp.interp_create(path.encode(), seg, 1)
    mi = np.ascontiguousarray(mix.T)[None]
    out = np.zeros((1, 4, 2, seg), np.float32)
    interp.interp_run(h, mi.ctypes.data, out.ctypes.data)
    interp.interp_free(h)
    om = orc.OracleModel(path)
    ref = om.segment(mix)
    om.close()
    err = np.abs(out[0] - ref).max() / np.abs(ref).max()
    assert np.isfinite(out).all() and err < 2e-5, err


def test_eigen_typed_shim_overloads_compile(tmp_path):
    """demucscpp_hip.hpp under -DDEMUCSCPP_HIP_WITH_EIGEN (reference signatures of src/model.hpp:569-666 on
    Eigen::MatrixXf / Eigen::Tensor<float,3>) type-checks and links; run on the GPU by test_gpu_parity.py."""
    exe = str(tmp_path / "shim_eigen")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-DDEMUCSCPP_HIP_WITH_EIGEN", "-I" + os.path.join(ROOT, "tests", "eigen_stub"),
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "demucs_cpp_amd", "host"), "-o", exe,
                           os.path.join(ROOT, "tests", "shim_harness.cpp"), "-L" + os.path.join(ROOT, "demucs_cpp_amd", "lib"),
                           "-ldemucs_hip", "-lpthread", "-Wl,-rpath," + os.path.join(ROOT, "demucs_cpp_amd", "lib")])
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == 2  # usage


def test_cli_rejects_unsupported_wav_encodings_cleanly(tmp_path):
    """cli/wav.hpp: a 4-bit ADPCM file (bits / 8 == 0) and a truncated WAVE_FORMAT_EXTENSIBLE fmt chunk end with
    the error message and exit code 1 (cf. /root/reference/cli-apps/demucs.cpp:30-48), not with a signal."""
    import struct
    exe = os.path.join(ROOT, "cli", "demucs.cpp.main")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")

    def wav(path, tag, bits, fmt_len=16, nch=2, rate=44100):
        data = b"\x00" * 64
        fmt = struct.pack("<HHIIHH", tag, nch, rate, rate * nch * max(bits, 8) // 8, nch * max(bits, 8) // 8, bits)
        fmt = fmt + b"\x00" * (fmt_len - 16)
        body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(data)) + data
        open(path, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)

    p1, p2 = str(tmp_path / "adpcm.wav"), str(tmp_path / "ext.wav")
    wav(p1, 2, 4)
    wav(p2, 0xFFFE, 16, fmt_len=18)  # claims EXTENSIBLE but carries no sub-format GUID
    for p in (p1, p2):
        r = subprocess.run([exe, "/nonexistent.bin", p, str(tmp_path / "o")], capture_output=True, text=True)
        assert r.returncode == 1 and "unsupported wav encoding" in r.stderr, (r.returncode, r.stderr)

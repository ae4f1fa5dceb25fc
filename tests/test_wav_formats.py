"""cli/wav.hpp decodes every WAV encoding it accepts sample by sample (CPU; tests/wav_harness.cpp dumps what the CLI's
reader produced). Expected values come from independent decoders: numpy for the PCM / float layouts, Python's `audioop`
for the G.711 companded encodings (ITU-T tables)."""
import os
import struct
import subprocess
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "_build", "wav_harness")


def write(path, tag, bits, nch, rate, payload, extensible=False):
    align = nch * bits // 8
    if extensible:
        guid = struct.pack("<H", tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
        fmt = struct.pack("<HHIIHH", 0xFFFE, nch, rate, rate * align, align, bits) + struct.pack("<HHI", 22, bits, 3) + guid
    else:
        fmt = struct.pack("<HHIIHH", tag, nch, rate, rate * align, align, bits)
    body = b"WAVE" + b"LIST" + struct.pack("<I", 4) + b"INFO" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
    if len(payload) & 1:
        body += b"\x00"
    open(path, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)


def decode(path, tmp_path):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", ROOT, "harness"], stdout=subprocess.DEVNULL)
    out = str(tmp_path / "dump.f32")
    r = subprocess.run([EXE, path, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.fromfile(out, np.float32).reshape(-1, 2).T, r.stdout


@pytest.mark.parametrize("nch", [1, 2])
@pytest.mark.parametrize("kind", ["pcm8", "pcm16", "pcm24", "pcm32", "f32", "f64", "ulaw", "alaw", "pcm16ext"])
def test_every_supported_encoding_decodes_exactly(kind, nch, tmp_path):
    rng = np.random.default_rng(11)
    n = 1001  # odd: the data chunk of the 8-bit mono files needs its pad byte
    ints = rng.integers(-32768, 32768, size=(n, nch))
    ints[0], ints[1], ints[2] = -32768, 32767, 0
    p = str(tmp_path / f"{kind}.wav")
    if kind == "pcm8":
        u = ((ints >> 8) + 128).astype(np.uint8)
        write(p, 1, 8, nch, 44100, u.tobytes())
        want = (u.astype(np.float32) - 128.0) / 128.0
    elif kind in ("pcm16", "pcm16ext"):
        v = ints.astype("<i2")
        write(p, 1, 16, nch, 44100, v.tobytes(), extensible=kind == "pcm16ext")
        want = v.astype(np.float32) / 32768.0
    elif kind == "pcm24":
        v = (ints.astype(np.int64) << 8) + rng.integers(0, 256, size=(n, nch))
        raw = b"".join(int(x).to_bytes(3, "little", signed=True) for x in v.reshape(-1))
        write(p, 1, 24, nch, 44100, raw)
        want = v.astype(np.float32) / 8388608.0
    elif kind == "pcm32":
        v = (ints.astype(np.int64) << 16) + rng.integers(0, 65536, size=(n, nch))
        write(p, 1, 32, nch, 44100, v.astype("<i4").tobytes())
        want = (v.astype(np.float64) / 2147483648.0).astype(np.float32)
        want = np.float32(v.astype(np.int32)) / np.float32(2147483648.0)  # the reader converts int32 -> float32 first
    elif kind == "f32":
        v = (ints / 32768.0).astype("<f4")
        write(p, 3, 32, nch, 44100, v.tobytes())
        want = v
    elif kind == "f64":
        v = (ints / 32768.0 + 1e-9).astype("<f8")
        write(p, 3, 64, nch, 44100, v.tobytes())
        want = v.astype(np.float32)
    else:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", DeprecationWarning)
            audioop = pytest.importorskip("audioop")
        lin = ints.astype("<i2").tobytes()
        comp = audioop.lin2ulaw(lin, 2) if kind == "ulaw" else audioop.lin2alaw(lin, 2)
        write(p, 7 if kind == "ulaw" else 6, 8, nch, 44100, comp)
        back = audioop.ulaw2lin(comp, 2) if kind == "ulaw" else audioop.alaw2lin(comp, 2)
        want = np.frombuffer(back, "<i2").reshape(n, nch).astype(np.float32) / 32768.0
        assert np.abs(want - ints / 32768.0).max() < 0.04  # companding error, not ours
    got, out = decode(p, tmp_path)
    assert "rate 44100" in out and got.shape == (2, n)
    want = np.asarray(want, np.float32).reshape(n, nch).T
    if nch == 1:
        want = np.repeat(want, 2, axis=0)  # mono is duplicated to both channels (cli-apps/demucs.cpp:56-64)
    assert np.array_equal(got, want), np.abs(got - want).max()


def test_all_256_g711_codes_match_audioop(tmp_path):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        audioop = pytest.importorskip("audioop")
    codes = bytes(range(256))
    for tag, dec in ((7, audioop.ulaw2lin), (6, audioop.alaw2lin)):
        p = str(tmp_path / f"g711_{tag}.wav")
        write(p, tag, 8, 1, 44100, codes)
        got, _ = decode(p, tmp_path)
        want = np.frombuffer(dec(codes, 2), "<i2").astype(np.float32) / 32768.0
        assert np.array_equal(got[0], want) and np.array_equal(got[1], want)

"""Minimal RIFF/WAVE reader for tests (PCM16/24/32/float32; skips unknown chunks such as LIST)."""
import struct

import numpy as np


def read_wav(path):
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        cid, sz = b[pos:pos + 4], struct.unpack_from("<I", b, pos + 4)[0]
        if cid == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", b, pos + 8)
        elif cid == b"data":
            data = b[pos + 8:pos + 8 + sz]
        pos += 8 + sz + (sz & 1)
    tag, nch, rate, _, _, bits = fmt
    if tag == 1 and bits == 16:
        a = np.frombuffer(data, "<i2").astype(np.float32) / 32768.0
    elif tag == 3 and bits == 32:
        a = np.frombuffer(data, "<f4").astype(np.float32)
    else:
        raise ValueError("unsupported wav")
    return rate, a.reshape(-1, nch).T.copy()


def write_wav_f32(path, audio, rate=44100):
    """audio (2, n) float32 -> stereo IEEE-float WAV (the format the CLIs write)."""
    import struct

    import numpy as np

    audio = np.asarray(audio, np.float32)
    data = np.ascontiguousarray(audio.T).tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 3, 2, rate, rate * 8, 8, 32))
        f.write(b"data" + struct.pack("<I", len(data)) + data)

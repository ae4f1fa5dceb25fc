#!/usr/bin/env python
"""SDR harness: counterpart of /root/reference/scripts/evaluate-demixed-output.py without museval / musdb.

The reference script calls museval.eval_mus_track (BSS Eval v4, "images" variant, 1 s windows, 1 s hop) and
.github/SDR_scores.md quotes the per-target median over windows. In that variant the true source image is
the reference stem itself, s_true = ref, and the three error terms sum to est - ref, so

    SDR(window) = 10 log10( sum_{c,n} ref^2 / sum_{c,n} (est - ref)^2 )

(both channels pooled; the 512-tap distortion filters only split the error into ISR / SIR / SAR). Windows whose
reference or estimate is silent are NaN and skipped by the median, as museval does. This file computes exactly
that from plain WAV stems:

    eval_sdr.py <reference dir> <estimate dir>
        <reference dir> : {drums,bass,other,vocals}.wav of one MUSDB18-HQ track (44.1 kHz stereo)
        <estimate dir>  : target_{i}_{name}.wav as written by the demucs*.cpp.main drivers
prints one line per target in the format of SDR_scores.md. Reference values for 'Zeno - Signs', shift offset
1337 (.github/SDR_scores.md:16-20): vocals 8.370, drums 10.002, bass 4.021, other 7.469 (4-source model).
"""
from __future__ import annotations

import os
import struct
import sys
from typing import Dict

import numpy as np

TARGETS = ["drums", "bass", "other", "vocals", "guitar", "piano"]  # target_{i}_{name}.wav, cli-apps/demucs.cpp:185-204
SDR_SCORES_MD_CPP_4S = {"vocals": 8.370, "drums": 10.002, "bass": 4.021, "other": 7.469}      # .github/SDR_scores.md:16-20
SDR_SCORES_MD_CPP_6S = {"vocals": 8.395, "drums": 9.922, "bass": 4.523, "other": 0.167}       # :38-42
SDR_SCORES_MD_CPP_FT = {"vocals": 8.679, "drums": 10.480, "bass": 4.590, "other": 7.370}      # :56-60
SDR_SCORES_MD_CPP_V3 = {"vocals": 8.332, "drums": 9.285, "bass": 3.668, "other": 7.130}       # :82-86 (hdemucs_mmi, demucs_v3.cpp)


def read_wav(path: str):
    """(rate, float32 array (n, channels)) for PCM16/24/32 and float32 WAV files."""
    with open(path, "rb") as f:
        b = f.read()
    if b[:4] != b"RIFF" or b[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        cid, sz = b[pos:pos + 4], struct.unpack_from("<I", b, pos + 4)[0]
        if cid == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", b, pos + 8)
            if fmt[0] == 0xFFFE and sz >= 26:
                fmt = (struct.unpack_from("<H", b, pos + 8 + 24)[0],) + fmt[1:]
        elif cid == b"data":
            data = b[pos + 8:pos + 8 + sz]
        pos += 8 + sz + (sz & 1)
    if fmt is None or data is None:
        raise ValueError(f"{path}: malformed wav")
    tag, nch, rate, _, _, bits = fmt
    if tag == 3 and bits == 32:
        a = np.frombuffer(data, "<f4").astype(np.float32)
    elif tag == 1 and bits == 16:
        a = np.frombuffer(data, "<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        a = np.frombuffer(data, "<i4").astype(np.float32) / 2147483648.0
    elif tag == 1 and bits == 24:
        r = np.frombuffer(data[:len(data) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
        a = (((r[:, 0] | (r[:, 1] << 8) | (r[:, 2] << 16)) << 8) >> 8).astype(np.float32) / 8388608.0
    else:
        raise ValueError(f"{path}: unsupported encoding (tag {tag}, {bits} bits)")
    return rate, a[:len(a) // nch * nch].reshape(-1, nch)


def sdr_framewise(ref: np.ndarray, est: np.ndarray, win: int = 44100, hop: int = 44100) -> np.ndarray:
    """ref, est: (n, channels). SDR per window (dB), NaN where the reference or the estimate is silent."""
    ref = np.asarray(ref, np.float64)
    est = np.asarray(est, np.float64)
    n = min(len(ref), len(est))
    ref, est = ref[:n], est[:n]
    starts = list(range(0, max(n - win, 0) + 1, hop)) if n >= win else [0]
    out = np.full(len(starts), np.nan)
    for k, s in enumerate(starts):
        r, e = ref[s:s + win], est[s:s + win]
        num, den = float((r * r).sum()), float(((e - r) ** 2).sum())
        if num == 0.0 or float((e * e).sum()) == 0.0:
            continue
        out[k] = np.inf if den == 0.0 else 10.0 * np.log10(num / den)
    return out


def track_sdr(refs: Dict[str, np.ndarray], ests: Dict[str, np.ndarray]) -> Dict[str, float]:
    """median over the 1 s windows per target (what eval_mus_track reports / SDR_scores.md quotes)."""
    res = {}
    for name, ref in refs.items():
        if name not in ests:
            continue
        f = sdr_framewise(ref, ests[name])
        f = f[np.isfinite(f)]
        res[name] = float(np.median(f)) if len(f) else float("nan")
    return res


def load_dirs(ref_dir: str, est_dir: str):
    refs, ests = {}, {}
    for i, name in enumerate(TARGETS):
        rp, ep = os.path.join(ref_dir, f"{name}.wav"), os.path.join(est_dir, f"target_{i}_{name}.wav")
        if os.path.exists(rp) and os.path.exists(ep):
            rate_r, refs[name] = read_wav(rp)
            rate_e, ests[name] = read_wav(ep)
            if rate_r != 44100 or rate_e != 44100:
                raise ValueError("44.1 kHz stems expected")
    return refs, ests


REFERENCE_TABLES = {"htdemucs_4s": SDR_SCORES_MD_CPP_4S, "htdemucs_6s": SDR_SCORES_MD_CPP_6S, "htdemucs_ft": SDR_SCORES_MD_CPP_FT,
                    "hdemucs_mmi": SDR_SCORES_MD_CPP_V3}


def main(argv):
    """eval_sdr.py <reference dir> <estimate dir>
    eval_sdr.py --track <reference dir> --estimates <estimate dir> [--reference-table htdemucs_4s|htdemucs_6s|htdemucs_ft|hdemucs_mmi]
                [--tolerance 0.1]      prints the reference's own C++ score (.github/SDR_scores.md) and the verdict beside each
                                       target; the exit code is the number of targets outside the tolerance"""
    ref_dir = est_dir = table = None
    tol = 0.1
    args = argv[1:]
    if len(args) == 2 and not args[0].startswith("--"):
        ref_dir, est_dir = args
    else:
        it = iter(args)
        for a in it:
            if a == "--track":
                ref_dir = next(it, None)
            elif a == "--estimates":
                est_dir = next(it, None)
            elif a == "--reference-table":
                table = next(it, None)
            elif a == "--tolerance":
                tol = float(next(it, "0.1"))
            else:
                print(main.__doc__)
                return 2
    if not ref_dir or not est_dir or (table is not None and table not in REFERENCE_TABLES):
        print(__doc__)
        print(main.__doc__)
        return 2
    refs, ests = load_dirs(ref_dir, est_dir)
    if not refs:
        print("no matching stems found", file=sys.stderr)
        return 1
    want = REFERENCE_TABLES.get(table, {})
    bad = 0
    for name, v in track_sdr(refs, ests).items():
        line = f"{name:15s} ==> SDR: {v:7.3f}"
        if name in want:
            ok = abs(v - want[name]) <= tol
            bad += 0 if ok else 1
            line += f"   reference C++ {want[name]:7.3f}   delta {v - want[name]:+.3f} dB   {'OK' if ok else 'OUTSIDE +-%.2f dB' % tol}"
        print(line)
    return bad


if __name__ == "__main__":
    sys.exit(main(sys.argv))

#!/usr/bin/env python
"""Throughput of the GPU sample-rate converter (csrc/resample.hip) with its input resident in HBM, against the HBM
roofline, with a CPU implementation of the same operation (scipy.signal.resample_poly with the product's own filter
taps - the library the oracle is pinned against; nothing under oracle/ is used here) timed beside it. One JSON line
per case.

Algorithmic bytes per launch = 4 B x planes x (n_in + n_out) (every input sample read once, every output written
once; the polyphase table, 20 KB for 48 <-> 44.1 kHz, stays in cache)."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demucs_cpp_amd import binding as dmx  # noqa: E402

L = dmx.lib()
L.dmx_resample_device.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
PEAK = 8000.0  # GB/s, MI355X_MICROARCH.md

for label, rin, rout, planes, seconds in (("track in: 48 kHz -> 44.1 kHz, interleaved stereo, 4 min", 48000, 44100, 2, 240),
                                          ("stems out: 44.1 kHz -> 48 kHz, 4 x 2 planes, 4 min", 44100, 48000, 8, 240),
                                          ("track in: 96 kHz -> 44.1 kHz, interleaved stereo, 4 min", 96000, 44100, 2, 240)):
    n = rin * seconds
    m = dmx.resample_length(n, rin, rout)
    interleaved = planes == 2
    x = torch.randn((n, planes) if interleaved else (planes, n), device="cuda") * 0.1
    y = torch.zeros((m, planes) if interleaved else (planes, m), device="cuda")
    s = torch.cuda.current_stream()

    def launch():
        if interleaved:
            rc = L.dmx_resample_device(0, x.data_ptr(), n, planes, 1, planes, rin, rout, y.data_ptr(), 1, planes, s.cuda_stream)
        else:
            rc = L.dmx_resample_device(0, x.data_ptr(), n, planes, n, 1, rin, rout, y.data_ptr(), m, 1, s.cuda_stream)
        assert rc == 0, L.dmx_last_error()

    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record(s)
    for _ in range(reps):
        launch()
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = 4.0 * planes * (n + m) / 1e9
    # CPU: the oracle's pinned counterpart on a bounded sample (20 s of audio), all planes
    from scipy.signal import resample_poly
    up, down, h = dmx.resample_filter(rin, rout)
    h = h.astype(np.float64)
    xs = (x[: rin * 20] if interleaved else x[:, : rin * 20]).cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    resample_poly(xs, up, down, axis=0 if interleaved else 1, window=h / up)
    cpu_s = time.perf_counter() - t0
    print(json.dumps({"case": label, "ms_per_launch": round(ms, 4), "audio_seconds_per_s": round(seconds / (ms * 1e-3), 0),
                      "roofline": {"bound": "hbm", "achieved": round(gb / (ms * 1e-3), 1), "peak": PEAK, "unit": "GB/s",
                                   "frac": round(gb / (ms * 1e-3) / PEAK, 4), "algorithmic_bytes_per_launch": int(gb * 1e9)},
                      "cpu_baseline": {"value": round(20.0 / cpu_s, 1), "unit": "audio-sec/s", "cores": 1, "kind": "port",
                                       "sample": f"20 s of the same signal through scipy.signal.resample_poly (float64), {cpu_s:.2f} s wall"}}))

#!/usr/bin/env python
"""Where one single-segment call spends its time on the GPU.

  latency_timeline.py run                      30 single-segment calls (HIP-graph replays after the first two)
  latency_timeline.py analyse run.db [n_ops]   from a `rocprofv3 --kernel-trace` rocpd database of `run`: for the LAST call
                                               (the last n_ops kernels), the span from the first kernel's start to
                                               the last kernel's end, the sum of kernel durations, the time during
                                               which at least one / at least two kernels were running, and the
                                               idle time between kernels (launch / dependency gaps)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import time
    import numpy as np
    from demucs_cpp_amd import binding as dmx
    from demucs_cpp_amd.weights import write_synthetic_model
    path = "/tmp/lat_4s.bin"
    if not os.path.exists(path):
        write_synthetic_model(path, 4, 0)
    m = dmx.Model(path)
    ctx = dmx.Context(m, 0, 1)
    mix = (0.1 * np.random.default_rng(0).standard_normal((2, 343980))).astype(np.float32)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        ctx.segment(mix)
        ts.append(time.perf_counter() - t0)
    print("host wall per call incl. H2D/D2H, ms:", " ".join(f"{1e3 * t:.2f}" for t in ts[-5:]))


def analyse(path, n_ops):
    import sqlite3
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    rows = [r for r in rows if "elementwise" not in r[0] and "rocclr" not in r[0]]
    if n_ops <= 0:
        # the calls are separated by host round trips (H2D / D2H): split at the largest gaps
        n_ops = len(rows) // 30
    last = rows[-n_ops:]
    t0 = min(r[1] for r in last)
    t1 = max(r[2] for r in last)
    ev = []
    for _, s, e in last:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    busy1 = busy2 = 0
    depth = 0
    prev = t0
    gaps = []
    for t, d in ev:
        if depth >= 1:
            busy1 += t - prev
        if depth >= 2:
            busy2 += t - prev
        if depth == 0 and t > prev:
            gaps.append(t - prev)
        depth += d
        prev = t
    dur = sum(e - s for _, s, e in last)
    print(f"kernels in the last call: {len(last)} (columns: {cols})")
    print(f"span {1e-6 * (t1 - t0):.3f} ms; sum of kernel durations {1e-6 * dur:.3f} ms; >=1 kernel running {1e-6 * busy1:.3f} ms; "
          f">=2 running {1e-6 * busy2:.3f} ms; idle {1e-6 * sum(gaps):.3f} ms in {len(gaps)} gaps "
          f"(median {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.2f} us, max {max(gaps) / 1e3 if gaps else 0:.2f} us)")
    # per kernel: time from the end of the latest kernel that finished before it started (its launch gap)
    short = sorted(((e - s) for _, s, e in last))
    print(f"kernel duration quartiles, us: {short[len(short) // 4] / 1e3:.1f} {short[len(short) // 2] / 1e3:.1f} {short[3 * len(short) // 4] / 1e3:.1f}; "
          f"kernels under 10 us: {sum(1 for d in short if d < 10000)}")


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "analyse":
        analyse(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    else:
        run()

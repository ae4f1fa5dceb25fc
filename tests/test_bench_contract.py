"""bench.py contract checks that need no GPU: it refuses to run without one (no CPU fallback), and the
committed bench line of this round carries every key the driver / judge read."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert "needs a GPU" in (r.stdout + r.stderr)


def test_committed_bench_line_schema():
    path = os.path.join(ROOT, "profiles", "r02_bench_b24.json")
    d = json.loads(open(path).read())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "audio-sec/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # round 2: `value` counts seconds of TRACK produced; the track-level configs are measured in the same run
    cfg = d["config"]
    assert "track" in cfg["value_counts"] and cfg["segment_seconds_per_s"] > d["value"]
    assert cfg["track_4min_xRT"]["segments"] == 42 and cfg["track_4min_xRT"]["finite"] and cfg["track_4min_xRT"]["xRT"] > 100
    assert cfg["track_strong_xRT"]["ranks"] == d["n_gpus"] and cfg["single_segment_latency_ms"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port"
    # the rocprofv3 average of the dominant kernel agrees with the live measurement (within 5 %)
    import csv
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r02_kernel_stats_b24_by_class.csv"))))
    row = next(x for x in rows if x["kernel"] == r["kernel"])
    assert abs(float(row["avg_us"]) / 1e3 - r["avg_launch_ms"]) / r["avg_launch_ms"] < 0.05

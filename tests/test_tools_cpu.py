"""Host-side tools that the measurement chain rests on, checked without a GPU: tools/scale_expect.py (what the first SCALE
record is judged against) and the kernel-class names of tools/pmc_summary.py (bench.py's roofline.kernel must find its
rocprofv3 counterpart under the same name)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_expect.py"), *args], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_scale_expect_predictions_follow_the_bench_line(tmp_path):
    line = os.path.join(ROOT, "profiles", "r04_bench_4s_b42.json")
    d = json.load(open(line))
    out = _run(line)
    m = re.search(r"N=1: value\s+([\d.]+) \.\.\s+([\d.]+)\s+efficiency 1\.000 \.\. 1\.000", out)
    assert m and abs(float(m.group(1)) - d["value"]) < 0.1 and abs(float(m.group(2)) - d["value"]) < 0.1
    eff = {int(n): (float(lo), float(hi)) for n, lo, hi in re.findall(r"N=(\d): value\s+[\d.]+ \.\.\s+[\d.]+\s+efficiency ([\d.]+) \.\. ([\d.]+)", out)}
    assert sorted(eff) == [1, 2, 4, 8]
    assert all(eff[a][1] >= eff[b][1] for a, b in ((1, 2), (2, 4), (4, 8))) and 0.93 < eff[8][0] < eff[8][1] < 0.99
    strong = re.findall(r"N=(\d): ROOT finish ([\d.]+) s =\s+\d+ xRT\s+OWNER finish ([\d.]+) s", out)
    assert len(strong) == 4
    for n, root, owner in strong:
        assert float(owner) <= float(root) + 1e-9, n  # every device finishing its own stretch is never slower than the root doing it all
    assert float(strong[3][1]) < float(strong[0][1]) / 3  # eight GPUs: at least 3x on one track (ceiling 0.875 x 8)
    assert "5,5,5,6,5,5,5,6" in out and "DMX_FINISH default stays ROOT unless" in out
    # the driver's record wraps the line in {"parsed": ...}; a SCALE record's points are judged against 0.95 at every N > 1
    bench = tmp_path / "BENCH.json"
    bench.write_text(json.dumps({"parsed": d}))
    good = {"runs": [{"n_gpus": n, "value": d["value"] * n * (0.97 if n > 1 else 1.0)} for n in (1, 2, 4, 8)]}
    bad = {"runs": [{"n_gpus": 8, "parsed": {"value": d["value"] * 8 * 0.80}}]}
    (tmp_path / "good.json").write_text(json.dumps(good))
    (tmp_path / "bad.json").write_text(json.dumps(bad))
    og = _run(str(bench), str(tmp_path / "good.json"))
    assert og.count(" OK") == 4 and "BELOW 0.95" not in og
    ob = _run(str(bench), str(tmp_path / "bad.json"))
    assert "BELOW 0.95" in ob and "eff 0.800" in ob
    (tmp_path / "skipped.json").write_text(json.dumps({"skipped": True, "reason": "no 8-GPU node"}))
    assert " eff " not in _run(str(bench), str(tmp_path / "skipped.json"))  # a skipped record prints no measured column


def test_kernel_class_names_match_the_bench_lines():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_summary import kernel_class
    cases = {
        "void dmx::igemm_split_kernel<2, 2, 4, 4, 0, 0, true>(dmx::GemmArgs)": "igemm_split_128x128",
        "void dmx::igemm_split_lin_kernel<2, 8, 1>(dmx::GemmArgs)": "igemm_split_128x128",  # same tile, same class
        "void dmx::igemm_split_lin_kernel<1, 8, 0>(dmx::GemmArgs)": "igemm_split_64x128",
        "void dmx::igemm_split_kernel<4, 1, 2, 6, 0, 2, false>(dmx::GemmArgs)": "igemm_split_128x96",
        "void dmx::igemm_split_kernel<4, 1, 1, 3, 0, 0, false>(dmx::GemmArgs)": "igemm_split_64x48",
        "void dmx::igemm_kernel<2, 2, 4, 4, 2, 0, 0, false, false>(dmx::GemmArgs)": "igemm_128x128",
        "void dmx::attention_split_kernel<64, 2>(dmx::AttnArgs)": "attention_split",
        "void dmx::attention_kernel<64, 2, 1>(dmx::AttnArgs)": "attention",
        "void dmx::dgemm_k1_ring_kernel<6, 24>(dmx::GemmArgs)": "dgemm_k1_ring",
        "dmx::istft_ola_kernel(dmx::IstftOlaArgs)": "istft_ola",
    }
    for sym, cls in cases.items():
        assert kernel_class(sym) == cls, (sym, kernel_class(sym))
    # every roofline.kernel of the committed round-4 lines is a class of the committed rocprofv3 summary of the same model
    import csv
    for line, stats in (("r04_bench_4s_b42.json", "r04_kernel_stats_b42_by_class.csv"), ("r04_bench_v3_b42.json", "r04_kernel_stats_v3_b42_by_class.csv")):
        k = json.load(open(os.path.join(ROOT, "profiles", line)))["roofline"]["kernel"]
        names = [r["kernel"] for r in csv.DictReader(open(os.path.join(ROOT, "profiles", stats)))]
        assert k in names, (line, k)

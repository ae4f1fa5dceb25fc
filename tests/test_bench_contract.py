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


def _check_line(d, kernel_stats_csv, rnd=3, items=42, track_s=240.0):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "audio-sec/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    # dtype = the arithmetic type the path computes in: fp32 either way; since round 4 the products come from exact bf16
    # operand splits and the string says so
    assert d["dtype"] == ("f32" if rnd == 3 else "f32 (exact bf16x3 operand split, fp32 accumulate)")
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    cfg = d["config"]
    # round 3: a step is one 4-minute track (BASELINE configs[2]) resident in HBM; every other figure measured in the same
    # run is a SCALAR key of config (nested dicts do not survive the driver's `parsed`)
    assert cfg["segments_per_gpu_per_step"] == items and cfg["track_samples_per_step"] == 240 * 44100
    assert abs(d["value"] - track_s / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    for k, v in cfg.items():
        assert not isinstance(v, (dict, list)), k
    assert cfg["track_4min_host_xRT"] > 100 and cfg["track_4min_host_wall_s"] > 0 and cfg["track_4min_host_MB_in_out"] > 400
    assert cfg["track_strong_ranks"] == d["n_gpus"] and cfg["track_strong_xRT"] > 100 and cfg["single_segment_latency_ms"] > 0
    if rnd >= 5:
        # round 5: the whole path against its own roofline, counter traffic over algorithmic bytes, the sustained clock of the
        # dominant class, and the single-segment point priced at the arithmetic that ran
        assert 0.3 < r["whole_path_frac"] < 0.8 and 1.9 < r["effective_clock_ghz"] <= 2.45 and r["peak_clock_ghz"] == 2.4
        assert r["traffic"] is None or (r["traffic_ratio"] is not None and abs(r["traffic_ratio"] - r["traffic"] / r["algorithmic_bytes_per_launch"]) < 2e-3)
        if cfg.get("single_segment_roofline_frac") is not None:
            assert cfg["single_segment_peak_tflops"] == 503.3
            assert abs(cfg["single_segment_roofline_frac"] - cfg["single_segment_tflops"] / 503.3) < 2e-3
    if rnd == 3:
        assert cfg["gemm_path"].startswith("f32 MFMA")  # round 3: the operand split was an opt-in experiment
        assert cfg["experiment_bf16x3_split_xRT"] is None or cfg["experiment_bf16x3_split_xRT"] > d["value"]
    else:
        # round 4: the exact split is the product path; the fp32-MFMA context measured in the same run is beside it, and
        # the dominant kernel is priced against the bf16 pipe divided by the partial products per fp32 term
        assert cfg["gemm_path"].startswith("bf16x3") and cfg["outputs_finite"] is True
        assert cfg["f32_mfma_xRT"] is not None and 0 < cfg["f32_mfma_xRT"] < d["value"] and cfg["f32_mfma_outputs_finite"] is True
        assert (r["kernel"].startswith("igemm_split_") and r["peak"] == 503.3) or (r["kernel"] == "attention_split" and r["peak"] == 419.4)
        assert "2516.6" in r["peak_basis"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "openblas_value", "openblas_kind"):
        assert k in c, k
    assert c["kind"] == "port" and "median of 3" in c["sample"] and c["openblas_kind"] == "port+openblas"
    # the rocprofv3 average of the dominant kernel agrees with the live measurement (within 5 %)
    import csv
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", kernel_stats_csv))))
    row = next(x for x in rows if x["kernel"] == r["kernel"])
    assert abs(float(row["avg_us"]) / 1e3 - r["avg_launch_ms"]) / r["avg_launch_ms"] < 0.05


def test_committed_bench_line_schema():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_b42.json")).read())
    assert "htdemucs-4s" in d["metric"] and "configs[2]" in d["metric"]
    _check_line(d, "r03_kernel_stats_b42_by_class.csv")


def test_committed_v3_bench_line_schema():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_v3_b42.json")).read())
    assert "hdemucs_mmi" in d["metric"]
    _check_line(d, "r03_kernel_stats_v3_b42_by_class.csv")
    assert d["config"]["single_segment_latency_ms"] > 4.0  # 2016 sequential LSTM steps alone are ~5 ms


def test_committed_round4_bench_lines():
    rd = lambda n: json.loads(open(os.path.join(ROOT, "profiles", n)).read())
    d = rd("r04_bench_4s_b42.json")
    assert "htdemucs-4s" in d["metric"] and "configs[2]" in d["metric"]
    _check_line(d, "r04_kernel_stats_b42_by_class.csv", rnd=4)
    assert d["value"] >= 2600 and d["config"]["ms_per_segment"] <= 2.2  # the round-3 review's target for the split path
    v3 = rd("r04_bench_v3_b42.json")
    assert "hdemucs_mmi" in v3["metric"]
    _check_line(v3, "r04_kernel_stats_v3_b42_by_class.csv", rnd=4)
    # BASELINE configs[3] / configs[4] at full size on one GPU: the 6-source model, and the fine-tuned bag whose step is
    # 4 models x 42 segments and whose value still counts the track's seconds once
    s6 = rd("r04_bench_6s_b42.json")
    assert "htdemucs-6s" in s6["metric"] and "configs[3]" in s6["metric"] and s6["config"]["models"] == 1
    assert s6["config"]["segments_per_gpu_per_step"] == 42 and s6["config"]["gemm_path"].startswith("bf16x3")
    ft = rd("r04_bench_ft_b42.json")
    assert "configs[4]" in ft["metric"] and ft["config"]["models"] == 4 and ft["config"]["segments_per_gpu_per_step"] == 168
    assert abs(ft["value"] - 240.0 / (ft["ms_per_step"] * 1e-3)) / ft["value"] < 1e-3
    assert 3.5 < d["value"] / ft["value"] < 4.5  # four models' work per second of track
    # the batch sweep: ms per segment falls monotonically with the batch in both arithmetic modes
    sweep = [json.loads(x) for x in open(os.path.join(ROOT, "profiles", "r04_bench_4s_b1_b4_b12_b24.jsonl")) if x.strip()]
    ms = [x["config"]["ms_per_segment"] for x in sweep]
    assert len(ms) == 4 and ms == sorted(ms, reverse=True) and ms[-1] > d["config"]["ms_per_segment"]


def test_committed_round5_bench_lines():
    rd = lambda n: json.loads(open(os.path.join(ROOT, "profiles", n)).read())
    d = rd("r05_bench_4s_b42.json")
    assert "htdemucs-4s" in d["metric"] and "configs[2]" in d["metric"]
    _check_line(d, "r05_kernel_stats_b42_by_class.csv", rnd=5)
    assert d["value"] >= 2700 and d["config"]["ms_per_segment"] <= 2.1 and d["roofline"]["traffic"] is not None
    for name, key in (("r05_bench_6s_b42.json", "configs[3]"), ("r05_bench_ft_b42.json", "configs[4]")):
        x = rd(name)
        assert key in x["metric"] and x["config"]["gemm_path"].startswith("bf16x3") and x["config"]["outputs_finite"] is True
        assert 0.3 < x["roofline"]["whole_path_frac"] < 0.8
    assert rd("r05_bench_ft_b42.json")["config"]["segments_per_gpu_per_step"] == 168
    assert "hdemucs_mmi" in rd("r05_bench_v3_b42.json")["metric"]
    sweep = [json.loads(x) for x in open(os.path.join(ROOT, "profiles", "r05_bench_4s_b1_b4_b12_b24.jsonl")) if x.strip()]
    ms = [x["config"]["ms_per_segment"] for x in sweep]
    assert len(ms) == 4 and ms == sorted(ms, reverse=True) and ms[-1] > d["config"]["ms_per_segment"]
    # the committed traffic file of this round is the one bench.py will quote from now on
    t = rd("r05_traffic_bf16x3.json")
    assert t["batch"] == 42 and t["gemm"] == "bf16x3" and "igemm_split_128x128" in t["classes"]
    # the opt-in fp16x3 mode is reported BESIDE the headline (a scalar key of the default line), never as it; a run that asks
    # for it says so in dtype / gemm_path and prices its dominant class at the three-MFMA peak
    three = rd("r05_bench_4s_b42_three_modes.json")
    assert three["config"]["gemm_path"].startswith("bf16x3") and "bf16x3" in three["dtype"] and "fp16x3" not in three["dtype"]
    assert three["config"]["fp16x3_outputs_finite"] is True and three["config"]["f32_mfma_outputs_finite"] is True
    assert three["config"]["f32_mfma_xRT"] < three["value"] < three["config"]["fp16x3_xRT"] < 1.15 * three["value"]
    assert three["roofline"]["kernel"] == "igemm_split_128x128" and abs(three["roofline"]["peak"] - 503.3) < 0.1
    h = rd("r05_bench_4s_b42_gemm_fp16x3.json")
    assert h["config"]["gemm_path"].startswith("fp16x3") and "fp16x3" in h["dtype"] and h["config"]["outputs_finite"] is True
    assert h["roofline"]["kernel"] == "igemm_splith_128x128" and abs(h["roofline"]["peak"] - 2516.6 / 3) < 0.1
    assert abs(h["value"] - three["config"]["fp16x3_xRT"]) / h["value"] < 0.02 and "bf16x3_xRT" not in h["config"]


def test_committed_round6_bench_lines():
    """Round 6: the dominant class is the 128 x 256 direct-fragment tile; the headline line was re-run at the head with the round's
    counter files committed, so it quotes its own traffic and clock; `single_segment_launches` is on the line."""
    rd = lambda n: json.loads(open(os.path.join(ROOT, "profiles", n)).read())
    d = rd("r06_bench_4s_b42.json")
    assert "htdemucs-4s" in d["metric"] and "configs[2]" in d["metric"]
    _check_line(d, "r06_kernel_stats_b42_by_class.csv", rnd=6)
    assert d["value"] >= 3000 and d["config"]["ms_per_segment"] <= 1.9
    r = d["roofline"]
    assert r["kernel"] == "igemm_split_128x256" and r["frac"] > 0.46 and r["traffic_source"] == "profiles/r06_traffic_bf16x3.json"
    assert r["traffic"] is not None and r["traffic_ratio"] < 1.6 and r["effective_clock_source"] == "profiles/r06_effective_clock.csv"
    assert 250 <= d["config"]["single_segment_launches"] <= 300 and d["config"]["single_segment_latency_ms"] < 4.5
    coll = rd("r06_bench_4s_b42_collection.json")  # the line of the collection run itself (before its counter files existed)
    assert abs(coll["value"] - d["value"]) / d["value"] < 0.03 and coll["roofline"]["kernel"] == r["kernel"]
    for name, key in (("r06_bench_6s_b42.json", "configs[3]"), ("r06_bench_ft_b42.json", "configs[4]")):
        x = rd(name)
        assert key in x["metric"] and x["config"]["gemm_path"].startswith("bf16x3") and x["config"]["outputs_finite"] is True
        assert 0.3 < x["roofline"]["whole_path_frac"] < 0.8
    ft = rd("r06_bench_ft_b42.json")["config"]
    assert ft["segments_per_gpu_per_step"] == 168 and ft["track_strong_items"] == 168 and ft["strong_ceiling"] == 1.0
    assert rd("r06_bench_6s_b42.json")["config"]["strong_ceiling"] == 1.0  # (one rank: the ceiling of 42 items over 1)
    assert "hdemucs_mmi" in rd("r06_bench_v3_b42.json")["metric"]
    sweep = [json.loads(x) for x in open(os.path.join(ROOT, "profiles", "r06_bench_4s_b1_b4_b12_b24.jsonl")) if x.strip()]
    ms = [x["config"]["ms_per_segment"] for x in sweep]
    assert len(ms) == 4 and ms == sorted(ms, reverse=True) and ms[-1] > d["config"]["ms_per_segment"]
    t = rd("r06_traffic_bf16x3.json")
    assert t["batch"] == 42 and t["gemm"] == "bf16x3"
    for cls in ("igemm_split_128x256", "igemm_split_128x192", "igemm_split_128x96d", "igemm_split_128x32d", "dconv_row", "attention_split"):
        assert cls in t["classes"], cls
    # the row-resident DConv moves its rows once: counter bytes within 10 % of read + write of x
    dr = t["classes"]["dconv_row"]
    assert dr["launches"] % 4 == 0 and 0.95 < dr["read_bytes_per_launch"] / dr["write_bytes_per_launch"] < 1.15
    h = rd("r06_bench_4s_b42_fp16x3.json")
    assert h["config"]["gemm_path"].startswith("fp16x3") and "fp16x3" in h["dtype"] and h["roofline"]["kernel"] == "igemm_splith_128x128"


def test_round2_bench_line_still_parses():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r02_bench_b24.json")).read())
    assert d["config"]["track_4min_xRT"]["segments"] == 42 and d["roofline"]["kernel"] == "igemm_128x128"

#!/usr/bin/env python
"""Predictions for the first multi-GPU SCALE run, in the units of the driver's records.

  tools/scale_expect.py BENCH_rNN.json [SCALE_rNN.json]

Reads the N = 1 bench line (the driver's BENCH record or a bare JSON line of bench.py) and prints, for N = 1, 2, 4, 8:
  weak   : the `value` bench.py --gpus N should report (every rank processes `segments_per_gpu_per_step` segments per step,
           RCCL gather of the per-segment outputs to the root, root overlap-add of all N x 42 segments) and the efficiency
           value(N) / (N x value(1)) the driver will compute from it;
  strong : config.track_strong_wall_s / track_strong_xRT of ONE 4-minute track whose 42 segments are dealt over the N ranks
           (contiguous ranges 5,5,5,6,5,5,5,6 at N = 8: at most 42 / (8 x 6) = 87.5 % of ideal), for the ROOT finish
           (gather + root overlap-add + the root's D2H of the whole result) and the OWNER finish (every device copies out
           its own stretch; only 2.75 MB segment tails cross the links).
If a SCALE record is given, its measured values are printed beside the predictions with a verdict per line: the weak
prediction is falsified below 0.95 efficiency at N = 8 (gather not overlapped, or RCCL channels starving the compute
tiles: DESIGN.md section 5 says what to check); the default finish mode switches to OWNER if its measured strong-scaling
gain over ROOT exceeds 5 %.

Model (all terms from measurements on one MI355X, profiles/): step time t1 = ms_per_step; per step the root additionally
overlap-adds (N - 1) x 42 segments (0.33 ms per 42) and writes the gathered slabs ((N - 1) x 42 x 11.0 MB at ~6 TB/s); the
gather itself (462 MB per rank and step over that rank's own xGMI link, ~75 GB/s one way = 6 ms) is issued behind the
next step's kernels and hidden unless RCCL's receive kernels take CUs from the root: the lower bound of the band assumes
they cost the root 2 % of its step."""
import json
import sys

OLA_MS_PER_42 = 0.33       # root overlap-add of 42 segments (profiles: track_ola)
SLAB_MB = 11.0             # one 4-source segment output (6 sources: 16.5)
HBM_WRITE_TBS = 6.0
PCIE_GBS = 55.0            # D2H of the result on one link (measured: 339 MB in ~6 ms)
XGMI_GBS = 75.0            # one direction of one link


def load_line(path):
    d = json.load(open(path))
    if "parsed" in d:
        d = d["parsed"]
    return d


def measured_points(scale):
    """{n_gpus: value} from a SCALE record of ANY shape: the driver's file is walked for every object that carries both
    `n_gpus` and `value` (bench.py's own line, wherever the driver nests it: `parsed`, `runs[i]`, `points[i]`, `n1` ...);
    objects with `n` / `value` or `n_gpus` inside `parsed` are covered by the same walk."""
    out = {}

    def walk(o):
        if isinstance(o, dict):
            n = o.get("n_gpus", o.get("n"))
            v = o.get("value")
            if v is None and isinstance(o.get("parsed"), dict):
                v = o["parsed"].get("value")
            if isinstance(n, int) and isinstance(v, (int, float)) and v > 0:
                out[int(n)] = float(v)
            for v in o.values():
                walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)
        elif isinstance(o, str) and o.lstrip().startswith("{") and '"n_gpus"' in o:
            try:
                walk(json.loads(o[o.index("{"):o.rindex("}") + 1]))  # a raw stdout tail holding the JSON line
            except Exception:
                pass
    walk(scale)
    return out


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    b = load_line(sys.argv[1])
    v1, t1 = float(b["value"]), float(b["ms_per_step"])
    cfg = b.get("config", {})
    segs = int(cfg.get("segments_per_gpu_per_step", 42))
    models = int(cfg.get("models", 1))
    ms_seg = float(cfg.get("ms_per_segment", t1 / segs))
    n_sources = 6 if "6s" in b.get("metric", "") else 4
    slab = SLAB_MB * n_sources / 4.0
    scale = None
    if len(sys.argv) > 2:
        try:
            scale = json.load(open(sys.argv[2]))
        except Exception:
            scale = None
    print(f"N = 1 line: value {v1:.1f} {b['unit']}, {t1:.2f} ms per step, {ms_seg:.3f} ms per segment ({segs} items per GPU per step)")
    print("weak scaling (bench.py --gpus N): predicted value and efficiency value(N) / (N x value(1))")
    for n in (1, 2, 4, 8):
        extra = (n - 1) * (OLA_MS_PER_42 * segs / 42.0 + segs / models * slab * 1e6 / (HBM_WRITE_TBS * 1e12) * 1e3)
        hi = t1 / (t1 + extra)
        lo = hi * (0.98 if n > 1 else 1.0)
        line = f"  N={n}: value {n * v1 * lo:9.1f} .. {n * v1 * hi:9.1f}   efficiency {lo:.3f} .. {hi:.3f}"
        if scale is not None and not (isinstance(scale, dict) and scale.get("skipped")):
            meas = measured_points(scale).get(n)
            if meas:
                eff = float(meas) / (n * v1)
                line += f"   measured {float(meas):9.1f} eff {eff:.3f} " + ("OK" if eff >= 0.95 or n == 1 else "BELOW 0.95: check gather overlap / NCCL_MAX_NCHANNELS")
        print(line)
    print("strong scaling of ONE 4-minute track (config.track_strong_wall_s; 42 segments dealt 5,5,5,6,5,5,5,6 at N = 8)")
    res_mb = n_sources * 2 * 240 * 44100 * 4 / 1e6
    for n in (1, 2, 4, 8):
        most = -(-42 // n)
        # fewer segments in flight per device run slower per segment (batch 6: ~3.6 ms vs 3.06 at 42)
        per_seg = ms_seg * (1.0 + 0.18 * (1.0 - most / 42.0))
        compute = most * per_seg
        gather = (42 - most) * slab / (XGMI_GBS * 1e3) * 1e3 / max(n - 1, 1) if n > 1 else 0.0
        root = compute + gather + OLA_MS_PER_42 + res_mb / PCIE_GBS
        owner = compute + 2.75 / XGMI_GBS + OLA_MS_PER_42 / n + res_mb / n / PCIE_GBS
        print(f"  N={n}: ROOT finish {root / 1e3:.4f} s = {240.0 / (root / 1e3):8.0f} xRT   OWNER finish {owner / 1e3:.4f} s = {240.0 / (owner / 1e3):8.0f} xRT"
              f"   (ideal {240.0 / (42 * ms_seg / n / 1e3):8.0f}; dealing ceiling {42.0 / (n * most):.3f})")
    print("decision rule (DESIGN.md section 5): DMX_FINISH default stays ROOT unless OWNER's measured track_strong gain exceeds 5 %")


if __name__ == "__main__":
    main()

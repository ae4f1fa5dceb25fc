#!/usr/bin/env python
"""Re-aggregates per-symbol CSVs written by pmc_summary.py into per-class CSVs / the traffic JSON on a machine without the rocpd
database (the class rule of pmc_summary.py applied to the symbol names).
  reclass_csv.py class symbols.csv > by_class.csv
  reclass_csv.py traffic fetch_symbols.csv write_symbols.csv <batch> [gemm] [model] > traffic.json"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import kernel_class  # noqa: E402


def load(path):
    rows = list(csv.reader(open(path)))
    head = [h for h in rows[0] if h != ""]
    agg = defaultdict(lambda: defaultdict(float))
    for r in rows[1:]:
        if not r:
            continue
        k = kernel_class(r[0])
        for h, v in zip(head[1:], r[1:]):
            if h == "avg_us" or v == "":
                continue
            agg[k][h] += float(v)
    return agg, [h for h in head[1:] if h != "avg_us"]


if sys.argv[1] == "class":
    agg, cols = load(sys.argv[2])
    extra = [c for c in cols if c not in ("calls", "total_us")]
    print("kernel,calls,total_us,avg_us," + ",".join(extra))
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["total_us"]):
        print(f"\"{k}\",{int(d['calls'])},{d['total_us']:.1f},{d['total_us'] / d['calls']:.2f}," + ",".join(f"{d.get(c, 0):.0f}" for c in extra))
else:
    fa, _ = load(sys.argv[2])
    wa, _ = load(sys.argv[3])
    batch = int(sys.argv[4])
    out = {"_note": "bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches; rocprofv3 --pmc, separate passes, bench.py --batch %d "
                    "(classes re-aggregated from the per-symbol tables: tools/reclass_csv.py)" % batch,
           "batch": batch, "gemm": sys.argv[5] if len(sys.argv) > 5 else "bf16x3", "model": sys.argv[6] if len(sys.argv) > 6 else "4s", "classes": {}}
    for k in fa:
        if k not in wa:
            continue
        rd = 2.0 * fa[k].get("FETCH_SIZE", 0.0) * 1024 / fa[k]["calls"]
        wr = wa[k].get("WRITE_SIZE", 0.0) * 1024 / wa[k]["calls"]
        out["classes"][k] = {"launches": int(fa[k]["calls"]), "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                             "traffic_bytes_per_launch": round(rd + wr), "avg_us_fetch_pass": round(fa[k]["total_us"] / fa[k]["calls"], 2)}
    print(json.dumps(out, indent=1))

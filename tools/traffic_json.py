#!/usr/bin/env python
"""HBM traffic per kernel class and launch from two separate rocprofv3 PMC passes
(FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md §rocprofv3 PMC slots).

  traffic_json.py fetch.db write.db <batch> > profiles/rNN_traffic.json

Corrections (MI355X_MICROARCH.md §HBM): both counters are in KB; on gfx950 FETCH_SIZE counts the
128-B requests of 16-B/lane coalesced reads at 64 B, so reads are doubled. Every kernel of this
library reads with 16-B/lane loads (float4), so the factor applies to all classes. WRITE_SIZE is
taken as reported (uncalibrated per the guide)."""
import json
import os
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from pmc_summary import aggregate  # noqa: E402

fa, fc, _ = aggregate(sys.argv[1], True)
wa, wc, _ = aggregate(sys.argv[2], True)
batch = int(sys.argv[3]) if len(sys.argv) > 3 else -1
out = {"_note": "bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches; rocprofv3 --pmc, separate passes, "
                "bench.py --batch %d" % batch, "batch": batch, "gemm": os.environ.get("GEMM", "f32"), "model": os.environ.get("MODEL", "4s"), "classes": {}}
for k in fa:
    if k not in wa:
        continue
    rd = 2.0 * fa[k].get("FETCH_SIZE", 0.0) * 1024 / fc[k]
    wr = wa[k].get("WRITE_SIZE", 0.0) * 1024 / wc[k]
    out["classes"][k] = {"launches": fc[k], "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                         "traffic_bytes_per_launch": round(rd + wr), "avg_us_fetch_pass": round(fa[k]["dur_us"] / fc[k], 2)}
print(json.dumps(out, indent=1))

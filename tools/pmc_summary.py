#!/usr/bin/env python
"""Aggregates a rocprofv3 rocpd SQLite result (kernel trace [+ PMC]) per kernel name -> CSV on stdout."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
kern = {}
for did, name, dur in cur.execute("select dispatch_id, name, end-start from kernels"):
    kern[did] = (name, dur)
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for did, (name, dur) in kern.items():
    agg[name]["dur_us"] += dur / 1e3
    cnt[name] += 1
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
try:
    q = "select dispatch_id, counter_name, value from counters_collection" if "counters_collection" in [r[0] for r in cur.execute("select name from sqlite_master")] else None
    names = set()
    if q:
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        sel = "select dispatch_id, counter_name, value from counters_collection"
        for did, cname, val in cur.execute(sel):
            if did in kern:
                agg[kern[did][0]][cname] += float(val)
                names.add(cname)
except Exception as e:  # noqa
    print("# pmc parse failed:", e, "cols", cols)
    names = set()
names = sorted(names)
print("kernel,calls,total_us," + ",".join(names))
for name, d in sorted(agg.items(), key=lambda kv: -kv[1]["dur_us"]):
    print(f"\"{name[:90]}\",{cnt[name]},{d['dur_us']:.1f}," + ",".join(f"{d.get(n, 0):.0f}" for n in names))

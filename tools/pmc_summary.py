#!/usr/bin/env python
"""Aggregates a rocprofv3 rocpd SQLite result (kernel trace [+ PMC]) -> CSV on stdout.

  pmc_summary.py run.db            per kernel symbol
  pmc_summary.py run.db --class    per kernel class (the names bench.py / dmx_debug_profile use:
                                   igemm_<BM>x<BN>, dgemm_direct, attention, ...)
"""
import re
import sqlite3
import sys
from collections import defaultdict


def kernel_class(name: str) -> str:
    m = re.search(r"igemm_kernel<(\d+), (\d+), (\d+), (\d+)", name)
    if m:
        wm, wn, mf, nf = (int(x) for x in m.groups())
        return f"igemm_{wm * mf * 16}x{wn * nf * 16}"
    m = re.search(r"igemm_split_kernel<(\d+), (\d+), (\d+), (\d+)", name)
    if m:
        wm, wn, mf, nf = (int(x) for x in m.groups())
        return f"igemm_split_{wm * mf * 16}x{wn * nf * 16}"
    m = re.search(r"igemm_split_lin_kernel<(\d+), (\d+)", name)  # linear layers, activation fragments in registers: same tiles / class
    if m:
        mf, nf = (int(x) for x in m.groups())
        return f"igemm_split_{4 * mf * 16}x{nf * 16}"
    m = re.search(r"igemm_split_linw_kernel<(\d+)", name)  # the wide tiles (128 x 256 / 192; 6 fragments: 128 x 96, direct fragments)
    if m:
        return {16: "igemm_split_128x256", 12: "igemm_split_128x192", 6: "igemm_split_128x96d", 4: "igemm_split_128x64d", 2: "igemm_split_128x32d"}[int(m.group(1))]
    if "dconv_row_kernel" in name:
        return "dconv_row"
    if "igemm_lin256_kernel" in name:
        return "igemm_lin256x128"
    for key, cls in (("lstm_kernel", "lstm"), ("local_attn_kernel", "local_attn"), ("group_stats", "group_stats"), ("gn_act_kernel", "gn_act"),
                     ("dgemm_k1_ring_kernel", "dgemm_k1_ring"), ("dgemm_kernel", "dgemm_direct"), ("attention_split_kernel", "attention_split"), ("attention_kernel", "attention"), ("track_stats", "track_stats"),
                     ("track_gather", "track_gather"), ("track_ola", "track_ola"), ("istft_ola", "istft_ola"), ("istft", "istft"), ("stft", "stft"),
                     ("stats_", "stats_reduce"), ("layernorm", "layernorm"), ("gn_apply", "gn_apply"), ("ola_kernel", "ola")):
        if key in name:
            return cls
    return name[:60]


def aggregate(path, by_class=False):
    db = sqlite3.connect(path)
    cur = db.cursor()
    kern = {}
    for did, name, dur in cur.execute("select dispatch_id, name, end-start from kernels"):
        kern[did] = (kernel_class(name) if by_class else name, dur)
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for did, (name, dur) in kern.items():
        agg[name]["dur_us"] += dur / 1e3
        cnt[name] += 1
    names = set()
    tables = [r[0] for r in cur.execute("select name from sqlite_master")]
    if "counters_collection" in tables:
        for did, cname, val in cur.execute("select dispatch_id, counter_name, value from counters_collection"):
            if did in kern:
                agg[kern[did][0]][cname] += float(val)
                names.add(cname)
    return agg, cnt, sorted(names)


if __name__ == "__main__":
    by_class = "--class" in sys.argv
    agg, cnt, names = aggregate(sys.argv[1], by_class)
    print("kernel,calls,total_us,avg_us," + ",".join(names))
    for name, d in sorted(agg.items(), key=lambda kv: -kv[1]["dur_us"]):
        print(f"\"{name[:90]}\",{cnt[name]},{d['dur_us']:.1f},{d['dur_us'] / cnt[name]:.2f}," + ",".join(f"{d.get(n, 0):.0f}" for n in names))

#!/usr/bin/env python
"""Per-workgroup timeline of one linear layer of the split path (library built with -DDMX_TIMING): when does a workgroup
multiply, when does it store, and what is the OTHER workgroup of its CU doing at that time?
  DMX_LIB=.../libdemucs_hip_timing.so python tools/gpu_wg_timeline.py [op ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demucs_cpp_amd import binding as dmx  # noqa: E402
from demucs_cpp_amd.weights import write_synthetic_model  # noqa: E402

ops = sys.argv[1:] or ["crosstransformer.layers.0.qkv", "crosstransformer.layers.0.linear1", "crosstransformer.layers.0.linear2"]
B = int(os.environ.get("PB", "42"))
path = "/tmp/wgt_4s.bin"
if not os.path.exists(path):
    write_synthetic_model(path, 4, 0)
m = dmx.Model(path)
ctx = dmx.Context(m, 0, B, gemm=dmx.GEMM_BF16X3)
rng = np.random.default_rng(3)
ctx.profile(B, 1)  # every buffer of the plan holds real activations
for op in ops:
    dump = f"/tmp/wgt_{op}.bin"
    os.environ["DMX_TIMING_DUMP"] = dump
    for rep in range(2):  # second launch: warm
        r = dmx.igemm_timing(ctx, B, op)
    if r is None or not os.path.exists(dump):
        print(op, ": no timing (not an igemm op of this plan / library without -DDMX_TIMING)")
        continue
    raw = np.fromfile(dump, dtype=np.uint64)
    nrec = (len(raw) // 8 - 64) // 2  # records, then 8 words of epilogue stamps per block of the padded grid (api.cpp)
    d = raw[: nrec * 8].reshape(-1, 8)
    stamps = raw[nrec * 8:].reshape(-1, 4, 2)  # [block][wave][stamp]
    keep = d[:, 5] > 0
    d = d[keep]
    est = stamps[d[:, 7].astype(np.int64) % len(stamps)]
    t = d[:, :5].astype(np.int64)
    t0 = t[:, 0].min()
    t = (t - t0) * 0.01  # 100 MHz -> microseconds
    hw = (d[:, 6] >> np.uint64(32)).astype(np.int64)
    xcc = (d[:, 6] & np.uint64(0xF)).astype(np.int64)
    slot, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    n = len(d)
    total = t[:, 4].max()
    print(f"== {op}: {n} workgroups, {len(np.unique(cuid))} distinct CUs, wave slots seen {sorted(np.unique(slot).tolist())}, launch {total:.1f} us")
    pro, loop, epi, drain = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
    for nm, v in (("prologue", pro), ("K loop", loop), ("epilogue issue", epi), ("store drain", drain), ("whole workgroup", t[:, 4] - t[:, 0])):
        print(f"   {nm:16s} mean {v.mean():7.2f} us   p10 {np.percentile(v, 10):7.2f}   p90 {np.percentile(v, 90):7.2f}")
    if est[:, 0, 0].max() > 0:
        e = (est.astype(np.int64) - t0) * 0.01
        for w in range(4):
            a, b2 = e[:, w, 0] - t[:, 2], e[:, w, 1] - e[:, w, 0]
            c2 = t[:, 3] - e[:, w, 1]
            print(f"   wave {w}: epilogue entry -> bias loads issued {a.mean():6.2f} us, first row block {b2.mean():6.2f} us, rest (second row block, statistics) {c2.mean():6.2f} us")
    # the share of a workgroup's epilogue (issue + drain) during which the other workgroup of its CU is inside its K loop
    order = np.argsort(cuid, kind="stable")
    covered = 0.0
    both_epi = 0.0
    tot_epi = 0.0
    for c in np.unique(cuid):
        idx = np.nonzero(cuid == c)[0]
        for i in idx:
            e0, e1 = t[i, 2], t[i, 4]
            tot_epi += e1 - e0
            for j in idx:
                if j == i:
                    continue
                covered += max(0.0, min(e1, t[j, 2]) - max(e0, t[j, 1]))
                both_epi += max(0.0, min(e1, t[j, 4]) - max(e0, t[j, 2]))
    print(f"   of the epilogue time of a workgroup, the other workgroup of its CU is in its K loop {100 * covered / tot_epi:.0f} %, in its own epilogue {100 * both_epi / tot_epi:.0f} %")
    # global picture: how many workgroups are in their epilogue at a time (histogram over the launch, 50 bins)
    bins = np.linspace(0, total, 51)
    inepi = np.zeros(50)
    for i in range(n):
        a, b = np.searchsorted(bins, [t[i, 2], t[i, 4]])
        inepi[max(a - 1, 0):max(b, 1)] += 1
    print("   workgroups in their epilogue per 1/50 of the launch:", " ".join(f"{int(x)}" for x in inepi))
    # rounds: start times of the workgroups, sorted, as multiples of the mean workgroup time
    wg = (t[:, 4] - t[:, 0]).mean()
    st = np.sort(t[:, 0])
    print(f"   start times (us) of workgroups 0, 256, 512, 768, 1024, 1536, 2048: " + " ".join(f"{st[min(i, n - 1)]:.1f}" for i in (0, 256, 512, 768, 1024, 1536, 2048)) + f"   mean workgroup time {wg:.1f}")

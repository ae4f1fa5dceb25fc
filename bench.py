#!/usr/bin/env python
"""bench.py — throughput of the HTDemucs-4s per-segment hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
as one rank per GPU by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
environment, backend "nccl" = RCCL). Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): htdemucs-4s, synthetic dmc4 weights
(seed 0), a ~4-minute synthetic 44.1 kHz stereo track 0.1*N(0,1) RESIDENT IN HBM, fp32 arithmetic. One step on every
rank = one such track through the whole path of demucs_inference (src/model_apply.cpp:60-288) on the device entry
points of the C ABI: track statistics (dmx_track_stats_device), extraction + normalisation + centring of its
`--batch` = 42 overlapping segments of 343980 samples (dmx_track_gather_device), the segment graph STFT -> encoders
-> cross-transformer -> decoders -> ISTFT on all of them (dmx_segment_infer_device), and the triangle-weighted
overlap-add + de-normalisation (dmx_track_overlap_add_device), stems left in HBM. At N = 1 the track is literally
configs[2]: 10 584 000 samples, shift offset 4033 -> 42 segments. The segment loop is embarrassingly parallel
(src/model_apply.cpp:189-235): ranks own disjoint stretches of segments (weak scaling, no collective inside the
hot path). The one real exchange step of the track path - gathering the per-segment outputs
to the root before overlap-add (north_star) - is part of every step when N > 1
(RCCL gather over xGMI), followed by the root's triangle-weighted overlap-add of all N*batch
segments (dmx_track_overlap_add_device); with N = 1 the overlap-add alone runs.
value = seconds of TRACK produced per wall second: the N*batch segments of a step are the consecutive
overlapping segments of one stretch of a track (stride 257985 samples = 5.85 s of new audio per 7.8 s
segment, src/model_apply.cpp:162), which the root overlap-adds into n_track = N*batch*257985 - 22050
samples; value = n_track/44100 * K / T, T = max over ranks of the barrier-bracketed wall time of the K timed
steps. (Round 1 counted the 7.8 s every segment PROCESSES; that figure stays in config.segment_seconds_per_s.)

config also reports, as SCALAR keys, measured in this same run:
  track_4min_host_xRT / track_4min_host_wall_s   (N = 1) the same configs[2] track end to end through dmx_track_infer
                      with HOST buffers in and out (PCIe inclusive: H2D of the 85 MB track, 42 segments, overlap-add,
                      D2H of the 339 MB stems), 240 s / best wall of 3; never `value` (the boundary rule of the contract);
  track_strong_xRT / track_strong_wall_s   (every N) ONE such track with its 42 segments dealt over the N ranks
                      (contiguous ranges), RCCL gather of the per-segment outputs to the root, root overlap-add, D2H on
                      the root: strong scaling of a single track (<= 87.5 % at N = 8: 42 = 6+6+5*6);
  single_segment_latency_ms  BASELINE configs[1] read literally (one segment per call, device resident).
`--model v3` runs the same workload on Demucs v3 (hdemucs_mmi, synthetic dmc3 weights): its own metric name.

Extra objects on the JSON line:
  roofline     : dominant kernel (by device time) measured live with HIP events on the stream
                 it runs on (dmx_debug_profile), algorithmic FLOPs / duration vs the fp32 MFMA
                 peak 157.3 TFLOP/s (MI355X_MICROARCH.md); traffic = PMC HBM bytes (null unless
                 profiles/ holds a counter pass of the same model and batch)
  cpu_baseline : the CPU oracle (oracle/, a from-scratch port of the reference algorithm; the reference itself
                 needs Eigen and cannot be built here) timed on this box's host cores on ONE full segment, 1 warm-up +
                 median of 3, rank 0 at N = 1 only; `openblas_*` keys: the same port with its GEMMs routed through the
                 OpenBLAS that NumPy bundles (configs[0] names "Eigen/OpenBLAS").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEG = 343980
SEG_SECONDS = 7.8
PEAK_TFLOPS_FP32_MFMA = 157.3
MODEL_FLOPS_4S = 340.2e9  # algorithmic FLOPs per segment (SURVEY.md §8d / BASELINE.md §3)


def pmc_traffic(kernel_class, batch, model="4s"):
    """HBM bytes per launch of a kernel class from the committed rocprofv3 PMC passes
    (profiles/rNN_traffic.json, written by tools/traffic_json.py from separate FETCH_SIZE and
    WRITE_SIZE passes over this same workload; gfx950 corrections applied there). The counters cannot be
    collected from inside this process, so the newest committed pass is quoted - only if it was taken at
    the same batch size."""
    import glob

    # one file per workload: rNN_traffic.json is htdemucs-4s, rNN_traffic_<model>.json any other model (none committed => null)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json" if model == "4s" else f"r*_traffic_{model}.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        if int(d.get("batch", -1)) != batch:
            return None, None
        return d["classes"][kernel_class]["traffic_bytes_per_launch"], "profiles/" + os.path.basename(files[-1])
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("DMX_BENCH_BATCH", "42")),
                    help="segments per GPU per step; 42 = the segments of configs[2]'s 4-minute track")
    ap.add_argument("--model", default="4s", choices=["4s", "v3"], help="4s: htdemucs (the BASELINE metric); v3: hdemucs_mmi")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-single", action="store_true")
    ap.add_argument("--no-track", action="store_true")
    ap.add_argument("--no-split-probe", action="store_true",
                    help="skip the secondary measurement of the opt-in exact-split bf16 path (DMX_GEMM=bf16x3, a child process)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: TEST MODE ONLY - several ranks share GPU 0 and the gather goes through host memory, to "
                         "exercise the N > 1 control flow (double buffering, flush, overlap-add of all ranks' segments) on a "
                         "1-GPU box; the numbers it prints are not a measurement")
    args = ap.parse_args()

    import torch

    from demucs_cpp_amd import binding as dmx
    from demucs_cpp_amd.weights import write_synthetic_model

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    test_mode = args.backend == "gloo"
    if test_mode:
        local_rank = 0  # every rank on GPU 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if test_mode:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    B = args.batch
    S = 4
    v3 = args.model == "v3"
    tmpdir = os.environ.get("TMPDIR", "/tmp")
    mpath = os.path.join(tmpdir, f"dmx_bench_model_{args.model}_{os.getpid()}.bin")
    write_synthetic_model(mpath, 4, 0, "default", "v3" if v3 else "v4")
    model = dmx.Model(mpath, local_rank)
    os.remove(mpath)
    ctx = dmx.Context(model, SEG, B)

    # Everything device-side is ordered on ONE torch stream: the library enqueues on it
    # (dmx_ctx_set_stream), torch ops run on it, RCCL collectives fork from / join into it. The host
    # never blocks inside the timed region, so the gather of step i (RCCL, xGMI) and the root's
    # overlap-add of step i-1 overlap the kernels of the following step on every rank.
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    torch.cuda.set_stream(stream)

    stride = int((1 - 0.25) * SEG)
    nseg_total = world * B
    # This rank's stretch of the track, interleaved stereo, resident in HBM. N = 1, batch 42: literally configs[2]
    # (10 584 000 samples, shift 4033 -> 42 segments). Otherwise a stretch whose segment loop
    # (`for offset < len; offset += stride`, len = n + 22050 - shift) has exactly B iterations at shift 0, and the
    # N*B segments of a step are overlap-added by the root as ONE track of n_track samples.
    literal = world == 1 and B == 42
    n_rank, shift = (240 * 44100, 4033) if literal else (B * stride - 22050, 0)
    n_track = n_rank if literal else nseg_total * stride - 22050
    gen = torch.Generator(device="cpu").manual_seed(1000 + rank)
    d_audio = (0.1 * torch.randn((n_rank, 2), generator=gen)).cuda()
    assert ctx.track_geometry(n_rank, shift)[1] == B
    mix = torch.zeros((B, SEG, 2), device="cuda")  # the step's segments (written by dmx_track_gather_device)
    seg_ids = list(range(B))
    outs = [torch.zeros((B, S, 2, SEG), device="cuda") for _ in range(2)]  # double buffered: step i -> slot i & 1
    out = outs[0]
    d_stats = torch.zeros(4, device="cuda")  # mean / std of the track's mono reference (dmx_track_stats_device)
    allseg = [None, None]   # root: [world*B][S][2][SEG] per slot, rank-major; the gather lands in views of it
    gathered = [None, None]
    track_out = None
    if rank == 0:
        if world > 1:
            allseg = [torch.zeros((world * B, S, 2, SEG), device="cuda") for _ in range(2)]
            gathered = [list(a.chunk(world, dim=0)) for a in allseg]
        else:
            allseg = outs
        track_out = torch.zeros((S, 2, n_track), device="cuda")
    works = [None, None]
    state = {"pending": None}
    torch.cuda.synchronize()

    class HostGather:
        """test mode: gloo gather through host memory with the interface of an async Work"""

        def __init__(self, src, dst_views):
            stream.synchronize()
            h = src.cpu()
            lst = [torch.empty_like(h) for _ in range(world)] if rank == 0 else None
            dist.gather(h, lst, dst=0)
            if rank == 0:
                for v, t in zip(dst_views, lst):
                    v.copy_(t)

        def wait(self):
            pass

    def finish(slot):
        """root: triangle-weighted overlap-add of the step held in `slot` (after its gather landed)"""
        if world > 1:
            works[slot].wait()  # stream-level: `stream` waits for the RCCL gather
        ctx.track_overlap_add_device(allseg[slot].data_ptr(), nseg_total, n_track, shift, d_stats.data_ptr(), track_out.data_ptr())

    def step(i):
        slot = i & 1
        if world > 1 and works[slot] is not None:
            works[slot].wait()  # the gather that last read outs[slot] (step i-2) is complete before it is overwritten
        ctx.track_stats_device(d_audio.data_ptr(), n_rank, d_stats.data_ptr())                      # model_apply.cpp:72-82
        ctx.track_gather_device(d_audio.data_ptr(), n_rank, d_stats.data_ptr(), shift, seg_ids, mix.data_ptr())  # :93-138,189-205,250-263
        ctx.segment_device(mix.data_ptr(), outs[slot].data_ptr(), B)                                # model_inference.cpp:48-475
        if world > 1 and test_mode:
            works[slot] = HostGather(outs[slot], gathered[slot] if rank == 0 else None)
        elif world > 1:
            works[slot] = dist.gather(outs[slot], gathered[slot] if rank == 0 else None, dst=0, async_op=True)
        if rank == 0:
            if state["pending"] is not None:
                finish(state["pending"])  # previous step's overlap-add, behind this step's kernels
            state["pending"] = slot

    def flush():
        if rank == 0 and state["pending"] is not None:
            finish(state["pending"])
        state["pending"] = None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    check = None
    if test_mode and rank == 0:
        check = {"track": None}

    for i in range(args.warmup):
        step(i)
    flush()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    flush()  # the timed region contains exactly K segment batches, K gathers, K overlap-adds
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    out = outs[(args.steps - 1) & 1] if args.steps > 0 else outs[0]
    if test_mode and world > 1:
        # every rank computed the same segments here?  No: ranks use different seeds; the root checks that the
        # track it overlap-added from the gathered slabs equals the overlap-add of the slabs recomputed locally
        torch.cuda.synchronize()
        slot = (args.steps - 1) & 1
        if rank == 0:
            ref = torch.zeros_like(track_out)
            parts = []
            st = torch.zeros(4, device="cuda")
            for r in reversed(range(world)):  # rank 0 last: `st` ends as the root's own statistics
                g = torch.Generator(device="cpu").manual_seed(1000 + r)
                ar = (0.1 * torch.randn((n_rank, 2), generator=g)).cuda()
                mr = torch.zeros((B, SEG, 2), device="cuda")
                o = torch.zeros((B, S, 2, SEG), device="cuda")
                ctx.track_stats_device(ar.data_ptr(), n_rank, st.data_ptr())
                ctx.track_gather_device(ar.data_ptr(), n_rank, st.data_ptr(), shift, seg_ids, mr.data_ptr())
                ctx.segment_device(mr.data_ptr(), o.data_ptr(), B)
                parts.insert(0, o)
            allref = torch.cat(parts, dim=0)
            ctx.track_overlap_add_device(allref.data_ptr(), nseg_total, n_track, shift, st.data_ptr(), ref.data_ptr())
            torch.cuda.synchronize()
            same = bool(torch.equal(ref, track_out)) and bool(torch.equal(allref, allseg[slot]))
            print(f"[test mode] world={world}: gathered slabs and overlap-added track bit-identical to a local recomputation: {same}", flush=True)
            if not same:
                raise SystemExit(3)

    finite = bool(torch.isfinite(out).all().item())

    # BASELINE.json configs[1] read literally: ONE segment per call (latency), device resident
    single_ms = None
    if rank == 0 and world == 1 and not args.no_single:
        torch.cuda.synchronize()
        ctx.set_stream(None)  # the context's own stream: repeated identical calls replay a captured HIP graph
        for _ in range(3):
            ctx.segment_device(mix.data_ptr(), out.data_ptr(), 1)
        ctx.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            ctx.segment_device(mix.data_ptr(), out.data_ptr(), 1)
            ctx.synchronize()  # latency of ONE call: wait for every result before the next call
        single_ms = (time.perf_counter() - t1) / 10 * 1e3
        ctx.set_stream(stream.cuda_stream)

    # ---- BASELINE configs[2]: a 4-minute track end to end (host buffers in and out), and the same track
    # strong-scaled over the ranks (segments dealt in contiguous ranges, RCCL gather, root overlap-add)
    track_4min = None
    track_strong = None
    if not args.no_track:
        n4 = 240 * 44100
        g4 = torch.Generator(device="cpu").manual_seed(1)
        audio_il = (0.1 * torch.randn((n4, 2), generator=g4))  # the same track on every rank
        ctx.set_stream(None)
        torch.cuda.synchronize()
        if rank == 0 and world == 1:
            a_planar = np.ascontiguousarray(audio_il.numpy().T)
            res = np.zeros((S, 2, n4), np.float32)
            ctx.track(a_planar[:, :3 * SEG], 4033)  # warm-up of the scratch buffers
            ctx.track(a_planar, 4033, out=res)
            ts = []
            for _ in range(3):
                t1 = time.perf_counter()
                ctx.track(a_planar, 4033, out=res)
                ts.append(time.perf_counter() - t1)
            track_4min = {"xRT": round(240.0 / min(ts), 1), "wall_s": [round(t, 4) for t in ts], "segments": 42,
                          "host_MB_in_out": round((a_planar.nbytes + res.nbytes) / 1e6, 1), "finite": bool(np.isfinite(res).all())}
            del res
        if not test_mode:
            from demucs_cpp_amd.distributed import HipBackend, track_infer_sharded

            be = HipBackend(ctx)
            pinned = audio_il.pin_memory()
            host_out = torch.empty((S, 2, n4), dtype=torch.float32).pin_memory() if rank == 0 else None
            ts = []
            for it in range(4):  # first pass = warm-up
                fence()
                t1 = time.perf_counter()
                d_audio = pinned.to("cuda", non_blocking=True)
                o = track_infer_sharded(be, d_audio, 4033, dist=dist, rank=rank, world=world)
                if rank == 0:
                    host_out.copy_(o, non_blocking=True)
                fence()
                ts.append(time.perf_counter() - t1)
            tt = torch.tensor(ts[1:], device="cuda", dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            best = float(tt.min().item())
            track_strong = {"xRT": round(240.0 / best, 1), "wall_s": [round(float(t), 4) for t in tt.tolist()], "segments": 42,
                            "ranks": world, "host_buffers": "pinned"}
        ctx.set_stream(stream.cuda_stream)

    roofline = None
    if rank == 0 and not args.no_roofline:
        prof = ctx.profile(B, 3)
        by_kernel = {}
        for nm, k, ms, fl, by in prof:
            d = by_kernel.setdefault(k, [0.0, 0.0, 0.0, 0])
            d[0] += ms
            d[1] += fl
            d[2] += by
            d[3] += 1
        dom = max(by_kernel.items(), key=lambda kv: kv[1][0])
        kname, (ms, fl, by, cnt) = dom
        tot_ms = sum(v[0] for v in by_kernel.values())
        achieved = fl / (ms * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(kname, B, args.model)
        roofline = {
            "bound": "mfma", "kernel": kname, "achieved": round(achieved, 2), "peak": PEAK_TFLOPS_FP32_MFMA,
            "unit": "TFLOP/s", "frac": round(achieved / PEAK_TFLOPS_FP32_MFMA, 4), "traffic": traffic,
            "traffic_source": traffic_src,
            "launches": cnt, "avg_launch_ms": round(ms / cnt, 4),
            "algorithmic_flops_per_launch": fl / cnt, "algorithmic_bytes_per_launch": by / cnt,
            "kernel_share_of_device_time": round(ms / tot_ms, 3),
            "sum_of_kernel_ms_per_step": round(tot_ms, 3),
            "whole_path_tflops": round((sum(v[1] for v in by_kernel.values()) if v3 else MODEL_FLOPS_4S * B) / (tot_ms * 1e-3) / 1e12, 2),
        }

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib as orc  # test infrastructure, timed here only as the reported CPU baseline

        mpath2 = os.path.join(tmpdir, f"dmx_bench_model_cpu_{os.getpid()}.bin")
        write_synthetic_model(mpath2, 4, 0, "default", "v3" if v3 else "v4")
        om = orc.OracleModel(mpath2)
        os.remove(mpath2)
        cm = np.ascontiguousarray(mix[0].cpu().numpy().T)

        def timed(runs=3):  # 1 warm-up + median of >= 3
            om.segment(cm)
            ts = []
            for _ in range(runs):
                t0 = time.perf_counter()
                om.segment(cm)
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts)), ts

        dt, ts = timed()
        cpu_baseline = {"value": round(SEG_SECONDS / dt, 4), "unit": "audio-sec/s", "cores": int(orc.lib().orc_num_threads()),
                        "kind": "port",
                        "sample": f"1 full 7.8 s segment (343980 samples), {'hdemucs_mmi' if v3 else 'htdemucs-4s'} synthetic weights, "
                                  f"1 warm-up + median of 3 ({dt:.1f} s; runs {', '.join(f'{t:.1f}' for t in ts)})",
                        "published_reference_context": "0.385x RT on 16 Zen3 cores, real weights (.github/PERFORMANCE.md:42-47)"}
        blas = orc.use_openblas(True)  # configs[0] names "Eigen/OpenBLAS": the same port on NumPy's bundled OpenBLAS sgemm
        if blas:
            dtb, tsb = timed()
            cpu_baseline.update({"openblas_value": round(SEG_SECONDS / dtb, 4), "openblas_kind": "port+openblas",
                                 "openblas_sample": f"same segment, GEMMs through {os.path.basename(blas)} cblas_sgemm, median of 3 ({dtb:.1f} s)"})
            orc.use_openblas(False)
        om.close()

    # EXPERIMENT, never `value`: the same step with the MFMA-bound convs / linears on the bf16 matrix pipe through exact
    # operand splits (csrc/igemm_split.hip). The switch is read once per process: measured in a child.
    split_probe = None
    if rank == 0 and world == 1 and not args.no_split_probe and os.environ.get("DMX_GEMM", "f32") != "bf16x3":
        import subprocess

        cmd = [sys.executable, os.path.abspath(__file__), "--batch", str(B), "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--model", args.model, "--no-cpu-baseline", "--no-track", "--no-single", "--no-roofline", "--no-split-probe"]
        try:
            r = subprocess.run(cmd, env=dict(os.environ, DMX_GEMM="bf16x3"), capture_output=True, text=True, timeout=600)
            child = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            split_probe = (child["value"], child["config"]["ms_per_segment"], child["config"]["outputs_finite"])
        except Exception as e:  # the probe is informational
            print(f"[bench] split probe failed: {e}", file=sys.stderr)

    if rank == 0:
        audio_s = n_track / 44100.0 * args.steps          # seconds of track produced
        seg_s = nseg_total * SEG_SECONDS * args.steps        # seconds of audio processed (segments overlap by 25 %)
        line = {
            "metric": ("audio-sec/s (xRT) hdemucs_mmi (Demucs v3) 44.1kHz stereo, ~4-min track with overlap-add" if v3 else
                       "audio-sec/s (xRT) htdemucs-4s 44.1kHz stereo, ~4-min track with overlap-add (BASELINE configs[2])"),
            "value": round(audio_s / elapsed, 2),
            "unit": "audio-sec/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("hdemucs_mmi (v3)" if v3 else "htdemucs-4s") + " f16-weights, fp32 MFMA compute: "
                                   + (f"configs[2] literally - one 4-minute track (10 584 000 samples, shift 4033, {B} segments of 343980) "
                                      if literal else f"a track stretch of {B} overlapping 343980-sample segments per GPU ")
                                   + "resident in HBM per step: statistics + segment extraction + segment graph + overlap-add"
                                   + (" + RCCL gather to root" if world > 1 else ""),
                       "value_counts": "seconds of track produced (5.85 s of new audio per 7.8 s segment, stride 257985)",
                       "segments_per_gpu_per_step": B, "segment_samples": SEG, "audio_seconds_per_segment": SEG_SECONDS,
                       "track_samples_per_step": n_track,
                       "segment_seconds_per_s": round(seg_s / elapsed, 2),
                       # scalars (nested objects are dropped by the driver's parser)
                       "track_4min_host_xRT": None if track_4min is None else track_4min["xRT"],
                       "track_4min_host_wall_s": None if track_4min is None else min(track_4min["wall_s"]),
                       "track_4min_host_MB_in_out": None if track_4min is None else track_4min["host_MB_in_out"],
                       "track_strong_xRT": None if track_strong is None else track_strong["xRT"],
                       "track_strong_wall_s": None if track_strong is None else min(track_strong["wall_s"]),
                       "track_strong_ranks": None if track_strong is None else track_strong["ranks"],
                       "ms_per_segment": round(elapsed / args.steps / B * 1e3, 3), "outputs_finite": finite,
                       "gemm_path": os.environ.get("DMX_GEMM", "f32") + (" (EXPERIMENT: exact bf16x3 operand split, fp32 accumulate)"
                                                                         if os.environ.get("DMX_GEMM") == "bf16x3" else " MFMA (v_mfma_f32_16x16x4_f32)"),
                       # opt-in experiment (DMX_GEMM=bf16x3), same step in a child process; NOT the headline
                       "experiment_bf16x3_split_xRT": None if split_probe is None else split_probe[0],
                       "experiment_bf16x3_split_ms_per_segment": None if split_probe is None else split_probe[1],
                       "single_segment_latency_ms": None if single_ms is None else round(single_ms, 3),
                       "single_segment_xRT": None if single_ms is None else round(SEG_SECONDS / (single_ms * 1e-3), 1),
                       # the latency point against the same roofline: 340.2 GFLOP in one call vs the fp32 MFMA peak
                       "single_segment_tflops": None if single_ms is None or v3 else round(MODEL_FLOPS_4S / (single_ms * 1e-3) / 1e12, 2),
                       "single_segment_roofline_frac": None if single_ms is None or v3 else round(MODEL_FLOPS_4S / (single_ms * 1e-3) / 1e12 / PEAK_TFLOPS_FP32_MFMA, 4),
                       "parallelism": f"segment-sharded x{world}"},
        }
        if roofline:
            line["roofline"] = roofline
        if cpu_baseline:
            line["cpu_baseline"] = cpu_baseline
        print(json.dumps(line), flush=True)
    ctx.close()
    model.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

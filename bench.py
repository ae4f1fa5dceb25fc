#!/usr/bin/env python
"""bench.py — throughput of the HTDemucs per-segment hot path + overlapping-segment loop on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
as one rank per GPU by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
environment, backend "nccl" = RCCL) - or, started from a bare shell without WORLD_SIZE, it launches those N ranks itself
(torch.distributed.run on a free local port). Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): htdemucs-4s, synthetic dmc4 weights
(seed 0), a ~4-minute synthetic 44.1 kHz stereo track 0.1*N(0,1) RESIDENT IN HBM, fp32 in / fp32 out. One step on every
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
steps.

--model selects the workload (each has its own metric name; the driver's default is 4s = the BASELINE metric):
  4s  htdemucs 4-source (configs[1] / configs[2])
  6s  htdemucs 6-source, the configs[3] model: the same 4-minute track, 42 segments per GPU per step
  ft  the fine-tuned bag of configs[4]: four 4-source models (synthetic htdemucs_ft_{drums,bass,other,vocals}), every
      model over all segments of the track with its own shift offset (the successive unseeded rand() % 22050 of
      cli-apps/demucs_ft.cpp:221-231: 4033, 12436, 5427, 6865), stem i of the result from model i (:238-241): 4 x 42 =
      168 (model, segment) items per GPU per step, `value` = seconds of track per second for the WHOLE bag
  v3  Demucs v3 hdemucs_mmi
--gemm f32|bf16x3|fp16x3 selects the GEMM arithmetic of the measured context (include/demucs_hip.h DMX_GEMM_*; default: the
library default = bf16x3, the exact operand-split path, unless the environment says otherwise through DMX_GEMM); at N = 1
the OTHER modes are measured on the same workload in the same process and reported as config.f32_mfma_xRT /
_ms_per_segment (config.bf16x3_xRT / ... when the run itself is the fp32 MFMA path) and config.fp16x3_xRT / ... - the
opt-in mode whose linear layers use fp16 terms under a per-row scale (bounded, not exact: never `value` unless asked for).

config also reports, as SCALAR keys, measured in this same run:
  track_4min_host_xRT / track_4min_host_wall_s   (N = 1) the same track end to end with HOST buffers in and out
                      (dmx_track_infer; for the bag dmx_engine_track_infer = the call the drop-in demucs_ft CLI makes):
                      PCIe inclusive, 240 s / best wall of 3; never `value` (the boundary rule of the contract);
  track_strong_xRT / track_strong_wall_s   (every N, single-model workloads) ONE such track with its 42 segments dealt
                      over the N ranks (contiguous ranges), RCCL gather of the per-segment outputs to the root, root
                      overlap-add, D2H on the root: strong scaling of a single track (<= 87.5 % at N = 8: 42 = 6+6+5*6);
  single_segment_latency_ms  BASELINE configs[1] read literally (one segment per call, device resident);
  single_segment_launches    kernel launches of that call (one plan run at batch 1).

Extra objects on the JSON line:
  roofline     : dominant kernel (by device time) measured live with HIP events on the stream
                 it runs on (dmx_debug_profile), algorithmic FLOPs / duration vs the MFMA peak of the arithmetic that
                 kernel runs (fp32 MFMA 157.3 TFLOP/s; exact bf16 split kernels: bf16 dense peak 2516.6 / 5 partial
                 products per term = 503.3 fp32-equivalent TFLOP/s); traffic = PMC HBM bytes (null
                 unless profiles/ holds a counter pass of the same model, mode and batch)
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
PEAK_TFLOPS_FP32_MFMA = 157.3            # v_mfma_f32_16x16x4_f32 (MI355X_MICROARCH.md)
PEAK_TFLOPS_BF16_MFMA = 2516.6           # 256 CUs x 4 SIMDs x 16384 FLOP / 16 cycles x 2.4 GHz (v_mfma_f32_16x16x32_bf16)
MODEL_FLOPS = {"4s": 340.2e9, "6s": 261.7e9}  # algorithmic FLOPs per segment (SURVEY.md §8d / BASELINE.md §3)
SHIFTS_GLIBC = (4033, 12436, 5427, 6865)  # successive unseeded rand() % 22050
FT_NAMES = ("drums", "bass", "other", "vocals")


def kernel_peak(kernel_class):
    """fp32-equivalent MFMA peak of a kernel class: the exact-split kernels issue 5 (GEMM: a1w1, a1w2, a2w1, a2w2, a3w1)
    or 6 (attention: both operands are activations) bf16 MFMAs per fp32 product term."""
    if kernel_class.startswith("igemm_splith"):
        return PEAK_TFLOPS_BF16_MFMA / 3  # fp16 terms (DMX_GEMM_FP16X3): three MFMAs per product term, same pipe rate as bf16
    if kernel_class.startswith("igemm_split"):
        return PEAK_TFLOPS_BF16_MFMA / 5
    if kernel_class.startswith("attention_split"):
        return PEAK_TFLOPS_BF16_MFMA / 6
    return PEAK_TFLOPS_FP32_MFMA


def pmc_traffic(kernel_class, batch, model="4s", gemm="f32"):
    """HBM bytes per launch of a kernel class from the committed rocprofv3 PMC passes
    (profiles/rNN_traffic*.json, written by tools/traffic_json.py from separate FETCH_SIZE and
    WRITE_SIZE passes over this same workload; gfx950 corrections applied there). The counters cannot be
    collected from inside this process, so the newest committed pass is quoted - only if it was taken at
    the same batch size, model and GEMM mode."""
    import glob

    tag = ("" if model == "4s" else f"_{model}") + ("" if gemm == "f32" else f"_{gemm}")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_traffic{tag}.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        if int(d.get("batch", -1)) != batch or d.get("gemm", "f32") != gemm:
            return None, None
        return d["classes"][kernel_class]["traffic_bytes_per_launch"], "profiles/" + os.path.basename(files[-1])
    except Exception:
        return None, None


def committed_clock(kernel_class, gemm="bf16x3"):
    """Sustained shader clock of a kernel class under the bench workload, from the newest committed GRBM_GUI_ACTIVE pass
    (profiles/rNN_effective_clock[_f32].csv, tools/gpu_clock.sh: busy cycles / 8 XCDs / duration). The peaks on the line are
    priced at the 2.4 GHz maximum; this says how much of the distance to the peak is clock, not kernel."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_effective_clock" + ("" if gemm != "f32" else "_f32") + ".csv")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            lines = [l for l in f if l.startswith("class,") or (l.count(",") == 3 and not l.startswith("W"))]
        for r in csv.DictReader(lines):
            if r["class"] == kernel_class:
                return float(r["effective_clock_GHz_per_XCD"]), "profiles/" + os.path.basename(files[-1])
    except Exception:
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("DMX_BENCH_BATCH", "42")),
                    help="segments per GPU (and per model of the bag) per step; 42 = the segments of configs[2]'s 4-minute track")
    ap.add_argument("--model", default="4s", choices=["4s", "6s", "ft", "v3"],
                    help="4s: htdemucs (the BASELINE metric); 6s: the 6-source model of configs[3]; ft: the fine-tuned bag of configs[4]; v3: hdemucs_mmi")
    ap.add_argument("--gemm", default=None, choices=["f32", "bf16x3", "fp16x3"], help="GEMM arithmetic of the measured context (default: library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-single", action="store_true")
    ap.add_argument("--no-track", action="store_true")
    ap.add_argument("--no-other-gemm", action="store_true", help="skip the secondary measurement of the other GEMM mode (N = 1)")
    ap.add_argument("--no-split-probe", action="store_true", help=argparse.SUPPRESS)  # round-3 spelling (tools/gpu_r3*.sh): same as --no-other-gemm
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
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a bare `python bench.py --gpus N`: launch the N ranks ourselves (same command line under torch.distributed.run on a
        # free local port); rank 0 of the child job prints the one JSON line, this process only forwards the exit code
        import socket
        import subprocess

        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}; the launcher's world size is used", file=sys.stderr)
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
    v3, ft = args.model == "v3", args.model == "ft"
    tmpdir = os.environ.get("TMPDIR", "/tmp")
    # ---- synthetic weight files in the reference's container format (no checkpoints exist here)
    specs = {"4s": [(4, 0, "v4")], "6s": [(6, 3, "v4")], "v3": [(4, 0, "v3")], "ft": [(4, 50 + i, "v4") for i in range(4)]}[args.model]
    mpaths = []
    for i, (ns, seed, arch) in enumerate(specs):
        name = f"ggml-model-htdemucs_ft_{FT_NAMES[i]}-4s-f16.bin" if ft else f"dmx_bench_model_{args.model}.bin"
        d = os.path.join(tmpdir, f"dmx_bench_{os.getpid()}")
        os.makedirs(d, exist_ok=True)
        mpaths.append(os.path.join(d, name))
        write_synthetic_model(mpaths[-1], ns, seed, "default", arch)
    models = [dmx.Model(p, local_rank) for p in mpaths]
    M = len(models)
    S = models[0].n_sources
    gemm_names = {"f32": dmx.GEMM_F32, "bf16x3": dmx.GEMM_BF16X3, "fp16x3": dmx.GEMM_FP16X3}
    primary = args.gemm or dmx.GEMM_NAMES[dmx.default_gemm()]
    # the other arithmetics, measured after the timed region on the same buffers (config.<key>_xRT): the fp32 MFMA path (or
    # bf16x3 when the run itself is fp32) and the opt-in fp16-term mode, which is never `value` unless --gemm fp16x3 asks
    others = [g for g in ("f32", "bf16x3", "fp16x3") if g != primary and not (primary == "fp16x3" and g == "bf16x3")]
    other_keys = {"f32": "f32_mfma", "bf16x3": "bf16x3", "fp16x3": "fp16x3"}
    ctx = dmx.Context(models[0], SEG, B, gemm=gemm_names[primary])

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
    # (10 584 000 samples, shift 4033 -> 42 segments; the bag: every model its own offset, 42 segments each). Otherwise a
    # stretch whose segment loop (`for offset < len; offset += stride`, len = n + 22050 - shift) has exactly B iterations
    # at shift 0, and the N*B segments of a step are overlap-added by the root as ONE track of n_track samples.
    literal = world == 1 and B == 42
    n_rank = 240 * 44100 if literal else B * stride - 22050
    shifts = list(SHIFTS_GLIBC[:M]) if literal else [0] * M
    n_track = n_rank if literal else nseg_total * stride - 22050
    gen = torch.Generator(device="cpu").manual_seed(1000 + rank)
    d_audio = (0.1 * torch.randn((n_rank, 2), generator=gen)).cuda()
    for sh in shifts:
        assert ctx.track_geometry(n_rank, sh)[1] == B
    mix = torch.zeros((B, SEG, 2), device="cuda")  # one model's segments of the step (written by dmx_track_gather_device)
    seg_ids = list(range(B))
    outs = [torch.zeros((M, B, S, 2, SEG), device="cuda") for _ in range(2)]  # double buffered: step i -> slot i & 1
    d_stats = torch.zeros(4, device="cuda")  # mean / std of the track's mono reference (dmx_track_stats_device)
    allseg = [None, None]   # root: [M][world*B][S][2][SEG] per slot, rank-major per model; the gathers land in views of it
    gathered = [None, None]
    track_out = bag_tmp = None
    if rank == 0:
        if world > 1:
            allseg = [torch.zeros((M, world * B, S, 2, SEG), device="cuda") for _ in range(2)]
            gathered = [[list(a[mi].chunk(world, dim=0)) for mi in range(M)] for a in allseg]
        else:
            allseg = outs
        track_out = torch.zeros((S, 2, n_track), device="cuda")
        if M > 1:
            bag_tmp = torch.zeros((S, 2, n_track), device="cuda")  # one model's overlap-added stems; stem mi is kept
    works = [[None] * M, [None] * M]
    state = {"pending": None, "ctx": ctx}
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
        """root: triangle-weighted overlap-add of the step held in `slot` (after its gathers landed); the bag keeps
        stem mi of model mi (demucs_ft.cpp:238-241)"""
        c = state["ctx"]
        for mi in range(M):
            if world > 1:
                works[slot][mi].wait()  # stream-level: `stream` waits for the RCCL gather
            dst = track_out if M == 1 else bag_tmp
            c.track_overlap_add_device(allseg[slot][mi].data_ptr(), nseg_total, n_track, shifts[mi], d_stats.data_ptr(), dst.data_ptr())
            if M > 1:
                track_out[mi].copy_(bag_tmp[mi])

    def step(i):
        c = state["ctx"]
        slot = i & 1
        if world > 1:
            for wk in works[slot]:
                if wk is not None:
                    wk.wait()  # the gather that last read outs[slot] (step i-2) is complete before it is overwritten
        c.track_stats_device(d_audio.data_ptr(), n_rank, d_stats.data_ptr())                      # model_apply.cpp:72-82
        for mi in range(M):
            if M > 1:
                c.set_model(models[mi])                                                             # demucs_ft.cpp:221-231
            c.track_gather_device(d_audio.data_ptr(), n_rank, d_stats.data_ptr(), shifts[mi], seg_ids, mix.data_ptr())  # :93-138,189-205,250-263
            c.segment_device(mix.data_ptr(), outs[slot][mi].data_ptr(), B)                         # model_inference.cpp:48-475
            if world > 1 and test_mode:
                works[slot][mi] = HostGather(outs[slot][mi], gathered[slot][mi] if rank == 0 else None)
            elif world > 1:
                works[slot][mi] = dist.gather(outs[slot][mi], gathered[slot][mi] if rank == 0 else None, dst=0, async_op=True)
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

    def timed_run():
        for i in range(args.warmup):
            step(i)
        flush()
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        flush()  # the timed region contains exactly K segment batches (x M models), K gathers, K overlap-adds
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        state["ctx"].synchronize()  # also surfaces a raised device status word (cooperative LSTM kernel of v3)
        return dt

    elapsed = timed_run()
    out = outs[(args.steps - 1) & 1] if args.steps > 0 else outs[0]
    if test_mode and world > 1:
        # the root checks that the track it overlap-added from the gathered slabs equals the overlap-add of the slabs
        # recomputed locally (ranks use different seeds)
        torch.cuda.synchronize()
        slot = (args.steps - 1) & 1
        if rank == 0:
            ref = torch.zeros_like(track_out)
            tmp = torch.zeros_like(track_out)
            st = torch.zeros(4, device="cuda")
            same = True
            for mi in range(M):
                parts = []
                if M > 1:
                    ctx.set_model(models[mi])
                for r in reversed(range(world)):  # rank 0 last: `st` ends as the root's own statistics
                    g = torch.Generator(device="cpu").manual_seed(1000 + r)
                    ar = (0.1 * torch.randn((n_rank, 2), generator=g)).cuda()
                    mr = torch.zeros((B, SEG, 2), device="cuda")
                    o = torch.zeros((B, S, 2, SEG), device="cuda")
                    ctx.track_stats_device(ar.data_ptr(), n_rank, st.data_ptr())
                    ctx.track_gather_device(ar.data_ptr(), n_rank, st.data_ptr(), shifts[mi], seg_ids, mr.data_ptr())
                    ctx.segment_device(mr.data_ptr(), o.data_ptr(), B)
                    parts.insert(0, o)
                allref = torch.cat(parts, dim=0)
                ctx.track_overlap_add_device(allref.data_ptr(), nseg_total, n_track, shifts[mi], st.data_ptr(), tmp.data_ptr())
                torch.cuda.synchronize()
                if M > 1:
                    ref[mi].copy_(tmp[mi])
                else:
                    ref.copy_(tmp)
                same = same and bool(torch.equal(allref, allseg[slot][mi]))
            torch.cuda.synchronize()
            same = same and bool(torch.equal(ref, track_out))
            print(f"[test mode] world={world}: gathered slabs and overlap-added track bit-identical to a local recomputation: {same}", flush=True)
            if not same:
                raise SystemExit(3)

    finite = bool(torch.isfinite(out).all().item())
    if rank == 0:
        finite = finite and bool(torch.isfinite(track_out).all().item())

    # ---- the other GEMM mode on the same workload, same process, same buffers (N = 1)
    other_runs = {}
    if world == 1 and not (args.no_other_gemm or args.no_split_probe) and not test_mode:
        for other in others:
            ctx2 = dmx.Context(models[0], SEG, B, gemm=gemm_names[other])
            ctx2.set_stream(stream.cuda_stream)
            state["ctx"] = ctx2
            dt2 = timed_run()
            other_runs[other] = {"xRT": round(n_track / 44100.0 * args.steps / dt2, 2), "ms_per_segment": round(dt2 / args.steps / (B * M) * 1e3, 3),
                                 "finite": bool(torch.isfinite(track_out).all().item())}
            state["ctx"] = ctx
            torch.cuda.synchronize()
            ctx2.close()
    if M > 1:
        ctx.set_model(models[0])

    # BASELINE.json configs[1] read literally: ONE segment per call (latency), device resident
    single_ms = None
    single_launches = None
    if rank == 0 and world == 1 and not args.no_single:
        torch.cuda.synchronize()
        ctx.set_stream(None)  # the context's own stream: repeated identical calls replay a captured HIP graph
        o1 = out[0]
        for _ in range(3):
            ctx.segment_device(mix.data_ptr(), o1.data_ptr(), 1)
        ctx.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            ctx.segment_device(mix.data_ptr(), o1.data_ptr(), 1)
            ctx.synchronize()  # latency of ONE call: wait for every result before the next call
        single_ms = (time.perf_counter() - t1) / 10 * 1e3
        ctx.set_stream(stream.cuda_stream)
        # kernel launches of one single-segment plan run (the ISTFT op runs inside the overlap-add kernel: no launch of its own)
        single_launches = sum(1 for nm, *_ in ctx.profile(1, 1) if nm != "istft")

    # ---- the same 4-minute track end to end (host buffers in and out), and strong-scaled over the ranks
    # (segments dealt in contiguous ranges, RCCL gather, root overlap-add)
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
            if M == 1:
                run = lambda a, o=None: ctx.track(a, 4033, out=o)
            else:  # the bag through the engine: the call sequence of the drop-in demucs_ft CLI (one device)
                old = dmx.default_gemm()
                dmx.set_default_gemm(gemm_names[primary])
                eng = dmx.Engine(mpaths, [local_rank], max_batch=21)
                dmx.set_default_gemm(old)
                run = lambda a, o=None: eng.track(a, list(SHIFTS_GLIBC), out=o)
            run(a_planar[:, :3 * SEG])  # warm-up of the scratch buffers
            run(a_planar, res)
            ts = []
            for _ in range(3):
                t1 = time.perf_counter()
                run(a_planar, res)
                ts.append(time.perf_counter() - t1)
            track_4min = {"xRT": round(240.0 / min(ts), 1), "wall_s": [round(t, 4) for t in ts], "segments": 42 * M,
                          "host_MB_in_out": round((a_planar.nbytes + res.nbytes) / 1e6, 1), "finite": bool(np.isfinite(res).all())}
            del res
            if M > 1:
                eng.close()
        if world > 1 or not test_mode:
            # ONE 4-minute track strong-scaled over the ranks: the 42 segments (the bag: the 168 (model, segment) items of
            # cli-apps/demucs_ft.cpp:221-241, model-major) dealt in contiguous balanced ranges, one gather, root overlap-add
            from demucs_cpp_amd.distributed import HipBackend, bag_infer_sharded, strong_ceiling, track_infer_sharded

            be = HipBackend(ctx, models if M > 1 else None)
            pinned = audio_il.pin_memory()
            host_out = torch.empty((S, 2, n4), dtype=torch.float32).pin_memory() if rank == 0 else None

            def host_gather(local, gathered):  # test mode (gloo has no device gather): through host memory
                h = local.cpu()
                lst = [torch.empty_like(h) for _ in range(world)] if rank == 0 else None
                dist.gather(h, lst, dst=0)
                if rank == 0:
                    for v, t_ in zip(gathered, lst):
                        v.copy_(t_)

            ts = []
            o = None
            for it in range(2 if test_mode else 4):  # first pass = warm-up
                fence()
                t1 = time.perf_counter()
                d_a = pinned.to("cuda", non_blocking=True)
                if M > 1:
                    o = bag_infer_sharded(be, d_a, list(SHIFTS_GLIBC[:M]), dist=dist, rank=rank, world=world,
                                          gather=host_gather if test_mode else None)
                else:
                    o = track_infer_sharded(be, d_a, 4033, dist=dist, rank=rank, world=world, gather=host_gather if test_mode else None)
                if rank == 0:
                    host_out.copy_(o, non_blocking=True)
                fence()
                ts.append(time.perf_counter() - t1)
            tt = torch.tensor(ts[1:], device="cuda", dtype=torch.float64)
            if world > 1:
                if test_mode:  # gloo reduces host tensors
                    tt = tt.cpu()
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            best = float(tt.min().item())
            track_strong = {"xRT": round(240.0 / best, 1), "wall_s": [round(float(t), 4) for t in tt.tolist()], "segments": 42 * M,
                            "ranks": world, "host_buffers": "pinned", "ceiling": round(strong_ceiling(42 * M, world), 4),
                            "finite": None if o is None else bool(torch.isfinite(o).all().item())}
            if test_mode and world > 1 and rank == 0:
                # the root checks the sharded result against the same track run by one rank alone
                if M > 1:
                    ref1 = bag_infer_sharded(be, d_a, list(SHIFTS_GLIBC[:M]))
                else:
                    ref1 = track_infer_sharded(be, d_a, 4033)
                same1 = bool(torch.equal(ref1, o))
                print(f"[test mode] world={world}: strong-scaled track ({42 * M} items) bit-identical to one rank alone: {same1}", flush=True)
                if not same1:
                    raise SystemExit(4)
            if M > 1:
                ctx.set_model(models[0])
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
        traffic, traffic_src = pmc_traffic(kname, B, args.model, primary)
        peak = kernel_peak(kname)
        seg_flops = MODEL_FLOPS.get("4s" if ft else args.model)
        # whole path against its own roofline: every op priced at max(algorithmic FLOPs / the matrix peak of ITS kernel class,
        # algorithmic bytes / 8 TB/s), summed, over the measured sum of kernel time (1 = every op on its roof)
        roof_ms = sum(max(fl_ / (kernel_peak(k_) * 1e12), by_ / 8.0e12) * 1e3 for _, k_, _, fl_, by_ in prof)
        clock_ghz, clock_src = committed_clock(kname, primary)
        roofline = {
            "bound": "mfma", "kernel": kname, "achieved": round(achieved, 2), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
            "traffic_source": traffic_src,
            "launches": cnt, "avg_launch_ms": round(ms / cnt, 4),
            "algorithmic_flops_per_launch": fl / cnt, "algorithmic_bytes_per_launch": by / cnt,
            "kernel_share_of_device_time": round(ms / tot_ms, 3),
            "sum_of_kernel_ms_per_step": round(tot_ms * M, 3),
            "whole_path_tflops": round((seg_flops * B if seg_flops else sum(v[1] for v in by_kernel.values())) / (tot_ms * 1e-3) / 1e12, 2),
            "whole_path_frac": round(roof_ms / tot_ms, 4),
            "traffic_ratio": None if not traffic else round(traffic / (by / cnt), 3),
            "effective_clock_ghz": clock_ghz, "effective_clock_source": clock_src, "peak_clock_ghz": 2.4,
            "peak_basis": ("fp32 MFMA v_mfma_f32_16x16x4_f32" if peak == PEAK_TFLOPS_FP32_MFMA else
                           f"fp16 MFMA dense {PEAK_TFLOPS_BF16_MFMA} TFLOP/s (the bf16 rate) / 3 partial products per fp32 term (per-row scaled fp16 terms: bounded, not exact)"
                           if kname.startswith("igemm_splith") else
                           f"bf16 MFMA dense {PEAK_TFLOPS_BF16_MFMA} TFLOP/s / {round(PEAK_TFLOPS_BF16_MFMA / peak)} exact partial products per fp32 term"),
        }

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib as orc  # test infrastructure, timed here only as the reported CPU baseline

        om = orc.OracleModel(mpaths[0])
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
        label = {"4s": "htdemucs-4s", "6s": "htdemucs-6s", "v3": "hdemucs_mmi", "ft": "one htdemucs_ft model (the bag runs four)"}[args.model]
        cpu_baseline = {"value": round(SEG_SECONDS / dt / M, 4), "unit": "audio-sec/s", "cores": int(orc.lib().orc_num_threads()),
                        "kind": "port",
                        "sample": f"1 full 7.8 s segment (343980 samples), {label}, synthetic weights, "
                                  f"1 warm-up + median of 3 ({dt:.1f} s; runs {', '.join(f'{t:.1f}' for t in ts)})"
                                  + (f"; value = 7.8 s / ({M} models x {dt:.1f} s)" if M > 1 else ""),
                        "published_reference_context": "0.385x RT on 16 Zen3 cores, real weights (.github/PERFORMANCE.md:42-47)"}
        blas = orc.use_openblas(True)  # configs[0] names "Eigen/OpenBLAS": the same port on NumPy's bundled OpenBLAS sgemm
        if blas:
            dtb, tsb = timed()
            cpu_baseline.update({"openblas_value": round(SEG_SECONDS / dtb / M, 4), "openblas_kind": "port+openblas",
                                 "openblas_sample": f"same segment, GEMMs through {os.path.basename(blas)} cblas_sgemm, median of 3 ({dtb:.1f} s)"})
            orc.use_openblas(False)
        om.close()

    if rank == 0:
        audio_s = n_track / 44100.0 * args.steps          # seconds of track produced
        seg_s = nseg_total * SEG_SECONDS * args.steps        # seconds of audio processed (segments overlap by 25 %)
        wl = {"4s": "htdemucs-4s", "6s": "htdemucs-6s (configs[3] model)", "v3": "hdemucs_mmi (v3)",
              "ft": "htdemucs_ft bag of 4 fine-tuned 4-source models (configs[4]), every model over every segment, stem i from model i"}[args.model]
        arith = {"f32": "fp32 MFMA compute (v_mfma_f32_16x16x4_f32)",
                 "bf16x3": "fp32 products from exact bf16 operand splits (a = a1+a2+a3, w = w1+w2) on the bf16 MFMA pipe, fp32 accumulate",
                 "fp16x3": "OPT-IN mode: as bf16x3, the transformer's linear layers with three fp16 activation terms under a per-row power-of-two scale x one fp16 weight (bounded, not exact)"}
        metric = {"4s": "audio-sec/s (xRT) htdemucs-4s 44.1kHz stereo, ~4-min track with overlap-add (BASELINE configs[2])",
                  "6s": "audio-sec/s (xRT) htdemucs-6s 44.1kHz stereo, ~4-min track with overlap-add (BASELINE configs[3] workload)",
                  "ft": "audio-sec/s (xRT) htdemucs_ft bag-of-4, 44.1kHz stereo, ~4-min track with overlap-add (BASELINE configs[4] workload)",
                  "v3": "audio-sec/s (xRT) hdemucs_mmi (Demucs v3) 44.1kHz stereo, ~4-min track with overlap-add"}[args.model]
        line = {
            "metric": metric,
            "value": round(audio_s / elapsed, 2),
            "unit": "audio-sec/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "f32 (exact bf16x3 operand split, fp32 accumulate)",
                      "fp16x3": "f32 (bf16x3 operand split; linear layers fp16x3 under a per-row scale, fp32 accumulate)"}[primary],
            "data": "synthetic",
            "config": {"workload": wl + " f16-weights, " + arith[primary] + ": "
                                   + (f"one 4-minute track literally (10 584 000 samples, shift {shifts if M > 1 else shifts[0]}, {B} segments of 343980"
                                      + (" per model) " if M > 1 else ") ")
                                      if literal else f"a track stretch of {B} overlapping 343980-sample segments per GPU" + (" and model " if M > 1 else " "))
                                   + "resident in HBM per step: statistics + segment extraction + segment graph + overlap-add"
                                   + (" + RCCL gather to root" if world > 1 else ""),
                       "value_counts": "seconds of track produced (5.85 s of new audio per 7.8 s segment, stride 257985)",
                       "segments_per_gpu_per_step": B * M, "models": M, "segment_samples": SEG, "audio_seconds_per_segment": SEG_SECONDS,
                       "track_samples_per_step": n_track,
                       "segment_seconds_per_s": round(seg_s * M / elapsed, 2),
                       # scalars (nested objects are dropped by the driver's parser)
                       "track_4min_host_xRT": None if track_4min is None else track_4min["xRT"],
                       "track_4min_host_wall_s": None if track_4min is None else min(track_4min["wall_s"]),
                       "track_4min_host_MB_in_out": None if track_4min is None else track_4min["host_MB_in_out"],
                       "track_strong_xRT": None if track_strong is None else track_strong["xRT"],
                       "track_strong_wall_s": None if track_strong is None else min(track_strong["wall_s"]),
                       "track_strong_ranks": None if track_strong is None else track_strong["ranks"],
                       # one track's items (42 segments; the bag: 168) over N ranks: the busiest rank's share bounds the speed-up
                       "track_strong_items": None if track_strong is None else track_strong["segments"],
                       "strong_ceiling": None if track_strong is None else track_strong["ceiling"],
                       "track_strong_outputs_finite": None if track_strong is None else track_strong["finite"],
                       "ms_per_segment": round(elapsed / args.steps / (B * M) * 1e3, 3), "outputs_finite": finite,
                       "gemm_path": primary + ": " + arith[primary],
                       # the other GEMM arithmetic on the same workload, measured in this process right after the timed region
                       **{f"{other_keys[g]}_{k2}": (None if g not in other_runs else other_runs[g][k1])
                          for g in others for k1, k2 in (("xRT", "xRT"), ("ms_per_segment", "ms_per_segment"), ("finite", "outputs_finite"))},
                       "single_segment_latency_ms": None if single_ms is None else round(single_ms, 3),
                       "single_segment_launches": single_launches,
                       "single_segment_xRT": None if single_ms is None else round(SEG_SECONDS / (single_ms * 1e-3), 1),
                       "parallelism": f"segment-sharded x{world}"},
        }
        seg_flops = MODEL_FLOPS.get("4s" if ft else args.model)
        if single_ms is not None and seg_flops:
            # the latency point (configs[1] read literally) against the matrix peak of the arithmetic that RAN: the model's
            # FLOPs in one call / (503.3 for the exact-split GEMMs, 157.3 for fp32 MFMA)
            single_peak = PEAK_TFLOPS_FP32_MFMA if primary == "f32" else PEAK_TFLOPS_BF16_MFMA / 5
            line["config"]["single_segment_tflops"] = round(seg_flops / (single_ms * 1e-3) / 1e12, 2)
            line["config"]["single_segment_roofline_frac"] = round(seg_flops / (single_ms * 1e-3) / 1e12 / single_peak, 4)
            line["config"]["single_segment_peak_tflops"] = round(single_peak, 1)
        if roofline:
            line["roofline"] = roofline
        if cpu_baseline:
            line["cpu_baseline"] = cpu_baseline
        print(json.dumps(line), flush=True)
    ctx.close()
    for m in models:
        m.close()
    for p in mpaths:
        try:
            os.remove(p)
        except OSError:
            pass
    try:
        os.rmdir(os.path.dirname(mpaths[0]))
    except OSError:
        pass
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

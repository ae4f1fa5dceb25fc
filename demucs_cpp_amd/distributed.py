"""Segment-sharded track inference over the GPUs of one node (one process per GPU).

The overlapping-segment loop of the reference (src/model_apply.cpp:189-235, /root/reference)
is embarrassingly parallel: every iteration reads a slice of the shifted track and produces an
independent (S, 2, segment) block; the only coupling is the weighted overlap-add. So:

  * every rank holds the weights and the (small) input track;
  * rank r owns the contiguous segment range [r*n_seg/world, (r+1)*n_seg/world) (all segments cost the
    same: short tails are zero-padded to a full segment, Q8) - the dealing of csrc/engine.cpp;
  * each rank runs its segments through the hot path in batches;
  * ONE exchange step: the per-segment outputs are gathered to the root (RCCL gather over xGMI;
    11 MB per 4-source segment), equal-sized slabs padded to ceil(n_seg/world) segments;
  * the root overlap-adds ALL segments in segment order (bit-identical to the 1-GPU result
    because every output sample accumulates its <= 2 covering segments in increasing index).

`backend` abstracts the three device operations so the bookkeeping can be tested on CPU with
the gloo backend (tests/test_distributed_cpu.py); the product backend is HipBackend.
"""
from __future__ import annotations

from typing import List, Optional

import torch


def owned_segments(n_segments: int, rank: int, world: int) -> List[int]:
    """Contiguous balanced ranges, the dealing of csrc/engine.cpp (dmx_engine_partition): rank r owns
    segments [r*n/world, (r+1)*n/world) - one contiguous slab of results per rank."""
    return list(range(rank * n_segments // world, (rank + 1) * n_segments // world))


def slab_size(n_segments: int, world: int) -> int:
    return (n_segments + world - 1) // world


def strong_ceiling(n_items: int, world: int) -> float:
    """Best possible strong-scaling efficiency of `n_items` equal items dealt in contiguous balanced ranges: the busiest
    rank runs ceil(n/world) of them (42 segments over 8 GPUs: 5.25 / 6 = 0.875; the bag's 168 items: 21 / 21 = 1.0)."""
    return n_items / (world * slab_size(n_items, world))


class HipBackend:
    """Device ops through the C ABI (include/demucs_hip.h) on torch-owned HBM tensors. `models`: the bag of
    cli-apps/demucs_ft.cpp (one context, dmx_ctx_set_model per model), else the context's own model."""

    def __init__(self, ctx, models=None):
        self.ctx = ctx
        self.S = ctx.S
        self.seg = ctx.seg
        self.max_batch = ctx.max_batch
        self.device = torch.device("cuda", ctx.model.device)
        self.models = models

    def set_model(self, mi: int):
        if self.models is not None:
            self.ctx.set_model(self.models[mi])

    def geometry(self, n, shift):
        return self.ctx.track_geometry(n, shift)  # (shifted_len, n_segments, stride)

    @staticmethod
    def _handover():
        # the library runs on its own non-blocking HIP stream: everything torch produced on its
        # current stream must be complete before the library reads it
        torch.cuda.current_stream().synchronize()

    def stats(self, audio_il: torch.Tensor) -> torch.Tensor:
        st = torch.zeros(4, device=self.device)
        self._handover()
        self.ctx.track_stats_device(audio_il.data_ptr(), audio_il.shape[0], st.data_ptr())
        return st

    def infer_segments(self, audio_il, stats, shift, seg_ids: List[int], out: torch.Tensor):
        """out[i] = hot path of segment seg_ids[i]; out: [len(seg_ids)][S][2][seg] on device."""
        n = audio_il.shape[0]
        mix = torch.empty((self.max_batch, self.seg, 2), device=self.device)
        self._handover()
        for i0 in range(0, len(seg_ids), self.max_batch):
            ids = seg_ids[i0:i0 + self.max_batch]
            self.ctx.track_gather_device(audio_il.data_ptr(), n, stats.data_ptr(), shift, ids, mix.data_ptr())
            self.ctx.segment_device(mix.data_ptr(), out[i0].data_ptr(), len(ids))
        self.ctx.synchronize()

    def overlap_add(self, seg_out: torch.Tensor, n_segments, n, shift, stats) -> torch.Tensor:
        out = torch.empty((self.S, 2, n), device=self.device)
        self._handover()
        self.ctx.track_overlap_add_device(seg_out.data_ptr(), n_segments, n, shift, stats.data_ptr(), out.data_ptr())
        self.ctx.synchronize()
        return out


def track_infer_sharded(backend, audio_il: torch.Tensor, shift: int, dist=None, rank: int = 0, world: int = 1,
                        root: int = 0, gather=None) -> Optional[torch.Tensor]:
    """audio_il: [n][2] interleaved stereo on the backend's device (same on every rank).
    Returns (S, 2, n) on the root, None elsewhere."""
    n = int(audio_il.shape[0])
    _, n_seg, _ = backend.geometry(n, shift)
    S, seg = backend.S, backend.seg
    stats = backend.stats(audio_il)  # every rank computes the same statistics locally
    mine = owned_segments(n_seg, rank, world)
    slab = slab_size(n_seg, world)
    local = torch.zeros((slab, S, 2, seg), device=audio_il.device, dtype=torch.float32)
    if mine:
        backend.infer_segments(audio_il, stats, shift, mine, local)
    if world > 1:
        gathered = [torch.empty_like(local) for _ in range(world)] if rank == root else None
        if gather is not None:  # (tests: a host-staged gather for process groups without device collectives)
            gather(local, gathered)
        else:
            dist.gather(local, gathered, dst=root)
        if rank != root:
            return None
        # slab r holds segments [r*n_seg/world, (r+1)*n_seg/world) in its first slots -> segment order
        all_seg = torch.cat([gathered[r][:len(owned_segments(n_seg, r, world))] for r in range(world)], dim=0)
    else:
        all_seg = local[:n_seg]
    return backend.overlap_add(all_seg, n_seg, n, shift, stats)


def bag_infer_sharded(backend, audio_il: torch.Tensor, shifts: List[int], dist=None, rank: int = 0, world: int = 1, root: int = 0,
                      gather=None) -> Optional[torch.Tensor]:
    """The fine-tuned bag (cli-apps/demucs_ft.cpp:221-241: every model over the whole track with its own shift offset, stem i
    kept from model i) strong-scaled over the ranks: the (model, segment) items, model-major, are dealt in contiguous balanced
    ranges like the segments of one model (csrc/engine.cpp deals the same list), ONE gather of equal slabs to the root, which
    overlap-adds every model's segments in segment order and keeps its stem. `gather(local, gathered_list_or_None)`
    replaces dist.gather (tests: a host-staged gather for backends without device collectives)."""
    n = int(audio_il.shape[0])
    M = len(shifts)
    n_segs = [backend.geometry(n, sh)[1] for sh in shifts]
    starts = [0]
    for k in n_segs:
        starts.append(starts[-1] + k)
    n_items = starts[-1]
    S, seg = backend.S, backend.seg
    stats = backend.stats(audio_il)
    mine = owned_segments(n_items, rank, world)
    slab = slab_size(n_items, world)
    local = torch.zeros((slab, S, 2, seg), device=audio_il.device, dtype=torch.float32)
    pos = 0
    for mi in range(M):
        ids = [i - starts[mi] for i in mine if starts[mi] <= i < starts[mi + 1]]
        if ids:
            backend.set_model(mi)
            backend.infer_segments(audio_il, stats, shifts[mi], ids, local[pos:pos + len(ids)])
            pos += len(ids)
    if world > 1:
        gathered = [torch.empty_like(local) for _ in range(world)] if rank == root else None
        if gather is not None:
            gather(local, gathered)
        else:
            dist.gather(local, gathered, dst=root)
        if rank != root:
            return None
        all_items = torch.cat([gathered[r][:len(owned_segments(n_items, r, world))] for r in range(world)], dim=0)
    else:
        all_items = local[:n_items]
    out = torch.empty((S, 2, n), device=audio_il.device, dtype=torch.float32)
    for mi in range(M):
        t = backend.overlap_add(all_items[starts[mi]:starts[mi + 1]].contiguous(), n_segs[mi], n, shifts[mi], stats)
        out[mi].copy_(t[mi])
    return out

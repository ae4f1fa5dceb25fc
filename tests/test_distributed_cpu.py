"""World-size-2 (gloo, CPU) test of the segment-sharding bookkeeping in
demucs_cpp_amd/distributed.py: ownership, slab padding, gather order, root overlap-add.
The device operations are replaced by a CPU backend whose "model" is s-scaled identity, so the
overlap-add of the reference (src/model_apply.cpp:171-246, triangle weights / sum_weight) must
reproduce (s+1) * audio exactly up to rounding, for any sharding."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demucs_cpp_amd.distributed import bag_infer_sharded, owned_segments, slab_size, strong_ceiling, track_infer_sharded  # noqa: E402

SEG, S, MAX_SHIFT = 4000, 3, 22050


class CpuBackend:
    """numpy mirror of the track-level kernels (test double; NOT the product path)."""
    S, seg, max_batch = S, SEG, 2

    def geometry(self, n, shift):
        ln = n + MAX_SHIFT - shift
        stride = int((1 - 0.25) * SEG)
        return ln, (ln + stride - 1) // stride, stride

    def stats(self, audio_il):
        ref = audio_il.mean(dim=1)
        return torch.tensor([ref.mean().item(), ref.std(unbiased=True).item(), 0, 0])

    def _shifted(self, audio_il, stats, shift):
        n = audio_il.shape[0]
        ln = n + MAX_SHIFT - shift
        sh = torch.zeros((ln, 2))
        lo = MAX_SHIFT - shift
        sh[lo:lo + n] = (audio_il - stats[0]) / stats[1]
        return sh

    def infer_segments(self, audio_il, stats, shift, seg_ids, out):
        sh = self._shifted(audio_il, stats, shift)
        ln, _, stride = self.geometry(audio_il.shape[0], shift)
        for i, g in enumerate(seg_ids):
            off = g * stride
            chunk = min(SEG, ln - off)
            left = (SEG - chunk) // 2
            mix = torch.zeros((SEG, 2))
            mix[left:left + chunk] = sh[off:off + chunk]
            for s in range(S):
                out[i, s] = (s + 1) * mix.t()

    def overlap_add(self, seg_out, n_seg, n, shift, stats):
        ln, _, stride = self.geometry(n, shift)
        w = torch.zeros(SEG)
        half = SEG // 2
        w[:half] = torch.arange(1, half + 1) / half
        w[SEG - half:] = torch.flip(w[:half], dims=[0])
        acc = torch.zeros((S, 2, ln))
        sw = torch.zeros(ln)
        for g in range(n_seg):
            off = g * stride
            chunk = min(SEG, ln - off)
            left = (SEG - chunk) // 2
            acc[:, :, off:off + chunk] += w[:chunk] * seg_out[g, :, :, left:left + chunk]
            sw[off:off + chunk] += w[:chunk]
        res = acc / sw
        lo = MAX_SHIFT - shift
        return res[:, :, lo:lo + n] * stats[1] + stats[0]


def _worker(rank, world, port, n, shift, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    audio = 0.1 * torch.randn((n, 2), generator=g)
    out = track_infer_sharded(CpuBackend(), audio, shift, dist=dist, rank=rank, world=world)
    if rank == 0:
        q.put(out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_ownership_and_slabs():
    assert owned_segments(7, 0, 2) == [0, 1, 2] and owned_segments(7, 1, 2) == [3, 4, 5, 6]  # contiguous, balanced
    assert slab_size(7, 2) == 4 and slab_size(42, 8) == 6
    assert sum((owned_segments(42, r, 8) for r in range(8)), []) == list(range(42))
    assert sorted(len(owned_segments(42, r, 8)) for r in range(8)) == [5] * 6 + [6] * 2
    # the same dealing as the in-process engine (csrc/engine.cpp, dmx_engine_partition)
    from demucs_cpp_amd import binding
    for n_seg, world in ((42, 8), (7, 2), (3, 8), (41, 5)):
        runs = binding.engine_partition([n_seg], world)
        for r in range(world):
            own = owned_segments(n_seg, r, world)
            assert runs[r] == ([(0, own[0], own[-1] + 1)] if own else [])


@pytest.mark.parametrize("n,shift", [(3 * SEG + 777, 4033), (SEG // 3, 12436)])
def test_world2_equals_world1_and_identity(n, shift):
    g = torch.Generator().manual_seed(0)
    audio = 0.1 * torch.randn((n, 2), generator=g)
    single = track_infer_sharded(CpuBackend(), audio, shift).numpy()
    for s in range(S):  # weights normalised: (s+1)*identity survives the overlap-add
        mean = audio.mean(dim=1).mean()
        expect = ((s + 1) * (audio - mean) + mean).t().numpy()
        assert np.abs(single[s] - expect).max() < 1e-5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, shift, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(res, single)  # same segment order on the root => bit-identical


class BagBackend(CpuBackend):
    """the bag's test double: "model" mi scales stem s by (s + 1) * (10 + mi), so the kept stem mi must read (mi + 1) * (10 + mi)"""

    def __init__(self):
        self.mi = 0

    def set_model(self, mi):
        self.mi = mi

    def infer_segments(self, audio_il, stats, shift, seg_ids, out):
        super().infer_segments(audio_il, stats, shift, seg_ids, out)
        out *= 10 + self.mi


def _bag_worker(rank, world, port, n, shifts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    audio = 0.1 * torch.randn((n, 2), generator=g)
    out = bag_infer_sharded(BagBackend(), audio, shifts, dist=dist, rank=rank, world=world)
    if rank == 0:
        q.put(out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_bag_items_sharded_over_world2_equal_world1():
    """configs[4] (cli-apps/demucs_ft.cpp:221-241): the (model, segment) items, model-major, in contiguous balanced ranges; the
    root keeps stem mi of model mi. Shift offsets that give the models DIFFERENT segment counts, so that rank boundaries fall
    inside a model and the per-model bookkeeping is exercised."""
    n, shifts = 3 * SEG + 777, [4033, 12436, 21000]  # S = 3 stems -> a bag of 3 "models"
    assert len({CpuBackend().geometry(n, sh)[1] for sh in shifts}) > 1
    g = torch.Generator().manual_seed(0)
    audio = 0.1 * torch.randn((n, 2), generator=g)
    single = bag_infer_sharded(BagBackend(), audio, shifts).numpy()
    mean = audio.mean(dim=1).mean()
    for mi in range(S):
        expect = ((mi + 1) * (10 + mi) * (audio - mean) + mean).t().numpy()
        assert np.abs(single[mi] - expect).max() < 1e-4
    assert strong_ceiling(42, 8) == 0.875 and strong_ceiling(168, 8) == 1.0 and strong_ceiling(42, 1) == 1.0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bag_worker, args=(r, 2, port, n, shifts, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(res, single)

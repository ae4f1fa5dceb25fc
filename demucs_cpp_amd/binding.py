"""ctypes binding of the C ABI in include/demucs_hip.h (libdemucs_hip.so, built in-tree
by `make` / __graft_entry__.build()).

This module is plumbing for tests and bench.py; the product is the shared library.
There is deliberately NO fallback: if the library is missing, or no GPU is usable, the
calls raise.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Tuple

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("DMX_LIB", os.path.join(ROOT, "demucs_cpp_amd", "lib", "libdemucs_hip.so"))

LAYOUT_EIGEN = 0
LAYOUT_PLANAR = 1
SEGMENT_SAMPLES = 343980
MAX_SHIFT = 22050

# every symbol include/demucs_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "dmx_last_error", "dmx_device_count", "dmx_model_load", "dmx_model_free", "dmx_model_n_sources",
    "dmx_model_n_tensors", "dmx_model_device", "dmx_ctx_create", "dmx_ctx_free", "dmx_ctx_segment_samples",
    "dmx_ctx_max_batch", "dmx_ctx_arena_bytes", "dmx_ctx_synchronize", "dmx_ctx_set_stream", "dmx_segment_infer",
    "dmx_segment_infer_device", "dmx_track_infer", "dmx_track_geometry", "dmx_track_stats_device",
    "dmx_track_gather_device", "dmx_track_overlap_add_device", "dmx_debug_tap", "dmx_debug_n_ops",
    "dmx_debug_profile", "dmx_ctx_set_model", "dmx_model_clone",
    "dmx_engine_create", "dmx_engine_free", "dmx_engine_n_devices", "dmx_engine_n_models", "dmx_engine_n_sources",
    "dmx_resample_length", "dmx_resample_filter", "dmx_resample_device", "dmx_resample",
    "dmx_ctx_create_gemm", "dmx_ctx_gemm", "dmx_default_gemm", "dmx_set_default_gemm", "dmx_debug_split_weights", "dmx_debug_split_activations", "dmx_debug_split_activations_fp16",
    "dmx_model_arch", "dmx_engine_arch", "dmx_engine_transport", "dmx_engine_set_finish", "dmx_engine_finish", "dmx_engine_root_ctx", "dmx_engine_track_infer", "dmx_engine_partition",
]

TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_P2P = 0, 1, 2
GEMM_F32, GEMM_BF16X3, GEMM_FP16X3 = 0, 1, 2  # include/demucs_hip.h DMX_GEMM_* (FP16X3: opt-in, linear layers with fp16 terms)
GEMM_NAMES = {GEMM_F32: "f32", GEMM_BF16X3: "bf16x3", GEMM_FP16X3: "fp16x3"}
FINISH_ROOT, FINISH_OWNER = 0, 1

_lib = None
PROGRESS_FN = ctypes.CFUNCTYPE(None, ctypes.c_float, ctypes.c_char_p, ctypes.c_void_p)


class DmxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[dmx error {code}] {msg}")
        self.code = code


def lib():
    global _lib
    if _lib is None:
        # PyTorch-ROCm bundles its own libamdhip64.so.7; a process must hold ONE HIP runtime.
        # Importing torch first makes the dynamic linker bind our NEEDED libamdhip64.so.7 to the
        # copy torch already loaded (same SONAME), so torch tensors / RCCL and this library share
        # the device context. (Loading ours first left torch with "No HIP GPUs are available".)
        if os.environ.get("DMX_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except Exception:  # torch is plumbing only; the library works without it
                pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `make` or __graft_entry__.build() (no fallback exists)")
        L = ctypes.CDLL(LIB_PATH)
        vp, i64, ci, fp = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
        L.dmx_last_error.restype = ctypes.c_char_p
        L.dmx_model_load.argtypes = [ctypes.c_char_p, ci, ctypes.POINTER(vp)]
        L.dmx_model_free.argtypes = [vp]
        for f in ("dmx_model_n_sources", "dmx_model_n_tensors", "dmx_model_device", "dmx_model_arch"):
            getattr(L, f).argtypes = [vp]
        L.dmx_ctx_create.argtypes = [vp, i64, ci, ctypes.POINTER(vp)]
        L.dmx_ctx_create_gemm.argtypes = [vp, i64, ci, ci, ctypes.POINTER(vp)]
        L.dmx_ctx_gemm.argtypes = [vp]
        L.dmx_set_default_gemm.argtypes = [ci]
        L.dmx_debug_split_weights.argtypes = [fp, i64, fp, fp]
        L.dmx_debug_split_weights.restype = i64
        L.dmx_debug_split_activations.argtypes = [ci, fp, i64, fp]
        L.dmx_debug_split_activations_fp16.argtypes = [ci, fp, i64, ci, fp]
        L.dmx_ctx_free.argtypes = [vp]
        L.dmx_ctx_segment_samples.argtypes = [vp]
        L.dmx_ctx_segment_samples.restype = i64
        L.dmx_ctx_max_batch.argtypes = [vp]
        L.dmx_ctx_arena_bytes.argtypes = [vp]
        L.dmx_ctx_arena_bytes.restype = i64
        L.dmx_ctx_synchronize.argtypes = [vp]
        L.dmx_ctx_set_stream.argtypes = [vp, vp]
        L.dmx_segment_infer.argtypes = [vp, fp, fp, ci]
        L.dmx_segment_infer_device.argtypes = [vp, fp, fp, ci]
        L.dmx_track_infer.argtypes = [vp, fp, i64, ci, fp, ci, vp, vp]
        L.dmx_track_geometry.argtypes = [vp, i64, ci, ctypes.POINTER(i64), ctypes.POINTER(ci), ctypes.POINTER(i64)]
        L.dmx_track_stats_device.argtypes = [vp, fp, i64, fp]
        L.dmx_track_gather_device.argtypes = [vp, fp, i64, fp, ci, vp, ci, fp]
        L.dmx_track_overlap_add_device.argtypes = [vp, fp, ci, i64, ci, fp, fp, ci]
        L.dmx_debug_tap.argtypes = [vp, ctypes.c_char_p, vp, fp]
        L.dmx_debug_n_ops.argtypes = [vp]
        L.dmx_debug_profile.argtypes = [vp, ci, ci, ctypes.c_char_p, ci]
        L.dmx_ctx_set_model.argtypes = [vp, vp]
        L.dmx_engine_create.argtypes = [ctypes.POINTER(ctypes.c_char_p), ci, ctypes.POINTER(ci), ci, ci, ci, ctypes.POINTER(vp)]
        L.dmx_engine_free.argtypes = [vp]
        L.dmx_engine_set_finish.argtypes = [vp, ci]
        for f in ("dmx_engine_n_devices", "dmx_engine_n_models", "dmx_engine_n_sources", "dmx_engine_transport", "dmx_engine_finish"):
            getattr(L, f).argtypes = [vp]
        L.dmx_engine_root_ctx.argtypes = [vp, ci]
        L.dmx_engine_root_ctx.restype = vp
        L.dmx_engine_track_infer.argtypes = [vp, fp, i64, ctypes.POINTER(ci), fp, ci, vp, vp]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise DmxError(rc, lib().dmx_last_error().decode(errors="replace"))


def device_count() -> int:
    return lib().dmx_device_count()


class Model:
    """demucscpp::demucs_model + load_demucs_model (src/model.hpp:285-554, :649)."""

    def __init__(self, path: str, device: int = 0):
        self.h = ctypes.c_void_p()
        _chk(lib().dmx_model_load(path.encode(), device, ctypes.byref(self.h)))
        self.n_sources = lib().dmx_model_n_sources(self.h)
        self.n_tensors = lib().dmx_model_n_tensors(self.h)
        self.arch = lib().dmx_model_arch(self.h)  # 4: HTDemucs v4, 3: Demucs v3 (hdemucs_mmi)
        self.device = device

    def close(self):
        if self.h:
            lib().dmx_model_free(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_gemm() -> int:
    return lib().dmx_default_gemm()


def set_default_gemm(gemm: int):
    """GEMM arithmetic of contexts (and engines) created from now on: GEMM_F32 | GEMM_BF16X3 | GEMM_FP16X3."""
    _chk(lib().dmx_set_default_gemm(gemm))


def split_weights(w: np.ndarray):
    """(w1, w2, n_inexact): bf16 bit patterns of the two-term weight split of GEMM_BF16X3 (host function)."""
    w = np.ascontiguousarray(w, np.float32).ravel()
    w1 = np.zeros(w.size, np.uint16)
    w2 = np.zeros(w.size, np.uint16)
    bad = lib().dmx_debug_split_weights(w.ctypes.data, w.size, w1.ctypes.data, w2.ctypes.data)
    return w1, w2, int(bad)


def split_activations(x: np.ndarray, device: int = 0) -> np.ndarray:
    """(3, n) bf16 bit patterns a1, a2, a3 of the kernels' three-term activation split (runs on the GPU)."""
    x = np.ascontiguousarray(x, np.float32).ravel()
    planes = np.zeros((3, x.size), np.uint16)
    _chk(lib().dmx_debug_split_activations(device, x.ctypes.data, x.size, planes.ctypes.data))
    return planes


def split_activations_fp16(x: np.ndarray, scale_exp: int = 0, device: int = 0) -> np.ndarray:
    """(3, n) fp16 bit patterns h1, h2, h3 of the fp16-term split of x * 2^scale_exp (GEMM_FP16X3; runs on the GPU)."""
    x = np.ascontiguousarray(x, np.float32).ravel()
    planes = np.zeros((3, x.size), np.uint16)
    _chk(lib().dmx_debug_split_activations_fp16(device, x.ctypes.data, x.size, int(scale_exp), planes.ctypes.data))
    return planes


class Context:
    def __init__(self, model: Model, segment_samples: int = 0, max_batch: int = 1, gemm: Optional[int] = None):
        self.model = model
        self.h = ctypes.c_void_p()
        if gemm is None:
            _chk(lib().dmx_ctx_create(model.h, segment_samples, max_batch, ctypes.byref(self.h)))
        else:
            _chk(lib().dmx_ctx_create_gemm(model.h, segment_samples, max_batch, gemm, ctypes.byref(self.h)))
        self.gemm = lib().dmx_ctx_gemm(self.h)
        self.seg = lib().dmx_ctx_segment_samples(self.h)
        self.max_batch = max_batch
        self.S = model.n_sources

    def close(self):
        if self.h:
            lib().dmx_ctx_free(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def arena_bytes(self) -> int:
        return lib().dmx_ctx_arena_bytes(self.h)

    def synchronize(self):
        _chk(lib().dmx_ctx_synchronize(self.h))

    def set_model(self, model: "Model"):
        """Rebind to another model of the same architecture on the same device (the fine-tuned bag shares one arena)."""
        _chk(lib().dmx_ctx_set_model(self.h, model.h))
        self.model = model

    def set_stream(self, hip_stream: Optional[int]):
        """Order the context's device work on a caller-owned hipStream_t (raw handle, e.g.
        torch.cuda.Stream().cuda_stream); None returns to the context's own stream."""
        _chk(lib().dmx_ctx_set_stream(self.h, ctypes.c_void_p(hip_stream) if hip_stream else None))

    # ---- host-pointer API
    def segment(self, mix: np.ndarray) -> np.ndarray:
        """mix (2, seg) planar -> (S, 2, seg) planar; demucscpp::model_inference."""
        mix = np.ascontiguousarray(mix, np.float32)
        assert mix.shape == (2, self.seg)
        out = np.zeros((self.S, 2, self.seg), np.float32)
        _chk(lib().dmx_segment_infer(self.h, mix.ctypes.data, out.ctypes.data, LAYOUT_PLANAR))
        return out

    def segment_eigen(self, mix_interleaved: np.ndarray) -> np.ndarray:
        """Eigen memory images: in (seg, 2) interleaved, out flat image of Tensor3dXf(S,2,seg)."""
        mix = np.ascontiguousarray(mix_interleaved, np.float32)
        out = np.zeros(self.S * 2 * self.seg, np.float32)
        _chk(lib().dmx_segment_infer(self.h, mix.ctypes.data, out.ctypes.data, LAYOUT_EIGEN))
        return out

    def track(self, audio: np.ndarray, shift_offset: int, progress=None, out: Optional[np.ndarray] = None) -> np.ndarray:
        """audio (2, n) planar -> (S, 2, n); demucscpp::demucs_inference. `out`: reuse a result buffer."""
        audio = np.ascontiguousarray(audio, np.float32)
        n = audio.shape[1]
        if out is None:
            out = np.zeros((self.S, 2, n), np.float32)
        assert out.shape == (self.S, 2, n) and out.dtype == np.float32 and out.flags.c_contiguous
        cb = PROGRESS_FN(lambda p, m, u: progress(p, m.decode())) if progress else None
        cbp = ctypes.cast(cb, ctypes.c_void_p) if cb else None
        _chk(lib().dmx_track_infer(self.h, audio.ctypes.data, n, shift_offset, out.ctypes.data, LAYOUT_PLANAR, cbp, None))
        return out

    def track_geometry(self, n: int, shift_offset: int) -> Tuple[int, int, int]:
        ln, st, ns = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        _chk(lib().dmx_track_geometry(self.h, n, shift_offset, ctypes.byref(ln), ctypes.byref(ns), ctypes.byref(st)))
        return ln.value, ns.value, st.value

    # ---- device-pointer API (raw addresses, e.g. torch.Tensor.data_ptr())
    def segment_device(self, d_mix: int, d_out: int, batch: int):
        _chk(lib().dmx_segment_infer_device(self.h, d_mix, d_out, batch))

    def track_stats_device(self, d_audio: int, n: int, d_stats: int):
        _chk(lib().dmx_track_stats_device(self.h, d_audio, n, d_stats))

    def track_gather_device(self, d_audio: int, n: int, d_stats: int, shift_offset: int, seg_idx: List[int], d_mix: int):
        arr = (ctypes.c_int * len(seg_idx))(*seg_idx)
        _chk(lib().dmx_track_gather_device(self.h, d_audio, n, d_stats, shift_offset, arr, len(seg_idx), d_mix))

    def track_overlap_add_device(self, d_seg_out: int, n_segments: int, n: int, shift_offset: int, d_stats: int, d_out: int,
                                 layout: int = LAYOUT_PLANAR):
        _chk(lib().dmx_track_overlap_add_device(self.h, d_seg_out, n_segments, n, shift_offset, d_stats, d_out, layout))

    # ---- debug
    def tap(self, name: str) -> Optional[np.ndarray]:
        shape = (ctypes.c_int64 * 8)()
        nd = lib().dmx_debug_tap(self.h, name.encode(), shape, None)
        if nd < 0:
            return None
        shp = [shape[i] for i in range(nd)]
        out = np.zeros(shp, np.float32)
        lib().dmx_debug_tap(self.h, name.encode(), shape, out.ctypes.data)
        return out

    def profile(self, batch: int = 1, reps: int = 3):
        """[(op name, kernel, ms per launch, algorithmic flops, algorithmic bytes)]"""
        cap = 1 << 17
        self.profile_geometry = {}
        buf = ctypes.create_string_buffer(cap)
        n = lib().dmx_debug_profile(self.h, batch, reps, buf, cap)
        if n < 0:
            raise RuntimeError("profile failed")
        rows = []
        for ln in buf.value.decode().split("\n"):
            if not ln:
                continue
            nm, k, ms, fl, by = ln.split("\t")[:5]
            rows.append((nm, k, float(ms), float(fl), float(by)))
            self.profile_geometry[nm] = ln.split("\t")[5] if ln.count("\t") >= 5 else ""
        return rows


def resample_length(n_in: int, rate_in: int, rate_out: int) -> int:
    L = lib()
    L.dmx_resample_length.restype = ctypes.c_int64
    L.dmx_resample_length.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    return L.dmx_resample_length(n_in, rate_in, rate_out)


def resample_filter(rate_in: int, rate_out: int):
    """(up, down, taps) of the polyphase filter of csrc/resample.hip (host function, no GPU)."""
    L = lib()
    L.dmx_resample_filter.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                      ctypes.POINTER(ctypes.c_int), ctypes.c_void_p, ctypes.c_int]
    up, down, nt = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _chk(L.dmx_resample_filter(rate_in, rate_out, ctypes.byref(up), ctypes.byref(down), ctypes.byref(nt), None, 0))
    taps = np.zeros(nt.value, np.float32)
    _chk(L.dmx_resample_filter(rate_in, rate_out, None, None, None, taps.ctypes.data, nt.value))
    return up.value, down.value, taps


def resample(x: np.ndarray, rate_in: int, rate_out: int, interleaved: bool = False, device: int = 0) -> np.ndarray:
    """x (planes, n) planar or (n, planes) interleaved float32 -> the same layout at rate_out (GPU, host buffers)."""
    L = lib()
    L.dmx_resample.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_void_p]
    x = np.ascontiguousarray(x, np.float32)
    n, planes = (x.shape[0], x.shape[1]) if interleaved else (x.shape[1], x.shape[0])
    m = resample_length(n, rate_in, rate_out)
    if m < 0:
        raise ValueError(f"invalid rates {rate_in} -> {rate_out}")
    out = np.zeros((m, planes) if interleaved else (planes, m), np.float32)
    _chk(L.dmx_resample(device, x.ctypes.data, n, planes, 1 if interleaved else 0, rate_in, rate_out, out.ctypes.data))
    return out


def engine_partition(n_segments, n_devices):
    """[(device l) -> [(model, g0, g1), ...]]: the contiguous balanced dealing of csrc/engine.cpp."""
    M = len(n_segments)
    ns = (ctypes.c_int * M)(*n_segments)
    out = (ctypes.c_int * (n_devices * M * 2))()
    L = lib()
    L.dmx_engine_partition.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    _chk(L.dmx_engine_partition(ns, M, n_devices, out))
    res = []
    for l in range(n_devices):
        res.append([(m, out[(l * M + m) * 2], out[(l * M + m) * 2 + 1]) for m in range(M)
                    if out[(l * M + m) * 2 + 1] > out[(l * M + m) * 2]])
    return res


class Engine:
    """Several GPUs and / or a bag of models in one process (csrc/engine.cpp): demucs_inference with the
    (model, segment) items sharded over `devices`; `devices` may repeat an id (logical devices on one GPU)."""

    def __init__(self, model_files, devices=None, max_batch: int = 4, transport: int = TRANSPORT_AUTO, finish: Optional[int] = None):
        files = (ctypes.c_char_p * len(model_files))(*[f.encode() for f in model_files])
        devs = (ctypes.c_int * len(devices))(*devices) if devices else None
        self.h = ctypes.c_void_p()
        _chk(lib().dmx_engine_create(files, len(model_files), devs, len(devices) if devices else 0, max_batch, transport,
                                     ctypes.byref(self.h)))
        self.S = lib().dmx_engine_n_sources(self.h)
        self.n_models = lib().dmx_engine_n_models(self.h)
        self.n_devices = lib().dmx_engine_n_devices(self.h)
        self.transport = lib().dmx_engine_transport(self.h)
        if finish is not None:
            self.set_finish(finish)

    def set_finish(self, finish: int):
        """FINISH_ROOT: blocks gathered and overlap-added on the first device; FINISH_OWNER: every device finishes
        the stretch of the track its segments cover (only segment tails are exchanged). Same bits."""
        _chk(lib().dmx_engine_set_finish(self.h, finish))

    @property
    def finish(self) -> int:
        return lib().dmx_engine_finish(self.h)

    def close(self):
        if self.h:
            lib().dmx_engine_free(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def track(self, audio: np.ndarray, shift_offsets, progress=None, out: Optional[np.ndarray] = None,
              layout: int = LAYOUT_PLANAR) -> np.ndarray:
        """audio (2, n) planar -> (S, 2, n); one shift offset per model. `out`: reuse a result buffer.
        layout=LAYOUT_EIGEN passes the track and receives the result through the C ABI as Eigen column-major
        images (what the C++ shim does); the arrays seen by the caller are the same."""
        audio = np.ascontiguousarray(audio, np.float32)
        n = audio.shape[1]
        so = (ctypes.c_int * self.n_models)(*shift_offsets)
        cb = PROGRESS_FN(lambda p, m, u: progress(p, m.decode())) if progress else None
        cbp = ctypes.cast(cb, ctypes.c_void_p) if cb else None
        if layout == LAYOUT_EIGEN:
            a = np.ascontiguousarray(audio.T)  # [n][2] == column-major 2 x n
            img = np.zeros((n, 2, self.S), np.float32)  # flat index s + S*(c + 2*i)
            _chk(lib().dmx_engine_track_infer(self.h, a.ctypes.data, n, so, img.ctypes.data, LAYOUT_EIGEN, cbp, None))
            res = np.ascontiguousarray(img.transpose(2, 1, 0))
            if out is not None:
                out[...] = res
                return out
            return res
        if out is None:
            out = np.zeros((self.S, 2, n), np.float32)
        assert out.shape == (self.S, 2, n) and out.dtype == np.float32 and out.flags.c_contiguous
        _chk(lib().dmx_engine_track_infer(self.h, audio.ctypes.data, n, so, out.ctypes.data, LAYOUT_PLANAR, cbp, None))
        return out


"""ctypes binding of the CPU oracle (oracle/_build/liboracle_demucs.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. The product package never imports this module."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_build", "liboracle_demucs.so")

_lib = None


def build_if_needed():
    src = os.path.join(ROOT, "oracle", "demucs_oracle.cpp")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build_if_needed()
        L = ctypes.CDLL(SO)
        vp, i64, f32p = ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p
        L.orc_last_error.restype = ctypes.c_char_p
        L.orc_model_load.restype = vp
        L.orc_model_load.argtypes = [ctypes.c_char_p]
        L.orc_model_free.argtypes = [vp]
        L.orc_model_n_sources.argtypes = [vp]
        L.orc_model_n_tensors.argtypes = [vp]
        L.orc_segment_infer.argtypes = [vp, f32p, i64, f32p]
        L.orc_track_infer.argtypes = [vp, f32p, i64, ctypes.c_int, i64, f32p]
        L.orc_geometry.argtypes = [i64, vp]
        L.orc_taps_enable.argtypes = [ctypes.c_int]
        L.orc_tap_numel.restype = i64
        L.orc_tap_numel.argtypes = [ctypes.c_char_p]
        L.orc_tap_shape.argtypes = [ctypes.c_char_p, vp]
        L.orc_tap_copy.argtypes = [ctypes.c_char_p, f32p]
        L.orc_stft.argtypes = [f32p, i64, f32p]
        L.orc_istft.argtypes = [f32p, ctypes.c_int, f32p, i64]
        L.orc_layer_norm.argtypes = [f32p, i64, i64, f32p, f32p, ctypes.c_float, f32p]
        L.orc_group_norm1.argtypes = [f32p, i64, i64, i64, f32p, f32p, ctypes.c_float, ctypes.c_int]
        L.orc_conv2d.argtypes = [f32p, i64, i64, i64, f32p, i64, i64, i64, f32p] + [ctypes.c_int] * 7 + [f32p, vp]
        L.orc_conv_tr_h.argtypes = [f32p, i64, i64, i64, f32p, i64, ctypes.c_int, ctypes.c_int, f32p, ctypes.c_int, f32p]
        L.orc_sin_embedding_2d.argtypes = [i64, i64, i64, f32p]
        L.orc_sin_embedding_1d.argtypes = [i64, i64, f32p]
        L.orc_set_num_threads.argtypes = [ctypes.c_int]
        L.orc_model_arch.argtypes = [vp]
        L.orc_use_blas.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        L.orc_group_norm_g.argtypes = [f32p, i64, i64, i64, f32p, f32p, ctypes.c_int, ctypes.c_float, ctypes.c_int]
        L.orc_lstm.argtypes = [vp, ctypes.c_char_p, f32p, i64, i64, f32p]
        L.orc_local_attention.argtypes = [vp, ctypes.c_char_p, f32p, i64, i64]
        _lib = L
    return _lib


def use_openblas(on=True):
    """cpu_baseline leg only: route the oracle's GEMMs through the OpenBLAS bundled with NumPy (ILP64 cblas_sgemm).
    Returns the library path, or None when it cannot be found / bound (the own SGEMM stays in place)."""
    import glob
    if not on:
        lib().orc_use_blas(None, None)
        return None
    base = os.path.dirname(os.path.dirname(np.__file__))
    cands = glob.glob(os.path.join(base, "numpy.libs", "libscipy_openblas64_*.so"))
    for c in cands:
        if lib().orc_use_blas(c.encode(), b"scipy_cblas_sgemm64_") == 0:
            return c
    return None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleModel:
    def __init__(self, path):
        self.h = lib().orc_model_load(path.encode())
        if not self.h:
            raise RuntimeError(lib().orc_last_error().decode())
        self.n_sources = lib().orc_model_n_sources(self.h)
        self.n_tensors = lib().orc_model_n_tensors(self.h)
        self.arch = lib().orc_model_arch(self.h)  # 4: HTDemucs, 3: Demucs v3 (hdemucs_mmi)

    def close(self):
        if self.h:
            lib().orc_model_free(self.h)
            self.h = None

    def segment(self, mix, taps=False):
        mix = _f32(mix)
        seg = mix.shape[1]
        out = np.zeros((self.n_sources, 2, seg), np.float32)
        lib().orc_taps_enable(1 if taps else 0)
        lib().orc_segment_infer(self.h, mix.ctypes.data, seg, out.ctypes.data)
        return out

    def track(self, audio, shift_offset, seg=343980):
        audio = _f32(audio)
        n = audio.shape[1]
        out = np.zeros((self.n_sources, 2, n), np.float32)
        lib().orc_track_infer(self.h, audio.ctypes.data, n, int(shift_offset), seg, out.ctypes.data)
        return out


    # ---- v3 primitives on this model's weights
    def lstm(self, prefix, x):
        """x (T, H) -> (T, 2H); prefix e.g. 'encoder.4.dconv.layers.0.3.lstm.'"""
        x = _f32(x)
        T, H = x.shape
        out = np.zeros((T, 2 * H), np.float32)
        lib().orc_lstm(self.h, prefix.encode(), x.ctypes.data, T, H, out.ctypes.data)
        return out

    def local_attention(self, prefix, x):
        """x (C, T) -> (C, T); prefix e.g. 'encoder.4.dconv.layers.0.4.'"""
        x = _f32(x).copy()
        lib().orc_local_attention(self.h, prefix.encode(), x.ctypes.data, x.shape[0], x.shape[1])
        return x


def group_norm_g(x, w, b, G, eps=1e-5, gelu=False):
    """x (D0, C, L): statistics per group over (D0, C/G, L) (reference generalized_group_norm)."""
    x = _f32(x).copy()
    w, b = _f32(w), _f32(b)
    lib().orc_group_norm_g(x.ctypes.data, x.shape[0], x.shape[1], x.shape[2], w.ctypes.data, b.ctypes.data, int(G), eps, int(gelu))
    return x


def tap(name):
    L = lib()
    n = L.orc_tap_numel(name.encode())
    if n < 0:
        raise KeyError(name)
    shape = (ctypes.c_int64 * 8)()
    nd = L.orc_tap_shape(name.encode(), shape)
    out = np.zeros(n, np.float32)
    L.orc_tap_copy(name.encode(), out.ctypes.data)
    return out.reshape([shape[i] for i in range(nd)])


def geometry(seg):
    v = (ctypes.c_int64 * 10)()
    lib().orc_geometry(seg, v)
    return dict(le=v[0], pad=v[1], pad_end=v[2], padded=v[3], nfr=v[4], Lt=[v[5 + i] for i in range(5)])


def stft(wave):
    wave = _f32(wave)
    n = wave.shape[1]
    nfr = n // 1024 + 1
    spec = np.zeros((2, 2049, nfr, 2), np.float32)
    got = lib().orc_stft(wave.ctypes.data, n, spec.ctypes.data)
    assert got == nfr
    return spec[..., 0] + 1j * spec[..., 1]


def istft(spec, n):
    nfr = spec.shape[2]
    s = np.zeros((2, 2049, nfr, 2), np.float32)
    s[..., 0], s[..., 1] = spec.real, spec.imag
    wave = np.zeros((2, n), np.float32)
    lib().orc_istft(s.ctypes.data, nfr, wave.ctypes.data, n)
    return wave


def layer_norm(x, w, b, eps=1e-5):
    x, w, b = _f32(x), _f32(w), _f32(b)
    y = np.zeros_like(x)
    lib().orc_layer_norm(x.ctypes.data, x.shape[0], x.shape[1], w.ctypes.data, b.ctypes.data, eps, y.ctypes.data)
    return y


def group_norm1(x, w, b, eps=1e-5, gelu=False):
    x = _f32(x).copy()
    w, b = _f32(w), _f32(b)
    lib().orc_group_norm1(x.ctypes.data, x.shape[0], x.shape[1], x.shape[2], w.ctypes.data, b.ctypes.data, eps, int(gelu))
    return x


def conv2d(x, w, b, stride=(1, 1), pad=(0, 0), dil=(1, 1), gelu=False):
    x, w, b = _f32(x), _f32(w), _f32(b)
    hw = (ctypes.c_int64 * 2)()
    args = [x.ctypes.data, x.shape[0], x.shape[1], x.shape[2], w.ctypes.data, w.shape[0], w.shape[2], w.shape[3],
            b.ctypes.data, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], int(gelu)]
    lib().orc_conv2d(*args, None, hw)
    y = np.zeros((w.shape[0], hw[0], hw[1]), np.float32)
    lib().orc_conv2d(*args, y.ctypes.data, hw)
    return y


def conv_tr_h(x, w, b, K=8, s=4, gelu=False):
    x, w, b = _f32(x), _f32(w), _f32(b)
    y = np.zeros((w.shape[1], (x.shape[1] - 1) * s + K, x.shape[2]), np.float32)
    lib().orc_conv_tr_h(x.ctypes.data, x.shape[0], x.shape[1], x.shape[2], w.ctypes.data, w.shape[1], K, s,
                        b.ctypes.data, int(gelu), y.ctypes.data)
    return y


def sin_embedding_2d(C, H, W):
    out = np.zeros((C, H, W), np.float32)
    lib().orc_sin_embedding_2d(C, H, W, out.ctypes.data)
    return out


def sin_embedding_1d(L, C):
    out = np.zeros((L, C), np.float32)
    lib().orc_sin_embedding_1d(L, C, out.ctypes.data)
    return out

"""Shared helpers of the GPU parity tests: run the HIP path (through the C ABI) and the
CPU oracle on the same input and compare every debug tap in the oracle's layout."""
import numpy as np

import oracle_lib as orc

TAPS = ["x_cac", "x_0", "xt_0", "x_1", "xt_1", "x_2", "xt_2", "x_3", "xt_3", "ct_x", "ct_xt",
        "dec_0", "tdec_0", "dec_1", "tdec_1", "dec_2", "tdec_2", "dec_3", "tdec_3"]


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def sdr_db(ref, est):
    num = float((np.asarray(ref, np.float64) ** 2).sum())
    den = float(((np.asarray(ref, np.float64) - np.asarray(est, np.float64)) ** 2).sum())
    return 10 * np.log10(max(num, 1e-300) / max(den, 1e-300))


# LOCAL parity metrics: the global max-abs / max-abs figure lets a stem, a channel or a quiet
# passage 40 dB below the loudest one be entirely wrong. These look at every stem and every block on its own scale.
LOCAL_TOL = 1e-3       # error of a block / channel relative to ITS OWN max-abs ...
LOCAL_FLOOR = 1e-2     # ... floored at 1 % (-40 dB) of the containing stem's / tensor's max-abs
MIN_STEM_SDR_DB = 60.0


def local_errors(got, ref, block=4096):
    """got, ref: (S, 2, n) stems. Returns (worst blockwise relative error, worst per-stem SDR in dB)."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    worst, worst_sdr = 0.0, np.inf
    for s in range(ref.shape[0]):
        smax = max(np.abs(ref[s]).max(), 1e-30)
        worst_sdr = min(worst_sdr, sdr_db(ref[s], got[s]))
        n = ref.shape[-1]
        nb = (n + block - 1) // block
        pad = nb * block - n
        r = np.pad(ref[s], ((0, 0), (0, pad))).reshape(ref.shape[1], nb, block)
        d = np.pad(got[s] - ref[s], ((0, 0), (0, pad))).reshape(ref.shape[1], nb, block)
        scale = np.maximum(np.abs(r).max(axis=-1), LOCAL_FLOOR * smax)
        worst = max(worst, float((np.abs(d).max(axis=-1) / scale).max()))
    return worst, float(worst_sdr)


def assert_local_parity(got, ref, block=4096, tol=LOCAL_TOL, what="output"):
    worst, wsdr = local_errors(got, ref, block)
    assert worst < tol, f"{what}: blockwise relative error {worst:.3e} >= {tol:g} (block {block}, floor -40 dB per stem)"
    assert wsdr > MIN_STEM_SDR_DB, f"{what}: worst per-stem SDR {wsdr:.1f} dB <= {MIN_STEM_SDR_DB} dB"


def channel_relerr(g, r):
    """Tap tensors in the oracle's layout (channel first): worst per-channel error relative to that channel's own
    max-abs, floored at -40 dB of the tensor's max-abs."""
    g = np.asarray(g, np.float64).reshape(g.shape[0], -1)
    r = np.asarray(r, np.float64).reshape(r.shape[0], -1)
    scale = np.maximum(np.abs(r).max(axis=1), LOCAL_FLOOR * max(np.abs(r).max(), 1e-30))
    return float((np.abs(g - r).max(axis=1) / scale).max())


def gpu_tap_as_oracle(ctx, name, b=0):
    """Returns the GPU tap `name` of batch element b re-laid-out like the oracle's tap."""
    a = ctx.tap(name)
    if a is None:
        return None
    a = a[b]
    if name == "x_cac" or name.startswith("x_") or name.startswith("dec_"):
        return a.transpose(2, 1, 0)  # [T][F][C] -> (C,F,T)
    if name.startswith("xt_") or name.startswith("tdec_"):
        return a.T[None]  # [L][C] -> (1,C,L)
    if name == "ct_x":
        tok, D = a.shape
        return a.reshape(tok // 8, 8, D).transpose(2, 1, 0)  # tokens t*8+f -> (C,8,T)
    if name == "ct_xt":
        return a.T
    raise KeyError(name)


def oracle_tap(name):
    """Oracle tensor matching what the GPU tap holds (decoder taps include the fused skip add)."""
    if name.startswith("dec_") and name != "dec_3":
        k = int(name[-1])
        return orc.tap(name) + orc.tap(f"x_{2 - k}")
    if name.startswith("tdec_") and name != "tdec_3":
        k = int(name[-1])
        return orc.tap(name) + orc.tap(f"xt_{2 - k}")
    return orc.tap(name)


# Demucs v3 (hdemucs_mmi): product tap -> (oracle tap, transpose of the product's [rows][C] / [T][F][C] image)
V3_TAPS = (["x_cac"] + [f"{p}_{i}" for i in range(4) for p in ("x", "xt")] +
           ["xt_4", "e4_lstm0", "e4_attn0", "e4_lstm1", "e4_attn1", "x_4", "e5_lstm0", "e5_attn0", "e5_lstm1", "e5_attn1", "x_5",
            "d1_in", "dec_in", "tdec_in"] + [f"{p}_{k}" for k in range(4) for p in ("dec", "tdec")])


def gpu_tap_as_oracle_v3(ctx, name, b=0):
    a = ctx.tap(name)
    if a is None:
        return None
    a = a[b]
    return a.transpose(2, 1, 0) if a.ndim == 3 else a.T  # [T][F][C] -> (C,F,T); [rows][C] -> (C,rows)


def oracle_tap_v3(name):
    """Oracle tensor matching what the v3 product tap holds (decoder taps include the fused skip add)."""
    if name in ("dec_0", "dec_1", "dec_2") or name in ("tdec_0", "tdec_1", "tdec_2"):
        return np.squeeze(oracle_tap(name))
    return np.squeeze(orc.tap(name))


LAST_LOCAL = {}  # name -> local (per-channel / blockwise) error of the last compare_segment call


def compare_segment(ctx, omodel, mix, b=0, taps=True, local=True):
    """mix (2, seg). Runs oracle + GPU; returns (errors dict, gpu_out, oracle_out). errs holds the global
    max-abs / max-abs figures; with local=True the per-channel (taps) and blockwise + per-stem-SDR (output) metrics
    are ASSERTED here as well (LOCAL_TOL, MIN_STEM_SDR_DB) and kept in LAST_LOCAL."""
    ref = omodel.segment(mix, taps=taps)
    out = ctx.segment(mix)
    errs = {}
    LAST_LOCAL.clear()
    if taps and getattr(omodel, "arch", 4) == 3:
        for name in V3_TAPS:
            g = gpu_tap_as_oracle_v3(ctx, name, b)
            if g is None:
                errs[name] = float("nan")
                continue
            r = oracle_tap_v3(name)
            g = np.squeeze(g)
            errs[name] = relerr(g, r) if g.shape == r.shape else float("nan")
            if g.shape == r.shape:
                LAST_LOCAL[name] = channel_relerr(g, r)
    elif taps:
        for name in TAPS:
            g = gpu_tap_as_oracle(ctx, name, b)
            if g is None:
                continue
            r = oracle_tap(name)
            errs[name] = relerr(g, r) if g.shape == r.shape else float("nan")
            if g.shape == r.shape and name != "x_cac":  # (x_cac: 4 CaC planes, channel axis = re/im of one spectrum)
                LAST_LOCAL[name] = channel_relerr(np.squeeze(g), np.squeeze(r))
    errs["out"] = relerr(out, ref)
    if local:
        bad = {k: v for k, v in LAST_LOCAL.items() if not (v < LOCAL_TOL)}
        assert not bad, f"per-channel relative error of taps above {LOCAL_TOL:g}: {bad}"
        assert_local_parity(out, ref, what="segment output")
        LAST_LOCAL["out_block"], LAST_LOCAL["out_min_stem_sdr_db"] = local_errors(out, ref)
    return errs, out, ref


def fp64_segment_forward(w, n_sources, mix):
    """The fp64 torch model of tests/golden/make_golden.py on (weights dict, mix (2, seg)). That script sets torch's
    DEFAULT dtype to float64 when imported (and relies on it): set it for the call only and restore what was there -
    a float64 default leaking into the other tests turns their `torch.zeros(..., device="cuda")` buffers into doubles."""
    import os
    import sys
    import torch
    prev = torch.get_default_dtype()
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if here not in sys.path:
        sys.path.insert(0, here)
    try:
        import make_golden as mg
        torch.set_default_dtype(torch.float64)
        return mg.segment_forward(w, n_sources, mix, {})
    finally:
        torch.set_default_dtype(prev)

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


def compare_segment(ctx, omodel, mix, b=0, taps=True):
    """mix (2, seg). Runs oracle + GPU; returns (errors dict, gpu_out, oracle_out)."""
    ref = omodel.segment(mix, taps=taps)
    out = ctx.segment(mix)
    errs = {}
    if taps and getattr(omodel, "arch", 4) == 3:
        for name in V3_TAPS:
            g = gpu_tap_as_oracle_v3(ctx, name, b)
            if g is None:
                errs[name] = float("nan")
                continue
            r = oracle_tap_v3(name)
            g = np.squeeze(g)
            errs[name] = relerr(g, r) if g.shape == r.shape else float("nan")
    elif taps:
        for name in TAPS:
            g = gpu_tap_as_oracle(ctx, name, b)
            if g is None:
                continue
            r = oracle_tap(name)
            errs[name] = relerr(g, r) if g.shape == r.shape else float("nan")
    errs["out"] = relerr(out, ref)
    return errs, out, ref

#!/usr/bin/env python
"""Golden vectors for Demucs v3 (hdemucs_mmi), generated in the BUILD container only.

An INDEPENDENT fp64 model of the v3 segment graph written with stock torch modules and
functional ops (torch.nn.LSTM, F.conv1d / F.conv2d / F.conv_transpose*, torch.einsum for the
LocalState attention exactly as facebookresearch/demucs' `LocalState.forward` writes it), i.e. a
different code path from the C++ oracle (own LSTM loop, own SGEMM, explicit softmax loops) and from
the HIP product. It models the REFERENCE's semantics, /root/reference:
  graph            src/model_inference.cpp:477-856
  levels 4 / 5     src/encdec.cpp:539-623, DConv with LSTM + LocalState src/layers.cpp:877-1113
  LSTM             src/lstm.cpp:68-147 == torch.nn.LSTM(bidirectional, 2 layers), run over the WHOLE sequence
                   (the reference does not reproduce PyTorch-demucs' max_steps=200 framing, .github/SDR_scores.md:67)
  LocalState       src/layers.cpp:533-721
  decoders         src/encdec.cpp:625-863
  Q3               GroupNorm with UNBIASED variance (src/layers.hpp:76-95,125-168), also for the 4-group norms
The reference's v3 tests only print (test/test_layers_v3.cpp has no EXPECT_), so these fp64 results pin
the oracle; "parity unpinned" against Eigen stands as for v4.

Outputs: golden_seg_v3.npz (reduced-size segment, taps subsampled), golden_prims_v3.npz.
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
from demucs_cpp_amd.weights import synth_weights  # noqa: E402
import make_golden as v4  # noqa: E402  (front end, back end, v4 encoder layers: shared with the v4 golden model)

torch.set_default_dtype(torch.float64)
W = v4.W


def gn_groups(x, weight, bias, G, eps=1e-5):
    """GroupNorm(G) on (1, C, *) with UNBIASED variance (Q3)."""
    C = x.shape[1]
    rest = x.shape[2:]
    xg = x.reshape(1, G, -1)
    mean = xg.mean(dim=2, keepdim=True)
    var = xg.var(dim=2, unbiased=True, keepdim=True)
    y = ((xg - mean) / torch.sqrt(var + eps)).reshape(1, C, *rest)
    shape = (1, C) + (1,) * len(rest)
    return y * weight.view(shape) + bias.view(shape)


def blstm(w, prefix, x):
    """x (1, H, T) -> (1, H, T): BLSTM.forward of demucs without framing: lstm, linear, + skip handled by caller."""
    H = x.shape[1]
    lstm = torch.nn.LSTM(input_size=H, hidden_size=H, num_layers=2, bidirectional=True).double()
    sd = {}
    for layer in range(2):
        for sfx in ("", "_reverse"):
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                key = f"{nm}_l{layer}{sfx}"
                sd[key] = W(w, prefix + "lstm." + key)
    lstm.load_state_dict(sd)
    with torch.no_grad():
        y, _ = lstm(x.permute(2, 0, 1))  # (T, 1, 2H)
        y = F.linear(y, W(w, prefix + "linear.weight"), W(w, prefix + "linear.bias"))  # (T, 1, H)
    return y.permute(1, 2, 0)


def local_state(w, prefix, x):
    """demucs.demucs.LocalState.forward (heads 4, nfreqs 0, ndecay 4). x (1, C, T)."""
    B, C, T = x.shape
    heads, ndecay = 4, 4
    conv = lambda nm: F.conv1d(x, W(w, prefix + nm + ".weight").reshape(-1, C, 1), W(w, prefix + nm + ".bias"))
    indexes = torch.arange(T, dtype=x.dtype)
    delta = indexes[:, None] - indexes[None, :]
    queries = conv("query").view(B, heads, -1, T)
    keys = conv("key").view(B, heads, -1, T)
    dots = torch.einsum("bhct,bhcs->bhts", keys, queries)
    dots = dots / keys.shape[2] ** 0.5
    decays = torch.arange(1, ndecay + 1, dtype=x.dtype)
    decay_q = conv("query_decay").view(B, heads, -1, T)
    decay_q = torch.sigmoid(decay_q) / 2
    decay_kernel = -decays.view(-1, 1, 1) * delta.abs() / ndecay ** 0.5
    dots = dots + torch.einsum("fts,bhfs->bhts", decay_kernel, decay_q)
    dots = dots.masked_fill(torch.eye(T, dtype=torch.bool), -100)
    weights = torch.softmax(dots, dim=2)
    content = conv("content").view(B, heads, -1, T)
    result = torch.einsum("bhts,bhct->bhcs", weights, content).reshape(B, -1, T)
    return x + F.conv1d(result, W(w, prefix + "proj.weight").reshape(C, C, 1), W(w, prefix + "proj.bias"))


def dconv_lstm(w, prefix, x, taps=None, tname=None):
    """x (1, C, T). src/layers.cpp:877-1113."""
    for j, d in ((0, 1), (1, 2)):
        p = f"{prefix}.dconv.layers.{j}."
        h = F.conv1d(x, W(w, p + "0.weight"), W(w, p + "0.bias"), padding=d, dilation=d)
        h = F.gelu(v4.gn1(h, W(w, p + "1.weight"), W(w, p + "1.bias")))
        h = blstm(w, p + "3.", h) + h
        if taps is not None:
            taps[f"{tname}_lstm{j}"] = h[0].numpy().copy()
        h = local_state(w, p + "4.", h)
        if taps is not None:
            taps[f"{tname}_attn{j}"] = h[0].numpy().copy()
        u = F.conv1d(h, W(w, p + "5.weight").unsqueeze(-1), W(w, p + "5.bias"))
        u = v4.gn1(u, W(w, p + "6.weight"), W(w, p + "6.bias"))
        u = F.glu(u, dim=1)
        x = x + u * W(w, p + "8.scale").view(1, -1, 1)
    return x


def segment_forward_v3(w, mix_np, taps=None):
    """mix (2, seg) -> (4, 2, seg). src/model_inference.cpp:477-856."""
    mix = torch.from_numpy(mix_np.astype(np.float64))
    seg = mix.shape[1]
    HOP, NFFT = v4.HOP, v4.NFFT
    le = math.ceil(seg / HOP)
    pad = HOP // 2 * 3
    pad_end = pad + le * HOP - seg
    padded = torch.from_numpy(np.pad(mix.numpy(), ((0, 0), (pad, pad_end)), mode="symmetric"))
    frames = padded.unfold(1, NFFT, HOP)
    z = torch.fft.rfft(frames * v4.hann(), dim=-1) / math.sqrt(NFFT)
    z = z[..., :2048].permute(0, 2, 1)
    x = torch.stack([z[0].real, z[0].imag, z[1].real, z[1].imag], dim=0)
    mean, std = x.mean(), x.std(unbiased=True)
    x = ((x - mean) / (std + 1e-5)).unsqueeze(0)
    xt = mix.clone()
    meant, stdt = xt.mean(), xt.std(unbiased=True)
    xt = ((xt - meant) / (stdt + 1e-5)).unsqueeze(0)
    saved, savedt, lens = [], [], [seg]
    for i in range(4):
        xt = v4.time_encoder(w, i, xt)
        x = v4.freq_encoder(w, i, x)
        if i == 0:
            emb = W(w, "freq_emb.embedding.weight").t() * (10.0 * 0.2)
            x = x + emb.view(1, 48, 512, 1)
        saved.append(x)
        savedt.append(xt)
        lens.append(xt.shape[-1])
        if taps is not None:
            taps[f"x_{i}"] = x[0].numpy().copy()
            taps[f"xt_{i}"] = xt.numpy().copy()
    # tencoder.4 (bare conv) -> injected into encoder.4
    xt4 = v4.strided_conv1d(xt, W(w, "tencoder.4.conv.weight"), W(w, "tencoder.4.conv.bias"))
    assert xt4.shape[-1] == le
    # encoder.4: Conv2d (8,1)/(4,1) no padding: F 8 -> 1
    y = F.conv2d(x, W(w, "encoder.4.conv.weight").unsqueeze(-1), W(w, "encoder.4.conv.bias"), stride=(4, 1))
    assert y.shape[2] == 1
    y = y[:, :, 0, :] + xt4
    y = F.gelu(gn_groups(y, W(w, "encoder.4.norm1.weight"), W(w, "encoder.4.norm1.bias"), 4))
    y = dconv_lstm(w, "encoder.4", y, taps, "e4")
    y = F.conv1d(y, W(w, "encoder.4.rewrite.weight").unsqueeze(-1), W(w, "encoder.4.rewrite.bias"))
    y = gn_groups(y, W(w, "encoder.4.norm2.weight"), W(w, "encoder.4.norm2.bias"), 4)
    x4 = F.glu(y, dim=1)  # (1, 768, T)
    # encoder.5: Conv1d k4 s2 p1 (ceil form = one extra right zero for odd T)
    T = x4.shape[-1]
    lo = math.ceil((T + 2 - 3 - 1) / 2) + 1
    need = (lo - 1) * 2 + 4 - (T + 2)
    y = F.conv1d(F.pad(x4, (1, 1 + max(need, 0))), W(w, "encoder.5.conv.weight"), W(w, "encoder.5.conv.bias"), stride=2)
    assert y.shape[-1] == lo
    y = F.gelu(gn_groups(y, W(w, "encoder.5.norm1.weight"), W(w, "encoder.5.norm1.bias"), 4))
    y = dconv_lstm(w, "encoder.5", y, taps, "e5")
    y = F.conv1d(y, W(w, "encoder.5.rewrite.weight").unsqueeze(-1), W(w, "encoder.5.rewrite.bias"))
    y = gn_groups(y, W(w, "encoder.5.norm2.weight"), W(w, "encoder.5.norm2.bias"), 4)
    x5 = F.glu(y, dim=1)  # (1, 1536, T5)
    if taps is not None:
        taps["xt_4"], taps["x_4"], taps["x_5"] = xt4.numpy().copy(), x4.numpy().copy(), x5.numpy().copy()
    # decoder.0
    y = F.conv1d(x5, W(w, "decoder.0.rewrite.weight"), W(w, "decoder.0.rewrite.bias"), padding=1)
    y = F.glu(gn_groups(y, W(w, "decoder.0.norm1.weight"), W(w, "decoder.0.norm1.bias"), 4), dim=1)
    y = F.conv_transpose1d(y, W(w, "decoder.0.conv_tr.weight"), W(w, "decoder.0.conv_tr.bias"), stride=2)
    y = F.gelu(gn_groups(y, W(w, "decoder.0.norm2.weight"), W(w, "decoder.0.norm2.bias"), 4))
    d0 = y[..., 1:1 + T]
    # decoder.1
    y = (d0 + x4).unsqueeze(2)  # (1, 768, 1, T)
    y = F.conv2d(y, W(w, "decoder.1.rewrite.weight"), W(w, "decoder.1.rewrite.bias"), padding=1)
    y = F.glu(gn_groups(y, W(w, "decoder.1.norm1.weight"), W(w, "decoder.1.norm1.bias"), 4), dim=1)
    pre = y
    y = F.conv_transpose2d(y, W(w, "decoder.1.conv_tr.weight").unsqueeze(-1), W(w, "decoder.1.conv_tr.bias"), stride=(4, 1))
    d1 = F.gelu(gn_groups(y, W(w, "decoder.1.norm2.weight"), W(w, "decoder.1.norm2.bias"), 4))  # (1, 384, 8, T)
    assert d1.shape[2] == 8
    # tdecoder.0
    y = F.conv_transpose1d(pre[:, :, 0, :], W(w, "tdecoder.0.conv_tr.weight"), W(w, "tdecoder.0.conv_tr.bias"), stride=4)
    y = F.gelu(gn_groups(y, W(w, "tdecoder.0.norm2.weight"), W(w, "tdecoder.0.norm2.bias"), 4))
    td0 = y[..., 2:2 + lens[4]]
    if taps is not None:
        taps["d0"], taps["d1"], taps["td0"] = d0.numpy().copy(), d1[0].numpy().copy(), td0.numpy().copy()
    x, xt = d1, td0
    for k in range(4):
        p, pt = f"decoder.{k + 2}", f"tdecoder.{k + 1}"
        y = F.glu(F.conv2d(x + saved[3 - k], W(w, p + ".rewrite.weight"), W(w, p + ".rewrite.bias"), padding=1), dim=1)
        y = F.conv_transpose2d(y, W(w, p + ".conv_tr.weight").unsqueeze(-1), W(w, p + ".conv_tr.bias"), stride=(4, 1))
        x = (F.gelu(y) if k < 3 else y)[:, :, 2:-2, :]
        y = F.glu(F.conv1d(xt + savedt[3 - k], W(w, pt + ".rewrite.weight"), W(w, pt + ".rewrite.bias"), padding=1), dim=1)
        y = F.conv_transpose1d(y, W(w, pt + ".conv_tr.weight"), W(w, pt + ".conv_tr.bias"), stride=4)
        xt = (F.gelu(y) if k < 3 else y)[..., 2:2 + lens[3 - k]]
        if taps is not None:
            taps[f"dec_{k}"] = x[0].numpy().copy()
            taps[f"tdec_{k}"] = xt.numpy().copy()
    S = 4
    x = x[0] * std + mean
    xt = xt[0] * stdt + meant
    nfr = le + 4
    win = v4.hann()
    wss = torch.zeros(NFFT + HOP * (nfr - 1))
    for f in range(nfr):
        wss[f * HOP:f * HOP + NFFT] += win * win
    out = torch.zeros(S, 2, seg)
    for s in range(S):
        for ch in range(2):
            spec = torch.zeros(2049, nfr, dtype=torch.complex128)
            spec[:2048, 2:2 + le] = torch.complex(x[s * 4 + 2 * ch], x[s * 4 + 2 * ch + 1])
            y = torch.fft.irfft(spec.t() * math.sqrt(NFFT), n=NFFT, dim=-1, norm="forward")
            acc = torch.zeros(NFFT + HOP * (nfr - 1))
            for f in range(nfr):
                acc[f * HOP:f * HOP + NFFT] += y[f] * win / NFFT / (wss[f * HOP:f * HOP + NFFT] + 1e-8)
            wave = acc[NFFT // 2:NFFT // 2 + (le + 3) * HOP]
            out[s, ch] = wave[pad:pad + seg] + xt[s * 2 + ch]
    return out.numpy()


def make_segment(seed, seg, fname):
    w = synth_weights(4, seed, "default", "v3")
    rng = np.random.default_rng(3000 + seed)
    mix = (0.1 * rng.standard_normal((2, seg))).astype(np.float32)
    taps = {}
    out = segment_forward_v3(w, mix, taps)
    save = dict(weight_seed=seed, seg=seg, mix=mix, out=out.astype(np.float32), out_absmax=float(np.abs(out).max()))
    for k, v in taps.items():
        idx, vals = v4.subsample(v)
        save[f"tap_{k}_shape"] = np.array(v.shape, dtype=np.int64)
        save[f"tap_{k}_idx"] = idx
        save[f"tap_{k}_val"] = vals.astype(np.float32)
        save[f"tap_{k}_absmax"] = float(np.abs(v).max())
    np.savez_compressed(os.path.join(HERE, fname), **save)
    print(fname, "out absmax", np.abs(out).max(), "std", out.std(), {k: float(np.abs(v).max()) for k, v in taps.items()})


def make_prims(seed):
    """Primitive vectors on the level-4 tensors of the SAME synthetic weight file (regenerated in the test)."""
    w = synth_weights(4, seed, "default", "v3")
    rng = np.random.default_rng(11)
    save = dict(weight_seed=seed)
    x = torch.from_numpy(rng.standard_normal((1, 192, 23)))
    save["lstm_x"] = x[0].numpy()
    with torch.no_grad():
        H = 192
        lstm = torch.nn.LSTM(input_size=H, hidden_size=H, num_layers=2, bidirectional=True).double()
        p = "encoder.4.dconv.layers.1.3."
        lstm.load_state_dict({f"{nm}_l{l}{s}": W(w, p + f"lstm.{nm}_l{l}{s}") for l in range(2) for s in ("", "_reverse")
                              for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")})
        y, _ = lstm(x.permute(2, 0, 1))
    save["lstm_y"] = y[:, 0, :].numpy()  # (T, 2H)
    save["attn_y"] = local_state(w, "encoder.4.dconv.layers.1.4.", x)[0].numpy()
    xg = torch.from_numpy(rng.standard_normal((1, 24, 5, 7)) * 2 + 0.7)
    wg = torch.from_numpy(1 + 0.1 * rng.standard_normal(24))
    bg = torch.from_numpy(0.1 * rng.standard_normal(24))
    save["gn4_x"], save["gn4_w"], save["gn4_b"] = xg[0].numpy(), wg.numpy(), bg.numpy()
    save["gn4_y"] = gn_groups(xg, wg, bg, 4)[0].numpy()
    save["gn4_y_gelu"] = F.gelu(gn_groups(xg, wg, bg, 4))[0].numpy()
    # transposed conv k4 s2 (decoder.0)
    xc = torch.from_numpy(rng.standard_normal((1, 5, 9)))
    wt = torch.from_numpy(rng.standard_normal((5, 3, 4)) * 0.3)
    bt = torch.from_numpy(rng.standard_normal(3) * 0.1)
    save["ct_x"], save["ct_w"], save["ct_b"] = xc[0].numpy(), wt.numpy(), bt.numpy()
    save["ct_y"] = F.conv_transpose1d(xc, wt, bt, stride=2)[0].numpy()
    np.savez_compressed(os.path.join(HERE, "golden_prims_v3.npz"), **save)


if __name__ == "__main__":
    torch.manual_seed(0)
    make_prims(5)
    make_segment(5, 20000, "golden_seg_v3.npz")

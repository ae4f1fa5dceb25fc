#!/usr/bin/env python
"""Generates the committed golden vectors under tests/golden/ (run in the BUILD
container only; the vectors travel, this script's inputs do not need to).

It is an INDEPENDENT fp64 model of the HTDemucs segment graph written with
torch.nn.functional ops (F.conv1d / F.conv2d / F.conv_transpose2d / torch.fft), i.e. a
different code path from both the C++ oracle (oracle/demucs_oracle.cpp: im2col + own
SGEMM + own FFT) and the HIP product. The reference's semantics that differ from stock
PyTorch are modelled explicitly and cite /root/reference:
  Q2  symmetric (edge-duplicating) padding          src/model_inference.cpp:35-45
  Q3  UNBIASED variance in LayerNorm / GroupNorm    src/layers.hpp:76-95
  Q5  ceil-form strided conv = right zero pad       src/conv.hpp:25-34
  Q6  "(t f)" token order                           src/crosstransformer.cpp:233-235
  self layers use norm2 as the FFN norm             src/crosstransformer.cpp:111-113
  norm_out = GroupNorm(1 group over all (C,T))      src/layers.cpp:517-530
The reference has no recorded outputs for this path (its layer tests only print), so
these fp64 results are what pins the oracle; see DESIGN.md §3.

Outputs (small, committed):
  golden_seg_4s.npz / golden_seg_6s.npz  reduced-size segment (seg=10000 / 6000),
       synthetic weights regenerated from (n_sources, seed) by demucs_cpp_amd.weights
  golden_prims.npz                        primitive known-answer vectors
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from demucs_cpp_amd.weights import synth_weights  # noqa: E402

torch.set_default_dtype(torch.float64)
NFFT, HOP = 4096, 1024


def W(w, name):
    return torch.from_numpy(w[name].astype(np.float64))


def gn1(x, weight, bias, eps=1e-5):
    """GroupNorm(1 group) over all but dim0 with UNBIASED variance (Q3). x (B,C,L)."""
    B = x.shape[0]
    flat = x.reshape(B, -1)
    mean = flat.mean(dim=1).view(B, 1, 1)
    var = flat.var(dim=1, unbiased=True).view(B, 1, 1)
    return (x - mean) / torch.sqrt(var + eps) * weight.view(1, -1, 1) + bias.view(1, -1, 1)


def ln(x, weight, bias, eps=1e-5):
    """LayerNorm over last dim with UNBIASED variance (Q3). x (..., C)."""
    mean = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, unbiased=True, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * weight + bias


def dconv(w, prefix, x):
    """x (B, C, L). src/layers.cpp:152-375."""
    for j, d in ((0, 1), (1, 2)):
        p = f"{prefix}.dconv.layers.{j}."
        h = F.conv1d(x, W(w, p + "0.weight"), W(w, p + "0.bias"), padding=d, dilation=d)
        h = F.gelu(gn1(h, W(w, p + "1.weight"), W(w, p + "1.bias")))
        u = F.conv1d(h, W(w, p + "3.weight").unsqueeze(-1), W(w, p + "3.bias"))
        u = gn1(u, W(w, p + "4.weight"), W(w, p + "4.bias"))
        u = F.glu(u, dim=1)
        x = x + u * W(w, p + "6.scale").view(1, -1, 1)
    return x


def strided_conv1d(x, weight, bias):
    """k8 s4 p2 with the reference's ceil-form length (Q5) = explicit right zero pad."""
    L = x.shape[-1]
    lo = math.ceil((L + 4 - 7 - 1) / 4) + 1
    need = (lo - 1) * 4 + 8 - (L + 4)  # extra right pad so that `lo` outputs exist
    x = F.pad(x, (2, 2 + max(need, 0)))
    y = F.conv1d(x, weight, bias, stride=4)
    assert y.shape[-1] == lo
    return y


def freq_encoder(w, i, x):
    """x (1, Cin, F, T). src/encdec.cpp:8-80."""
    p = f"encoder.{i}"
    y = F.conv2d(x, W(w, p + ".conv.weight").unsqueeze(-1), W(w, p + ".conv.bias"), stride=(4, 1), padding=(2, 0))
    y = F.gelu(y)
    B, C, Fr, T = y.shape
    yb = y.permute(0, 2, 1, 3).reshape(Fr, C, T)  # freq rows are the batch
    yb = dconv(w, p, yb)
    y = yb.reshape(1, Fr, C, T).permute(0, 2, 1, 3)
    y = F.conv2d(y, W(w, p + ".rewrite.weight").view(2 * C, C, 1, 1), W(w, p + ".rewrite.bias"))
    return F.glu(y, dim=1)


def time_encoder(w, i, x):
    """x (1, Cin, L). src/encdec.cpp:82-164."""
    p = f"tencoder.{i}"
    y = F.gelu(strided_conv1d(x, W(w, p + ".conv.weight"), W(w, p + ".conv.bias")))
    y = dconv(w, p, y)
    C = y.shape[1]
    y = F.conv1d(y, W(w, p + ".rewrite.weight").view(2 * C, C, 1), W(w, p + ".rewrite.bias"))
    return F.glu(y, dim=1)


def freq_decoder(w, k, x, skip):
    """src/encdec.cpp:166-256."""
    p = f"decoder.{k}"
    y = x + skip
    y = F.conv2d(y, W(w, p + ".rewrite.weight"), W(w, p + ".rewrite.bias"), padding=1)
    y = F.glu(y, dim=1)
    B, C, Fr, T = y.shape
    yb = dconv(w, p, y.permute(0, 2, 1, 3).reshape(Fr, C, T))
    y = yb.reshape(1, Fr, C, T).permute(0, 2, 1, 3)
    y = F.conv_transpose2d(y, W(w, p + ".conv_tr.weight").unsqueeze(-1), W(w, p + ".conv_tr.bias"), stride=(4, 1))
    if k < 3:
        y = F.gelu(y)
    return y[:, :, 2:-2, :]


def time_decoder(w, k, x, skip, out_len):
    """src/encdec.cpp:258-361."""
    p = f"tdecoder.{k}"
    y = F.conv1d(x + skip, W(w, p + ".rewrite.weight"), W(w, p + ".rewrite.bias"), padding=1)
    y = F.glu(y, dim=1)
    y = dconv(w, p, y)
    y = F.conv_transpose1d(y, W(w, p + ".conv_tr.weight"), W(w, p + ".conv_tr.bias"), stride=4)
    if k < 3:
        y = F.gelu(y)
    return y[..., 2:2 + out_len]


def sin_emb_2d(C, H, Wd, max_period=10000.0):
    """src/crosstransformer.cpp:7-53 (same as demucs' create_2d_sin_embedding)."""
    pe = torch.zeros(C, H, Wd)
    dm = C // 2
    div = torch.exp(torch.arange(0.0, dm, 2) * -(math.log(max_period) / dm))
    pos_w = torch.arange(0.0, Wd).unsqueeze(1)
    pos_h = torch.arange(0.0, H).unsqueeze(1)
    pe[0:dm:2] = torch.sin(pos_w * div).t().unsqueeze(1).repeat(1, H, 1)
    pe[1:dm:2] = torch.cos(pos_w * div).t().unsqueeze(1).repeat(1, H, 1)
    pe[dm::2] = torch.sin(pos_h * div).t().unsqueeze(2).repeat(1, 1, Wd)
    pe[dm + 1::2] = torch.cos(pos_h * div).t().unsqueeze(2).repeat(1, 1, Wd)
    return pe


def sin_emb_1d(L, C, max_period=10000.0):
    """src/crosstransformer.cpp:55-77."""
    half = C // 2
    pos = torch.arange(0.0, L).view(-1, 1)
    adim = torch.arange(0.0, half).view(1, -1)
    phase = pos / (max_period ** (adim / (half - 1)))
    return torch.cat([torch.cos(phase), torch.sin(phase)], dim=-1)


def encoder_layer(w, prefix, q, k, self_attn):
    """q (T, C), k (S, C). src/layers.cpp:377-531."""
    H = 8
    T, C = q.shape
    attn = prefix + (".self_attn" if self_attn else ".cross_attn")
    qn = ln(q, W(w, prefix + ".norm1.weight"), W(w, prefix + ".norm1.bias"))
    kn = qn if self_attn else ln(k, W(w, prefix + ".norm2.weight"), W(w, prefix + ".norm2.bias"))
    ipw, ipb = W(w, attn + ".in_proj_weight"), W(w, attn + ".in_proj_bias")
    Q = F.linear(qn, ipw[:C], ipb[:C])
    K = F.linear(kn, ipw[C:2 * C], ipb[C:2 * C])
    V = F.linear(kn, ipw[2 * C:], ipb[2 * C:])
    hs = C // H
    Qh = Q.view(T, H, hs).transpose(0, 1)
    Kh = K.view(-1, H, hs).transpose(0, 1)
    Vh = V.view(-1, H, hs).transpose(0, 1)
    att = torch.softmax(Qh @ Kh.transpose(1, 2) / math.sqrt(hs), dim=-1) @ Vh
    att = att.transpose(0, 1).reshape(T, C)
    q = q + F.linear(att, W(w, attn + ".out_proj.weight"), W(w, attn + ".out_proj.bias")) * W(w, prefix + ".gamma_1.scale")
    n3 = ".norm2" if self_attn else ".norm3"
    h = ln(q, W(w, prefix + n3 + ".weight"), W(w, prefix + n3 + ".bias"))
    h = F.gelu(F.linear(h, W(w, prefix + ".linear1.weight"), W(w, prefix + ".linear1.bias")))
    h = F.linear(h, W(w, prefix + ".linear2.weight"), W(w, prefix + ".linear2.bias"))
    q = q + h * W(w, prefix + ".gamma_2.scale")
    # norm_out: GroupNorm(1 group) over all (T, C), per-channel affine
    mean, var = q.mean(), q.var(unbiased=True)
    return (q - mean) / torch.sqrt(var + 1e-5) * W(w, prefix + ".norm_out.weight") + W(w, prefix + ".norm_out.bias")


def crosstransformer(w, x, xt):
    """x (C, Fr, T1), xt (C, T2). src/crosstransformer.cpp:205-339."""
    C, Fr, T1 = x.shape
    pe = sin_emb_2d(C, Fr, T1)
    xs = x.permute(2, 1, 0).reshape(T1 * Fr, C)  # "(t1 fr) c"
    pes = pe.permute(2, 1, 0).reshape(T1 * Fr, C)
    xs = ln(xs, W(w, "crosstransformer.norm_in.weight"), W(w, "crosstransformer.norm_in.bias")) + pes
    T2 = xt.shape[1]
    xts = ln(xt.t(), W(w, "crosstransformer.norm_in_t.weight"), W(w, "crosstransformer.norm_in_t.bias")) + sin_emb_1d(T2, C)
    for layer in range(5):
        pf, pt = f"crosstransformer.layers.{layer}", f"crosstransformer.layers_t.{layer}"
        if layer % 2 == 0:
            xs = encoder_layer(w, pf, xs, xs, True)
            xts = encoder_layer(w, pt, xts, xts, True)
        else:
            old = xs
            xs = encoder_layer(w, pf, xs, xts, False)
            xts = encoder_layer(w, pt, xts, old, False)
    return xs.reshape(T1, Fr, C).permute(2, 1, 0), xts.t()


def hann():
    return torch.hann_window(NFFT, periodic=True, dtype=torch.float64)


def segment_forward(w, n_sources, mix_np, taps=None):
    """mix (2, seg) -> (S, 2, seg). src/model_inference.cpp:48-475."""
    mix = torch.from_numpy(mix_np.astype(np.float64))
    seg = mix.shape[1]
    le = math.ceil(seg / HOP)
    pad = HOP // 2 * 3
    pad_end = pad + le * HOP - seg
    padded = torch.from_numpy(np.pad(mix.numpy(), ((0, 0), (pad, pad_end)), mode="symmetric"))  # Q2
    frames = padded.unfold(1, NFFT, HOP)  # (2, le, 4096): reference frames 2..le+1
    assert frames.shape[1] == le
    z = torch.fft.rfft(frames * hann(), dim=-1) / math.sqrt(NFFT)  # (2, le, 2049)
    z = z[..., :2048].permute(0, 2, 1)  # (2, 2048, le)
    x = torch.stack([z[0].real, z[0].imag, z[1].real, z[1].imag], dim=0)  # CaC (4, 2048, le)
    mean, std = x.mean(), x.std(unbiased=True)
    x = (x - mean) / (std + 1e-5)
    xt = mix.clone()
    meant, stdt = xt.mean(), xt.std(unbiased=True)
    xt = (xt - meant) / (stdt + 1e-5)
    x = x.unsqueeze(0)
    xt = xt.unsqueeze(0)
    saved, savedt, lens = [], [], [seg]
    for i in range(4):
        xt = time_encoder(w, i, xt)
        x = freq_encoder(w, i, x)
        if i == 0:
            emb = W(w, "freq_emb.embedding.weight").t() * (10.0 * 0.2)  # (48, 512)
            x = x + emb.view(1, 48, 512, 1)
        saved.append(x)
        savedt.append(xt)
        lens.append(xt.shape[-1])
        if taps is not None:
            taps[f"x_{i}"] = x[0].numpy().copy()
            taps[f"xt_{i}"] = xt.numpy().copy()
    x3, xt3 = x[0], xt[0]
    if n_sources == 4:
        x3 = F.conv1d(x3.reshape(1, 384, -1), W(w, "channel_upsampler.weight").unsqueeze(-1), W(w, "channel_upsampler.bias")).reshape(512, 8, -1)
        xt3 = F.conv1d(xt3.unsqueeze(0), W(w, "channel_upsampler_t.weight").unsqueeze(-1), W(w, "channel_upsampler_t.bias"))[0]
    x3, xt3 = crosstransformer(w, x3, xt3)
    if taps is not None:
        taps["ct_x"] = x3.numpy().copy()
        taps["ct_xt"] = xt3.numpy().copy()
    if n_sources == 4:
        x3 = F.conv1d(x3.reshape(1, 512, -1), W(w, "channel_downsampler.weight").unsqueeze(-1), W(w, "channel_downsampler.bias")).reshape(384, 8, -1)
        xt3 = F.conv1d(xt3.unsqueeze(0), W(w, "channel_downsampler_t.weight").unsqueeze(-1), W(w, "channel_downsampler_t.bias"))[0]
    x, xt = x3.unsqueeze(0), xt3.unsqueeze(0)
    for k in range(4):
        x = freq_decoder(w, k, x, saved[3 - k])
        xt = time_decoder(w, k, xt, savedt[3 - k], lens[3 - k])
        if taps is not None:
            taps[f"dec_{k}"] = x[0].numpy().copy()
            taps[f"tdec_{k}"] = xt.numpy().copy()
    S = n_sources
    x = x[0] * std + mean  # (4S, 2048, le)
    xt = xt[0] * stdt + meant  # (2S, seg)
    nfr = le + 4
    win = hann()
    wss = torch.zeros(NFFT + HOP * (nfr - 1))
    for f in range(nfr):
        wss[f * HOP:f * HOP + NFFT] += win * win
    out = torch.zeros(S, 2, seg)
    for s in range(S):
        for ch in range(2):
            spec = torch.zeros(2049, nfr, dtype=torch.complex128)
            spec[:2048, 2:2 + le] = torch.complex(x[s * 4 + 2 * ch], x[s * 4 + 2 * ch + 1])
            # Q1: X*sqrt(N), unscaled inverse, then *w/N/(wss+1e-8); src/dsp.cpp:151-185
            y = torch.fft.irfft(spec.t() * math.sqrt(NFFT), n=NFFT, dim=-1, norm="forward")  # (nfr, 4096)
            acc = torch.zeros(NFFT + HOP * (nfr - 1))
            for f in range(nfr):
                acc[f * HOP:f * HOP + NFFT] += y[f] * win / NFFT / (wss[f * HOP:f * HOP + NFFT] + 1e-8)
            wave = acc[NFFT // 2:NFFT // 2 + (le + 3) * HOP]
            out[s, ch] = wave[pad:pad + seg] + xt[s * 2 + ch]
    return out.numpy()


def subsample(a, n=4096):
    flat = np.asarray(a).reshape(-1)
    idx = np.linspace(0, flat.size - 1, num=min(n, flat.size)).astype(np.int64)
    return idx, flat[idx]


def make_segment(n_sources, seed, seg, fname):
    w = synth_weights(n_sources, seed)
    rng = np.random.default_rng(1000 + seed)
    mix = (0.1 * rng.standard_normal((2, seg))).astype(np.float32)
    taps = {}
    out = segment_forward(w, n_sources, mix, taps)
    save = dict(n_sources=n_sources, weight_seed=seed, seg=seg, mix=mix, out=out.astype(np.float32), out_absmax=float(np.abs(out).max()))
    for k, v in taps.items():
        idx, vals = subsample(v)
        save[f"tap_{k}_shape"] = np.array(v.shape, dtype=np.int64)
        save[f"tap_{k}_idx"] = idx
        save[f"tap_{k}_val"] = vals.astype(np.float32)
        save[f"tap_{k}_absmax"] = float(np.abs(v).max())
    np.savez_compressed(os.path.join(HERE, fname), **save)
    print(fname, "out absmax", np.abs(out).max(), "std", out.std())


def make_prims():
    rng = np.random.default_rng(7)
    save = {}
    # LayerNorm KAT inputs of /root/reference/test/test_layers.cpp:2161-2187, unbiased (Q3)
    x = torch.tensor([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]])
    wv, bv = torch.tensor([0.75, -0.5, -1.35]), torch.tensor([0.5, -0.25, 0.75])
    save["ln_kat_x"], save["ln_kat_w"], save["ln_kat_b"] = x.numpy(), wv.numpy(), bv.numpy()
    save["ln_kat_y"] = ln(x, wv, bv).numpy()
    save["ln_kat_y_pytorch_biased"] = F.layer_norm(x, (3,), wv, bv, 1e-5).numpy()
    # LayerNormBigger inputs of test_layers.cpp:2189-2234 (closed-form fill), 64 rows only
    xb = torch.ones(64, 512)
    xb[:, 0::2] = -1.0
    i = torch.arange(512.0)
    wb = torch.where(i % 2 == 0, -0.25 + i * 0.03, torch.full_like(i, 0.25))
    bb = torch.where(i % 2 == 0, torch.full_like(i, 0.5), -0.5 + i * 0.57)
    save["ln_big_y_row0"] = ln(xb, wb, bb)[0].numpy()
    # GemmConv KAT of test_layers.cpp:856-930: x (1,13,9) 0.5/-0.75 by column parity,
    # w (5,1,2,3) = -0.1*counter, bias 0, stride 1, no pad
    xc = torch.full((1, 1, 13, 9), -0.75)
    xc[..., 0::2] = 0.5
    wc = -(0.1 * torch.arange(1.0, 31.0)).view(5, 1, 2, 3)
    save["conv_kat_y"] = F.conv2d(xc, wc)[0].numpy()
    # random conv cases (strided k8 s4 p2 with ragged length; dilated k3; 3x3; transposed)
    xr = torch.from_numpy(rng.standard_normal((1, 6, 37)))
    wr = torch.from_numpy(rng.standard_normal((10, 6, 8)) * 0.2)
    br = torch.from_numpy(rng.standard_normal(10) * 0.1)
    save["c1_x"], save["c1_w"], save["c1_b"] = xr[0].numpy(), wr.numpy(), br.numpy()
    save["c1_y_gelu"] = F.gelu(strided_conv1d(xr, wr, br))[0].numpy()
    wd = torch.from_numpy(rng.standard_normal((4, 6, 3)) * 0.3)
    bd = torch.from_numpy(rng.standard_normal(4) * 0.1)
    save["c2_w"], save["c2_b"] = wd.numpy(), bd.numpy()
    save["c2_y_d2"] = F.conv1d(xr, wd, bd, padding=2, dilation=2)[0].numpy()
    x2 = torch.from_numpy(rng.standard_normal((1, 5, 7, 11)))
    w2 = torch.from_numpy(rng.standard_normal((8, 5, 3, 3)) * 0.2)
    b2 = torch.from_numpy(rng.standard_normal(8) * 0.1)
    save["c3_x"], save["c3_w"], save["c3_b"] = x2[0].numpy(), w2.numpy(), b2.numpy()
    save["c3_y"] = F.conv2d(x2, w2, b2, padding=1)[0].numpy()
    wt = torch.from_numpy(rng.standard_normal((5, 3, 8)) * 0.2)
    bt = torch.from_numpy(rng.standard_normal(3) * 0.1)
    save["c4_w"], save["c4_b"] = wt.numpy(), bt.numpy()
    save["c4_y"] = F.conv_transpose2d(x2, wt.unsqueeze(-1), bt, stride=(4, 1))[0].numpy()
    save["c4_y_gelu"] = F.gelu(F.conv_transpose2d(x2, wt.unsqueeze(-1), bt, stride=(4, 1)))[0].numpy()
    # GroupNorm(1)+GELU, unbiased
    xg = torch.from_numpy(rng.standard_normal((3, 6, 17)))
    wg = torch.from_numpy(1 + 0.1 * rng.standard_normal(6))
    bg = torch.from_numpy(0.1 * rng.standard_normal(6))
    save["gn_x"], save["gn_w"], save["gn_b"] = xg.numpy(), wg.numpy(), bg.numpy()
    save["gn_y"] = gn1(xg, wg, bg).numpy()
    save["gn_y_gelu"] = F.gelu(gn1(xg, wg, bg)).numpy()
    # STFT frames 2..le+1 of a random (2, 5*1024+3072) signal == torch.stft normalized
    sig = torch.from_numpy(rng.standard_normal((2, 8 * 1024)) * 0.3)
    fr = sig.unfold(1, NFFT, HOP)
    zz = torch.fft.rfft(fr * hann(), dim=-1) / math.sqrt(NFFT)  # (2, 5, 2049)
    save["stft_x"] = sig.numpy().astype(np.float32)
    save["stft_z_re"] = zz.real.permute(0, 2, 1).numpy()  # (2, 2049, 5) = reference frames 2..6
    save["stft_z_im"] = zz.imag.permute(0, 2, 1).numpy()
    # positional embeddings
    save["pe2d"] = sin_emb_2d(16, 3, 5).numpy()
    save["pe1d"] = sin_emb_1d(7, 12).numpy()
    np.savez_compressed(os.path.join(HERE, "golden_prims.npz"), **save)
    print("LN KAT row0:", save["ln_kat_y"][0], " (SURVEY: [-0.24999625, -0.25, -0.59999325])")


if __name__ == "__main__":
    torch.manual_seed(0)
    make_prims()
    make_segment(4, 0, 10000, "golden_seg_4s.npz")
    make_segment(6, 3, 6000, "golden_seg_6s.npz")

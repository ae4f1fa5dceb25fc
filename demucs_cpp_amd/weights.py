"""dmc4 / dmc6 / dmc3 weight files: tensor catalogue, synthetic generator, writer, reader.

The file format is the reference's ggml-style container and is a KEPT SURFACE:
  writer  : /root/reference/scripts/convert-pth-to-ggml.py:111-140
  reader  : /root/reference/src/model_load.cpp:79-147 (+ fp16 widening :1092-1300)
little-endian  u32 magic ("dmc4" 0x646d6334 | "dmc6" 0x646d6336 | "dmc3" 0x646d6333 = Demucs v3
hdemucs_mmi, reader /root/reference/src/model_load.cpp:1302-2166) then records
  {i32 n_dims; i32 name_len; i32 ne[n_dims]; char name[name_len]; f16 data[prod(ne)]}
in C order, shapes after ``squeeze()`` (size-1 axes removed).

There are no real checkpoints in this environment (no network), so benchmarks and
parity tests use synthetic weights in exactly this format.
The product loader (csrc/model_pack.cpp) and the oracle's loader both read these files.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

MAGIC_4S = 0x646D6334
MAGIC_6S = 0x646D6336
MAGIC_V3 = 0x646D6333


def _squeeze(shape):
    return tuple(int(s) for s in shape if s != 1)


def tensor_catalogue(n_sources: int = 4) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, squeezed shape) for every tensor of htdemucs (4s: 533, 6s: 525).

    Shapes: /root/reference/src/model.hpp:26-554, names:
    /root/reference/src/model_load.cpp:156-1062."""
    assert n_sources in (4, 6)
    D = 512 if n_sources == 4 else 384
    FF = 4 * D
    S = n_sources
    ch = [48, 96, 192, 384]
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def add(name, shape):
        out.append((name, _squeeze(shape)))

    def dconv(prefix, C):
        for j in range(2):
            p = f"{prefix}.dconv.layers.{j}"
            add(f"{p}.0.weight", (C // 8, C, 3))
            add(f"{p}.0.bias", (C // 8,))
            add(f"{p}.1.weight", (C // 8,))
            add(f"{p}.1.bias", (C // 8,))
            add(f"{p}.3.weight", (2 * C, C // 8, 1))
            add(f"{p}.3.bias", (2 * C,))
            add(f"{p}.4.weight", (2 * C,))
            add(f"{p}.4.bias", (2 * C,))
            add(f"{p}.6.scale", (C,))

    for i in range(4):
        C = ch[i]
        cin_f = 4 if i == 0 else ch[i - 1]
        cin_t = 2 if i == 0 else ch[i - 1]
        add(f"encoder.{i}.conv.weight", (C, cin_f, 8, 1))
        add(f"encoder.{i}.conv.bias", (C,))
        add(f"encoder.{i}.rewrite.weight", (2 * C, C, 1, 1))
        add(f"encoder.{i}.rewrite.bias", (2 * C,))
        dconv(f"encoder.{i}", C)
        add(f"tencoder.{i}.conv.weight", (C, cin_t, 8))
        add(f"tencoder.{i}.conv.bias", (C,))
        add(f"tencoder.{i}.rewrite.weight", (2 * C, C, 1))
        add(f"tencoder.{i}.rewrite.bias", (2 * C,))
        dconv(f"tencoder.{i}", C)
    for k in range(4):
        Cd = ch[3 - k]
        cout_f = ch[2 - k] if k < 3 else 4 * S
        cout_t = ch[2 - k] if k < 3 else 2 * S
        add(f"decoder.{k}.conv_tr.weight", (Cd, cout_f, 8, 1))
        add(f"decoder.{k}.conv_tr.bias", (cout_f,))
        add(f"decoder.{k}.rewrite.weight", (2 * Cd, Cd, 3, 3))
        add(f"decoder.{k}.rewrite.bias", (2 * Cd,))
        dconv(f"decoder.{k}", Cd)
        add(f"tdecoder.{k}.conv_tr.weight", (Cd, cout_t, 8))
        add(f"tdecoder.{k}.conv_tr.bias", (cout_t,))
        add(f"tdecoder.{k}.rewrite.weight", (2 * Cd, Cd, 3))
        add(f"tdecoder.{k}.rewrite.bias", (2 * Cd,))
        dconv(f"tdecoder.{k}", Cd)
    add("freq_emb.embedding.weight", (512, 48))
    if n_sources == 4:
        for nm, (o, i_) in (("channel_upsampler", (512, 384)), ("channel_downsampler", (384, 512)),
                            ("channel_upsampler_t", (512, 384)), ("channel_downsampler_t", (384, 512))):
            add(f"{nm}.weight", (o, i_, 1))
            add(f"{nm}.bias", (o,))
    for nm in ("norm_in", "norm_in_t"):
        add(f"crosstransformer.{nm}.weight", (D,))
        add(f"crosstransformer.{nm}.bias", (D,))
    for layer in range(5):
        for sfx in ("", "_t"):
            p = f"crosstransformer.layers{sfx}.{layer}"
            attn = "self_attn" if layer % 2 == 0 else "cross_attn"
            add(f"{p}.{attn}.in_proj_weight", (3 * D, D))
            add(f"{p}.{attn}.in_proj_bias", (3 * D,))
            add(f"{p}.{attn}.out_proj.weight", (D, D))
            add(f"{p}.{attn}.out_proj.bias", (D,))
            add(f"{p}.linear1.weight", (FF, D))
            add(f"{p}.linear1.bias", (FF,))
            add(f"{p}.linear2.weight", (D, FF))
            add(f"{p}.linear2.bias", (D,))
            norms = ("norm1", "norm2", "norm_out") if layer % 2 == 0 else ("norm1", "norm2", "norm3", "norm_out")
            for n in norms:
                add(f"{p}.{n}.weight", (D,))
                add(f"{p}.{n}.bias", (D,))
            add(f"{p}.gamma_1.scale", (D,))
            add(f"{p}.gamma_2.scale", (D,))
    return out


def tensor_catalogue_v3() -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, squeezed shape) for every tensor of Demucs v3 `hdemucs_mmi` (4 sources).

    Shapes: /root/reference/src/model.hpp:694-1236, names: /root/reference/src/model_load.cpp:1388-2136.
    Six-level hybrid: encoder / tencoder 0-3 as in v4 but DConv compress 4 (hidden C/4) and no transformer;
    tencoder.4 is a bare conv whose output is injected into encoder.4; encoder.4 / encoder.5 carry
    GroupNorm(4 groups) and a DConv with a 2-layer BiLSTM and LocalState attention; the decoders have no DConv."""
    ch = [48, 96, 192, 384]
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def add(name, shape):
        out.append((name, _squeeze(shape)))

    def dconv(prefix, C):
        H = C // 4
        for j in range(2):
            p = f"{prefix}.dconv.layers.{j}"
            add(f"{p}.0.weight", (H, C, 3))
            add(f"{p}.0.bias", (H,))
            add(f"{p}.1.weight", (H,))
            add(f"{p}.1.bias", (H,))
            add(f"{p}.3.weight", (2 * C, H, 1))
            add(f"{p}.3.bias", (2 * C,))
            add(f"{p}.4.weight", (2 * C,))
            add(f"{p}.4.bias", (2 * C,))
            add(f"{p}.6.scale", (C,))

    def dconv_lstm(prefix, C):
        H = C // 4
        for j in range(2):
            p = f"{prefix}.dconv.layers.{j}"
            add(f"{p}.0.weight", (H, C, 3))
            add(f"{p}.0.bias", (H,))
            add(f"{p}.1.weight", (H,))
            add(f"{p}.1.bias", (H,))
            for layer in range(2):
                for sfx in ("", "_reverse"):
                    add(f"{p}.3.lstm.weight_ih_l{layer}{sfx}", (4 * H, H if layer == 0 else 2 * H))
                    add(f"{p}.3.lstm.weight_hh_l{layer}{sfx}", (4 * H, H))
                    add(f"{p}.3.lstm.bias_ih_l{layer}{sfx}", (4 * H,))
                    add(f"{p}.3.lstm.bias_hh_l{layer}{sfx}", (4 * H,))
            add(f"{p}.3.linear.weight", (H, 2 * H))
            add(f"{p}.3.linear.bias", (H,))
            for nm in ("content", "query", "key"):
                add(f"{p}.4.{nm}.weight", (H, H, 1))
                add(f"{p}.4.{nm}.bias", (H,))
            add(f"{p}.4.query_decay.weight", (16, H, 1))
            add(f"{p}.4.query_decay.bias", (16,))
            add(f"{p}.4.proj.weight", (H, H, 1))
            add(f"{p}.4.proj.bias", (H,))
            add(f"{p}.5.weight", (2 * C, H, 1))
            add(f"{p}.5.bias", (2 * C,))
            add(f"{p}.6.weight", (2 * C,))
            add(f"{p}.6.bias", (2 * C,))
            add(f"{p}.8.scale", (C,))

    for i in range(4):
        C = ch[i]
        cin_f = 4 if i == 0 else ch[i - 1]
        add(f"encoder.{i}.conv.weight", (C, cin_f, 8, 1))
        add(f"encoder.{i}.conv.bias", (C,))
        add(f"encoder.{i}.rewrite.weight", (2 * C, C, 1, 1))
        add(f"encoder.{i}.rewrite.bias", (2 * C,))
        dconv(f"encoder.{i}", C)
    for i, (C, cin, k) in ((4, (768, 384, (8, 1))), (5, (1536, 768, (4,)))):
        add(f"encoder.{i}.conv.weight", (C, cin) + k)
        add(f"encoder.{i}.conv.bias", (C,))
        add(f"encoder.{i}.norm1.weight", (C,))
        add(f"encoder.{i}.norm1.bias", (C,))
        add(f"encoder.{i}.rewrite.weight", (2 * C, C, 1))
        add(f"encoder.{i}.rewrite.bias", (2 * C,))
        add(f"encoder.{i}.norm2.weight", (2 * C,))
        add(f"encoder.{i}.norm2.bias", (2 * C,))
        dconv_lstm(f"encoder.{i}", C)
    add("decoder.0.conv_tr.weight", (1536, 768, 4))
    add("decoder.0.conv_tr.bias", (768,))
    add("decoder.0.norm2.weight", (768,))
    add("decoder.0.norm2.bias", (768,))
    add("decoder.0.rewrite.weight", (3072, 1536, 3))
    add("decoder.0.rewrite.bias", (3072,))
    add("decoder.0.norm1.weight", (3072,))
    add("decoder.0.norm1.bias", (3072,))
    add("decoder.1.conv_tr.weight", (768, 384, 8, 1))
    add("decoder.1.conv_tr.bias", (384,))
    add("decoder.1.norm2.weight", (384,))
    add("decoder.1.norm2.bias", (384,))
    add("decoder.1.rewrite.weight", (1536, 768, 3, 3))
    add("decoder.1.rewrite.bias", (1536,))
    add("decoder.1.norm1.weight", (1536,))
    add("decoder.1.norm1.bias", (1536,))
    for k in range(4):
        Cd = ch[3 - k]
        cout_f = ch[2 - k] if k < 3 else 16
        add(f"decoder.{k + 2}.conv_tr.weight", (Cd, cout_f, 8, 1))
        add(f"decoder.{k + 2}.conv_tr.bias", (cout_f,))
        add(f"decoder.{k + 2}.rewrite.weight", (2 * Cd, Cd, 3, 3))
        add(f"decoder.{k + 2}.rewrite.bias", (2 * Cd,))
    for i in range(4):
        C = ch[i]
        cin_t = 2 if i == 0 else ch[i - 1]
        add(f"tencoder.{i}.conv.weight", (C, cin_t, 8))
        add(f"tencoder.{i}.conv.bias", (C,))
        add(f"tencoder.{i}.rewrite.weight", (2 * C, C, 1))
        add(f"tencoder.{i}.rewrite.bias", (2 * C,))
        dconv(f"tencoder.{i}", C)
    add("tencoder.4.conv.weight", (768, 384, 8))
    add("tencoder.4.conv.bias", (768,))
    add("tdecoder.0.conv_tr.weight", (768, 384, 8))
    add("tdecoder.0.conv_tr.bias", (384,))
    add("tdecoder.0.norm2.weight", (384,))
    add("tdecoder.0.norm2.bias", (384,))
    for k in range(4):
        Cd = ch[3 - k]
        cout_t = ch[2 - k] if k < 3 else 8
        add(f"tdecoder.{k + 1}.conv_tr.weight", (Cd, cout_t, 8))
        add(f"tdecoder.{k + 1}.conv_tr.bias", (cout_t,))
        add(f"tdecoder.{k + 1}.rewrite.weight", (2 * Cd, Cd, 3))
        add(f"tdecoder.{k + 1}.rewrite.bias", (2 * Cd,))
    add("freq_emb.embedding.weight", (512, 48))
    return out


def _is_norm_param(name: str) -> bool:
    """GroupNorm / LayerNorm affine tensors by name (both architectures)."""
    if ".norm" in name:
        return True
    if ".dconv.layers." in name:
        tail = name.split(".dconv.layers.")[1].split(".", 1)[1]  # e.g. "1.weight", "3.lstm.bias_ih_l0"
        # index 1: GroupNorm(hidden); plain DConv: index 4 is GroupNorm(2C); LSTM DConv (levels 4/5 of v3): index 6
        return tail in ("1.weight", "1.bias", "4.weight", "4.bias", "6.weight", "6.bias")
    return False


def synth_weights(n_sources: int = 4, seed: int = 0, variant: str = "default", arch: str = "v4") -> Dict[str, np.ndarray]:
    """Synthetic fp16 weights: N(0, 1/fan_in) for linear /
    conv kernels, norm weights 1 +- 0.1, biases 0.01 N(0,1), LayerScale tensors
    U(0.05, 0.5) so that every residual branch moves the output measurably.

    Stress variants (parity where the default model is blind, tests/test_gpu_parity.py):
      "dc"       every conv / linear bias = 3 + 0.5 N(0,1) and norm biases N(0,1), norm weights U(0.2, 3):
                 the inputs of GroupNorm / LayerNorm have |mean| >> sigma, which is where a one-pass
                 sum / sum-of-squares variance loses digits against the reference's two-pass
                 calculate_variance (/root/reference/src/layers.hpp:76-95);
      "illcond"  "dc" + every DConv 1x1 weight (`.3.weight`, 2C x C/8) rebuilt with singular values from 1
                 down to 1e-3 and the last quarter exactly 0 (rank deficient): exercises the factored
                 W^T W statistics of the DConv (csrc/model_pack.cpp, EPI_STATS_FACT);
      "initscale" LayerScale tensors = 1e-4, the value Demucs initialises them with."""
    assert variant in ("default", "dc", "illcond", "initscale")
    assert arch in ("v4", "v3")
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    dc = variant in ("dc", "illcond")
    cat = tensor_catalogue(n_sources) if arch == "v4" else tensor_catalogue_v3()
    for name, shape in cat:
        n = int(np.prod(shape))
        if ".lstm." in name:
            # torch.nn.LSTM init: U(-1/sqrt(H), 1/sqrt(H)) for all four tensors; H = 4H rows / 4
            hdim = shape[0] // 4
            a = rng.uniform(-1.0, 1.0, size=shape) / np.sqrt(float(hdim))
            if dc and ".bias_" in name:
                a = a + 0.5
        elif name.endswith(".scale"):
            a = rng.uniform(0.05, 0.5, size=shape)
            if variant == "initscale":
                a = np.full(shape, 1e-4)
        elif name == "freq_emb.embedding.weight":
            a = 0.5 * rng.standard_normal(shape)
        elif name.endswith("bias") or name.endswith("in_proj_bias"):
            is_norm = _is_norm_param(name) if arch == "v3" else (".norm" in name or ".1.bias" in name or ".4.bias" in name)
            if is_norm:
                a = (1.0 if dc else 0.05) * rng.standard_normal(shape)
            else:
                a = 0.01 * rng.standard_normal(shape)
                if dc:
                    a = 3.0 + 0.5 * rng.standard_normal(shape)
        elif (_is_norm_param(name) if arch == "v3" else (".norm" in name or name.endswith(".1.weight") or name.endswith(".4.weight"))):
            a = 1.0 + 0.1 * rng.standard_normal(shape)
            if dc:
                a = rng.uniform(0.2, 3.0, size=shape)
        else:
            if "conv_tr" in name:
                fan_in = 2 * shape[0]  # every output sample sees 2 taps x Cin (k8 s4 and k4 s2 alike)
            else:
                fan_in = n // shape[0]
            a = rng.standard_normal(shape) / np.sqrt(float(fan_in))
            if variant == "illcond" and ".dconv." in name and name.endswith(".3.weight"):
                rows, cols = shape[0], int(np.prod(shape[1:]))
                u, _, vt = np.linalg.svd(rng.standard_normal((rows, cols)), full_matrices=False)
                sv = np.logspace(0, -3, cols)
                sv[cols - cols // 4:] = 0.0
                a = ((u * sv) @ vt).reshape(shape) * np.sqrt(float(rows) / max(1, cols - cols // 4)) / np.sqrt(float(fan_in)) * 3.0
        out[name] = np.ascontiguousarray(a.astype(np.float16))
    return out


def write_model(path: str, tensors: Dict[str, np.ndarray], n_sources: int, arch: str = "v4") -> None:
    with open(path, "wb") as f:
        f.write(struct.pack("<I", MAGIC_V3 if arch == "v3" else (MAGIC_4S if n_sources == 4 else MAGIC_6S)))
        for name, arr in tensors.items():
            a = np.ascontiguousarray(arr.astype(np.float16))
            nm = name.encode("utf-8")
            f.write(struct.pack("<ii", a.ndim, len(nm)))
            for d in a.shape:
                f.write(struct.pack("<i", int(d)))
            f.write(nm)
            f.write(a.tobytes())


def read_model(path: str) -> Tuple[int, Dict[str, np.ndarray]]:
    """Independent numpy reader (used by the fp64 golden model and by tests)."""
    with open(path, "rb") as f:
        buf = f.read()
    (magic,) = struct.unpack_from("<I", buf, 0)
    if magic == MAGIC_4S:
        ns = 4
    elif magic == MAGIC_6S:
        ns = 6
    elif magic == MAGIC_V3:
        ns = 3  # architecture tag, not a stem count: Demucs v3 has 4 stems
    else:
        raise ValueError("bad magic")
    pos = 4
    out: Dict[str, np.ndarray] = {}
    while pos < len(buf):
        nd, ln = struct.unpack_from("<ii", buf, pos)
        pos += 8
        shape = struct.unpack_from("<" + "i" * nd, buf, pos)
        pos += 4 * nd
        name = buf[pos:pos + ln].decode("utf-8")
        pos += ln
        n = int(np.prod(shape)) if nd else 1
        out[name] = np.frombuffer(buf, dtype="<f2", count=n, offset=pos).reshape(shape).copy()
        pos += 2 * n
    return ns, out


def write_synthetic_model(path: str, n_sources: int = 4, seed: int = 0, variant: str = "default", arch: str = "v4") -> None:
    write_model(path, synth_weights(n_sources, seed, variant, arch), n_sources, arch)

#!/usr/bin/env python
"""PyTorch HTDemucs / HDemucs checkpoint -> dmc4 / dmc6 / dmc3 weight file (the counterpart of
/root/reference/scripts/convert-pth-to-ggml.py:111-140, without its dependency on the `demucs`
package and the torch-hub download):

    python tools/convert_pth_to_dmc.py  955717e8-8726e21a.th  out/ggml-model-htdemucs-4s-f16.bin
    python tools/convert_pth_to_dmc.py  5c90dfd2-34c22ccb.th  out/ggml-model-htdemucs-6s-f16.bin
    python tools/convert_pth_to_dmc.py  75fc33f5-1941ce65.th  out/ggml-model-hdemucs_mmi-v3-f16.bin   (Demucs v3, magic dmc3)

Input: a file `torch.load` can read that holds either the state dict itself or the hub checkpoint
`{"state": state_dict, ...}` (what facebookresearch/demucs publishes). Every tensor is written in
state-dict order as {i32 n_dims, i32 name_len, i32 shape[n_dims], name, f16 data} after `squeeze()`,
behind the magic "dmc4" / "dmc6" chosen from the number of sources (the last decoder's output
channels). The result is validated against the tensor catalogue the loaders expect
(demucs_cpp_amd/weights.py): missing, unexpected or mis-shaped tensors are reported."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demucs_cpp_amd.weights import tensor_catalogue, tensor_catalogue_v3, write_model  # noqa: E402


def convert(state, strict=True):
    """state: name -> array-like (torch tensors or numpy). Returns (n_sources, ordered dict of fp16 arrays, problems);
    n_sources == 3 tags the Demucs v3 (hdemucs_mmi) architecture, which has 4 stems (weights.read_model convention)."""
    tensors = {}
    for name, t in state.items():
        a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
        tensors[name] = np.ascontiguousarray(np.squeeze(a).astype(np.float16))
    if "encoder.4.dconv.layers.0.3.lstm.weight_ih_l0" in tensors:  # the BiLSTM of levels 4 / 5 exists in v3 only
        ns, cat = 3, dict(tensor_catalogue_v3())
    else:
        key = "decoder.3.conv_tr.bias"
        if key not in tensors:
            raise ValueError(f"not an HTDemucs / HDemucs state dict: '{key}' missing")
        n_out = int(tensors[key].shape[0])  # 4 * S (complex-as-channels x stereo x sources)
        if n_out not in (16, 24):
            raise ValueError(f"unsupported number of sources: decoder.3.conv_tr.bias has {n_out} channels")
        ns = n_out // 4
        cat = dict(tensor_catalogue(ns))
    problems = []
    for name, shape in cat.items():
        if name not in tensors:
            problems.append(f"missing tensor {name}")
        elif tuple(tensors[name].shape) != tuple(shape):
            problems.append(f"shape of {name}: {tuple(tensors[name].shape)} != expected {tuple(shape)}")
    for name in tensors:
        if name not in cat:
            problems.append(f"unexpected tensor {name}")
    if problems and strict:
        raise ValueError("; ".join(problems[:8]) + (f" (+{len(problems) - 8} more)" if len(problems) > 8 else ""))
    return ns, tensors, problems


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("checkpoint")
    ap.add_argument("output")
    ap.add_argument("--lenient", action="store_true", help="write the file even if the catalogue check reports problems")
    args = ap.parse_args()
    import torch

    ck = torch.load(args.checkpoint, map_location="cpu", weights_only=False)
    state = ck["state"] if isinstance(ck, dict) and "state" in ck else ck
    ns, tensors, problems = convert(state, strict=not args.lenient)
    for pmsg in problems:
        print("warning:", pmsg, file=sys.stderr)
    os.makedirs(os.path.dirname(os.path.abspath(args.output)), exist_ok=True)
    write_model(args.output, tensors, 4 if ns == 3 else ns, "v3" if ns == 3 else "v4")
    print(f"wrote {args.output}: dmc{ns}, {len(tensors)} tensors, {os.path.getsize(args.output) / 1e6:.1f} MB")


if __name__ == "__main__":
    main()

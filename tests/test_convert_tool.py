"""tools/convert_pth_to_dmc.py: a PyTorch-style checkpoint with the HTDemucs
state-dict names and un-squeezed shapes converts to a dmc file that the independent reader and the
oracle's loader accept, with identical values. (No real checkpoint exists in this environment.)"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle_lib as orc
from demucs_cpp_amd.weights import read_model, synth_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "convert_pth_to_dmc.py")


def unsqueezed_state(ns, seed):
    """synthetic weights in the shapes PyTorch stores them: conv kernels keep their singleton axes"""
    w = synth_weights(ns, seed)
    state = {}
    for name, a in w.items():
        t = torch.from_numpy(a.astype(np.float32))
        if name.endswith("conv.weight") and name.startswith(("encoder.", "decoder.")) and t.ndim == 3:
            t = t.unsqueeze(-1)                      # Conv2d (C, Cin, 8, 1)
        if ".rewrite.weight" in name and name.startswith("encoder.") and t.ndim == 2:
            t = t.reshape(t.shape[0], t.shape[1], 1, 1)  # Conv2d 1x1
        if ".dconv.layers." in name and name.endswith(".3.weight") and t.ndim == 2:
            t = t.unsqueeze(-1)                      # Conv1d 1x1
        state[name] = t
    return w, state


@pytest.mark.parametrize("ns,wrap", [(4, True), (6, False)])
def test_checkpoint_round_trip(ns, wrap, tmp_path):
    w, state = unsqueezed_state(ns, 7 + ns)
    ck = str(tmp_path / "model.th")
    torch.save({"state": state, "klass": "HTDemucs"} if wrap else state, ck)
    out = str(tmp_path / "sub" / f"ggml-model-htdemucs-{ns}s-f16.bin")
    r = subprocess.run([sys.executable, TOOL, ck, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got_ns, got = read_model(out)
    assert got_ns == ns and list(got.keys()) == list(w.keys())  # state-dict order kept
    for name in w:
        assert got[name].shape == w[name].shape and np.array_equal(got[name], w[name].astype(np.float16)), name
    m = orc.OracleModel(out)  # the oracle's loader (same format rules as the product's) accepts it
    m.close()


def test_v3_checkpoint_round_trip(tmp_path):
    """hdemucs_mmi: the converter recognises the v3 state dict by its LSTM tensors and writes a dmc3 file
    (counterpart of scripts/convert-pth-to-ggml.py --v3)."""
    w = synth_weights(4, 11, "default", "v3")
    state = {}
    for name, a in w.items():
        t = torch.from_numpy(a.astype(np.float32))
        if name in ("encoder.4.conv.weight",) or (name.startswith(("encoder.", "decoder.")) and name.endswith(("conv.weight", "conv_tr.weight"))
                                                 and t.ndim == 3 and not name.startswith(("encoder.5", "decoder.0"))):
            t = t.unsqueeze(-1)  # Conv2d / ConvTranspose2d kernels (k, 1)
        if ".dconv.layers." in name and t.ndim == 2 and name.endswith((".3.weight", ".5.weight", "content.weight", "query.weight",
                                                                       "key.weight", "query_decay.weight", "proj.weight")):
            t = t.unsqueeze(-1)  # Conv1d 1x1
        state[name] = t
    ck = str(tmp_path / "v3.th")
    torch.save({"state": state, "klass": "HDemucs"}, ck)
    out = str(tmp_path / "ggml-model-hdemucs_mmi-v3-f16.bin")
    r = subprocess.run([sys.executable, TOOL, ck, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "dmc3" in r.stdout, r.stdout + r.stderr
    tag, got = read_model(out)
    assert tag == 3 and list(got.keys()) == list(w.keys())
    for name in w:
        assert got[name].shape == w[name].shape and np.array_equal(got[name], w[name].astype(np.float16)), name
    m = orc.OracleModel(out)
    assert m.arch == 3 and m.n_tensors == 395
    m.close()


def test_rejects_foreign_and_incomplete_state(tmp_path):
    _, state = unsqueezed_state(4, 1)
    del state["freq_emb.embedding.weight"]
    ck = str(tmp_path / "broken.th")
    torch.save(state, ck)
    r = subprocess.run([sys.executable, TOOL, ck, str(tmp_path / "x.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "missing tensor freq_emb.embedding.weight" in (r.stdout + r.stderr)
    torch.save({"foo": torch.zeros(3)}, ck)
    r = subprocess.run([sys.executable, TOOL, ck, str(tmp_path / "x.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "state dict" in (r.stdout + r.stderr)

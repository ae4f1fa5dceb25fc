"""dmc4/dmc6 container (kept surface): catalogue, writer, readers.
Format: /root/reference/scripts/convert-pth-to-ggml.py:111-140, src/model_load.cpp:79-147."""
import os

import numpy as np

import oracle_lib as orc
from demucs_cpp_amd.weights import read_model, synth_weights, tensor_catalogue, write_model


def test_catalogue_counts_match_reference_readme():
    assert len(tensor_catalogue(4)) == 533  # /root/reference/README.md:100
    assert len(tensor_catalogue(6)) == 525


def test_write_read_roundtrip_and_size(tmp_path):
    w = synth_weights(4, 0)
    p = str(tmp_path / "m.bin")
    write_model(p, w, 4)
    assert abs(os.path.getsize(p) / 1024 / 1024 - 80.08) < 0.1  # "80.08 MB", README.md:100
    ns, r = read_model(p)
    assert ns == 4 and list(r.keys()) == list(w.keys())
    for k in w:
        assert r[k].shape == w[k].shape and np.array_equal(r[k], w[k])


def test_oracle_loader_agrees(tmp_models):
    m = orc.OracleModel(tmp_models[6])
    assert m.n_sources == 6 and m.n_tensors == 525
    m.close()

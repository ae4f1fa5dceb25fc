"""C-ABI surface checks that need no GPU: the library loads, exports every symbol the
header declares, and fails loudly (no CPU fallback) - loader error behaviour mirrors
/root/reference/src/model_load.cpp:64-69,97-102,1065-1070,1096-1105."""
import ctypes
import os
import re
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dmx():
    so = os.path.join(ROOT, "demucs_cpp_amd", "lib", "libdemucs_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", ROOT, "demucs_cpp_amd/lib/libdemucs_hip.so"], stdout=subprocess.DEVNULL)
    from demucs_cpp_amd import binding
    return binding


def test_header_and_library_export_the_same_symbols(dmx):
    hdr = open(os.path.join(ROOT, "include", "demucs_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(dmx_[a-z_0-9]+)\s*\(", hdr)) - {"dmx_progress_fn"})
    assert declared == sorted(dmx.EXPORTS)
    L = dmx.lib()
    for s in declared:
        assert hasattr(L, s), s


def test_no_torch_types_in_the_abi():
    hdr = open(os.path.join(ROOT, "include", "demucs_hip.h")).read()
    assert "torch" not in hdr.lower() and "std::" not in hdr and "Eigen::" not in hdr.replace("Eigen::MatrixXf", "").replace("Eigen::Tensor3dXf", "")


def test_load_missing_file_fails(dmx):
    with pytest.raises(dmx.DmxError) as e:
        dmx.Model("/nonexistent/model.bin")
    assert e.value.code == 1 and "failed to open" in str(e.value)


def test_load_bad_magic_fails(dmx, tmp_path):
    p = tmp_path / "bad.bin"
    p.write_bytes(struct.pack("<I", 0x646D6335) + b"\0" * 64)  # "dmc5": neither dmc4 / dmc6 (v4) nor dmc3 (v3)
    with pytest.raises(dmx.DmxError) as e:
        dmx.Model(str(p))
    assert e.value.code == 2 and "bad magic" in str(e.value)


def _write(path, ns, tensors):
    from demucs_cpp_amd.weights import write_model
    write_model(str(path), tensors, ns)


def test_load_unknown_tensor_and_wrong_size_fail(dmx, tmp_path):
    from demucs_cpp_amd.weights import synth_weights
    w = synth_weights(4, 0)
    bad = dict(w)
    bad["encoder.0.conv.bogus"] = np.zeros(3, np.float16)
    _write(tmp_path / "unk.bin", 4, bad)
    with pytest.raises(dmx.DmxError) as e:
        dmx.Model(str(tmp_path / "unk.bin"))
    assert e.value.code == 2 and "failed to load encoder.0.conv.bogus" in str(e.value)
    bad = dict(w)
    bad["encoder.0.conv.bias"] = np.zeros(47, np.float16)
    _write(tmp_path / "size.bin", 4, bad)
    with pytest.raises(dmx.DmxError) as e:
        dmx.Model(str(tmp_path / "size.bin"))
    assert e.value.code == 2 and "wrong size" in str(e.value)
    bad = dict(w)
    del bad["freq_emb.embedding.weight"]
    _write(tmp_path / "missing.bin", 4, bad)
    with pytest.raises(dmx.DmxError) as e:
        dmx.Model(str(tmp_path / "missing.bin"))
    assert e.value.code == 2 and "missing" in str(e.value)


def test_six_source_file_with_four_source_tensor_shapes_fails(dmx, tmp_path):
    from demucs_cpp_amd.weights import synth_weights
    w = synth_weights(4, 0)  # 4s shapes written under the dmc6 magic: resamplers are unknown there
    _write(tmp_path / "mix.bin", 6, w)
    with pytest.raises(dmx.DmxError) as e:
        dmx.Model(str(tmp_path / "mix.bin"))
    assert e.value.code == 2


def test_v3_file_failure_cases(dmx, tmp_path):
    """dmc3 (Demucs v3 hdemucs_mmi): same loader contract as the reference's load_demucs_v3_model
    (/root/reference/src/model_load.cpp:1302-2166) + the missing-tensor check."""
    from demucs_cpp_amd.weights import synth_weights, write_model
    w = synth_weights(4, 5, "default", "v3")
    bad = dict(w)
    bad["encoder.4.dconv.layers.0.3.lstm.weight_ih_l2"] = np.zeros(3, np.float16)
    write_model(str(tmp_path / "unk.bin"), bad, 4, "v3")
    with pytest.raises(dmx.DmxError) as e:
        dmx.Model(str(tmp_path / "unk.bin"))
    assert e.value.code == 2 and "failed to load encoder.4.dconv.layers.0.3.lstm.weight_ih_l2" in str(e.value)
    bad = dict(w)
    bad["encoder.5.dconv.layers.1.4.query_decay.bias"] = np.zeros(15, np.float16)
    write_model(str(tmp_path / "size.bin"), bad, 4, "v3")
    with pytest.raises(dmx.DmxError) as e:
        dmx.Model(str(tmp_path / "size.bin"))
    assert e.value.code == 2 and "wrong size" in str(e.value)
    bad = dict(w)
    del bad["tdecoder.0.norm2.bias"]
    write_model(str(tmp_path / "missing.bin"), bad, 4, "v3")
    with pytest.raises(dmx.DmxError) as e:
        dmx.Model(str(tmp_path / "missing.bin"))
    assert e.value.code == 2 and "missing" in str(e.value)
    # v4 tensors under the v3 magic (and the reverse) are rejected by name
    write_model(str(tmp_path / "v4as3.bin"), synth_weights(4, 0), 4, "v3")
    with pytest.raises(dmx.DmxError) as e:
        dmx.Model(str(tmp_path / "v4as3.bin"))
    assert e.value.code == 2
    write_model(str(tmp_path / "v3as4.bin"), w, 4, "v4")
    with pytest.raises(dmx.DmxError) as e:
        dmx.Model(str(tmp_path / "v3as4.bin"))
    assert e.value.code == 2


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="this check is for GPU-less hosts")
def test_no_gpu_means_loud_failure_not_a_cpu_fallback(dmx, tmp_models):
    assert dmx.device_count() == 0
    with pytest.raises(dmx.DmxError) as e:
        dmx.Model(tmp_models[4])
    assert e.value.code == 3 and "no CPU fallback" in str(e.value)


def test_product_does_not_reference_the_oracle():
    # the oracle is test infrastructure: nothing under the product tree may mention it
    for base in ("demucs_cpp_amd", "include", "cli"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "oracle_lib" not in txt and "liboracle" not in txt and "cpu_interp" not in txt.replace("tests/cpu_interp.cpp", ""), os.path.join(dp, f)
                    # no include / import / dlopen of anything that lives under oracle/ (comments may cite it)
                    import re
                    for ln in txt.splitlines():
                        assert not re.search(r"(#\s*include|\bimport\b|\bfrom\b|dlopen|CDLL)[^\n]*(oracle|threaded_split)", ln), (os.path.join(dp, f), ln)


def test_weight_split_is_exact_for_every_fp16_value(dmx):
    """DMX_GEMM_BF16X3 (include/demucs_hip.h): a weight - an fp16 number in the dmc4 / dmc6 / dmc3 files
    (/root/reference/scripts/convert-pth-to-ggml.py:111-140) - is the sum of two bf16 terms by round-to-nearest splits.
    Pure host function: checked here for EVERY finite fp16 bit pattern (subnormals included - bf16 has the fp32 exponent
    range), with the bound |w2| <= 2^-8 |w| that makes the dropped a3 w2 product <= 2^-24 |a w|; values that are not
    fp16-exact are reported (such ops keep the fp32 kernel, api.cpp split_ok)."""
    h = np.arange(0, 1 << 16, dtype=np.uint16).view(np.float16)
    w = h[np.isfinite(h)].astype(np.float32)
    w1, w2, bad = dmx.split_weights(w)
    assert bad == 0
    f1 = (w1.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    f2 = (w2.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    assert np.array_equal(f1 + f2, w.astype(np.float64))
    assert (np.abs(f2) <= 2.0 ** -8 * np.abs(w.astype(np.float64))).all()
    assert (np.abs(f1 - w) <= 2.0 ** -8 * np.abs(w)).all()
    # derived (non-fp16) values: some need a third term, and the function says so instead of rounding silently
    v = np.array([1.0 + 2.0 ** -9 + 2.0 ** -20, np.pi, 1e-30, 3.0], np.float32)
    _, _, bad = dmx.split_weights(v)
    assert bad == 3
    # non-finite weights stay non-finite in the first plane
    w1, _, _ = dmx.split_weights(np.array([np.inf, -np.inf, np.nan], np.float32))
    t = (w1.astype(np.uint32) << 16).view(np.float32)
    assert t[0] == np.inf and t[1] == -np.inf and np.isnan(t[2])


def test_dmx_gemm_environment_value_selects_the_process_default_and_typos_are_reported():
    """DMX_GEMM is read once by the first dmx_default_gemm() of a process (no GPU needed): unset / bf16x3 -> DMX_GEMM_BF16X3 (the
    default), f32 -> DMX_GEMM_F32, fp16x3 -> DMX_GEMM_FP16X3 (opt-in: the ONLY way besides an explicit dmx_ctx_create_gemm /
    dmx_set_default_gemm to get that mode); anything else is named on stderr and the default is used - a typo must not silently
    select another arithmetic."""
    import subprocess
    import sys

    code = "from demucs_cpp_amd import binding as b; print('MODE', b.default_gemm(), b.GEMM_NAMES[b.default_gemm()])"
    for val, want, warn in ((None, 1, False), ("", 1, False), ("bf16x3", 1, False), ("f32", 0, False), ("fp16x3", 2, False),
                            ("fp16", 1, True), ("BF16X3", 1, True), ("fp32", 1, True)):
        env = dict(os.environ)
        env.pop("DMX_GEMM", None)
        if val is not None:
            env["DMX_GEMM"] = val
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert f"MODE {want} " in r.stdout, (val, r.stdout)
        assert ("is not one of {f32, bf16x3, fp16x3}" in r.stderr) == warn, (val, r.stderr)
        if warn:
            assert f"DMX_GEMM={val}" in r.stderr


def test_every_environment_switch_of_the_product_is_documented():
    """INTEGRATION.md's environment table names every DMX_* variable the library, the shim and the CLIs read."""
    import re
    names = set()
    for sub in ("demucs_cpp_amd/csrc", "demucs_cpp_amd/host", "cli"):
        for dp, _, fs in os.walk(os.path.join(ROOT, sub)):
            for f in fs:
                if f.endswith((".cpp", ".hip", ".h", ".hpp")):
                    names |= set(re.findall(r'getenv\("(DMX_[A-Z0-9_]+)"\)', open(os.path.join(dp, f), errors="replace").read()))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert len(names) > 20 and not missing, missing

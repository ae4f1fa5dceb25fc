"""GPU parity tests of Demucs v3 (hdemucs_mmi) - run with -m gpu on an MI355X.

The HIP path (dmc3 loader, 6-level plan, cooperative BiLSTM kernel, LocalState attention, 4-group GroupNorm;
csrc/plan.cpp build_plan_v3, csrc/v3.hip) through the C ABI against the CPU oracle's restatement of
/root/reference/src/model_inference.cpp:477-856 on the same seeded inputs, and against the committed fp64 torch
golden vectors (tests/golden/make_golden_v3.py). Tolerance as for v4: max-abs error relative to the reference
tensor's max-abs < 1e-4 (the reference's NEAR_TOLERANCE), plus SDR > 60 dB per stem."""
import os

import numpy as np
import pytest

import oracle_lib as orc
import parity_utils as pu

pytestmark = pytest.mark.gpu
TOL = 1e-4
SEG_FULL = 343980


@pytest.fixture(scope="module")
def oracle_threads():
    orc.lib().orc_set_num_threads(min(32, os.cpu_count() or 1))


def sdr_db(ref, est):
    num = float((ref.astype(np.float64) ** 2).sum())
    den = float(((ref.astype(np.float64) - est.astype(np.float64)) ** 2).sum())
    return 10 * np.log10(num / max(den, 1e-300))


def test_v3_reduced_segment_all_layers_vs_oracle_and_golden(dmx, tmp_models, golden_dir, oracle_threads):
    g = np.load(os.path.join(golden_dir, "golden_seg_v3.npz"))
    m = dmx.Model(tmp_models[3])
    assert m.arch == 3 and m.n_sources == 4 and m.n_tensors == 395
    ctx = dmx.Context(m, int(g["seg"]), 1)
    om = orc.OracleModel(tmp_models[3])
    errs, out, ref = pu.compare_segment(ctx, om, g["mix"])
    bad = {k: v for k, v in errs.items() if not (v < TOL)}
    assert not bad, bad
    assert pu.relerr(out, g["out"]) < TOL  # independent fp64 torch model (torch.nn.LSTM, einsum LocalState)
    for s in range(4):
        assert sdr_db(ref[s], out[s]) > 60.0
    ctx.close(); m.close(); om.close()


@pytest.mark.parametrize("seg", [4096, 7000])
def test_v3_short_and_odd_frame_counts(seg, dmx, tmp_models, oracle_threads):
    # T = 4 (LSTM over 4 / 2 steps) and T = 7 (odd: encoder.5's ceil-form output, decoder.0's crop)
    rng = np.random.default_rng(seg)
    mix = (0.1 * rng.standard_normal((2, seg))).astype(np.float32)
    m = dmx.Model(tmp_models[3]); ctx = dmx.Context(m, seg, 1); om = orc.OracleModel(tmp_models[3])
    errs, _, _ = pu.compare_segment(ctx, om, mix)
    bad = {k: v for k, v in errs.items() if not (v < TOL)}
    assert not bad, bad
    ctx.close(); m.close(); om.close()


def test_v3_full_size_segment_vs_oracle(dmx, tmp_models, oracle_threads):
    # the production segment: T = 336 LSTM steps at level 4, 168 at level 5, LocalState over 336 / 168 positions
    rng = np.random.default_rng(33)
    mix = (0.1 * rng.standard_normal((2, SEG_FULL))).astype(np.float32)
    m = dmx.Model(tmp_models[3]); ctx = dmx.Context(m, 0, 1); om = orc.OracleModel(tmp_models[3])
    assert ctx.seg == SEG_FULL
    errs, out, ref = pu.compare_segment(ctx, om, mix)
    bad = {k: v for k, v in errs.items() if not (v < TOL)}
    assert not bad, bad
    for s in range(4):
        assert sdr_db(ref[s], out[s]) > 60.0
    ctx.close(); m.close(); om.close()


def test_v3_batch_equals_singles_bitwise(dmx, tmp_models):
    """18 segments in flight = two 16-column groups of the cooperative LSTM kernel, the second one padded: every
    output bit equals the one-segment-per-call result, run to run."""
    import torch
    seg, B = 8192, 18
    rng = np.random.default_rng(21)
    mixes = (0.1 * rng.standard_normal((B, 2, seg))).astype(np.float32)
    m = dmx.Model(tmp_models[3]); ctx = dmx.Context(m, seg, B)
    d_mix = torch.from_numpy(np.ascontiguousarray(mixes.transpose(0, 2, 1))).cuda()
    d_out = torch.zeros((B, 4, 2, seg), device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        ctx.segment_device(d_mix.data_ptr(), d_out.data_ptr(), B)
        ctx.synchronize()
    got = d_out.cpu().numpy()
    for b in (0, 1, 7, 15, 16, 17):
        single = ctx.segment(mixes[b])
        assert np.array_equal(got[b], single), b
        assert np.array_equal(ctx.segment(mixes[b]), single)
    ctx.close(); m.close()


def test_v3_track_vs_oracle(dmx, tmp_models, oracle_threads):
    """demucs_v3_inference (model_apply.cpp:290-535): normalise, shift, overlapping segments, overlap-add."""
    seg, n, off = 16384, 50000, 1337
    rng = np.random.default_rng(5)
    audio = (0.2 * rng.standard_normal((2, n))).astype(np.float32)
    m = dmx.Model(tmp_models[3]); ctx = dmx.Context(m, seg, 4); om = orc.OracleModel(tmp_models[3])
    out = ctx.track(audio, off)
    ref = om.track(audio, off, seg)
    assert pu.relerr(out, ref) < TOL
    for s in range(4):
        assert sdr_db(ref[s], out[s]) > 60.0
    ctx.close(); m.close(); om.close()


def test_v3_dc_stress_weights(dmx, tmp_path, oracle_threads):
    from demucs_cpp_amd.weights import write_synthetic_model
    p = str(tmp_path / "v3_dc.bin")
    write_synthetic_model(p, 4, 9, "dc", "v3")
    seg = 12000
    rng = np.random.default_rng(9)
    mix = (0.1 * rng.standard_normal((2, seg)) + 0.05).astype(np.float32)
    m = dmx.Model(p); ctx = dmx.Context(m, seg, 1); om = orc.OracleModel(p)
    errs, out, ref = pu.compare_segment(ctx, om, mix)
    bad = {k: v for k, v in errs.items() if not (v < 1e-3)}  # |mean| >> sigma regime: DESIGN.md section 3 envelope
    assert not bad, bad
    assert errs["out"] < TOL, errs["out"]
    ctx.close(); m.close(); om.close()


def test_v3_cli_drop_in_vs_library_and_oracle(dmx, tmp_models, golden_dir, tmp_path, oracle_threads):
    """cli/demucs_v3.cpp.main keeps the reference's argv contract and output naming
    (/root/reference/cli-apps/demucs_v3.cpp:108-232); its stems equal the library's bit for bit and the ORACLE's
    demucs_v3_inference restatement within the tolerance, on the reference's own fixture (test/data/gspi_stereo_short.wav)."""
    import subprocess
    from wavio import read_wav
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "cli", "demucs_v3.cpp.main")
    assert os.path.exists(exe), "CLI not built"
    wav = os.path.join(golden_dir, "gspi_stereo_short.wav")
    out_dir = tmp_path / "stems"
    env = dict(os.environ, DMX_SHIFT_OFFSET="1337", DMX_BATCH="2")
    r = subprocess.run([exe, tmp_models[3], wav, str(out_dir)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "demucs_model_load returned true" in r.stdout and "Starting Demucs v3 MMI inference" in r.stdout
    _, audio = read_wav(wav)
    m = dmx.Model(tmp_models[3]); ctx = dmx.Context(m, 0, 2); om = orc.OracleModel(tmp_models[3])
    lib = ctx.track(audio, 1337)
    ref = om.track(audio, 1337)
    for i, name in enumerate(["drums", "bass", "other", "vocals"]):
        rate, stem = read_wav(str(out_dir / f"target_{i}_{name}.wav"))
        assert rate == 44100 and stem.shape == audio.shape
        assert np.array_equal(stem, lib[i])
        assert pu.relerr(stem, ref[i]) < TOL and sdr_db(ref[i], stem) > 60.0
    # wrong usage, unreadable model, and a v4 file ("bad magic" for load_demucs_v3_model) exit 1 like the reference
    assert subprocess.run([exe], capture_output=True).returncode == 1
    assert subprocess.run([exe, "/nonexistent.bin", wav, str(out_dir)], capture_output=True).returncode == 1
    r = subprocess.run([exe, tmp_models[4], wav, str(out_dir)], capture_output=True, text=True)
    assert r.returncode == 1 and "bad magic" in r.stderr
    # and the v4 CLI refuses the v3 file the same way
    r = subprocess.run([os.path.join(root, "cli", "demucs.cpp.main"), tmp_models[3], wav, str(out_dir)], capture_output=True, text=True)
    assert r.returncode == 1 and "bad magic" in r.stderr
    ctx.close(); m.close(); om.close()


def test_v3_cli_mt_and_sharded_devices(dmx, tmp_models, tmp_path):
    """demucs_v3_mt.cpp.main (/root/reference/cli-apps/demucs_v3_mt.cpp:108-227): <num threads> coarse chunks recombined as
    oracle/threaded_split.py restates cli-apps/threaded_inference.hpp:196-369; and the v3 CLI sharded over two logical
    devices (DMX_DEVICES=0,0) gives the one-device bits."""
    import subprocess
    import sys
    from wavio import read_wav, write_wav_f32
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    from threaded_split import threaded_split
    exe, exe_mt = os.path.join(root, "cli", "demucs_v3.cpp.main"), os.path.join(root, "cli", "demucs_v3_mt.cpp.main")
    assert os.path.exists(exe) and os.path.exists(exe_mt), "CLIs not built"
    audio = (0.1 * np.random.default_rng(23).standard_normal((2, 4 * 44100))).astype(np.float32)
    wav = str(tmp_path / "noise4s.wav")
    write_wav_f32(wav, audio)
    env = dict(os.environ, DMX_SHIFT_OFFSET="1337", DMX_BATCH="2")
    names = ["drums", "bass", "other", "vocals"]
    r = subprocess.run([exe_mt, tmp_models[3], wav, str(tmp_path / "mt"), "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "[THREAD 1]" in r.stdout
    m = dmx.Model(tmp_models[3]); ctx = dmx.Context(m, 0, 2)
    ref = threaded_split(audio, 2, 4, lambda i, chunk: ctx.track(chunk, 1337))
    for i, name in enumerate(names):
        _, stem = read_wav(str(tmp_path / "mt" / f"target_{i}_{name}.wav"))
        assert stem.shape == audio.shape and np.abs(stem - ref[i]).max() <= 2e-6 * max(1.0, np.abs(ref[i]).max())
    assert subprocess.run([exe_mt, tmp_models[3], wav, str(tmp_path / "mt")], capture_output=True).returncode == 1
    ctx.close(); m.close()
    # two segments' worth of audio dealt over two logical devices
    n = 257985 + 5000
    audio2 = (0.1 * np.random.default_rng(24).standard_normal((2, n))).astype(np.float32)
    wav2 = str(tmp_path / "two.wav")
    write_wav_f32(wav2, audio2)
    outs = []
    for devs in ("0", "0,0"):
        od = tmp_path / ("dev" + devs.replace(",", "_"))
        r = subprocess.run([exe, tmp_models[3], wav2, str(od)], env=dict(env, DMX_DEVICES=devs), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append([read_wav(str(od / f"target_{i}_{nm}.wav"))[1] for i, nm in enumerate(names)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_v3_concurrent_contexts_at_full_batches_share_the_lstm_lane(dmx, tmp_models):
    """Four contexts on ONE GPU, each running a 42-segment batch at the same time (engine with logical devices
    [0, 0, 0, 0]): every launch of the cooperative LSTM kernel keeps 144 workgroups spinning on their partners and the
    device holds 512, so four unordered launches could starve each other's partners of slots (the bounded spin would
    report an error). csrc/api.cpp orders the LSTM launches of a device in a lane; the result must be the single-context
    track, bit for bit, twice in a row."""
    stride = 257985
    nseg = 4 * 42
    n = (nseg - 1) * stride + 1000
    audio = (0.1 * np.random.default_rng(61).standard_normal((2, n))).astype(np.float32)
    m = dmx.Model(tmp_models[3]); ctx = dmx.Context(m, 0, 42)
    assert ctx.track_geometry(n, 0)[1] == nseg
    ref = ctx.track(audio, 0)
    ctx.close(); m.close()
    eng = dmx.Engine([tmp_models[3]], [0, 0, 0, 0], max_batch=42)
    for _ in range(2):
        got = eng.track(audio, [0])
        assert np.array_equal(got, ref)
    eng.close()

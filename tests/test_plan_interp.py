"""Validates the product's host-side plan (demucs_cpp_amd/csrc/plan.cpp: op list and
descriptors) and weight repacking (model_pack.cpp) on the CPU by interpreting the plan with
tests/cpu_interp.cpp (the executable specification of every HIP kernel) and comparing with the
fp64 golden vectors and the oracle. No GPU needed; the HIP kernels are checked against the same
oracle in test_gpu_parity.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "_build", "libcpu_interp.so")


@pytest.fixture(scope="module")
def interp():
    srcs = [os.path.join(ROOT, "tests", "cpu_interp.cpp")] + [os.path.join(ROOT, "demucs_cpp_amd", "csrc", f) for f in ("plan.cpp", "plan.h", "model_pack.cpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(SO) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", ROOT, "interp"], stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(SO)
    L.interp_create.restype = ctypes.c_void_p
    L.interp_create.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int]
    L.interp_free.argtypes = [ctypes.c_void_p]
    L.interp_run.argtypes = [ctypes.c_void_p] * 3
    L.interp_n_ops.argtypes = [ctypes.c_void_p]
    return L


def run(L, path, mixes):
    """mixes (B, 2, seg) -> (B, S, 2, seg)"""
    B, _, seg = mixes.shape
    h = L.interp_create(path.encode(), seg, B)
    assert h
    mi = np.ascontiguousarray(mixes.transpose(0, 2, 1)).astype(np.float32)
    ns = 6 if "6s" in path else 4  # htdemucs-4s and hdemucs_mmi (v3) separate 4 stems
    out = np.zeros((B, ns, 2, seg), np.float32)
    L.interp_run(h, mi.ctypes.data, out.ctypes.data)
    L.interp_free(h)
    return out


@pytest.mark.parametrize("ns", [4, 6])
def test_plan_matches_fp64_golden(ns, interp, golden_dir, tmp_models):
    g = np.load(os.path.join(golden_dir, f"golden_seg_{ns}s.npz"))
    out = run(interp, tmp_models[ns], g["mix"][None])
    err = np.abs(out[0] - g["out"]).max() / np.abs(g["out"]).max()
    assert err < 2e-5, err


def test_v3_plan_matches_fp64_golden(interp, golden_dir, tmp_models):
    """Demucs v3 (hdemucs_mmi): dmc3 packing (LSTM gate interleave, fused LocalState projections, k4/s2 transposed
    conv) + build_plan_v3, interpreted on the CPU, against the fp64 torch model of tests/golden/make_golden_v3.py."""
    g = np.load(os.path.join(golden_dir, "golden_seg_v3.npz"))
    out = run(interp, tmp_models[3], g["mix"][None])
    err = np.abs(out[0] - g["out"]).max() / np.abs(g["out"]).max()
    assert err < 2e-5, err


def test_v3_plan_batch_and_odd_frames_vs_oracle(interp, tmp_models):
    rng = np.random.default_rng(31)
    seg = 7000  # T = 7: encoder.5's ceil-form length and decoder.0's crop on an odd frame count
    mixes = (0.1 * rng.standard_normal((2, 2, seg))).astype(np.float32)
    both = run(interp, tmp_models[3], mixes)
    m = orc.OracleModel(tmp_models[3])
    for b in range(2):
        single = run(interp, tmp_models[3], mixes[b:b + 1])
        assert np.abs(both[b] - single[0]).max() < 1e-6
        ref = m.segment(mixes[b])
        assert np.abs(both[b] - ref).max() / np.abs(ref).max() < 2e-5
    m.close()


def test_plan_batch_equals_singles_and_oracle(interp, tmp_models):
    rng = np.random.default_rng(3)
    seg = 6000
    mixes = (0.1 * rng.standard_normal((2, 2, seg))).astype(np.float32)
    both = run(interp, tmp_models[6], mixes)
    m = orc.OracleModel(tmp_models[6])
    for b in range(2):
        single = run(interp, tmp_models[6], mixes[b:b + 1])
        assert np.abs(both[b] - single[0]).max() < 1e-6
        ref = m.segment(mixes[b])
        assert np.abs(both[b] - ref).max() / np.abs(ref).max() < 2e-5
    m.close()


def test_weight_gain_2_is_an_fp32_conditioning_limit_not_a_plan_error(interp, tmp_path):
    """Weights x2 (norm affines, biases, LayerScale untouched): the fp32 ORACLE is ~1e-3 away from the fp64 model of
    tests/golden/make_golden.py - the un-normalised encoders grow activations 4x per level and the GLU gates saturate -
    and the interpreted product plan is no further from the exact result than the oracle is (GPU counterpart:
    test_weight_scale_sweep_ill_conditioned_side_vs_fp64)."""
    import parity_utils as pu
    from demucs_cpp_amd.weights import synth_weights, write_model, tensor_catalogue
    w = synth_weights(4, 0)
    for name, _ in tensor_catalogue(4):
        is_norm = ".norm" in name or name.endswith(".1.weight") or name.endswith(".4.weight")
        if (name.endswith("weight") or name.endswith("in_proj_weight")) and not is_norm and "freq_emb" not in name:
            w[name] = (w[name].astype(np.float32) * 2.0).astype(np.float16)
    path = str(tmp_path / "gain2-4s.bin")
    write_model(path, w, 4)
    mix = (0.1 * np.random.default_rng(18).standard_normal((1, 2, 10000))).astype(np.float32)
    exact = pu.fp64_segment_forward(w, 4, mix[0])
    out = run(interp, path, mix)[0]
    m = orc.OracleModel(path)
    ref = m.segment(mix[0])
    m.close()
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    e_plan, e_orc = rel(out, exact), rel(ref, exact)
    assert e_orc > 1e-4 and e_plan < 3 * e_orc + 1e-5, (e_plan, e_orc)


@pytest.mark.parametrize("ns", [4, 6, 3])
def test_two_stream_waits_cover_every_hazard(ns, interp, tmp_models):
    """The engine runs the freq and time branches on two HIP streams joined only by the waits
    plan.cpp derives from the ops' arena ranges. Executing the plan in the two most skewed
    interleavings those waits permit must reproduce plan order bit for bit."""
    interp.interp_run_order.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int]
    interp.interp_n_waits.argtypes = [ctypes.c_void_p]
    rng = np.random.default_rng(11)
    seg = 6000
    mix = (0.1 * rng.standard_normal((1, seg, 2))).astype(np.float32)
    outs = []
    for order in (0, 1, 2):
        h = interp.interp_create(tmp_models[ns].encode(), seg, 1)
        nw = interp.interp_n_waits(h)
        out = np.zeros((1, 6 if ns == 6 else 4, 2, seg), np.float32)  # key 3 = Demucs v3 (4 stems)
        interp.interp_run_order(h, mix.ctypes.data, out.ctypes.data, order)
        interp.interp_free(h)
        outs.append(out)
    assert 4 <= nw <= 64, nw  # joins exist (STFT fan-out, cross layers, final sum) and stay few
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0], outs[2])


TILES = {0: (128, 128), 1: (64, 64), 2: (128, 96), 3: (128, 48), 4: (256, 16), 5: (128, 32), 6: (128, 64), 7: (64, 128), 8: (16, 256),
         9: (64, 64), 10: (64, 96), 11: (64, 48), 12: (64, 32), 13: (64, 64), 14: (128, 16), 15: (32, 128), 16: (32, 64),
         17: (256, 128), 18: (256, 128), 19: (256, 128), 20: (256, 96)}  # 17 / 18: experiment tiles (DMX_TALL); 19: igemm_lin256.hip, linear layers
# cfg -> family (column decomposition: waves x fragments along N); siblings of a family give identical bits,
# the families with row statistics never cross (plan.h)
FAMILY = {0: "2x4", 7: "2x4", 15: "2x4", 17: "2x4", 18: "2x4", 19: "2x4", 20: "1x6", 9: "2x2", 16: "2x2", 2: "1x6", 10: "1x6", 3: "1x3", 11: "1x3", 5: "1x2", 12: "1x2",
          6: "1x4", 13: "1x4", 4: "1x1", 14: "1x1", 8: "direct", 1: "2x2o"}


@pytest.mark.parametrize("ns", [4, 6])
def test_tile_choice_keeps_the_column_tiling_at_every_batch_size(ns, interp, tmp_models):
    """refine_cfg picks a tile per launch from the actual row count (cost model, half- and quarter-height siblings).
    What makes that safe: an output element is the same k-ordered fmaf chain in every tile shape; for the ops that
    also write row statistics, the COLUMN extent of the tile (hence NB, the layout of the partials and their summation
    order) and the column decomposition are the same at every batch size; and the choice only ever shrinks the tile,
    and only for launches that would not fill the chip a few times over."""
    interp.interp_plan_dump.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    interp.interp_create_plan.restype = ctypes.c_void_p
    interp.interp_create_plan.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int]
    plans = {}
    for b in (1, 2, 3, 4, 12, 24):
        h = interp.interp_create_plan(tmp_models[ns].encode(), 343980, b)
        buf = ctypes.create_string_buffer(1 << 18)
        assert interp.interp_plan_dump(h, buf, 1 << 18) > 0
        interp.interp_free(h)
        plans[b] = [ln.split() for ln in buf.value.decode().splitlines()]
    ref = plans[24]
    shrunk = 0
    for b, ops in plans.items():
        assert [o[0] for o in ops] == [o[0] for o in ref]
        for o, r in zip(ops, ref):
            name, cfg, M, N, K, tiles, stat = o[0], int(o[1]), int(o[2]), int(o[3]), int(o[4]), int(o[5]), int(o[6])
            rcfg = int(r[1])
            if stat:  # row statistics: same BN (hence NB and the layout of the partials) and same column decomposition
                assert TILES[cfg][1] == TILES[rcfg][1], (name, b, cfg, rcfg)
                assert FAMILY[cfg] == FAMILY[rcfg], (name, b, cfg, rcfg)
            assert TILES[cfg][0] <= TILES[rcfg][0]  # fewer rows in flight never get the bigger tile
            bm, bn = TILES[cfg]
            assert tiles == -(-M // bm) * -(-N // bn) or cfg == 8
            if TILES[cfg][0] < TILES[rcfg][0]:
                shrunk += 1
                big = -(-M // TILES[rcfg][0]) * -(-N // TILES[rcfg][1])
                assert big <= 3 * 512, (name, b, big)  # only launches that would not fill the chip a few times over
    assert shrunk > 20  # one and two segments per call do use the small tiles


@pytest.mark.parametrize("ns", [4, 6])
def test_fp16x3_plans_give_an_op_one_arithmetic_at_every_batch_size(ns, interp, tmp_models, monkeypatch):
    """GEMM_FP16X3 plans (plan.cpp; host only): the ops that may take the fp16-term kernel are marked by the PLAN (IGemm::hterms),
    they are exactly the transformer's linear layers, its K / V plane projections and the 1x1 channel up- / down-samplers around it
    (one `linear` builder) - the same set at every batch size - and they
    sit on the 128- / 64-row tiles of the 128-wide family at every batch size (the only tiles igemm_split_lin_kernel has), so an
    op never changes arithmetic with the number of segments in flight: batch = singles stays bitwise. (First form of the mode:
    the 1x1 convs of the DConv blocks also qualified as "linear" on some tiles and not on others - caught on the GPU by
    test_full_size_track_properties.) Everything else is laid out exactly as in a GEMM_BF16X3 plan."""
    interp.interp_plan_dump.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    interp.interp_create_plan.restype = ctypes.c_void_p
    interp.interp_create_plan.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int]

    def dump(mode, b, seg=343980):
        monkeypatch.setenv("DMX_INTERP_GEMM", str(mode))
        h = interp.interp_create_plan(tmp_models[ns].encode(), seg, b)
        buf = ctypes.create_string_buffer(1 << 18)
        assert interp.interp_plan_dump(h, buf, 1 << 18) > 0
        interp.interp_free(h)
        return [ln.split() for ln in buf.value.decode().splitlines()]

    EPI_KPL, EPI_VT = 7, 8
    marked_ref = None
    for b in (1, 2, 3, 4, 5, 8, 12, 24, 42, 64):
        ops2, ops1 = dump(2, b), dump(1, b)
        assert [o[0] for o in ops2] == [o[0] for o in ops1]
        marked = [o[0] for o in ops2 if int(o[7])]
        marked_ref = marked_ref or marked
        assert marked == marked_ref and len(marked) >= 50
        for o2, o1 in zip(ops2, ops1):
            name, cfg, N, K, ht, epi = o2[0], int(o2[1]), int(o2[3]), int(o2[4]), int(o2[7]), int(o2[8])
            assert int(o1[7]) == 0  # bf16x3 plans never mark anything
            if ht:
                assert name.startswith(("crosstransformer.", "channel_")) and cfg in (0, 7) and N % 128 == 0 and K % 32 == 0, (b, o2)
            else:
                assert o2[:7] == o1[:7], (b, o2, o1)  # same tile, same everything as the bf16x3 plan
                assert not name.startswith("crosstransformer.") or epi not in (EPI_KPL, EPI_VT)
        lin = [o for o in ops2 if o[0].startswith("crosstransformer.") and o[0].rsplit(".", 1)[-1] in ("linear1", "linear2", "out_proj", "qk", "q", "k", "v", "qkv", "kv")]
        assert lin and all(int(o[7]) for o in lin), [o for o in lin if not int(o[7])]
    # short segments (no whole 64-key tiles: no planes) keep the marking on the plain projections
    ops = dump(2, 3, 20000)
    assert any(int(o[7]) for o in ops) and not any(int(o[8]) in (EPI_KPL, EPI_VT) for o in ops)
    assert all(int(o[1]) in (0, 7) for o in ops if int(o[7]))


def test_tile_choice_counts_workgroups_per_xcd(interp, tmp_models):
    """The igemm tile map deals ROW tiles round-robin to the 8 XCDs (all column tiles of a row tile on one XCD), so the
    cost model counts the workgroups of the busiest XCD: decoder.0.rewrite at 4 segments is 84 x 6 tiles of 128x128 =
    11 row tiles on four of the XCDs = 66 workgroups for 64 slots; measured 617 us against 428 us with the 64x128 sibling
    (profiles/DESIGN_history_r1-r4.md 7.1). The double-height experiment tile (cfg 17) is never chosen without DMX_TALL."""
    interp.interp_plan_dump.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    interp.interp_create_plan.restype = ctypes.c_void_p
    interp.interp_create_plan.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int]
    assert "DMX_TALL" not in os.environ
    for b, want in ((4, 7), (42, 0)):
        h = interp.interp_create_plan(tmp_models[4].encode(), 343980, b)
        buf = ctypes.create_string_buffer(1 << 18)
        assert interp.interp_plan_dump(h, buf, 1 << 18) > 0
        interp.interp_free(h)
        ops = {ln.split()[0]: ln.split() for ln in buf.value.decode().splitlines()}
        assert int(ops["decoder.0.rewrite"][1]) == want, (b, ops["decoder.0.rewrite"])
        assert all(int(o[1]) != 17 for o in ops.values())

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The driver's GPU step has a time limit (1200 s; round 5's suite took 726 s with every test in all three arithmetics). By default:
#  * the opt-in fp16x3 mode runs the tests that are ABOUT the arithmetic (parity at reduced and full size, batch = singles, its own
#    bounds and kernels, non-finite inputs, the 4-minute 4s track); plumbing tests do not repeat in it, and v3 has no transformer
#    (a v3 fp16x3 context builds the bf16x3 plan bit for bit: asserted once);
#  * tests of host-side plumbing whose device work is mode-independent by construction (engine dealing / finish modes / RCCL
#    transport, CLI sharding, the bag through the engine) run in the default arithmetic (bf16x3) only.
#    Exception (round 6): the eight-logical-device run of the 6-source model also runs in f32 - eight contexts competing for one
#    GPU is what exposed a barrier missing from the fp32 attention kernel's head-dim-48 form (attention.hip), which no quiet run shows.
# DMX_TEST_ALL_MODES=1 runs everything in every mode (profiles/r05_gpu_tests.txt is such a run).
FP16X3_KEEPS = ("test_reduced_segment_all_layers_vs_oracle_and_golden", "test_full_size_segment_vs_oracle", "test_batch_equals_singles_bitwise_and_layouts",
                "test_bench_batch_and_awkward_lengths_equal_singles", "test_full_4min_track_end_to_end", "test_track_vs_oracle_reduced",
                "test_stream_schedule_does_not_change_a_bit", "test_single_segment_graph_replay_is_bit_identical",
                "test_kv_operand_planes_equal_the_fp32_kv_path", "test_two_contexts_on_one_gpu_do_not_disturb_each_other",
                "test_engine_two_logical_devices_equals_one_context_bitwise")
DEFAULT_MODE_ONLY = ("test_full_ft_bag_4min_track_over_one_and_eight_logical_devices", "test_cli_mt_and_ft_drop_in",
                     "test_cli_shards_over_dmx_devices_and_finish_modes", "test_engine_rccl_self_exchange_moves_the_slabs",
                     "test_engine_rccl_agrees_before_the_exchange", "test_engine_rccl_transport_binds_and_builds_a_communicator",
                     "test_engine_ft_bag_equals_four_sequential_runs_bitwise", "test_engine_owner_finish_mode_equals_root_gather_bitwise",
                     "test_cli_mono_input_is_duplicated_to_stereo", "test_argument_errors",
                     "test_shim_is_reentrant_and_eigen_overloads_match", "test_caller_stream_ordering_without_host_sync")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("DMX_TEST_ALL_MODES", "0") not in ("", "0"):
        return
    skip_h = pytest.mark.skip(reason="not repeated in the opt-in fp16x3 mode by default (tests/conftest.py); DMX_TEST_ALL_MODES=1 runs it")
    skip_p = pytest.mark.skip(reason="host-side plumbing, mode-independent: default arithmetic only (tests/conftest.py); DMX_TEST_ALL_MODES=1 runs it")
    for it in items:
        cs = getattr(it, "callspec", None)
        if cs is None or "dmx" not in cs.params:
            continue
        mode, name = cs.params["dmx"], it.originalname or it.name
        if mode == "fp16x3" and (os.path.basename(str(it.fspath)) == "test_gpu_v3.py" or name not in FP16X3_KEEPS):
            it.add_marker(skip_h)
        elif mode != "bf16x3" and name in DEFAULT_MODE_ONLY:
            it.add_marker(skip_p)
        elif mode == "f32" and name == "test_full_4min_track_end_to_end" and cs.params.get("ns") == 6:
            it.add_marker(skip_p)  # (the f32 family is covered by the 4s track; 6s adds the source count, not the arithmetic)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def tmp_models(tmp_path_factory):
    """Synthetic dmc4 / dmc6 / dmc3 weight files (seeds match tests/golden/make_golden*.py); key 3 = Demucs v3."""
    from demucs_cpp_amd.weights import write_synthetic_model

    d = tmp_path_factory.mktemp("models")
    p4 = str(d / "ggml-model-htdemucs-4s-f16.bin")
    p6 = str(d / "ggml-model-htdemucs-6s-f16.bin")
    write_synthetic_model(p4, 4, 0)
    write_synthetic_model(p6, 6, 3)
    p3 = str(d / "ggml-model-hdemucs_mmi-v3-f16.bin")  # README.md:82 of the reference
    write_synthetic_model(p3, 4, 5, "default", "v3")  # seed matches tests/golden/make_golden_v3.py
    return {4: p4, 6: p6, 3: p3}


GEMM_MODES = ["f32", "bf16x3", "fp16x3"]


@pytest.fixture(scope="module", params=GEMM_MODES)
def dmx(request):
    """The ctypes binding with the process default GEMM arithmetic set to the parameter: every `-m gpu` test that takes
    this fixture runs once per mode, IN PROCESS (contexts and engines created by the test get the mode; child processes -
    the CLIs, bench.py - inherit it through DMX_GEMM). f32: fp32 MFMA; bf16x3: exact bf16 operand splits, fp32 accumulate;
    fp16x3 (opt-in mode): as bf16x3, the linear layers with fp16 terms under a per-row scale
    (include/demucs_hip.h DMX_GEMM_*)."""
    from demucs_cpp_amd import binding

    assert binding.device_count() >= 1, "no HIP device: the product has no CPU fallback"
    mode = {"f32": binding.GEMM_F32, "bf16x3": binding.GEMM_BF16X3, "fp16x3": binding.GEMM_FP16X3}[request.param]
    old_mode, old_env = binding.default_gemm(), os.environ.get("DMX_GEMM")
    binding.set_default_gemm(mode)
    os.environ["DMX_GEMM"] = request.param
    binding.gemm_mode_name = request.param
    yield binding
    binding.set_default_gemm(old_mode)
    if old_env is None:
        os.environ.pop("DMX_GEMM", None)
    else:
        os.environ["DMX_GEMM"] = old_env

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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

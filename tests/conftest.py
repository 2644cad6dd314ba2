import os
import sys

import pytest

try:  # BEFORE anything loads libminiasm_amd.so: PyTorch-ROCm carries a HIP runtime of its own, and a process that loaded the system's first (through our library) finds
    # "No HIP GPUs are available" when torch initialises later (round 5, visit 2: tests/test_gpu_ingest.py run without the modules that import torch at collection time).
    # Loaded in this order the two share one runtime -- the order bench.py and the full suite have always had.
    import torch  # noqa: F401
except Exception:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built():
    """make sure the in-tree artefacts exist (hipcc cross-compiles without a GPU)"""
    import miniasm_amd as ma
    need = [ma.LIB_PATH, ma.CLI_PATH, ma.PAFGEN_PATH, os.path.join(ROOT, "oracle", "libma_oracle.so"),
            os.path.join(ROOT, "miniasm_amd", "lib", "libma_core_host.so")]
    if not all(os.path.exists(p) for p in need):
        ma.build()
    return True


@pytest.fixture(scope="session")
def tmpdir_s(tmp_path_factory):
    return str(tmp_path_factory.mktemp("ma"))


@pytest.fixture(scope="session")
def gpu_ctx():
    import miniasm_amd as ma
    c = ma.Ctx(0)
    yield c
    c.close()
